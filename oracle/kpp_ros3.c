/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, table-driven CPU restatement of the reference's per-cell chemistry integrator (Mistra-UEA/Mistra,
 * KPP-2.2.4 generated Rosenbrock path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library; the product (mistra_amd/) never does.
 *
 * Parity status: PINNED — bit-exact against the reference itself (oracle/_ref/libmistra_ref.so, flang -O2
 * -ffp-contract=off, built from /root/reference/src by oracle/build_ref.sh) on every function below and on whole
 * INTEGRATE_x calls captured from the running reference model; see tests/test_oracle.py and
 * tests/golden/ (the reference ships no golden vectors of its own).
 *
 * The mechanism (reaction products, stoichiometric sums, LU sparsity) comes from the committed tables
 * mistra_amd/mech/<mech>.mech (tools/extract_mech.py); this file holds the algorithm.  Reference lines followed
 * (gas.f; aer.f and tot.f are the same generated code):
 *   INTEGRATE_g            gas.f:710-773     kpp_integrate          (hard-wired options: Ros3, scalar tol 1e-3/1e-25, Hstart 1e-3)
 *   Rosenbrock_g           gas.f:777-1108    kpp_integrate          (defaults: Hmin 0, Hmax |Tend-Tstart|, Fac 0.2/6/0.1/0.9, 100000 steps)
 *   RosenbrockIntegrator_g gas.f:1112-1337   ros_integrator
 *   ros_ErrorNorm_g        gas.f:1341-1372   ros_error_norm
 *   ros_FunTimeDerivative_g gas.f:1375-1400  ros_fun_time_derivative
 *   ros_PrepareMatrix_g    gas.f:1404-1470   ros_prepare_matrix
 *   Ros3_g                 gas.f:1570-1628   ROS3_* constants
 *   Fun_g                  gas.f:2043-2617   kpp_fun
 *   Jac_SP_g               gas.f:2656-6092   kpp_jac_sp
 *   KppDecomp_g            gas.f:6142-6176   kpp_decomp
 *   KppSolve_g             gas.f:6206-6608   kpp_solve   (the generated code is the unrolled form of the CSR loops here)
 *   WAXPY_g                gas.f:6641-6675   waxpy
 * Arithmetic is IEEE double, one rounding per operation, in the reference's source order; build with
 * -ffp-contract=off (oracle/Makefile does).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define KMCH_MAGIC 0x48434D4B
#define KMCH_VERSION 2

typedef struct kpp_mech {
  int nvar, nfix, nreact, nnz, n_afac, nb, n_bfac, n_vd, n_jv, nconst;
  const int32_t *crow, *icol, *diag;          /* 0-based CSR of the LU pattern, diag[k] = position of (k,k) */
  const int32_t *a_ptr, *a_fac;               /* A(i) = RCT(i) * prod X[a_fac[a_ptr[i]..a_ptr[i+1])]        */
  const int32_t *b_rct, *b_ptr, *b_fac;       /* B(m) = RCT(b_rct[m]) * prod X[b_fac[...]]                  */
  const int32_t *vd_ptr, *vd_idx;             /* Vdot(j) = sum vd_coef * A(vd_idx)                          */
  const int32_t *jv_ptr, *jv_idx;             /* JVS(k)  = sum jv_coef * B(jv_idx)   (empty -> 0)            */
  const double *vd_coef, *jv_coef, *consts;   /* X = [V | F | consts]                                       */
  void *blob;
} kpp_mech;

kpp_mech *kpp_mech_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *blob = (char *)malloc((size_t)sz);
  if (fread(blob, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(blob); return NULL; }
  fclose(f);
  const int32_t *h = (const int32_t *)blob;
  if (h[0] != KMCH_MAGIC || h[1] != KMCH_VERSION) { free(blob); return NULL; }
  kpp_mech *m = (kpp_mech *)calloc(1, sizeof *m);
  m->blob = blob;
  m->nvar = h[2]; m->nfix = h[3]; m->nreact = h[4]; m->nnz = h[5]; m->n_afac = h[6]; m->nb = h[7];
  m->n_bfac = h[8]; m->n_vd = h[9]; m->n_jv = h[10]; m->nconst = h[11];
  size_t off = 48;
#define TAKE_I(field, n) m->field = (const int32_t *)(blob + off); off += 4 * (size_t)(n)
  TAKE_I(crow, m->nvar + 1); TAKE_I(icol, m->nnz); TAKE_I(diag, m->nvar);
  TAKE_I(a_ptr, m->nreact + 1); TAKE_I(a_fac, m->n_afac);
  TAKE_I(b_rct, m->nb); TAKE_I(b_ptr, m->nb + 1); TAKE_I(b_fac, m->n_bfac);
  TAKE_I(vd_ptr, m->nvar + 1); TAKE_I(vd_idx, m->n_vd);
  TAKE_I(jv_ptr, m->nnz + 1); TAKE_I(jv_idx, m->n_jv);
  off += (8 - off % 8) % 8;
#define TAKE_D(field, n) m->field = (const double *)(blob + off); off += 8 * (size_t)(n)
  TAKE_D(vd_coef, m->n_vd); TAKE_D(jv_coef, m->n_jv); TAKE_D(consts, m->nconst);
  if ((long)off != sz) { free(blob); free(m); return NULL; }
  return m;
}

void kpp_mech_free(kpp_mech *m) { if (m) { free(m->blob); free(m); } }
int kpp_mech_dim(const kpp_mech *m, int which) {
  switch (which) { case 0: return m->nvar; case 1: return m->nfix; case 2: return m->nreact; case 3: return m->nnz; }
  return -1;
}

/* Sensitivity variants — NOT the reference's arithmetic; used only by tests/tools to measure how far the result moves
 * under legal re-associations (what a different compiler or summation order does to the reference itself):
 *   bit 0: backward sweep of KppSolve subtracts its terms in DESCENDING column order (the kernel's readiness order)
 *   bit 1: a*b+c contracted to fma() in KppDecomp, KppSolve and the stoichiometric sums (-ffp-contract=fast on FMA hardware)
 *   bit 2: quotients by a pivot formed as a product with its reciprocal (KppDecomp multipliers, KppSolve backward sweep)
 * Default 0 = the pinned, bit-exact restatement. */
static int kpp_variant = 0;
void kpp_set_variant(int v) { kpp_variant = v; }

/* Study knobs — NOT INTEGRATE_x's values (gas.f:743-746 fixes RTOL 1e-3, ATOL 1e-25, Hstart 1e-3): a tighter tolerance gives
 * the "truth" the opt-in Hstart-reuse mode of the kernel is measured against (tools/hstart_study.py).  0 = the reference's value. */
static double opt_rtol = 0.0, opt_atol = 0.0, opt_hstart = 0.0;
void kpp_set_options(double rtol, double atol, double hstart) { opt_rtol = rtol; opt_atol = atol; opt_hstart = hstart; }
/* IPAR(3) of Rosenbrock_x, "maximum number of integration steps" (gas.f:845-846): INTEGRATE_x leaves it 0 = the default 100000 (gas.f:729-732).
 * A smaller value makes the IERR = -6 exit (gas.f:1199-1202) reachable in a test; 0 = the reference's value. */
static int opt_max_steps = 0;
void kpp_set_max_steps(int n) { opt_max_steps = n; }

/* factor lookup into X = [V | F | consts] */
static inline double xval(const kpp_mech *m, const double *V, const double *F, int code) {
  if (code < m->nvar) return V[code];
  if (code < m->nvar + m->nfix) return F[code - m->nvar];
  return m->consts[code - m->nvar - m->nfix];
}

/* signed stoichiometric sum, left to right as written in the generated code (first term is not added to zero) */
static inline double signed_sum(const int32_t *idx, const double *coef, int lo, int hi, const double *src) {
  if (lo == hi) return 0.0;
  double acc = (coef[lo] == 1.0) ? src[idx[lo]] : (coef[lo] == -1.0) ? -src[idx[lo]] : coef[lo] * src[idx[lo]];
  for (int t = lo + 1; t < hi; t++) {
    double c = coef[t];
    if (c == 1.0) acc = acc + src[idx[t]];
    else if (c == -1.0) acc = acc - src[idx[t]];
    else if (kpp_variant & 2) acc = fma(c, src[idx[t]], acc);
    else acc = acc + c * src[idx[t]];
  }
  return acc;
}

/* Fun_x (gas.f:2043): A(i) = RCT(i)*V(..)[*V|F(..)...], Vdot(j) = signed sum of A */
void kpp_fun(const kpp_mech *m, const double *V, const double *F, const double *RCT, double *Vdot, double *A_work) {
  double *A = A_work;
  for (int i = 0; i < m->nreact; i++) {
    double a = RCT[i];
    for (int p = m->a_ptr[i]; p < m->a_ptr[i + 1]; p++) a = a * xval(m, V, F, m->a_fac[p]);
    A[i] = a;
  }
  for (int j = 0; j < m->nvar; j++) Vdot[j] = signed_sum(m->vd_idx, m->vd_coef, m->vd_ptr[j], m->vd_ptr[j + 1], A);
}

/* Jac_SP_x (gas.f:2656): B(m) = dA/dV, JVS(k) = signed sum of B, explicit 0 for fill-in slots */
void kpp_jac_sp(const kpp_mech *m, const double *V, const double *F, const double *RCT, double *JVS, double *B_work) {
  double *B = B_work;
  for (int i = 0; i < m->nb; i++) {
    double b = RCT[m->b_rct[i]];
    for (int p = m->b_ptr[i]; p < m->b_ptr[i + 1]; p++) b = b * xval(m, V, F, m->b_fac[p]);
    B[i] = b;
  }
  for (int k = 0; k < m->nnz; k++) JVS[k] = signed_sum(m->jv_idx, m->jv_coef, m->jv_ptr[k], m->jv_ptr[k + 1], B);
}

/* KppDecomp_x (gas.f:6142): in-place row-wise sparse LU, no pivoting; returns 0 or the 1-based row of a zero pivot */
int kpp_decomp(const kpp_mech *m, double *JVS, double *W) {
  const int32_t *crow = m->crow, *icol = m->icol, *diag = m->diag;
  for (int k = 0; k < m->nvar; k++) {
    if (JVS[diag[k]] == 0.0) return k + 1;
    for (int kk = crow[k]; kk < crow[k + 1]; kk++) W[icol[kk]] = JVS[kk];
    for (int kk = crow[k]; kk < diag[k]; kk++) {
      int j = icol[kk];
      double a = (kpp_variant & 4) ? -(W[j] * (1.0 / JVS[diag[j]])) : -W[j] / JVS[diag[j]];
      W[j] = -a;
      if (kpp_variant & 2)
        for (int jj = diag[j] + 1; jj < crow[j + 1]; jj++) W[icol[jj]] = fma(a, JVS[jj], W[icol[jj]]);
      else
        for (int jj = diag[j] + 1; jj < crow[j + 1]; jj++) W[icol[jj]] = W[icol[jj]] + a * JVS[jj];
    }
    for (int kk = crow[k]; kk < crow[k + 1]; kk++) JVS[kk] = W[icol[kk]];
  }
  return 0;
}

/* KppSolve_x (gas.f:6206): forward (unit L) then backward (divide by diagonal) substitution */
void kpp_solve(const kpp_mech *m, const double *JVS, double *X) {
  const int32_t *crow = m->crow, *icol = m->icol, *diag = m->diag;
  if (kpp_variant) {   /* sensitivity variants, see kpp_set_variant */
    const int use_fma = kpp_variant & 2, desc = kpp_variant & 1;
    for (int i = 0; i < m->nvar; i++) {
      double acc = X[i];
      for (int k = crow[i]; k < diag[i]; k++) acc = use_fma ? fma(-JVS[k], X[icol[k]], acc) : acc - JVS[k] * X[icol[k]];
      X[i] = acc;
    }
    for (int i = m->nvar - 1; i >= 0; i--) {
      double acc = X[i];
      if (desc) for (int k = crow[i + 1] - 1; k > diag[i]; k--) acc = use_fma ? fma(-JVS[k], X[icol[k]], acc) : acc - JVS[k] * X[icol[k]];
      else for (int k = diag[i] + 1; k < crow[i + 1]; k++) acc = use_fma ? fma(-JVS[k], X[icol[k]], acc) : acc - JVS[k] * X[icol[k]];
      X[i] = (kpp_variant & 4) ? acc * (1.0 / JVS[diag[i]]) : acc / JVS[diag[i]];
    }
    return;
  }
  for (int i = 0; i < m->nvar; i++) {
    double acc = X[i];
    for (int k = crow[i]; k < diag[i]; k++) acc = acc - JVS[k] * X[icol[k]];
    X[i] = acc;
  }
  for (int i = m->nvar - 1; i >= 0; i--) {
    double acc = X[i];
    for (int k = diag[i] + 1; k < crow[i + 1]; k++) acc = acc - JVS[k] * X[icol[k]];
    X[i] = acc / JVS[diag[i]];
  }
}

/* WAXPY_x (gas.f:6641) */
static void waxpy(int n, double alpha, const double *x, double *y) {
  if (alpha == 0.0) return;
  for (int i = 0; i < n; i++) y[i] = y[i] + alpha * x[i];
}

/* Ros3_x (gas.f:1596-1626) */
static const double ROS3_A1 = 1.0;
static const double ROS3_C[3] = {-0.10156171083877702091975600115545e+01, 0.40759956452537699824805835358067e+01,
                                 0.92076794298330791242156818474003e+01};
static const double ROS3_M[3] = {0.1e+01, 0.61697947043828245592553615689730e+01, -0.42772256543218573326238373806514e+00};
static const double ROS3_E[3] = {0.5e+00, -0.29079558716805469821718236208017e+01, 0.22354069897811569627360909276199e+00};
static const double ROS3_ALPHA[3] = {0.0, 0.43586652150845899941601945119356e+00, 0.43586652150845899941601945119356e+00};
static const double ROS3_GAMMA[3] = {0.43586652150845899941601945119356e+00, 0.24291996454816804366592249683314e+00,
                                     0.21851380027664058511513169485832e+01};
static const double ROS3_ELO = 3.0;

enum { ST_NFUN, ST_NJAC, ST_NSTP, ST_NACC, ST_NREJ, ST_NDEC, ST_NSOL, ST_NSNG };

typedef struct {
  double *Ynew, *Fcn0, *Fcn, *K, *dFdT, *Yerr, *Jac0, *Ghimj, *W, *AB;
} ros_work;

/* Fortran MIN/MAX as flang lowers them for REAL*8 (second operand wins unless the first compares strictly) */
static inline double fmin_f(double a, double b) { return (a < b || b != b) ? a : b; }
static inline double fmax_f(double a, double b) { return (a > b || b != b) ? a : b; }

/* ros_ErrorNorm_x (gas.f:1341), scalar tolerances (VectorTol is false because INTEGRATE sets IPAR(2)=1) */
static double ros_error_norm(int n, const double *Y, const double *Ynew, const double *Yerr, double abstol, double reltol) {
  double err = 0.0;
  for (int i = 0; i < n; i++) {
    double ymax = fmax_f(fabs(Y[i]), fabs(Ynew[i]));
    double scale = abstol + reltol * ymax;
    double q = Yerr[i] / scale;
    err = err + q * q;
  }
  return sqrt(err / (double)n);
}

/* returns IERR (1 = success, negative = error code of ros_ErrorMsg_x, gas.f:1474) */
static int ros_integrator(const kpp_mech *m, double *Y, const double *FIX, const double *RCONST, double Tstart, double Tend,
                          double *T_out, double *Hexit_out, int32_t *st, ros_work *w) {
  const int n = m->nvar, nnz = m->nnz;
  const double Roundoff = DBL_EPSILON, Hmin = 0.0, FacMin = 0.2, FacMax = 6.0, FacRej = 0.1, FacSafe = 0.9;
  const double AbsTol = opt_atol > 0.0 ? opt_atol : 1.0e-25, RelTol = opt_rtol > 0.0 ? opt_rtol : 1.0e-3, DeltaMin = 1.0e-5;
  const int Max_no_steps = opt_max_steps > 0 ? opt_max_steps : 100000;
  const double Hmax = fabs(Tend - Tstart);
  const double Hstart = fmin_f(fabs(opt_hstart > 0.0 ? opt_hstart : 1.0e-3), fabs(Tend - Tstart));
  double *K1 = w->K, *K2 = w->K + n, *K3 = w->K + 2 * n;
  double *Kst[3] = {K1, K2, K3};

  double T = Tstart, Hexit = 0.0, H = fmin_f(Hstart, Hmax), Hnew, Err, Fac;
  if (fabs(H) <= 10.0 * Roundoff) H = DeltaMin;
  const int Direction = (Tend >= Tstart) ? 1 : -1;
  int RejectLastH = 0, RejectMoreH = 0;
  *T_out = T; *Hexit_out = Hexit;

  while (fabs(Tend - T) >= Roundoff) {
    if (st[ST_NSTP] > Max_no_steps) { *T_out = T; *Hexit_out = Hexit; return -6; }
    if (((T + 0.1 * H) == T) || (H <= Roundoff)) { *T_out = T; *Hexit_out = Hexit; return -7; }
    Hexit = H;
    H = fmin_f(H, fabs(Tend - T));

    kpp_fun(m, Y, FIX, RCONST, w->Fcn0, w->AB); st[ST_NFUN]++;
    { /* ros_FunTimeDerivative_x: Fun does not depend on T, the difference is an exact zero (kept for signed-zero parity) */
      double Delta = sqrt(Roundoff) * fmax_f(1.0e-6, fabs(T));
      kpp_fun(m, Y, FIX, RCONST, w->dFdT, w->AB); st[ST_NFUN]++;
      waxpy(n, -1.0, w->Fcn0, w->dFdT);
      double inv = 1.0 / Delta;
      for (int i = 0; i < n; i++) w->dFdT[i] = inv * w->dFdT[i];
    }
    kpp_jac_sp(m, Y, FIX, RCONST, w->Jac0, w->AB); st[ST_NJAC]++;

    for (;;) { /* until the step is accepted */
      { /* ros_PrepareMatrix_x */
        int nconsecutive = 0, singular = 1;
        while (singular) {
          for (int i = 0; i < nnz; i++) w->Ghimj[i] = -w->Jac0[i];
          double ghinv = 1.0 / ((double)Direction * H * ROS3_GAMMA[0]);
          for (int i = 0; i < n; i++) w->Ghimj[m->diag[i]] = w->Ghimj[m->diag[i]] + ghinv;
          int ising = kpp_decomp(m, w->Ghimj, w->W); st[ST_NDEC]++;
          if (ising == 0) singular = 0;
          else {
            st[ST_NSNG]++; nconsecutive++;
            if (nconsecutive <= 5) H = H * 0.5;
            else { *T_out = T; *Hexit_out = Hexit; return -8; }
          }
        }
      }
      for (int istage = 0; istage < 3; istage++) {
        double *Ki = Kst[istage];
        if (istage == 0) memcpy(w->Fcn, w->Fcn0, sizeof(double) * n);
        else if (istage == 1) { /* ros_NewF(2) = .TRUE.; ros_NewF(3) = .FALSE. */
          memcpy(w->Ynew, Y, sizeof(double) * n);
          waxpy(n, ROS3_A1, K1, w->Ynew);
          kpp_fun(m, w->Ynew, FIX, RCONST, w->Fcn, w->AB); st[ST_NFUN]++;
        }
        memcpy(Ki, w->Fcn, sizeof(double) * n);
        for (int j = 0; j < istage; j++) {
          double HC = ROS3_C[istage * (istage - 1) / 2 + j] / ((double)Direction * H);
          waxpy(n, HC, Kst[j], Ki);
        }
        if (ROS3_GAMMA[istage] != 0.0) {
          double HG = (double)Direction * H * ROS3_GAMMA[istage];
          waxpy(n, HG, w->dFdT, Ki);
        }
        kpp_solve(m, w->Ghimj, Ki); st[ST_NSOL]++;
      }
      memcpy(w->Ynew, Y, sizeof(double) * n);
      for (int j = 0; j < 3; j++) waxpy(n, ROS3_M[j], Kst[j], w->Ynew);
      for (int i = 0; i < n; i++) w->Yerr[i] = 0.0;
      for (int j = 0; j < 3; j++) waxpy(n, ROS3_E[j], Kst[j], w->Yerr);
      Err = ros_error_norm(n, Y, w->Ynew, w->Yerr, AbsTol, RelTol);

      Fac = fmin_f(FacMax, fmax_f(FacMin, FacSafe / pow(Err, 1.0 / ROS3_ELO)));
      Hnew = H * Fac;
      st[ST_NSTP]++;
      if ((Err <= 1.0) || (H <= Hmin)) {
        st[ST_NACC]++;
        memcpy(Y, w->Ynew, sizeof(double) * n);
        T = T + (double)Direction * H;
        Hnew = fmax_f(Hmin, fmin_f(Hnew, Hmax));
        if (RejectLastH) Hnew = fmin_f(Hnew, H);
        RejectLastH = 0; RejectMoreH = 0;
        H = Hnew;
        break;
      } else {
        if (RejectMoreH) Hnew = H * FacRej;
        RejectMoreH = RejectLastH;
        RejectLastH = 1;
        H = Hnew;
        if (st[ST_NACC] >= 1) st[ST_NREJ]++;
      }
    }
  }
  *T_out = T; *Hexit_out = Hexit;
  return 1;
}

size_t kpp_work_doubles(const kpp_mech *m) {
  size_t ab = (size_t)(m->nreact > m->nb ? m->nreact : m->nb);
  return (size_t)m->nvar * 9 + (size_t)m->nnz * 2 + ab;
}

/* INTEGRATE_x(TIN,TOUT) on one cell.  var: in/out.  stats: 8 x int32 = COMMON /Statistics/ after the call
 * (Nfun,Njac,Nstp,Nacc,Nrej,Ndec,Nsol,Nsng).  texit/hexit: what INTEGRATE_x stores into TIN and STEPMIN. */
int kpp_integrate(const kpp_mech *m, double *var, const double *fix, const double *rconst, double tin, double tout,
                  int32_t *stats, double *texit, double *hexit, double *work) {
  ros_work w;
  const int n = m->nvar;
  double *p = work;
  w.Ynew = p; p += n; w.Fcn0 = p; p += n; w.Fcn = p; p += n; w.K = p; p += 3 * n; w.dFdT = p; p += n;
  w.Yerr = p; p += n; w.W = p; p += n; w.Jac0 = p; p += m->nnz; w.Ghimj = p; p += m->nnz; w.AB = p;
  for (int i = 0; i < 8; i++) stats[i] = 0;
  return ros_integrator(m, var, fix, rconst, tin, tout, texit, hexit, stats, &w);
}

/* batch over cells, cell-major (AoS) arrays: var[ncell][nvar], fix[ncell][nfix], rconst[ncell][nreact] */
void kpp_integrate_batch(const kpp_mech *m, int ncell, double *var, const double *fix, const double *rconst, double tin,
                         double tout, int32_t *ierr, int32_t *stats /* [ncell][8] */) {
  double *work = (double *)malloc(sizeof(double) * kpp_work_doubles(m));
  for (int c = 0; c < ncell; c++) {
    double te, he;
    ierr[c] = kpp_integrate(m, var + (size_t)c * m->nvar, fix + (size_t)c * m->nfix, rconst + (size_t)c * m->nreact, tin,
                            tout, stats + (size_t)c * 8, &te, &he, work);
  }
  free(work);
}

/* The same with a first step size per cell (hstart[c] <= 0: INTEGRATE_x's 1e-3) — the checker's side of the kernel's OPT-IN
 * Hstart-reuse mode (include/mistra_chem.h: mistra_chem_integrate_device_hstart; NOT the reference's behaviour, gas.f:743). */
void kpp_integrate_batch_hstart(const kpp_mech *m, int ncell, double *var, const double *fix, const double *rconst, double tin,
                                double tout, const double *hstart, int32_t *ierr, int32_t *stats /* [ncell][8] */) {
  double *work = (double *)malloc(sizeof(double) * kpp_work_doubles(m));
  const double saved = opt_hstart;
  for (int c = 0; c < ncell; c++) {
    double te, he;
    opt_hstart = hstart[c] > 0.0 ? hstart[c] : 0.0;
    ierr[c] = kpp_integrate(m, var + (size_t)c * m->nvar, fix + (size_t)c * m->nfix, rconst + (size_t)c * m->nreact, tin,
                            tout, stats + (size_t)c * 8, &te, &he, work);
  }
  opt_hstart = saved;
  free(work);
}
