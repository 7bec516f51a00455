// The hand-over halves of the reference's per-layer drivers on the device, one workgroup per layer ("cell"):
//   pack     what x_drive does before Update_RCONST_x: C <- s1 / s3 through the species maps (gas.f:132-138), FIX from air, h2o and
//            the liquid water contents (gas.f:146-148 | tot.f:221-248), the liquid-phase species from sl1 / sion1 (aer_mk.dat,
//            tot.f:252-599), after clamping the layer's sl1 / sion1 to >= 0 in place where the driver does (aer.f, tot.f:226-227)
//   budgets  bud_x: bg(1,i) = RCONST(i) * reactants (the rate product A(i) of Fun_x, bud_g.f:67+), bg(2,i) += dt*bg(1,i);
//            bud_s_x: the selected sulphur / DMS rates bgs(1,slot) and their accumulation (bud_s_g.f:71-98)
//   unpack   the hand-over after the integration: s1 / s3 through the inverse maps (gas.f:191-197), sl1 / sion1 (aer_km.dat, tot.f:619-982)
//   env      the concentrations Update_RCONST_x reads (C(ind_Hplz) ...) copied from the packed C into the rate evaluator's input
// Integer / index work and plain products: every result is bit-identical to the reference's (built with -ffp-contract=off; the
// float32 literals 0.21, 0.79, 55.55 spelled out).  Memory-bound: a layer is ~10 KB in, ~4 KB out.
#include "pack.hpp"

namespace mistra {

namespace {

constexpr int kNT = 256;

// MAX(0.d0, x) as flang compiles it (x86 maxsd with the constant first: "0 > x ? 0 : x"): -0.0 stays -0.0 and a NaN stays a NaN —
// checked with the box's flang at -O2 on the scalar and on the whole-array form (the drivers use both, tot.f:226-227, gas.f:151-156)
__device__ __forceinline__ double fmax0(double v) { return 0.0 > v ? 0.0 : v; }

__global__ __launch_bounds__(kNT) void pack_kernel(const PackDev P, int ncell, const double* __restrict__ s1, const double* __restrict__ s3,
                                                   double* __restrict__ sl1, double* __restrict__ sion1, const double* __restrict__ scal,
                                                   double* __restrict__ var, double* __restrict__ fix) {
  const int cell = blockIdx.x, t = threadIdx.x;
  if (cell >= ncell) return;
  const int nl = P.j2 * P.nkc, ni = P.j6 * P.nkc;
  double* L = sl1 + (size_t)cell * nl;
  double* I = sion1 + (size_t)cell * ni;
  double* V = var + (size_t)cell * P.nvar;
  double* F = fix + (size_t)cell * P.nfix;
  auto put = [&](int c0, double v) {
    if (c0 < P.nvar) V[c0] = v;
    else F[c0 - P.nvar] = v;
  };
  if (P.preclamp) {      // sl1(:,:,k)=max(0.d0,sl1(:,:,k)); sion1 likewise (tot.f:226-227): the model's arrays themselves
    for (int i = t; i < nl; i += kNT) L[i] = fmax0(L[i]);
    for (int i = t; i < ni; i += kNT) I[i] = fmax0(I[i]);
    __syncthreads();
  }
  // C(gas_m2k(1,j)) = s1(gas_m2k(2,j),k); C(rad_m2k(1,j)) = s3(rad_m2k(2,j),k)
  for (int j = t; j < P.j1; j += kNT) put(P.gas_m2k[2 * j] - 1, s1[(size_t)cell * P.j1 + (P.gas_m2k[2 * j + 1] - 1)]);
  for (int j = t; j < P.j5; j += kNT) put(P.rad_m2k[2 * j] - 1, s3[(size_t)cell * P.j5 + (P.rad_m2k[2 * j + 1] - 1)]);
  // FIX: 0.21*air, h2o, 0.79*air, 55.55/cvv (default-REAL literals in a REAL*8 expression)
  const double* sc = scal + (size_t)cell * 6;      // air, h2o, cvv1..4
  for (int i = t; i < P.n_fix; i += kNT) {
    const int c0 = P.fix[3 * i], kind = P.fix[3 * i + 1], kc = P.fix[3 * i + 2];
    double v;
    if (kind == 0) v = (double)0.21f * sc[0];
    else if (kind == 1) v = (double)0.79f * sc[0];
    else if (kind == 2) v = sc[1];
    else v = sc[2 + kc] > 0.0 ? (double)55.55f / sc[2 + kc] : 0.0;
    put(c0, v);
  }
  for (int i = t; i < P.n_pack; i += kNT) {
    const int c0 = P.pack[4 * i], arr = P.pack[4 * i + 1], at = P.pack[4 * i + 2], clamp = P.pack[4 * i + 3];
    double v = arr == 0 ? L[at] : I[at];
    if (clamp) v = fmax0(v);
    put(c0, v);
  }
}

__global__ __launch_bounds__(kNT) void unpack_kernel(const PackDev P, int ncell, const double* __restrict__ var, double* __restrict__ s1,
                                                     double* __restrict__ s3, double* __restrict__ sl1, double* __restrict__ sion1) {
  const int cell = blockIdx.x, t = threadIdx.x;
  if (cell >= ncell) return;
  const double* V = var + (size_t)cell * P.nvar;      // (only VAR entries are handed over: the maps and lists hold no FIX index)
  for (int j = t; j < P.j1; j += kNT) s1[(size_t)cell * P.j1 + j] = V[P.gas_k2m[j] - 1];
  for (int j = t; j < P.j5; j += kNT) s3[(size_t)cell * P.j5 + j] = V[P.rad_k2m[j] - 1];
  double* L = sl1 + (size_t)cell * P.j2 * P.nkc;
  double* I = sion1 + (size_t)cell * P.j6 * P.nkc;
  for (int i = t; i < P.n_unpack; i += kNT) {
    const int arr = P.unpack[4 * i], at = P.unpack[4 * i + 1], c0 = P.unpack[4 * i + 2], clamp = P.unpack[4 * i + 3];
    double v = V[c0];
    if (clamp) v = fmax0(v);
    (arr == 0 ? L : I)[at] = v;
  }
}

__global__ __launch_bounds__(kNT) void budgets_kernel(const PackDev P, int ncell, const double* __restrict__ var, const double* __restrict__ fix,
                                                      const double* __restrict__ rconst, double dt, double* __restrict__ bg, double* __restrict__ bgs) {
  const int cell = blockIdx.x, t = threadIdx.x;
  if (cell >= ncell) return;
  const double* V = var + (size_t)cell * P.nvar;
  const double* F = fix + (size_t)cell * P.nfix;
  const double* K = rconst + (size_t)cell * P.nreact;
  auto X = [&](int c0) { return c0 < P.nvar ? V[c0] : c0 < P.nvar + P.nfix ? F[c0 - P.nvar] : P.consts[c0 - P.nvar - P.nfix]; };
  if (bg) {      // bud_x: bg(1,i,kl) = RCONST(i)*reactants, left to right; bg(2,i,kl) += dtg*bg(1,i,kl)
    double* B = bg + (size_t)cell * 2 * P.nreact;      // (2, nreact) as the Fortran holds it: [i][0] instantaneous, [i][1] cumulative
    for (int r = t; r < P.nreact; r += kNT) {
      double p = K[r];
      for (int q = P.a_ptr[r]; q < P.a_ptr[r + 1]; q++) p = p * X(P.a_fac[q]);
      B[2 * r] = p;
      B[2 * r + 1] = B[2 * r + 1] + dt * p;
    }
  }
  if (bgs) {     // bud_s_x: bgs(1,slot,k) = +-RCONST(r)*C(..)*.. +- ..; the listed slot ranges accumulate
    double* S = bgs + (size_t)cell * 2 * kBudSlots;
    for (int s = t; s < P.n_slots; s += kNT) {
      double acc = 0.0;
      for (int q = P.slot_first[s]; q < P.slot_first[s + 1]; q++) {
        const int sign = P.terms[3 * q], r = P.terms[3 * q + 1], w0 = P.terms[3 * q + 2], w1 = q + 1 < P.n_terms ? P.terms[3 * (q + 1) + 2] : P.n_words;
        double p = K[r];
        for (int w = w0; w < w1; w++) p = p * V[P.term_words[w]];      // (C indices of the budget terms are all variable species; the host checks)
        acc = q == P.slot_first[s] ? (sign < 0 ? -p : p) : (sign < 0 ? acc - p : acc + p);
      }
      S[2 * (P.slot_id[s] - 1)] = acc;
    }
    __syncthreads();
    for (int a = 0; a < P.n_acc; a++)
      for (int i = P.acc[2 * a] - 1 + t; i < P.acc[2 * a + 1]; i += kNT) S[2 * i + 1] = S[2 * i + 1] + dt * S[2 * i];
  }
}

__global__ __launch_bounds__(kNT) void env_from_c_kernel(const PackDev P, int ncell, int nenv, const double* __restrict__ var,
                                                         const double* __restrict__ fix, double* __restrict__ env) {
  const int i = blockIdx.x * kNT + threadIdx.x;
  if (i >= ncell * P.n_envc) return;
  const int cell = i / P.n_envc, e = i % P.n_envc, slot = P.envc[2 * e], c0 = P.envc[2 * e + 1];
  env[(size_t)cell * nenv + slot] = c0 < P.nvar ? var[(size_t)cell * P.nvar + c0] : fix[(size_t)cell * P.nfix + (c0 - P.nvar)];
}

// vterm(a, t, p) (str.f90:2793-2863): terminal velocity of a droplet of radius a [m] — Stokes with the Cunningham correction up to 10 um,
// Beard's polynomial in the logarithm of the Best number above (Pruppacher & Klett ch. 10).  The PARAMETER constants are formed as the
// compiler folds them (left to right in double precision); one rounding per operation; a**3 = (a*a)*a as flang expands it; log / exp are
// the device library's (regime 2 only: last-place differences against the host libm).
__device__ double vterm_dev(double a, double t, double p) {
  constexpr double g = 9.80665, r0 = 8.3144743 / 28.96546e-3, rhow = 1000.0;      // constants.f90:48,61,65,73,79
  constexpr double b0 = -.318657e+1, b1 = .992696e+0, b2 = -.153193e-2, b3 = -.987059e-3, b4 = -.578878e-3, b5 = +.855176e-4, b6 = -.327815e-5;
  constexpr double c1 = 2.0 * g / 9.0, c2 = 1.26, P0 = 101325.0, T0 = 293.15, lambda0 = 6.6e-8, c3 = c2 * lambda0 * P0 / T0, c4 = 32.0 * g / 3.0;
  const double rho_a = p / (r0 * t);
  const double eta = 3.7957e-06 + 4.9e-08 * t;
  if (a <= 1.0e-5) return c1 * a * a * (rhow - rho_a) / eta * (1.0 + c3 * t / (a * p));
  const double best = c4 * ((a * a) * a) * (rhow - rho_a) * rho_a / (eta * eta);
  const double x = log(best);
  double y = b6 * x + b5;
  y = y * x + b4;
  y = y * x + b3;
  y = y * x + b2;
  y = y * x + b1;
  y = y * x + b0;
  return eta * exp(y) / (2.0 * rho_a * a);
}

// fast_k_mt_a / fast_k_mt_t (kpp.f90:2683-2947 | 2421-2676): the mass-transfer coefficient of every exchanged species l into every
// active chemical bin kc of a layer, integrated over the bin's part of the 2-D particle spectrum ff(jt,ia):
//     xk1 = sum_ia sum_jt  vmean / (r/freep + 4/(3 alpha)) * r*r * ff(jt,ia) * 1e6       r = rq(jt,ia)*1e-6 [m]
//     xkmt(lex(l),kc) = 4 pi / (3 cw(kc)) * xk1                                          where cm(kc) > 0 and cw(kc) > 0
// and, "whatever LWC" (kpp.f90:2421-2432: also where cm(kc) = 0), the LWC-weighted sedimentation velocity of the bin that SR sedl reads
// (str.f90:2704, 2751):
//     xx1 = sum_ia sum_jt  r*r*r * vterm(r, t, p) * ff(jt,ia) * 1e6                       (the routine's l = 1 pass)
//     vt(kc) = 4 pi / (3 cw(kc)) * xx1                                                    where cw(kc) > 0
// One workgroup of sixteen waves per (layer, bin), a pipeline of three stages with one barrier per chunk of 56 grid cells.  A sum's ORDER is its value (ia
// outer, jt inner; one rounding per operation) but its terms are not ordered:
//   wave 15, lane = CELL: what does not depend on the species — the cell's radius, its ff, the quotient r/freep and the vt term (Beard's polynomial,
//            the expensive one: once per cell here instead of once per cell and wave) — for chunk i+1, into LDS;
//   waves 1..14, lane = SPECIES, four cells each: the terms of chunk i (one IEEE division per cell and species) out of the cell data, into LDS; a cell
//            outside the bin's jt range contributes +0.0, which leaves a non-negative sum as it is; lane 63 passes the vt term on;
//   wave 0, lane = species: adds the terms of chunk i-1, cell by cell in the reference's order (its reads run ahead of the add chain).
// Bit-identical to the thread per species that walked the whole bin alone (the round-3 kernel: 0.64 ms per column).
constexpr int kKmtWaves = 16, kKmtU = 4, kKmtCells = (kKmtWaves - 2) * kKmtU;      // 56 cells per chunk (<= 64: one lane of wave 15 each)
__global__ __launch_bounds__(kKmtWaves * 64) void fast_k_mt_kernel(const KmtDev K, int nlayer, const double* __restrict__ ff, const double* __restrict__ rq,
                                                                    const double* __restrict__ cw, const double* __restrict__ cm,
                                                                    const double* __restrict__ freep, const double* __restrict__ alpha,
                                                                    const double* __restrict__ vmean, double* __restrict__ xkmt,
                                                                    const double* __restrict__ tt, const double* __restrict__ pp, double* __restrict__ vt) {
  __shared__ double term[2][kKmtCells][64];      // 2 x 28 KB
  __shared__ double cell[2][4][64];              // per cell of a chunk: r [m], ff, r/freep, the vt term (0: outside the bin)
  __shared__ int inbin[2][64];
  const int layer = blockIdx.x, kc = blockIdx.y + 1, w = threadIdx.x / 64, l = threadIdx.x % 64;      // kc 1-based as in the Fortran
  if (layer >= nlayer || kc > K.nkc_l) return;                                                          // (uniform in the workgroup, like every exit below)
  const double cmk = cm[(size_t)layer * K.nkc + (kc - 1)], cwk = cw[(size_t)layer * K.nkc + (kc - 1)];
  const bool llchem = cmk > 0.0, do_vt = vt != nullptr;      // (the dry case only integrates vt: lmax = 1, kpp.f90:2560-2566)
  if (!llchem && !do_vt) return;
  const bool species = llchem && l < K.nx;
  const int sp = species ? K.lex[l] - 1 : 0;
  const double al = species ? alpha[(size_t)layer * K.nspec + sp] : 0.0, vm = species ? vmean[(size_t)layer * K.nspec + sp] : 0.0, fp = freep[layer];
  const double tk = do_vt ? tt[layer] : 0.0, pk = do_vt ? pp[layer] : 0.0;
  double x1 = 0.0;
  if (al > 0.0) x1 = 4.0 / (3.0 * al);            // 4./(3.*alpha): default-REAL literals, exact
  int ia0, ia1;                                   // summation limits (1): the aerosol-size axis, 1-based inclusive
  if (kc == 1 || kc == 3) { ia0 = K.ifeed == 2 ? 2 : 1; ia1 = K.ka; }
  else { ia0 = K.ka + 1; ia1 = K.nka; }
  const bool low_jt = kc == 1 || kc == 2;         // limits (2), the water axis: jt <= kw(ia) | jt > kw(ia)
  const int ncell = ia1 >= ia0 ? (ia1 - ia0 + 1) * K.nkt : 0, nchunk = (ncell + kKmtCells - 1) / kKmtCells;
  const double* F = ff + (size_t)layer * K.nka * K.nkt;      // ff(jt,ia,k): jt fastest
  double acc = 0.0;                               // wave 0: xk1 of species l; lane 63: xx1
  for (int i = 0; i < nchunk + 2; i++) {
    if (w == kKmtWaves - 1) {
      if (i < nchunk && l < kKmtCells) {          // chunk i: this lane's cell
        const int n = i * kKmtCells + l, ia = ia0 + n / K.nkt, jt = 1 + n % K.nkt;
        bool in = false;
        double rqq = 0.0, fv = 0.0, q = 0.0, tv = 0.0;
        if (n < ncell) {
          const int kwa = K.kw[ia - 1];
          in = low_jt ? jt <= kwa : jt > kwa;
          if (in) {
            rqq = rq[(size_t)(ia - 1) * K.nkt + (jt - 1)] * 1.0e-6;      // rqm = rq * 1.d-6
            fv = F[(size_t)(ia - 1) * K.nkt + (jt - 1)];
            q = rqq / fp;
            if (do_vt) {
              const double xvs = vterm_dev(rqq, tk, pk);
              tv = ((((rqq * rqq) * rqq) * xvs) * fv) * 1.0e6;
            }
          }
        }
        cell[i & 1][0][l] = rqq; cell[i & 1][1][l] = fv; cell[i & 1][2][l] = q; cell[i & 1][3][l] = tv;
        inbin[i & 1][l] = in ? 1 : 0;
      }
    } else if (w > 0) {
      if (i >= 1 && i <= nchunk) {                // chunk i-1: this wave's four cells
        const int b = (i - 1) & 1;
#pragma unroll
        for (int u = 0; u < kKmtU; u++) {
          const int c = (w - 1) * kKmtU + u;
          double t = 0.0;
          if (inbin[b][c]) {                      // (uniform in the wave)
            const double rqq = cell[b][0][c], fv = cell[b][1][c], q = cell[b][2][c];
            if (species) {
              const double x2 = vm / (q + x1);
              t = (((x2 * rqq) * rqq) * fv) * 1.0e6;
            }
            if (l == 63) t = cell[b][3][c];
          }
          term[b][c][l] = t;
        }
      }
    } else if (i >= 2) {                          // chunk i-2
      const double (*B)[64] = term[i & 1];
#pragma unroll 8
      for (int c = 0; c < kKmtCells; c++) acc = acc + B[c][l];
    }
    __syncthreads();
  }
  if (w != 0 || !(cwk > 0.0)) return;
  constexpr double z4pi3 = 4.0 * 3.1415926535897932 / 3.0;      // z4pi3 = 4._dp * pi / 3._dp (constants.f90:54)
  if (species) xkmt[((size_t)layer * K.nkc + (kc - 1)) * K.nspec + sp] = z4pi3 / cwk * acc;
  if (do_vt && l == 63) vt[(size_t)layer * K.nkc + (kc - 1)] = z4pi3 / cwk * acc;
}

}  // namespace

// henry_a | henry_t (kpp.f90:1914-2145 | 1676-1907): one thread per (layer, species).  A species the routine does not set is 0
// ("k_H = infinity"); a number stays a number, a temperature law is a0*exp(b0*Tfact), Tfact = 1/T - 3.3540d-3; every positive value then
// becomes the inverse dimensionless constant 1/(k_H * FCT), FCT = 0.0820577*T.  One rounding per operation as in the reference; exp is
// the device library's (last-place differences against the host libm, like the rate laws).
__global__ __launch_bounds__(256) void henry_kernel(const LiqDev L, int nlayer, const double* __restrict__ tt, double* __restrict__ henry) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (j >= L.nspec || k >= nlayer) return;
  const double T = tt[k];
  const int kind = L.h_kind[j];
  double h = 0.0;
  if (kind == 0) h = L.h_a0[j];
  else if (kind == 1) {
    const double tfact = 1.0 / T - L.henry_tref;
    h = L.h_a0[j] * exp(L.h_b0[j] * tfact);
  }
  if (h > 0.0) {
    const double fct = L.henry_fct * T;
    h = 1.0 / (h * fct);
  }
  henry[(size_t)k * L.nspec + j] = h;
}

// equil_co_a | equil_co_t (kpp.f90:3162-3363 | 2954-3155): one thread per (layer, bin, species).  A bin without liquid water
// (conv2 <= 0) is zeroed for every species; elsewhere the species the routine sets get the left-to-right product of their factors
// (number | funa(a0,b0) = a0*exp(b0*(1/T - 3.354d-3)) | conv2 | xgamma(i)) and the others keep what they hold.
__device__ double liq_product(const LiqDev& L, int first, int last, double T, double cv2, const double* __restrict__ xg) {
  double v = 0.0;
  for (int f = first; f < last; f++) {
    const int kind = L.fkind[f];
    double x;
    if (kind == 0) x = L.fa[f];
    else if (kind == 1) x = L.fa[f] * exp(L.fb[f] * (1.0 / T - L.equil_tref));
    else if (kind == 2) x = cv2;
    else x = xg[L.farg[f] - 1];
    v = f == first ? x : v * x;
  }
  return v;
}
__global__ __launch_bounds__(256) void equil_co_kernel(const LiqDev L, int nlayer, int nkc, int j6, const double* __restrict__ tt,
                                                        const double* __restrict__ conv2, const double* __restrict__ xgamma,
                                                        double* __restrict__ xkef, double* __restrict__ xkeb) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, kc = blockIdx.y, k = blockIdx.z;
  if (j >= L.nspec || kc >= L.nkc_eq || k >= nlayer) return;
  const double cv2 = conv2[(size_t)k * nkc + kc];
  const size_t at = ((size_t)k * nkc + kc) * L.nspec + j;
  if (!(cv2 > 0.0)) {
    xkef[at] = 0.0;
    xkeb[at] = 0.0;
    return;
  }
  const int e = L.e_of[j];
  if (e < 0) return;
  const double T = tt[k];
  const double* xg = xgamma + ((size_t)k * nkc + kc) * j6;
  xkef[at] = liq_product(L, L.foff[e], L.boff[e], T, cv2, xg);
  xkeb[at] = liq_product(L, L.boff[e], L.foff[e + 1], T, cv2, xg);
}

// cw_rc (kpp.f90:2152-2414) | dry_cw_rc (kpp.f90:4580-4690): one workgroup per layer.  The reference walks the particle grid — dry-aerosol classes ia
// ascending, droplet classes jt ascending inside — with three running sums per bin; every sum is a serial chain of ~1 200 additions whose order is its value.
// The TERMS do not depend on that order: all 256 threads form them for a chunk of grid rows (coalesced reads of ff and rq, the products into LDS), then
// wave kc walks bin kc's cells of the chunk in the reference's order, lanes 0..2 adding one of the three sums each out of LDS (the reads run ahead of the
// additions: nothing but the add chain is serial).  Per layer ~10 us instead of the ~1 ms of a thread that fetched and multiplied as it went.
//   bin 1: ia <= ka, jt <= kw(ia)    bin 3: ia <= ka, jt > kw(ia)    bin 2: ia > ka, jt <= kw(ia)    bin 4: ia > ka, jt > kw(ia)
//   x0 = ff*xpi*rq**3 (rq**3 as the compiler expands it: rq*rq*rq); cw += x0; rc += x0*rq; cm += ff*e(jt)
// xpi = 4._dp/3._dp*pi in cw_rc and 4./3.*pi — a single-precision 4/3 — in dry_cw_rc.  One rounding per operation (-ffp-contract=off).
constexpr int kCwRcChunk = 2048;      // grid cells per chunk: 3 x 16 KB of LDS
__global__ __launch_bounds__(256) void cw_rc_kernel(const CwRcArgs A) {
  constexpr double pi = 3.1415926535897932;                              // constants.f90:54
  constexpr double cwm = 1.0e-1, cwmd = 1.0e2;                           // kpp.f90:2195-2196
  __shared__ double term[3][kCwRcChunk];
  __shared__ double sums[4][3];
  const int nbin = A.dry ? 2 : 4;
  const int k = blockIdx.x, t = threadIdx.x, kc = t / 64, which = t % 64;
  const double xpi = A.dry ? (double)(4.0f / 3.0f) * pi : 4.0 / 3.0 * pi;
  const bool small_ia = (kc == 0 || kc == 2);
  const int ia0 = small_ia ? A.ial : A.ka + 1, ia1 = small_ia ? A.ka : A.nka;      // this wave's bin: 1-based, inclusive
  const double* ffk = A.ff + (size_t)k * A.nka * A.nkt;
  const int rows = kCwRcChunk / A.nkt;                                   // grid rows (ia) per chunk; the host checks nkt <= kCwRcChunk
  double acc = 0.0;
  for (int r0 = 1; r0 <= A.nka; r0 += rows) {                            // rows r0 .. r1 of the grid
    const int r1 = r0 + rows - 1 < A.nka ? r0 + rows - 1 : A.nka, ncell = (r1 - r0 + 1) * A.nkt;
    __syncthreads();
    for (int c = t; c < ncell; c += 256) {
      const size_t g = (size_t)(r0 - 1) * A.nkt + c;
      const double f = ffk[g], r = A.rq[g];
      const double x0 = (f * xpi) * ((r * r) * r);
      term[0][c] = x0;
      term[1][c] = x0 * r;
      if (!A.dry) term[2][c] = f * A.e[c % A.nkt];
    }
    __syncthreads();
    if (kc < nbin && which < (A.dry ? 2 : 3)) {
      const double* tm = term[which];
      const int lo = ia0 > r0 ? ia0 : r0, hi = ia1 < r1 ? ia1 : r1;
      for (int ia = lo; ia <= hi; ia++) {
        const int kwa = A.kw[ia - 1];
        const int jt0 = kc < 2 ? 1 : kwa + 1, jt1 = kc < 2 ? kwa : A.nkt;
        const double* row = tm + (ia - r0) * A.nkt - 1;
        int jt = jt0;
        for (; jt + 7 <= jt1; jt += 8) {            // eight reads in flight, then the eight additions in order
          const double v0 = row[jt], v1 = row[jt + 1], v2 = row[jt + 2], v3 = row[jt + 3], v4 = row[jt + 4], v5 = row[jt + 5], v6 = row[jt + 6], v7 = row[jt + 7];
          acc = acc + v0; acc = acc + v1; acc = acc + v2; acc = acc + v3; acc = acc + v4; acc = acc + v5; acc = acc + v6; acc = acc + v7;
        }
        for (; jt <= jt1; jt++) acc = acc + row[jt];
      }
    }
  }
  if (kc < nbin && which < 3) sums[kc][which] = acc;
  __syncthreads();
  if (t >= nbin) return;      // thread b finishes bin b
  const int b = t;
  const double cws = sums[b][0], rcs = sums[b][1], cms = A.dry ? 0.0 : sums[b][2];
  const size_t at = (size_t)k * nbin + b;
  A.rc[at] = cws > 0.0 ? (rcs / cws) * 1.0e-6 : 0.0;
  A.cw[at] = cws * 1.0e-12;
  if (A.dry) return;
  const double feu = A.feu[k];
  const double xmin = A.xcryssulf < A.xcrysss ? A.xcryssulf : A.xcrysss;      // min(xcryssulf,xcrysss)
  bool on;
  if (feu < xmin) {
    on = false;
    if (b == 0) A.below[k] = 1;
  } else {
    if (b == 0) A.below[k] = 0;
    if (b == 0) on = cws >= cwm && ((A.cloud[(size_t)k * 4 + 0] != 0 && feu >= A.xcryssulf) || feu >= A.xdelisulf);
    else if (b == 1) on = cws >= cwm && ((A.cloud[(size_t)k * 4 + 1] != 0 && feu >= A.xcrysss) || feu >= A.xdeliss);
    else on = cws >= cwmd;
  }
  A.cm[at] = on ? cms * 1.0e-3 : 0.0;
  A.conv2[at] = on ? 1.0e9 / cws : 0.0;
}
hipError_t launch_cw_rc(const CwRcArgs& A, hipStream_t stream) {
  if (A.nlayer <= 0) return hipSuccess;
  if (A.nkt > kCwRcChunk) return hipErrorInvalidValue;      // one grid row per chunk at least (capi.cpp refuses such a grid with a message)
  hipLaunchKernelGGL(cw_rc_kernel, dim3((unsigned)A.nlayer), dim3(256), 0, stream, A);
  return hipGetLastError();
}

// dry_rates_g | dry_rates_a | dry_rates_t (kpp.f90:4697-4853 | 4860-5073 | 5079-5198): mass-transfer coefficients of HNO3, N2O5, NH3, H2SO4 onto the DRY
// aerosol of bins 1 and 2, one thread per layer.  All three: xeq(HNO3) = funa(1.54d+1,8700.d0), funa(a0,b0) = a0*exp(b0*(1/t - 3.354d-3));
// x1 = 1./(rcd*(rcd/freep + 4./(3.*zgamma))) where rcd > 0, else 0 (zgamma = 0.02, 0.02, 0.05, 0.1 in both bins); xkmtd = vmean*x1.  dry_rates_g forms the
// mean molecular speeds itself, func(M) = sqrt(tt/M)*4.60138 with M = 6.3d-2, 1.08d-1, 1.7d-2, 9.8d-2, and the Henry constant of HNO3,
// henry = func3(2.5d6/xeq,8694.d0), func3(a0,b0) = a0*exp(b0*((1/tt) - 3.3557d-3)), then henry = 1./(henry*FCT), FCT = 0.0820577*tt (a default-real
// literal), for each of the four species whose entry is positive.  One rounding per operation; exp is the device library's.
__global__ __launch_bounds__(64) void dry_rates_kernel(const DryRatesArgs A) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= A.nlayer) return;
  const double zgamma[4] = {0.02, 0.02, 0.05, 0.1}, mass[4] = {6.3e-2, 1.08e-1, 1.7e-2, 9.8e-2};
  const double t = A.tt[k], freep = A.freep[k];
  const double xeq = 1.54e+1 * exp(8700.0 * (1.0 / t - 3.354e-3));
  A.xeq[k] = xeq;
  double vm[4];
  if (A.gas) {
    double h[4];
    for (int l = 0; l < 4; l++) h[l] = A.henry4[(size_t)k * 4 + l];
    h[0] = (2.5e6 / xeq) * exp(8694.0 * ((1.0 / t) - 3.3557e-3));
    const double fct = (double)0.0820577f * t;
    for (int l = 0; l < 4; l++) {
      if (h[l] > 0.0) h[l] = 1.0 / (h[l] * fct);
      A.henry4[(size_t)k * 4 + l] = h[l];
      vm[l] = sqrt(t / mass[l]) * (double)4.60138f;
    }
  } else {
    for (int l = 0; l < 4; l++) vm[l] = A.vmean4[(size_t)k * 4 + l];
  }
  for (int kc = 0; kc < 2; kc++) {
    const double rcd = A.rcd[(size_t)k * 2 + kc];
    for (int l = 0; l < 4; l++) {
      double x1 = 0.0;
      if (zgamma[l] > 0.0 && rcd > 0.0) x1 = 1.0 / (rcd * (rcd / freep + 4.0 / (3.0 * zgamma[l])));
      A.xkmtd[((size_t)k * 2 + kc) * 4 + l] = vm[l] * x1;
    }
  }
}
hipError_t launch_dry_rates(const DryRatesArgs& A, hipStream_t stream) {
  if (A.nlayer <= 0) return hipSuccess;
  hipLaunchKernelGGL(dry_rates_kernel, dim3((unsigned)((A.nlayer + 63) / 64)), dim3(64), 0, stream, A);
  return hipGetLastError();
}

// v_mean_a | v_mean_t (kpp.f90:1472-1670 | 1268-1465): the mean molecular speed sqrt(8 R T / (pi M)) as the reference writes it,
// func(a,k) = sqrt(tt(k)/a)*4.60138 (the factor a default-real literal), one thread per (layer, species); a species the routine does not
// set stays 0 (`vmean(:,:) = 0._dp`).  Quotient, square root and product each round once, as compiled Fortran does: bit-identical.
__global__ __launch_bounds__(256) void v_mean_kernel(const double* __restrict__ mass, int nspec, double coef, int nlayer,
                                                      const double* __restrict__ tt, double* __restrict__ vmean) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (j >= nspec || k >= nlayer) return;
  const double a = mass[j];
  double v = 0.0;
  if (a != 0.0) v = sqrt(tt[k] / a) * coef;
  vmean[(size_t)k * nspec + j] = v;
}
hipError_t launch_v_mean(const double* mass, int nspec, double coef, int nlayer, const double* tt, double* vmean, hipStream_t stream) {
  if (nlayer <= 0) return hipSuccess;
  hipLaunchKernelGGL(v_mean_kernel, dim3((unsigned)((nspec + 255) / 256), (unsigned)nlayer), dim3(256), 0, stream, mass, nspec, coef, nlayer, tt, vmean);
  return hipGetLastError();
}

hipError_t launch_henry(const LiqDev& L, int nlayer, const double* tt, double* henry, hipStream_t stream) {
  if (nlayer <= 0) return hipSuccess;
  hipLaunchKernelGGL(henry_kernel, dim3((unsigned)((L.nspec + 255) / 256), (unsigned)nlayer), dim3(256), 0, stream, L, nlayer, tt, henry);
  return hipGetLastError();
}
hipError_t launch_equil_co(const LiqDev& L, int nlayer, int nkc, int j6, const double* tt, const double* conv2, const double* xgamma, double* xkef,
                           double* xkeb, hipStream_t stream) {
  if (nlayer <= 0) return hipSuccess;
  if (nlayer > 65535) return hipErrorInvalidValue;      // (grid z)
  hipLaunchKernelGGL(equil_co_kernel, dim3((unsigned)((L.nspec + 255) / 256), (unsigned)L.nkc_eq, (unsigned)nlayer), dim3(256), 0, stream, L, nlayer, nkc,
                     j6, tt, conv2, xgamma, xkef, xkeb);
  return hipGetLastError();
}

hipError_t launch_fast_k_mt(const KmtDev& K, int nlayer, const double* ff, const double* rq, const double* cw, const double* cm, const double* freep,
                            const double* alpha, const double* vmean, double* xkmt, const double* tt, const double* pp, double* vt, hipStream_t stream) {
  if (nlayer <= 0) return hipSuccess;
  if (K.nx > 63 || K.nka > kKmtMaxNka) return hipErrorInvalidValue;      // (lane 63 carries the vt sum)
  hipLaunchKernelGGL(fast_k_mt_kernel, dim3((unsigned)nlayer, (unsigned)K.nkc_l), dim3(kKmtWaves * 64), 0, stream, K, nlayer, ff, rq, cw, cm, freep, alpha, vmean, xkmt, tt, pp, vt);
  return hipGetLastError();
}

hipError_t launch_pack(const PackDev& P, int ncell, const double* s1, const double* s3, double* sl1, double* sion1, const double* scal,
                       double* var, double* fix, hipStream_t stream) {
  if (ncell <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ncell), dim3(kNT), 0, stream, P, ncell, s1, s3, sl1, sion1, scal, var, fix);
  return hipGetLastError();
}
hipError_t launch_unpack(const PackDev& P, int ncell, const double* var, double* s1, double* s3, double* sl1, double* sion1, hipStream_t stream) {
  if (ncell <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)ncell), dim3(kNT), 0, stream, P, ncell, var, s1, s3, sl1, sion1);
  return hipGetLastError();
}
hipError_t launch_budgets(const PackDev& P, int ncell, const double* var, const double* fix, const double* rconst, double dt, double* bg,
                          double* bgs, hipStream_t stream) {
  if (ncell <= 0) return hipSuccess;
  hipLaunchKernelGGL(budgets_kernel, dim3((unsigned)ncell), dim3(kNT), 0, stream, P, ncell, var, fix, rconst, dt, bg, bgs);
  return hipGetLastError();
}
hipError_t launch_env_from_c(const PackDev& P, int ncell, int nenv, const double* var, const double* fix, double* env, hipStream_t stream) {
  if (ncell <= 0 || P.n_envc == 0) return hipSuccess;
  const int n = ncell * P.n_envc;
  hipLaunchKernelGGL(env_from_c_kernel, dim3((unsigned)((n + kNT - 1) / kNT)), dim3(kNT), 0, stream, P, ncell, nenv, var, fix, env);
  return hipGetLastError();
}

}  // namespace mistra
