/* TEST INFRASTRUCTURE — stands in for libmistra_chem.so in the CPU validation build of the two-pass kpp_driver patch
 * (oracle/build_two_pass.sh): the entry points the Fortran shim calls, implemented with the REFERENCE's own integrator
 * (INTEGRATE_x of gas.f / aer.f / tot.f, reached as captured_integrate_x_ -> __real_integrate_x_).  It exists so that the
 * patched model — the reference's kpp_driver with the six added lines of shim/kpp_two_pass.patch, the unmodified shim and
 * batch module of shim/ — can be RUN in this container, where there is no GPU, and compared call by call with the unpatched
 * model: identical records prove the plumbing (layer order, pack/unpack halves, budgets), not the kernel.  Never part of
 * the product; the product library has no CPU path. */
#include <stdint.h>
#include <string.h>

#define DECL(sfx, NVAR, NFIX, NREACT)                                                              \
  extern struct { double c[NVAR + NFIX]; double rconst[NREACT]; double time, dt;                  \
                  double atol[NVAR], rtol[NVAR]; double stepmin, stepmax; } gdata_##sfx##_;       \
  void captured_integrate_##sfx##_(double *tin, double *tout);
DECL(g, 102, 3, 331)
DECL(a, 257, 5, 979)
DECL(t, 417, 7, 1627)
extern struct { int32_t v[8]; } statistics_;

static const int kDims[3][3] = {{102, 3, 331}, {257, 5, 979}, {417, 7, 1627}};

static void call_one(int mech, double *tin, double *tout) {
  if (mech == 0) captured_integrate_g_(tin, tout);
  else if (mech == 1) captured_integrate_a_(tin, tout);
  else captured_integrate_t_(tin, tout);
}
static double *gdata_c(int mech) { return mech == 0 ? gdata_g_.c : mech == 1 ? gdata_a_.c : gdata_t_.c; }
static double *gdata_rconst(int mech) { return mech == 0 ? gdata_g_.rconst : mech == 1 ? gdata_a_.rconst : gdata_t_.rconst; }
static double *gdata_stepmin(int mech) { return mech == 0 ? &gdata_g_.stepmin : mech == 1 ? &gdata_a_.stepmin : &gdata_t_.stepmin; }

const char *mistra_chem_last_error(void) { return ""; }
int mistra_chem_init_devices(int n, const int *ids) { (void)n; (void)ids; return 0; }
int mistra_chem_integrate_env_ex(int mech, int ncell, const double *v, const double *f, const double *e, double t0, double t1, double *o,
                                 int32_t *ierr, int32_t *stats, double *t_h) {      /* not used by the two-pass validation (RCONST path) */
  (void)mech; (void)ncell; (void)v; (void)f; (void)e; (void)t0; (void)t1; (void)o; (void)ierr; (void)stats; (void)t_h;
  return 1;
}
int mistra_chem_singular_rows(int mech, int cell, int32_t *rows8) { (void)mech; (void)cell; (void)rows8; return 0; }   /* never asked: nsng = 0 below */

int mistra_chem_integrate_ex(int mech, int ncell, const double *var_in, const double *fix, const double *rconst, double tin,
                             double tout, double *var_out, int32_t *ierr, int32_t *stats, double *t_h) {
  const int nv = kDims[mech][0], nf = kDims[mech][1], nr = kDims[mech][2];
  double save_c[424], save_r[1627];                  /* the caller's COMMON block is borrowed for every cell and put back */
  memcpy(save_c, gdata_c(mech), sizeof(double) * (size_t)(nv + nf));
  memcpy(save_r, gdata_rconst(mech), sizeof(double) * (size_t)nr);
  for (int c = 0; c < ncell; c++) {
    memcpy(gdata_c(mech), var_in + (size_t)c * nv, sizeof(double) * (size_t)nv);
    memcpy(gdata_c(mech) + nv, fix + (size_t)c * nf, sizeof(double) * (size_t)nf);
    memcpy(gdata_rconst(mech), rconst + (size_t)c * nr, sizeof(double) * (size_t)nr);
    double t0 = tin, t1 = tout;
    call_one(mech, &t0, &t1);
    memcpy(var_out + (size_t)c * nv, gdata_c(mech), sizeof(double) * (size_t)nv);
    if (ierr) ierr[c] = 1;                           /* the reference prints its own messages; the code is not returned */
    if (stats) memcpy(stats + (size_t)c * 8, statistics_.v, sizeof statistics_.v);
    if (t_h) { t_h[3 * c] = t0; t_h[3 * c + 1] = *gdata_stepmin(mech); t_h[3 * c + 2] = *gdata_stepmin(mech); }
  }
  memcpy(gdata_c(mech), save_c, sizeof(double) * (size_t)(nv + nf));
  memcpy(gdata_rconst(mech), save_r, sizeof(double) * (size_t)nr);
  return 0;
}

int mistra_chem_integrate_common_status(int mech, void *gdata, double *tin, double *tout, int32_t *ierr, double *t_err,
                                        double *h_err, int32_t *nsng) {
  (void)gdata;                                       /* the COMMON block itself: the reference works on it in place */
  call_one(mech, tin, tout);
  if (ierr) *ierr = 1;
  if (t_err) *t_err = *tin;
  if (h_err) *h_err = *gdata_stepmin(mech);
  if (nsng) *nsng = 0;
  return 0;
}
