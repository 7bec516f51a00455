/* TEST INFRASTRUCTURE — fixture capture of Update_RCONST_x calls of the running reference model (oracle/build_ref.sh `model`).
 *
 * Linked with -Wl,--wrap=update_rconst_{g,a,t}_ : every call x_drive makes to Update_RCONST_x() (gas.f:172 | aer.f:216 |
 * tot.f:603) lands here.  Before the real routine runs, the layer's inputs are packed by the PRODUCT's own Fortran routine
 * MISTRA_RATES_ENV_x (shim/mistra_kpp_rates.f90, generated from mistra_amd/mech/<mech>.rates_env.json) — the routine a
 * maintainer would call in the model — and after it the reference's RCONST of COMMON /GDATA_x/ is recorded next to them,
 * together with C = VAR | FIX of that layer.  A device evaluation of the recorded env that reproduces the recorded RCONST thus
 * checks table, evaluator AND the Fortran packing against the reference in the model's own state.  No reference source is
 * modified.
 *
 *   MISTRA_CAPTURE_RATES_FILE      output file; unset = pass-through only
 *   MISTRA_CAPTURE_RATES_SKIP_x / _EVERY_x / _MAX_x     first calls of mechanism x to skip (0), keep every n-th (1), at most (32)
 *
 * record: int32 {magic 'RATE', mech, nenv, nspec, nreact, callno}, then doubles env[nenv], c[nspec], rconst[nreact] */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#define DECL(sfx, NVAR, NFIX, NREACT)                                                               \
  extern struct { double c[NVAR + NFIX]; double rconst[NREACT]; double time, dt;                   \
                  double atol[NVAR], rtol[NVAR]; double stepmin, stepmax; } gdata_##sfx##_;        \
  void __real_update_rconst_##sfx##_(void);                                                        \
  void mistra_rates_env_##sfx(double *env);
DECL(g, 102, 3, 331)
DECL(a, 257, 5, 979)
DECL(t, 417, 7, 1627)

static FILE *fp;
static int inited;
static long ncall[3], nrec[3], skip[3], every[3] = {1, 1, 1}, maxrec[3] = {32, 32, 32};

static long envl(const char *base, char sfx, long dflt) {
  char name[64];
  snprintf(name, sizeof name, "%s_%c", base, sfx);
  const char *v = getenv(name);
  return v ? atol(v) : dflt;
}
static void done(void) { if (fp) fclose(fp); }
static void init(void) {
  inited = 1;
  const char *f = getenv("MISTRA_CAPTURE_RATES_FILE");
  if (f) fp = fopen(f, "wb");
  const char sfx[3] = {'g', 'a', 't'};
  for (int m = 0; m < 3; m++) {
    skip[m] = envl("MISTRA_CAPTURE_RATES_SKIP", sfx[m], 0);
    every[m] = envl("MISTRA_CAPTURE_RATES_EVERY", sfx[m], 1);
    maxrec[m] = envl("MISTRA_CAPTURE_RATES_MAX", sfx[m], 32);
    if (every[m] < 1) every[m] = 1;
  }
  atexit(done);
}
static int want(int m) {
  long n = ncall[m]++;
  if (!fp || nrec[m] >= maxrec[m] || n < skip[m]) return 0;
  return ((n - skip[m]) % every[m]) == 0;
}

/* the packed inputs of the last Update_RCONST_x call of a mechanism (oracle/capture_drive_wrap.f90 records them with the driver call) */
static double last_env[3][544];
void capture_last_env(int mech, double *out) { memcpy(out, last_env[mech], sizeof(double) * (size_t)(mech == 0 ? 74 : mech == 1 ? 330 : 544)); }

#define DEF_WRAP(sfx, M, NVAR, NFIX, NREACT, NENV)                                                 \
  void __wrap_update_rconst_##sfx##_(void) {                                                       \
    if (!inited) init();                                                                            \
    double *env = last_env[M];                                                                      \
    const int keep = want(M);                                                                       \
    mistra_rates_env_##sfx(env);                                                                    \
    __real_update_rconst_##sfx##_();                                                                \
    if (keep) {                                                                                     \
      int32_t h[6] = {0x52415445, M, NENV, NVAR + NFIX, NREACT, (int32_t)(ncall[M] - 1)};          \
      fwrite(h, sizeof h, 1, fp);                                                                   \
      fwrite(env, 8, NENV, fp);                                                                     \
      fwrite(gdata_##sfx##_.c, 8, NVAR + NFIX, fp);                                                 \
      fwrite(gdata_##sfx##_.rconst, 8, NREACT, fp);                                                 \
      nrec[M]++;                                                                                    \
    }                                                                                               \
  }
DEF_WRAP(g, 0, 102, 3, 331, 74)
DEF_WRAP(a, 1, 257, 5, 979, 330)
DEF_WRAP(t, 2, 417, 7, 1627, 544)
