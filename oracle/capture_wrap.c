/* TEST INFRASTRUCTURE — fixture capture for the reference model (used only by oracle/build_ref.sh `model`).
 *
 * Linked into the full reference model with -Wl,--wrap=integrate_{g,a,t}_ : every call the model makes to
 * INTEGRATE_x(TIN,TOUT) (gas.f:710 | aer.f:1408 | tot.f:2812, called from x_drive at gas.f:173 | aer.f:217 | tot.f:604)
 * lands here, the COMMON block /GDATA_x/ (gas_Global.h:29-58 and siblings) is dumped before and after the real routine,
 * plus COMMON /Statistics/ (gas.f:913-915).  No reference source is modified.
 *
 * Environment:
 *   MISTRA_CAPTURE_FILE   output file (binary records, see write_rec); unset = pass-through only
 *   MISTRA_CAPTURE_SKIP_x first calls of mechanism x (g|a|t) to skip      (default 0)
 *   MISTRA_CAPTURE_EVERY_x keep every n-th call of mechanism x after that (default 1)
 *   MISTRA_CAPTURE_MAX_x  stop after this many records of mechanism x     (default 64)
 *   MISTRA_CAPTURE_SEQ_FROM / MISTRA_CAPTURE_SEQ_TO   whole column steps: keep EVERY call, whatever its mechanism, whose
 *                         position in the model's sequence of INTEGRATE_x calls is in [FROM, TO); the record's call
 *                         number is then that global position (kpp_driver calls one of the three per layer and 10-s
 *                         step, kpp.f90:4310-4470, so a column step is a run of consecutive positions)
 *   MISTRA_RESET_DUMMIES=1  before every real call, zero the KPP dummy product species of the mechanism (DUMM1, and DUMM2 in
 *                         aer/tot: gas_Parameters.h:60, aer_Parameters.h:60-63).  x_drive never initialises them, so in the
 *                         reference they carry whatever the PREVIOUS layer's integration left in COMMON /GDATA_x/ — a
 *                         layer-to-layer coupling through the error norm only, which no batched evaluation can reproduce.
 *                         tests/test_two_pass.py switches it off in both models to compare them bit for bit.
 * A one-line call/step census per mechanism is printed at exit.
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#define DECL_GDATA(sfx, NVAR, NFIX, NREACT)                                                         \
  extern struct { double c[NVAR + NFIX]; double rconst[NREACT]; double time, dt;                   \
                  double atol[NVAR], rtol[NVAR]; double stepmin, stepmax; } gdata_##sfx##_;        \
  void __real_integrate_##sfx##_(double *tin, double *tout);

DECL_GDATA(g, 102, 3, 331)
DECL_GDATA(a, 257, 5, 979)
DECL_GDATA(t, 417, 7, 1627)

extern struct { int32_t nfun, njac, nstp, nacc, nrej, ndec, nsol, nsng; } statistics_;

static FILE *fp;
static long ncall[3], nrec[3], nstep_tot[3];
static long skip[3], every[3] = {1, 1, 1}, maxrec[3] = {64, 64, 64};
static int inited, reset_dummies;
static long seq, seq_from = -1, seq_to = -1;      /* global call position and the window kept (SEQ mode) */
static long this_seq;

static void census(void) {
  static const char *nm[3] = {"gas", "aer", "tot"};
  for (int m = 0; m < 3; m++)
    fprintf(stderr, "[capture] %s: %ld calls, %ld steps, %ld records\n", nm[m], ncall[m], nstep_tot[m], nrec[m]);
  if (fp) fclose(fp);
}

static long envl(const char *base, char sfx, long dflt) {
  char name[64];
  snprintf(name, sizeof name, "%s_%c", base, sfx);
  const char *v = getenv(name);
  return v ? atol(v) : dflt;
}

static void init(void) {
  inited = 1;
  const char *f = getenv("MISTRA_CAPTURE_FILE");
  if (f) fp = fopen(f, "wb");
  const char sfx[3] = {'g', 'a', 't'};
  for (int m = 0; m < 3; m++) {
    skip[m] = envl("MISTRA_CAPTURE_SKIP", sfx[m], 0);
    every[m] = envl("MISTRA_CAPTURE_EVERY", sfx[m], 1);
    maxrec[m] = envl("MISTRA_CAPTURE_MAX", sfx[m], 64);
    if (every[m] < 1) every[m] = 1;
  }
  reset_dummies = getenv("MISTRA_RESET_DUMMIES") != NULL;
  if (getenv("MISTRA_CAPTURE_SEQ_FROM")) seq_from = atol(getenv("MISTRA_CAPTURE_SEQ_FROM"));
  if (getenv("MISTRA_CAPTURE_SEQ_TO")) seq_to = atol(getenv("MISTRA_CAPTURE_SEQ_TO"));
  atexit(census);
}

static int want(int m) {
  long n = ncall[m]++;
  this_seq = seq++;
  if (seq_from >= 0) return fp && this_seq >= seq_from && this_seq < seq_to;
  if (!fp || nrec[m] >= maxrec[m] || n < skip[m]) return 0;
  return ((n - skip[m]) % every[m]) == 0;
}

/* record: int32 {magic, mech, nvar, nfix, nreact, callno, stats[8]}  then doubles
 *         {tin, tout, c_in[nvar+nfix], rconst[nreact], var_out[nvar], tin_out, stepmin_out}                       */
static void write_rec(int m, int nvar, int nfix, int nreact, double tin, double tout, const double *c_in,
                      const double *rconst, const double *var_out, double tin_out, double stepmin_out) {
  int32_t h[6] = {0x4d495354, m, nvar, nfix, nreact, (int32_t)(seq_from >= 0 ? this_seq : ncall[m] - 1)};
  fwrite(h, sizeof h, 1, fp);
  fwrite(&statistics_, sizeof statistics_, 1, fp);
  fwrite(&tin, 8, 1, fp);
  fwrite(&tout, 8, 1, fp);
  fwrite(c_in, 8, nvar + nfix, fp);
  fwrite(rconst, 8, nreact, fp);
  fwrite(var_out, 8, nvar, fp);
  fwrite(&tin_out, 8, 1, fp);
  fwrite(&stepmin_out, 8, 1, fp);
  nrec[m]++;
}

/* C = VAR | FIX as the last INTEGRATE_x call of a mechanism received it (oracle/capture_drive_wrap.f90 asks for it) */
static double last_c_in[3][424];
void capture_last_c_in(int mech, double *out) { memcpy(out, last_c_in[mech], sizeof(double) * (size_t)(mech == 0 ? 105 : mech == 1 ? 262 : 424)); }

#define DEF_WRAP(sfx, M, NVAR, NFIX, NREACT)                                                        \
  void __wrap_integrate_##sfx##_(double *tin, double *tout) {                                      \
    if (!inited) init();                                                                            \
    int keep = want(M);                                                                             \
    if (reset_dummies) { gdata_##sfx##_.c[3] = 0.0; if (M > 0) gdata_##sfx##_.c[4] = 0.0; }          \
    double t0 = *tin, t1 = *tout;                                                                   \
    static double c_in[NVAR + NFIX];                                                                \
    memcpy(last_c_in[M], gdata_##sfx##_.c, sizeof c_in);                                            \
    if (keep) memcpy(c_in, gdata_##sfx##_.c, sizeof c_in);                                          \
    __real_integrate_##sfx##_(tin, tout);                                                           \
    nstep_tot[M] += statistics_.nstp;                                                               \
    if (keep)                                                                                       \
      write_rec(M, NVAR, NFIX, NREACT, t0, t1, c_in, gdata_##sfx##_.rconst, gdata_##sfx##_.c, *tin, \
                gdata_##sfx##_.stepmin);                                                            \
  }

DEF_WRAP(g, 0, 102, 3, 331)
DEF_WRAP(a, 1, 257, 5, 979)
DEF_WRAP(t, 2, 417, 7, 1627)
