/* TEST INFRASTRUCTURE — wall-clock time of liq_parm (kpp.f90:516-657) and of each routine it calls, inside the RUNNING reference model
 * (oracle/build_ref.sh `model` links oracle/_ref/mistra_time with -Wl,--wrap=<routine>_ for every name below and nothing else wrapped).
 * No reference source is modified.  Every wrapper forwards six pointer arguments (the routines take 0-4 by reference, none on the stack),
 * clocks the real routine with CLOCK_MONOTONIC and adds to a per-routine sum; at exit the table goes to MISTRA_TIME_FILE:
 *   routine  calls  total_ms  mean_us  max_us
 * The figures are the CPU side of the f3 rows of INTEGRATION.md §4b, taken on this container's host cores. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define ROUTINES(X) X(liq_parm) X(gasdrydep) X(cw_rc) X(v_mean_a) X(henry_a) X(st_coeff_a) X(equil_co_a) X(fast_k_mt_a) X(v_mean_t) X(henry_t) \
  X(st_coeff_t) X(equil_co_t) X(fast_k_mt_t) X(dry_cw_rc) X(dry_rates_g) X(dry_rates_a) X(dry_rates_t) X(activ) X(pitzer) X(kpp_driver)

enum {
#define X(n) ID_##n,
  ROUTINES(X)
#undef X
  NROUTINES
};
static const char* const names[NROUTINES] = {
#define X(n) #n,
    ROUTINES(X)
#undef X
};
static double total_us[NROUTINES], max_us[NROUTINES];
static long calls[NROUTINES];
static int registered;

static void report(void) {
  const char* path = getenv("MISTRA_TIME_FILE");
  FILE* f = path && *path ? fopen(path, "w") : stderr;
  if (!f) return;
  fprintf(f, "%-14s %8s %12s %12s %12s\n", "routine", "calls", "total_ms", "mean_us", "max_us");
  for (int i = 0; i < NROUTINES; ++i)
    if (calls[i]) fprintf(f, "%-14s %8ld %12.3f %12.2f %12.2f\n", names[i], calls[i], total_us[i] * 1e-3, total_us[i] / calls[i], max_us[i]);
  if (f != stderr) fclose(f);
}

static inline double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

#define X(n)                                                                                  \
  void __real_##n##_(void*, void*, void*, void*, void*, void*);                               \
  void __wrap_##n##_(void* a, void* b, void* c, void* d, void* e, void* f) {                  \
    if (!registered) { registered = 1; atexit(report); }                                      \
    const double t0 = now_us();                                                               \
    __real_##n##_(a, b, c, d, e, f);                                                          \
    const double dt = now_us() - t0;                                                          \
    total_us[ID_##n] += dt; calls[ID_##n]++;                                                  \
    if (dt > max_us[ID_##n]) max_us[ID_##n] = dt;                                             \
  }
ROUTINES(X)
#undef X
