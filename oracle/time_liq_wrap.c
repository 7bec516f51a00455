/* TEST INFRASTRUCTURE — wall-clock time of liq_parm (kpp.f90:516-657) and of each routine it calls, inside the RUNNING reference model
 * (oracle/build_ref.sh `model` links oracle/_ref/mistra_time with -Wl,--wrap=<routine>_ for every name below and nothing else wrapped).
 * No reference source is modified.  Every wrapper forwards eight pointer arguments (the routines take 0-7 by reference; reading two stack slots more than a routine was given is harmless),
 * clocks the real routine with CLOCK_MONOTONIC and adds to a per-routine sum; at exit the table goes to MISTRA_TIME_FILE:
 *   routine  calls  total_ms  mean_us  max_us
 * The figures are the CPU side of the f3 rows of INTEGRATION.md §4b, taken on this container's host cores. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define ROUTINES(X) X(liq_parm) X(gasdrydep) X(cw_rc) X(v_mean_a) X(henry_a) X(st_coeff_a) X(equil_co_a) X(fast_k_mt_a) X(v_mean_t) X(henry_t) \
  X(st_coeff_t) X(equil_co_t) X(fast_k_mt_t) X(dry_cw_rc) X(dry_rates_g) X(dry_rates_a) X(dry_rates_t) X(activ) X(pitzer) X(kpp_driver) \
  /* the GPU build (build_gpu_model.sh `time`): liq_parm calls the drop-ins of shim/mistra_kpp_model.f90 instead, kpp_driver ends in KPP_DRIVE_RUN */ \
  X(cw_rc_hip) X(v_mean_hip_a) X(henry_hip_a) X(st_coeff_hip_a) X(equil_co_hip_a) X(fast_k_mt_hip_a) X(v_mean_hip_t) X(henry_hip_t) X(st_coeff_hip_t) \
  X(equil_co_hip_t) X(fast_k_mt_hip_t) X(dry_cw_rc_hip) X(dry_rates_hip_g) X(dry_rates_hip_a) X(dry_rates_hip_t) X(kpp_drive_run) X(stem_kpp)

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
static double total_us[NROUTINES], max_us[NROUTINES], first_us[NROUTINES];
static long calls[NROUTINES];
static int registered;

static void report(void) {
  const char* path = getenv("MISTRA_TIME_FILE");
  FILE* f = path && *path ? fopen(path, "w") : stderr;
  if (!f) return;
  fprintf(f, "%-16s %8s %12s %12s %12s %12s\n", "routine", "calls", "total_ms", "mean_us", "max_us", "first_us");
  for (int i = 0; i < NROUTINES; ++i)
    if (calls[i]) fprintf(f, "%-16s %8ld %12.3f %12.2f %12.2f %12.2f\n", names[i], calls[i], total_us[i] * 1e-3, total_us[i] / calls[i], max_us[i], first_us[i]);
  if (f != stderr) fclose(f);
}

static inline double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

#define X(n)                                                                                  \
  void __real_##n##_(void*, void*, void*, void*, void*, void*, void*, void*);                 \
  void __wrap_##n##_(void* a, void* b, void* c, void* d, void* e, void* f, void* g, void* h) { \
    if (!registered) { registered = 1; atexit(report); }                                      \
    const double t0 = now_us();                                                               \
    __real_##n##_(a, b, c, d, e, f, g, h);                                                        \
    const double dt = now_us() - t0;                                                          \
    if (!calls[ID_##n]) first_us[ID_##n] = dt;                                                \
    total_us[ID_##n] += dt; calls[ID_##n]++;                                                  \
    if (dt > max_us[ID_##n]) max_us[ID_##n] = dt;                                             \
  }
ROUTINES(X)
#undef X
