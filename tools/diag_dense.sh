#!/bin/bash
# Diagnostic builds of the library (results of a cut build are garbage; the timing is what is read):
#   tools/diag_dense.sh pN      dense block's factorisation cut after N panels   -> tools/diaglib/libdiag_pN.so
#   tools/diag_dense.sh stamps  cycle stamps inside dense_lu (tools/diag_dense_stamps.py reads them) -> libdiag_stamps.so
cd "$(dirname "$0")/.."
mkdir -p tools/diaglib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for V in "$@"; do
  if [ "$V" = env ]; then      # the schedule compiler's A/B switches (MISTRA_DIAG_*), read from the environment -> libdiag_env.so
    hipcc --offload-arch=gfx950 $FLAGS -DMISTRA_DIAG_ENV -c mistra_amd/csrc/capi.cpp -o /tmp/capi_env.o &&
    hipcc --offload-arch=gfx950 $FLAGS -DMISTRA_DIAG_ENV -c mistra_amd/csrc/schedule.cpp -o /tmp/schedule_env.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diaglib/libdiag_env.so mistra_amd/build/ros3_kernel.o /tmp/capi_env.o /tmp/schedule_env.o mistra_amd/build/mech_tables.o mistra_amd/build/rates.o mistra_amd/build/pack.o -ldl
    continue
  fi
  case $V in
    p*) DEF="-DMISTRA_DIAG_DENSE_PANELS=${V#p}";;
    stamps) DEF="-DMISTRA_DIAG_STAMPS";;
    callervm) DEF="-DMISTRA_DIAG_CALLER_RUNS_VM -DMISTRA_DIAG_LATE_LOADS";;
    stampslate) DEF="-DMISTRA_DIAG_STAMPS -DMISTRA_DIAG_LATE_LOADS";;
  esac
  hipcc --offload-arch=gfx950 $FLAGS $DEF -c mistra_amd/csrc/ros3_kernel.hip -o /tmp/ros3_diag_$V.o &&
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diaglib/libdiag_$V.so /tmp/ros3_diag_$V.o mistra_amd/build/capi.o mistra_amd/build/schedule.o mistra_amd/build/mech_tables.o mistra_amd/build/rates.o mistra_amd/build/pack.o -ldl
done
