#!/bin/bash
# A variant build of the whole library for same-box A/B runs (tools/ab_bench.sh):  tools/build_variant.sh NAME [-DMACRO=..]...
#   -> tools/diaglib/libNAME.so      e.g.  tools/build_variant.sh aer256 -DMISTRA_AER_NT=256 -DMISTRA_AER_WPS=2
# The product library (python -m mistra_amd.build) is built without any of these macros.
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p tools/diaglib
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
T=/tmp/variant_$NAME; rm -rf $T; mkdir -p $T
for f in schedule.cpp mech_tables.cpp capi.cpp ros3_kernel.hip rates.hip pack.hip; do
  hipcc --offload-arch=gfx950 $FLAGS "$@" -c mistra_amd/csrc/$f -o $T/${f%.*}.o || exit 1
done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/diaglib/lib$NAME.so $T/*.o -ldl && echo tools/diaglib/lib$NAME.so
