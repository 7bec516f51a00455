#!/usr/bin/env bash
# TEST INFRASTRUCTURE — runs a build of the reference model (oracle/_ref/mistra_capture: unpatched; oracle/_ref/mistra_gpu: chemistry on the GPU,
# build_gpu_model.sh) on one of the staged namelists for a number of model minutes, from the DATA files staged under oracle/_ref/model_inputs
# (build_gpu_model.sh) — nothing is read from /root/reference, so the same call works on the GPU box.
#   model_run.sh <binary> <case: Joyce2014_basecase | base1 | BTZ96> <minutes> <run dir> [ENV=VALUE ...]
# The namelist copy written into the run dir has netcdf=F (no netCDF in the image) and chem=T.  Prints the tail of the model's stderr (the time loop's
# wall-time line of oracle/column_driver.f90); MISTRA_COLUMN_DUMP=<file> in the environment list makes the model leave its chemical end state there.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
BIN="$1"; CASE="$2"; MINUTES="$3"; RUN="$4"; shift 4
IN="${MISTRA_MODEL_INPUTS:-$HERE/_ref/model_inputs}"
[ -f "$IN/namelists/namelist.$CASE" ] || { echo "no staged namelist for $CASE under $IN (oracle/build_gpu_model.sh stages them)" >&2; exit 1; }
rm -rf "$RUN"; mkdir -p "$RUN/out"
sed -e 's/^\( *netcdf *= *\)T/\1F/' -e 's/^\( *chem *= *\)F/\1T/' "$IN/namelists/namelist.$CASE" > "$RUN/namelist"
cd "$RUN"
env INPDIR="$IN/input/" MECHDIR="$IN/mech/" OUTDIR="$RUN/out/" NAMELIST="$RUN/namelist" MISTRA_COLUMN_MINUTES="$MINUTES" "$@" "$BIN" > "$RUN/stdout.log" 2> "$RUN/stderr.log" \
  || { echo "model failed:"; tail -5 "$RUN/stderr.log"; exit 1; }
grep -a "chemistry stem" "$RUN/stderr.log" | tail -1
