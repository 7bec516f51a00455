#!/usr/bin/env bash
# TEST INFRASTRUCTURE — runs the reference model built by build_ref.sh (`oracle/_ref/mistra_capture`) on one of the
# reference's shipped namelists and captures real INTEGRATE_x calls (see capture_wrap.c) into oracle/_ref/capture_<case>.bin.
# The namelist is read from /root/reference/namelists, a modified COPY (netcdf=F: the image has no netCDF; chem=T;
# lstmax=<hours>) is written to the scratch run directory under oracle/_ref/ — nothing under /root/reference is touched.
# MISTRA_MODEL_BIN: another build of the model to run instead (oracle/_ref/mistra_two_pass, build_two_pass.sh).
# Usage: capture_run.sh <namelist-suffix e.g. BTZ96> <hours> [extra env assignments for capture_wrap.c ...]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REFROOT="${MISTRA_REFERENCE:-/root/reference}"
CASE="$1"; HOURS="$2"; shift 2
RUN="$HERE/_ref/run_${CASE}${MISTRA_RUN_TAG:-}"
rm -rf "$RUN/out"; mkdir -p "$RUN/out"
# MISTRA_NAMELIST_SED: one more sed expression for the copy (e.g. to switch a module of a shipped case off)
sed -e 's/^\( *netcdf *= *\)T/\1F/' -e 's/^\( *chem *= *\)F/\1T/' -e "s/^\( *lstmax *= *\)[0-9]*/\1$HOURS/" \
    -e "${MISTRA_NAMELIST_SED:-s/^$//}" "$REFROOT/namelists/namelist.$CASE" > "$RUN/namelist"
cd "$RUN"
env INPDIR="$REFROOT/input/" MECHDIR="$REFROOT/src/mech/" OUTDIR="$RUN/out/" NAMELIST="$RUN/namelist" \
    MISTRA_CAPTURE_FILE="$HERE/_ref/capture_${CASE}${MISTRA_RUN_TAG:-}.bin" "$@" "${MISTRA_MODEL_BIN:-$HERE/_ref/mistra_capture}" > "$RUN/stdout.log" 2> "$RUN/stderr.log"
tail -4 "$RUN/stderr.log"
