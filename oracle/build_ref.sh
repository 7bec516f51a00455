#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds the *reference itself* (Mistra-UEA/Mistra, Fortran) from the sources where they
# lie under /root/reference/src, with the AMD flang that ships in the ROCm image.  Outputs go ONLY to oracle/_ref/
# (git-ignored, not gpurun-ignored).  No reference source is copied into the repo.
#
#   oracle/_ref/libmistra_ref.so   gas.f / aer.f / tot.f (drivers, Update_RCONST_x, INTEGRATE_x, Fun, Jac_SP, KppDecomp,
#                                  KppSolve) + kpp.f90 rate laws + the small modules they USE.  Used through ctypes by
#                                  tests/ and by bench.py's cpu_baseline leg (kind "reference").
#   oracle/_ref/mistra_capture     every routine of the model (all of src/ except out_netCDF.f, whose netcdf.inc this
#                                  image lacks) sequenced by oracle/column_driver.f90 — my own stripped time loop that
#                                  never calls an output routine — instead of the reference's main program (which calls
#                                  write_grid -> netCDF unconditionally, str.f90:216).  No stand-in for netCDF is written:
#                                  the 4 netCDF entry points stay unresolved and unreachable.  Linked with
#                                  -Wl,--wrap=integrate_{g,a,t}_ and oracle/capture_wrap.c, which dumps /GDATA_x/ before
#                                  and after every real INTEGRATE_x call.  Used once, here, to make tests/golden/.
#
# Flags: -O2 (the reference Makefile's level, src/Makefile:21), -ffp-contract=off, generic x86-64 (no FMA), so the
# arithmetic is plain IEEE double mul/add in source order.  Compiler and flags are recorded in oracle/_ref/BUILD_INFO.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MISTRA_REFERENCE_SRC:-/root/reference/src}"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
FFLAGS="${FFLAGS:--O2 -ffp-contract=off -fPIC}"
WHAT="${1:-lib}"          # lib | model | all

[ -d "$REF" ] || { echo "reference tree $REF not present: nothing to build (prebuilt oracle/_ref is used as is)"; exit 0; }
[ -x "$FC" ]  || { echo "no flang at $FC" >&2; exit 1; }
mkdir -p "$OUT/obj"
cd "$OUT/obj"

compile() {  # compile $1 (basename in $REF) unless its object is newer
  local src="$REF/$1" obj="${1%.*}.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then
    echo "  FC $1"
    "$FC" $FFLAGS -I"$REF" -c "$src" -o "$obj"
  fi
}

# module order follows the dependency chain of src/Makefile:79-131
MODS="precision.f90 constants.f90 global_params.f90 common_modules.f90 data_surface.f90 file_unit.f90 config.f90"
CHEM="gas.f aer.f tot.f kpp.f90 bud_g.f bud_a.f bud_t.f bud_s_g.f bud_s_a.f bud_s_t.f"
REST="mod_out_netCDF.f90 activity.f90 utils.f90 radinit.f90 nrad.f90 outp.f90 nuc.f90 jrate.f str.f90"

for f in $MODS; do compile "$f"; done
# the three mechanism files are independent of each other: build them side by side (tot.f takes about a minute)
for f in gas.f aer.f tot.f; do compile "$f" & done; wait
for f in kpp.f90 bud_g.f bud_a.f bud_t.f bud_s_g.f bud_s_a.f bud_s_t.f; do compile "$f"; done

objs() { for f in "$@"; do printf '%s ' "${f%.*}.o"; done; }

if [ "$WHAT" = lib ] || [ "$WHAT" = all ]; then
  echo "  LD libmistra_ref.so"
  # undefined references (pitzer, rgl, vterm, close_netcdf: called only from non-hot routines of kpp.o/config.o)
  # stay undefined; they are functions reached through the PLT and never called on the integrator path.
  "$FC" -shared -o "$OUT/libmistra_ref.so" $(objs $MODS $CHEM) -Wl,-z,lazy
fi

if [ "$WHAT" = model ] || [ "$WHAT" = all ]; then
  for f in $REST; do compile "$f"; done
  gcc -O2 -c "$HERE/capture_wrap.c" -o capture_wrap.o
  # Update_RCONST_x calls recorded with the inputs packed by the product's own Fortran routine (shim/mistra_kpp_rates.f90)
  gcc -O2 -c "$HERE/capture_rates_wrap.c" -o capture_rates_wrap.o
  "$FC" $FFLAGS -c "$HERE/../shim/mistra_kpp_rates.f90" -o mistra_kpp_rates.o
  # whole x_drive calls (pack, budgets, hand-over: SURVEY §8 f2), recorded from Fortran: the layer's data sits in module gas_common
  "$FC" $FFLAGS -I"$OUT/obj" -c "$HERE/capture_drive_wrap.f90" -o capture_drive_wrap.o
  # fast_k_mt_a / fast_k_mt_t calls of liq_parm
  "$FC" $FFLAGS -I"$OUT/obj" -c "$HERE/capture_kmt_wrap.f90" -o capture_kmt_wrap.o
  # henry_a/t and equil_co_a/t calls of liq_parm
  "$FC" $FFLAGS -I"$OUT/obj" -c "$HERE/capture_liq_wrap.f90" -o capture_liq_wrap.o
  # cw_rc and dry_cw_rc calls of liq_parm
  "$FC" $FFLAGS -I"$OUT/obj" -c "$HERE/capture_cwrc_wrap.f90" -o capture_cwrc_wrap.o
  # dry_rates_g/a/t calls of liq_parm; the species indices of the routines' idr list come from the mechanisms' parameter headers
  "$FC" $FFLAGS -I"$OUT/obj" -I"$REF" -c "$HERE/capture_dry_wrap.f90" -o capture_dry_wrap.o
  # the reference's own main program stays in str.o but under another name (it is not the entry point here)
  /opt/rocm/lib/llvm/bin/llvm-objcopy --redefine-sym main=mistra_reference_main --redefine-sym _QQmain=mistra_reference_qqmain \
      str.o str_lib.o
  "$FC" $FFLAGS -I"$REF" -c "$HERE/column_driver.f90" -o column_driver.o
  echo "  LD mistra_capture"
  "$FC" -o "$OUT/mistra_capture" column_driver.o $(objs $MODS $CHEM ${REST% str.f90}) str_lib.o capture_wrap.o \
      capture_rates_wrap.o mistra_kpp_rates.o capture_drive_wrap.o capture_kmt_wrap.o capture_liq_wrap.o capture_cwrc_wrap.o capture_dry_wrap.o \
      -Wl,--wrap=integrate_g_ -Wl,--wrap=integrate_a_ -Wl,--wrap=integrate_t_ \
      -Wl,--wrap=update_rconst_g_ -Wl,--wrap=update_rconst_a_ -Wl,--wrap=update_rconst_t_ \
      -Wl,--wrap=gas_drive_ -Wl,--wrap=aer_drive_ -Wl,--wrap=tot_drive_ -Wl,--wrap=fast_k_mt_a_ -Wl,--wrap=fast_k_mt_t_ \
      -Wl,--wrap=henry_a_ -Wl,--wrap=henry_t_ -Wl,--wrap=equil_co_a_ -Wl,--wrap=equil_co_t_ -Wl,--wrap=v_mean_a_ -Wl,--wrap=v_mean_t_ -Wl,--wrap=st_coeff_a_ -Wl,--wrap=st_coeff_t_ -Wl,--wrap=cw_rc_ -Wl,--wrap=dry_cw_rc_ -Wl,--wrap=dry_rates_g_ -Wl,--wrap=dry_rates_a_ -Wl,--wrap=dry_rates_t_ \
      -Wl,--unresolved-symbols=ignore-all
  # the same model with nothing captured: every routine liq_parm calls (and liq_parm, pitzer, kpp_driver) behind a clock (time_liq_wrap.c)
  gcc -O2 -c "$HERE/time_liq_wrap.c" -o time_liq_wrap.o
  echo "  LD mistra_time"
  TW=""; for r in liq_parm gasdrydep cw_rc v_mean_a henry_a st_coeff_a equil_co_a fast_k_mt_a v_mean_t henry_t st_coeff_t equil_co_t fast_k_mt_t \
      dry_cw_rc dry_rates_g dry_rates_a dry_rates_t activ pitzer kpp_driver; do TW="$TW -Wl,--wrap=${r}_"; done
  "$FC" -o "$OUT/mistra_time" column_driver.o $(objs $MODS $CHEM ${REST% str.f90}) str_lib.o time_liq_wrap.o $TW -Wl,--unresolved-symbols=ignore-all
fi

{ echo "compiler: $("$FC" --version | head -1)"; echo "flags: $FFLAGS"; echo "reference: $REF";
  echo "built: $(date -u +%FT%TZ)"; } > "$OUT/BUILD_INFO"
echo "oracle/_ref ready"
