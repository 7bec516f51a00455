#!/usr/bin/env bash
# TEST INFRASTRUCTURE — the reference model END TO END with its chemistry on the GPU (BASELINE.json configs[4]).
# Builds oracle/_ref/mistra_gpu: the reference's own routines (objects of oracle/build_ref.sh `model`), sequenced by oracle/column_driver.f90, with
#   * scratch copies of kpp.f90 / gas.f / aer.f / tot.f carrying BOTH shipped patches — shim/kpp_drive.patch (one device call per mechanism and 10-s
#     step) and shim/kpp_liq.patch (liq_parm's fifteen kernel calls go to the drop-ins) —, deleted again when the objects exist,
#   * the UNMODIFIED shim of shim/ and the PRODUCT library mistra_amd/lib/libmistra_chem.so (found through an rpath relative to the binary):
#     no stand-in anywhere — what runs is what a maintainer would link.
# and stages the DATA files the model reads at run time (initial profiles, photolysis tables, species lists, three namelists; no source file) under
# oracle/_ref/model_inputs/, because /root/reference does not exist on the GPU box.  Everything lands in oracle/_ref/ (git-ignored, travels with gpurun).
# tests/test_gpu_model.py runs it on the MI355X against the end state of the unpatched model (oracle/_ref/mistra_capture) computed here.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REFROOT="${MISTRA_REFERENCE:-/root/reference}"
REF="$REFROOT/src"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
FFLAGS="${FFLAGS:--O2 -ffp-contract=off -fPIC}"
OBJCOPY=/opt/rocm/lib/llvm/bin/llvm-objcopy
[ -d "$REF" ] || { echo "reference tree $REF not present: nothing to build (prebuilt oracle/_ref is used as is)"; exit 0; }
[ -f "$OUT/obj/str_lib.o" ] || "$HERE/build_ref.sh" model
[ -f "$HERE/../mistra_amd/lib/libmistra_chem.so" ] || { echo "build the product library first (python -m mistra_amd.build)" >&2; exit 1; }
TP="$OUT/gpu_model"
mkdir -p "$TP/obj"
python3 "$HERE/two_pass_patch.py" --mode drive "$REF" "$TP/src"
python3 "$HERE/two_pass_patch.py" --mode liq "$TP/src" "$TP/src"      # (kpp.f90 again: liq_parm's calls, on top of the driver's)
cd "$TP/obj"
S="$HERE/../shim"
"$FC" $FFLAGS -c "$S/mistra_kpp_batch.f90" -o mistra_kpp_batch.o
"$FC" $FFLAGS -c "$S/mistra_kpp_shim.f90" -o shim.o
"$FC" $FFLAGS -c "$S/mistra_kpp_rates.f90" -o mistra_kpp_rates.o
"$FC" $FFLAGS -c "$S/mistra_kpp_drive.f90" -o mistra_kpp_drive.o
"$FC" $FFLAGS -c "$S/mistra_kpp_liq.f90" -o mistra_kpp_liq.o
"$FC" $FFLAGS -I"$OUT/obj" -I"$REF" -c "$S/mistra_kpp_model.f90" -o mistra_kpp_model.o
for f in gas.f aer.f tot.f; do "$FC" $FFLAGS -I"$REF" -I"$OUT/obj" -c "$TP/src/$f" -o "${f%.*}.o" & done; wait
"$FC" $FFLAGS -I"$REF" -I"$OUT/obj" -c "$TP/src/kpp.f90" -o kpp.o
# the shim's INTEGRATE_x take the place of the generated ones (a box run or a serial fall-back reaches them; the batched driver does not)
"$OBJCOPY" --redefine-sym integrate_g_=__wrap_integrate_g_ --redefine-sym integrate_a_=__wrap_integrate_a_ \
           --redefine-sym integrate_t_=__wrap_integrate_t_ shim.o shim_wrap.o
O="$OUT/obj"
REST="mod_out_netCDF activity utils radinit nrad outp nuc jrate"
MODS="precision constants global_params common_modules data_surface file_unit config"
BUD="bud_g bud_a bud_t bud_s_g bud_s_a bud_s_t"
objs=""; for m in $MODS $BUD $REST; do objs="$objs $O/$m.o"; done
"$FC" -o "$OUT/mistra_gpu" "$O/column_driver.o" $objs gas.o aer.o tot.o kpp.o "$O/str_lib.o" mistra_kpp_batch.o shim_wrap.o mistra_kpp_rates.o \
    mistra_kpp_drive.o mistra_kpp_liq.o mistra_kpp_model.o \
    -Wl,--wrap=integrate_g_ -Wl,--wrap=integrate_a_ -Wl,--wrap=integrate_t_ \
    -L"$HERE/../mistra_amd/lib" -lmistra_chem -Wl,-rpath,'$ORIGIN/../../mistra_amd/lib' -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-all
# the same with a clock around liq_parm, kpp_driver and what they call (time_liq_wrap.c): where a step's time goes inside the running model
gcc -O2 -c "$HERE/time_liq_wrap.c" -o time_liq_wrap.o
TW=""; for r in stem_kpp liq_parm gasdrydep activ pitzer kpp_driver kpp_drive_run cw_rc_hip v_mean_hip_a henry_hip_a st_coeff_hip_a equil_co_hip_a fast_k_mt_hip_a \
    v_mean_hip_t henry_hip_t st_coeff_hip_t equil_co_hip_t fast_k_mt_hip_t dry_cw_rc_hip dry_rates_hip_g dry_rates_hip_a dry_rates_hip_t; do TW="$TW -Wl,--wrap=${r}_"; done
"$FC" -o "$OUT/mistra_gpu_time" "$O/column_driver.o" $objs gas.o aer.o tot.o kpp.o "$O/str_lib.o" mistra_kpp_batch.o shim_wrap.o mistra_kpp_rates.o \
    mistra_kpp_drive.o mistra_kpp_liq.o mistra_kpp_model.o time_liq_wrap.o $TW \
    -Wl,--wrap=integrate_g_ -Wl,--wrap=integrate_a_ -Wl,--wrap=integrate_t_ \
    -L"$HERE/../mistra_amd/lib" -lmistra_chem -Wl,-rpath,'$ORIGIN/../../mistra_amd/lib' -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-all
rm -rf "$TP/src"      # the patched scratch copies are build inputs only: nothing of the reference's text stays under the repo
# ---- run-time DATA of the model (no source): initial profiles and radiation / photolysis tables, species lists and index tables, namelists
IN="$OUT/model_inputs"
rm -rf "$IN"; mkdir -p "$IN/input" "$IN/mech" "$IN/namelists"
cp -r "$REFROOT"/input/*.dat "$REFROOT"/input/photolys "$IN/input/"
cp "$REFROOT"/src/mech/*.csv "$REFROOT"/src/mech/*.dat "$IN/mech/"
for c in Joyce2014_basecase base1 BTZ96 Buys13_0D Bott2020; do cp "$REFROOT/namelists/namelist.$c" "$IN/namelists/"; done
echo "oracle/_ref/mistra_gpu ready (inputs: $(du -sh "$IN" | cut -f1))"
