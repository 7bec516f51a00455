#!/usr/bin/env bash
# TEST INFRASTRUCTURE — CPU validation of the two-pass kpp_driver patch (shim/kpp_two_pass.patch).
# Builds oracle/_ref/mistra_two_pass: the reference model with
#   * scratch copies of kpp.f90 / gas.f / aer.f / tot.f carrying the patch (oracle/two_pass_patch.py; copies live under
#     oracle/_ref/two_pass/src only while this script runs),
#   * the UNMODIFIED shim of shim/ (mistra_kpp_batch.f90, mistra_kpp_shim.f90) taking over INTEGRATE_x by --wrap,
#   * oracle/two_pass_standin.c in the place of libmistra_chem.so: the batched calls are served by the reference's own
#     integrator, one cell after the other (there is no GPU in the build container),
#   * oracle/capture_wrap.c recording every real INTEGRATE_x call as in the unpatched capture build.
# tests/test_two_pass.py runs both models on the same case and compares the records.
# Needs oracle/build_ref.sh model to have run (objects of the unpatched files are reused).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MISTRA_REFERENCE_SRC:-/root/reference/src}"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
FFLAGS="${FFLAGS:--O2 -ffp-contract=off -fPIC}"
OBJCOPY=/opt/rocm/lib/llvm/bin/llvm-objcopy
[ -d "$REF" ] || { echo "reference tree $REF not present: nothing to build"; exit 0; }
[ -f "$OUT/obj/str_lib.o" ] || "$HERE/build_ref.sh" model
TP="$OUT/two_pass"
mkdir -p "$TP/obj"
python3 "$HERE/two_pass_patch.py" "$REF" "$TP/src" "$HERE/../shim/kpp_two_pass.patch"
cd "$TP/obj"
# the batch module first (the patched files USE it); the patched sources see the reference's headers and modules
"$FC" $FFLAGS -c "$HERE/../shim/mistra_kpp_batch.f90" -o mistra_kpp_batch.o
for f in gas.f aer.f tot.f; do "$FC" $FFLAGS -I"$REF" -I"$OUT/obj" -c "$TP/src/$f" -o "${f%.*}.o" & done; wait
"$FC" $FFLAGS -I"$REF" -I"$OUT/obj" -c "$TP/src/kpp.f90" -o kpp.o
"$FC" $FFLAGS -c "$HERE/../shim/mistra_kpp_shim.f90" -o shim.o
# the shim's INTEGRATE_x become the --wrap targets; the recorder's wrappers become plain functions the stand-in calls
"$OBJCOPY" --redefine-sym integrate_g_=__wrap_integrate_g_ --redefine-sym integrate_a_=__wrap_integrate_a_ \
           --redefine-sym integrate_t_=__wrap_integrate_t_ shim.o shim_wrap.o
gcc -O2 -c "$HERE/capture_wrap.c" -o capture.o
"$OBJCOPY" --redefine-sym __wrap_integrate_g_=captured_integrate_g_ --redefine-sym __wrap_integrate_a_=captured_integrate_a_ \
           --redefine-sym __wrap_integrate_t_=captured_integrate_t_ capture.o capture_fn.o
gcc -O2 -c "$HERE/two_pass_standin.c" -o standin.o
O="$OUT/obj"
REST="mod_out_netCDF activity utils radinit nrad outp nuc jrate"
MODS="precision constants global_params common_modules data_surface file_unit config"
BUD="bud_g bud_a bud_t bud_s_g bud_s_a bud_s_t"
objs=""; for m in $MODS $BUD $REST; do objs="$objs $O/$m.o"; done
"$FC" -o "$OUT/mistra_two_pass" "$O/column_driver.o" $objs gas.o aer.o tot.o kpp.o "$O/str_lib.o" mistra_kpp_batch.o shim_wrap.o \
    capture_fn.o standin.o -Wl,--wrap=integrate_g_ -Wl,--wrap=integrate_a_ -Wl,--wrap=integrate_t_ -Wl,--unresolved-symbols=ignore-all
rm -rf "$TP/src"      # the patched scratch copies are build inputs only: nothing of the reference's text stays under the repo
echo "oracle/_ref/mistra_two_pass ready"
