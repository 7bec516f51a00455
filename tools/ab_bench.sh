#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by several % in clock): tools/ab_bench.sh libA.so libB.so [mech]
# libmistra_chem.so is the product (mistra_amd/lib/); every other name is a variant or diagnostic build in tools/diaglib/.  Prints timesteps/s of alternating runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
libpath() { if [ "$1" = libmistra_chem.so ]; then echo $PWD/mistra_amd/lib/$1; else echo $PWD/tools/diaglib/$1; fi; }
A=$1; B=$2; MECH=${3:-tot}; CELLS=${4:-25600}
for rep in 1 2; do
  for L in $A $B; do
    MISTRA_MECH_DIR=$PWD/mistra_amd/mech MISTRA_CHEM_LIB=$(libpath $L) timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-extra --mech $MECH --cells-per-gpu $CELLS --steps 2 --warmup 1 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', '%.0f' % d['value'], 'timesteps/s  kernel_ms %.1f' % d['roofline']['kernel_ms'])"
  done
done
