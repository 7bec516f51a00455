#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by several % in clock): tools/ab_bench.sh libA.so libB.so [mech]
# Libraries are looked up in mistra_amd/lib/.  Prints timesteps/s of alternating runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; MECH=${3:-tot}; CELLS=${4:-25600}
for rep in 1 2; do
  for L in $A $B; do
    MISTRA_CHEM_LIB=$PWD/mistra_amd/lib/$L timeout -k 10 200 python bench.py --no-cpu-baseline --mech $MECH --cells-per-gpu $CELLS --steps 2 --warmup 1 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', '%.0f' % d['value'], 'timesteps/s  kernel_ms %.1f' % d['roofline']['kernel_ms'])"
  done
done
