#!/bin/bash
# Same-box comparison of several library builds: tools/ab_many.sh MECH CELLS libA.so libB.so ...   (two alternating passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
libpath() { if [ "$1" = libmistra_chem.so ]; then echo $PWD/mistra_amd/lib/$1; else echo $PWD/tools/diaglib/$1; fi; }
MECH=$1; CELLS=$2; shift 2
for rep in 1 2; do
  for L in "$@"; do
    MISTRA_MECH_DIR=$PWD/mistra_amd/mech MISTRA_CHEM_LIB=$(libpath $L) timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-extra --mech $MECH --cells-per-gpu $CELLS --steps 2 --warmup 1 2>/tmp/ab_err.txt \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$MECH $L', '%.0f' % d['value'], 'timesteps/s  kernel_ms %.1f' % d['roofline']['kernel_ms'])" || { echo "$MECH $L FAILED:"; tail -3 /tmp/ab_err.txt; }
  done
done
