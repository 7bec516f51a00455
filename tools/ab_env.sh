#!/bin/bash
# Same-box A/B of one build under two environments: tools/ab_env.sh VAR=value [mech] [cells]
# (e.g. MISTRA_DIAG_PLAIN_DEAL=1) on the diagnostic library that reads such switches (tools/diag_dense.sh env -> libdiag_env.so;
# the product library ignores them).  Prints timesteps/s of alternating runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SETTING=$1; MECH=${2:-tot}; CELLS=${3:-25600}
for rep in 1 2; do
  for V in 0 1; do
    if [ $V = 1 ]; then PRE="env $SETTING MISTRA_MECH_DIR=$PWD/mistra_amd/mech MISTRA_CHEM_LIB=$PWD/tools/diaglib/libdiag_env.so"; else PRE="env MISTRA_MECH_DIR=$PWD/mistra_amd/mech MISTRA_CHEM_LIB=$PWD/tools/diaglib/libdiag_env.so"; fi
    $PRE timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-extra --mech $MECH --cells-per-gpu $CELLS --steps 2 --warmup 1 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('${SETTING} applied=$V', '%.0f' % d['value'], 'timesteps/s')"
  done
done
