#!/bin/bash
# Same-box comparison of schedule-compiler switches on the diagnostic library (tools/diag_dense.sh env -> libdiag_env.so):
#   tools/ab_envs.sh MECH CELLS "VAR=1 VAR2=x" "VAR3=1" ...     ("-" = no switch); two alternating passes, timesteps/s each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
MECH=$1; CELLS=$2; shift 2
for rep in 1 2; do
  for S in "$@"; do
    [ "$S" = "-" ] && SET="" || SET="$S"
    env $SET MISTRA_MECH_DIR=$PWD/mistra_amd/mech MISTRA_CHEM_LIB=$PWD/tools/diaglib/libdiag_env.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --no-extra --mech $MECH --cells-per-gpu $CELLS --steps 2 --warmup 1 2>/tmp/ab_err.txt \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$MECH [$S]', '%.0f' % d['value'], 'timesteps/s')" || { echo "$MECH [$S] FAILED"; tail -3 /tmp/ab_err.txt; }
  done
done
