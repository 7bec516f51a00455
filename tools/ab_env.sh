cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for V in 0 1; do
    if [ $V = 1 ]; then export MISTRA_DIAG_PLAIN_DEAL=1; else unset MISTRA_DIAG_PLAIN_DEAL; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --cells-per-gpu 25600 --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plain_deal=$V', '%.0f' % d['value'], 'timesteps/s')"
  done
done
