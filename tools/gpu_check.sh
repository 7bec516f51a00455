# One gpurun call after a kernel change: bit-identity tests gate a same-box A/B against tools/diaglib/libprev.so and the HBM traffic counters.  tools/gpu_check.sh TAG -> gpurun_out/TAG
set -o pipefail
cd $GRAFT_REPO_ROOT; D=gpurun_out/${1:-r03n}; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_gpu_phases.py tests/test_gpu_parity.py -m gpu -x -q > $D/tests.log 2>&1 || { tail -15 $D/tests.log; exit 1; }
tail -2 $D/tests.log
(tools/ab_many.sh tot 25600 libprev.so libmistra_chem.so; tools/ab_many.sh aer 51200 libprev.so libmistra_chem.so; tools/ab_many.sh gas 102400 libprev.so libmistra_chem.so) > $D/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|amdgpu.ids" $D/ab.log
grep -q FAILED $D/ab.log && exit 1
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for m in aer gas tot; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$D/pmc_${m}_$c -- python3 $R/bench.py --mech $m --no-cpu-baseline --no-parity --no-extra --steps 1 --warmup 0 > /dev/null 2> $R/$D/pmc_${m}_$c.err || exit 1
done; done
cd $R; python3 - $D <<'PY'
import csv,glob,sys
for m in ("aer","gas","tot"):
  for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv"%(sys.argv[1],m,c), recursive=True):
        s=0
        for r in csv.DictReader(open(f)):
            if "ros3" in r["Kernel_Name"]: s+=float(r["Counter_Value"])
        print(m, c, s)
PY
