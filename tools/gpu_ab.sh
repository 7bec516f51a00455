# One gpurun call after a change: the bit-identity / parity tests gate a same-box A/B of the product library against tools/diaglib/libprev.so (all three mechanisms).  tools/gpu_ab.sh TAG -> gpurun_out/TAG
set -o pipefail
cd $GRAFT_REPO_ROOT; D=gpurun_out/${1:-ab}; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_gpu_phases.py tests/test_gpu_parity.py -m gpu -x -q > $D/tests.log 2>&1 || { tail -15 $D/tests.log; exit 1; }
tail -2 $D/tests.log
(tools/ab_many.sh tot 25600 libprev.so libmistra_chem.so; tools/ab_many.sh aer 51200 libprev.so libmistra_chem.so; tools/ab_many.sh gas 102400 libprev.so libmistra_chem.so) > $D/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|amdgpu.ids" $D/ab.log
grep -q FAILED $D/ab.log && exit 1
exit 0
