# tools/gpu_ab3.sh TAG LIB...: parity tests on the product, then a same-box comparison of the named libraries (tot), product vs libprev.so on aer and gas
set -o pipefail
cd $GRAFT_REPO_ROOT; D=gpurun_out/${1:-ab}; shift; mkdir -p $D
timeout -k 10 600 python -m pytest tests/test_gpu_phases.py tests/test_gpu_parity.py -m gpu -x -q > $D/tests.log 2>&1 || { tail -25 $D/tests.log; exit 1; }
tail -2 $D/tests.log
(tools/ab_many.sh tot 25600 "$@"; tools/ab_many.sh aer 51200 libprev.so libmistra_chem.so; tools/ab_many.sh gas 102400 libprev.so libmistra_chem.so) > $D/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|amdgpu.ids" $D/ab.log
timeout -k 10 200 python3 tools/diag_dense_stamps.py > $D/stamps.log 2>&1; grep -v amdgpu.ids $D/stamps.log
timeout -k 10 200 python3 tools/profile_phases.py > $D/phases.log 2>&1; grep "profile] tot" $D/phases.log
exit 0
