# The round's closing GPU session (one gpurun call): the whole -m gpu suite, a same-box A/B of the product against tools/diaglib/libprev.so, the
# profile collection of tools/collect_profiles.sh and the large parity samples.   tools/gpu_final.sh TAG -> gpurun_out/final_TAG, gpurun_out/profiles_TAG
set -o pipefail
cd $GRAFT_REPO_ROOT; TAG=${1:-r04}; D=gpurun_out/final_$TAG; mkdir -p $D
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $D/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $D/gputests.log
(tools/ab_many.sh tot 25600 libprev.so libmistra_chem.so; tools/ab_many.sh aer 51200 libprev.so libmistra_chem.so; tools/ab_many.sh gas 102400 libprev.so libmistra_chem.so) > $D/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|amdgpu.ids" $D/ab.log
bash tools/collect_profiles.sh $TAG > $D/collect.log 2>&1; tail -3 $D/collect.log
(timeout -k 10 400 python3 tools/parity_sample.py 32768 tot; timeout -k 10 200 python3 tools/parity_sample.py 16384 aer; timeout -k 10 200 python3 tools/parity_sample.py 32768 gas) 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_$TAG/${TAG}_parity_samples.txt
cat gpurun_out/profiles_$TAG/${TAG}_parity_samples.txt
cat gpurun_out/profiles_$TAG/${TAG}_phase_profile.txt
python3 -c "
import json;d=json.load(open('gpurun_out/profiles_$TAG/${TAG}_bench_tot_1gpu.json'));print('tot',d['value'],'aer',d['extra']['aer']['value'],'gas',d['extra']['gas']['value'],'cpu',d['cpu_baseline']['value'],'parity',d['parity']['max_rel_all'],d['parity']['stats_identical'])"
