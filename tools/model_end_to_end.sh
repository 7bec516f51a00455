#!/bin/bash
# The reference model end to end on ONE host, chemistry on the CPU (oracle/_ref/mistra_capture, unpatched) and on the GPU (oracle/_ref/mistra_gpu,
# oracle/build_gpu_model.sh): wall time of the time loop and of its chemistry stem for the same model minutes.  Run on the GPU box:
#   tools/model_end_to_end.sh > gpurun_out/model_end_to_end.txt      (BTZ96 on one CPU core takes ~80 s per 10 model minutes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "Joyce2014_basecase 30" "base1 10" "BTZ96 10"; do
  set -- $spec
  for bin in mistra_capture mistra_gpu; do
    [ -x oracle/_ref/$bin ] || { echo "$1 $bin: not built"; continue; }
    line=$(timeout -k 10 900 oracle/model_run.sh $PWD/oracle/_ref/$bin $1 $2 /tmp/e2e_$1_$bin MISTRA_COLUMN_DUMP=/tmp/e2e_$1_$bin.bin 2>&1 | grep -a "chemistry stem" | tail -1)
    echo "$1, $2 model minutes, $bin: $line"
  done
  python3 - /tmp/e2e_$1_mistra_capture.bin /tmp/e2e_$1_mistra_gpu.bin <<'PY'
import sys, numpy as np
def load(p):
    raw = open(p, "rb").read(); j1, j5, a, b, n = (int(x) for x in np.frombuffer(raw, np.int32, 5)); d = np.frombuffer(raw, np.float64, offset=20); o = 0; out = []
    for w in (j1, j5, a, b): out.append(d[o:o + w * n].reshape(n, w)); o += w * n
    return out
try:
    x, y = load(sys.argv[1]), load(sys.argv[2])
    worst = 0.0
    for p, q in zip(x, y):
        sc = np.abs(p).max(axis=0, keepdims=True); m = np.abs(p) > 1e-3 * sc
        if m.any(): worst = max(worst, float((np.abs(p - q)[m] / np.abs(p[m])).max()))
    print("   end states (entries above 1e-3 of their species' column maximum): max relative difference %.2e" % worst)
except Exception as e:
    print("   end states not compared:", e)
PY
done
