#!/bin/bash
# The reference model end to end on ONE host, chemistry on the CPU (oracle/_ref/mistra_capture, unpatched) and on the GPU (oracle/_ref/mistra_gpu,
# oracle/build_gpu_model.sh): wall time of the time loop and of its chemistry stem for the same model minutes.  Run on the GPU box:
#   tools/model_end_to_end.sh > gpurun_out/model_end_to_end.txt      (BTZ96 on one CPU core takes ~80 s per 10 model minutes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# MISTRA_E2E_SPECS="case minutes;case minutes" replaces the default list (e.g. "BTZ96 60" for a model hour: ~4 min of CPU for the unpatched build)
IFS=";" read -ra SPECS <<< "${MISTRA_E2E_SPECS:-Joyce2014_basecase 30;base1 10;BTZ96 10}"
for spec in "${SPECS[@]}"; do
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
    for name, p, q in zip(("s1", "s3", "sl1", "sion1"), x, y):
        sc = np.abs(p).max(axis=0, keepdims=True); m = np.abs(p) > 1e-3 * sc
        if not m.any(): continue
        rel = np.where(m, np.abs(p - q) / np.where(m, np.abs(p), 1.0), 0.0)
        k, j = np.unravel_index(rel.argmax(), rel.shape)
        r = rel[m]
        print("   %-5s %6d entries above 1e-3 of their species' column maximum: relative difference median %.1e, 99 %% %.1e, 99.9 %% %.1e, max %.2e (layer %d, column %d: %.6e | %.6e)"
              % (name, int(m.sum()), np.median(r), np.percentile(r, 99), np.percentile(r, 99.9), r.max(), k + 1, j + 1, p[k, j], q[k, j]))
        worst = max(worst, float(r.max()))
    print("   end states: max relative difference %.2e" % worst)
except Exception as e:
    print("   end states not compared:", e)
PY
done
