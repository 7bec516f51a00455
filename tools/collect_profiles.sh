#!/bin/bash
# Collects the evidence kept under profiles/ (run on the GPU box through gpurun; ~5 GPU-minutes):
#   tools/collect_profiles.sh r01      -> gpurun_out/profiles_r01/*   (copy what is to be judged into profiles/)
# One rocprofv3 invocation per counter group; --pmc is never combined with the trace domains gpurun refuses.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== bench (tot, with cpu baseline)"; timeout -k 10 400 python3 $R/bench.py > $OUT/${TAG}_bench_tot_1gpu.json 2> $OUT/bench_tot.err
for m in gas aer; do
  echo "== bench $m"; timeout -k 10 200 python3 $R/bench.py --mech $m --no-cpu-baseline --no-parity --no-extra > $OUT/${TAG}_bench_${m}_1gpu.json 2> $OUT/bench_$m.err
done
echo "== kernel trace + stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-extra > $OUT/${TAG}_bench_tot_1gpu_under_rocprof.json 2> $OUT/stats.err
for m in tot aer gas; do
  echo "== FETCH_SIZE $m"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$m -- python3 $R/bench.py --mech $m --no-cpu-baseline --no-parity --no-extra --steps 1 --warmup 0 > /dev/null 2> $OUT/fetch_$m.err
  echo "== WRITE_SIZE $m"
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$m -- python3 $R/bench.py --mech $m --no-cpu-baseline --no-parity --no-extra --steps 1 --warmup 0 > /dev/null 2> $OUT/write_$m.err
done
echo "== L2 hits / misses (tot)"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-extra --steps 1 --warmup 0 > /dev/null 2> $OUT/tcc.err
echo "== SQ counters (4096 cells)"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH"; do
  d=$OUT/sq_$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-extra --cells-per-gpu 4096 --steps 1 --warmup 0 > /dev/null 2>> $OUT/sq.err
done
echo "== phase profile"; timeout -k 10 200 python3 $R/tools/profile_phases.py > $OUT/phases.log 2>&1
echo "== dense tail block, cycles per section"; timeout -k 10 200 python3 $R/tools/diag_dense_stamps.py > $OUT/${TAG}_dense_lu_sections.txt 2>&1
# summaries
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def find(d, pat):
    r = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return r[0] if r else None
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.reader(open(f)))
    open(os.path.join(out, tag + "_bench_tot_kernel_stats.csv"), "w").write("\n".join(",".join(r) for r in rows[:8]) + "\n")
f = find("stats", "*kernel_trace.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "ros3" in r["Kernel_Name"]]
    with open(os.path.join(out, tag + "_bench_tot_kernel_trace_ros3.csv"), "w") as g:
        g.write("Kernel_Name,Start_Timestamp,End_Timestamp,duration_ms,Grid_Size,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count\n")
        for r in keep:
            g.write(",".join([r["Kernel_Name"].replace(",", ";"), r["Start_Timestamp"], r["End_Timestamp"], "%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)] +
                             [r.get(k, "") for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")]) + "\n")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
cells = {"tot": 100000, "aer": 100000, "gas": 100000}
with open(os.path.join(out, tag + "_bench_pmc_fetch_write.csv"), "w") as g:
    g.write("Mechanism,Counter_Name,Kernel_Name,Counter_Value_sum\n")
    for mech in ("tot", "aer", "gas"):
        vals = {}
        for d in ("fetch_" + mech, "write_" + mech):
            f = find(d, "*counter_collection.csv")
            if not f: continue
            acc = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                if "ros3" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
            for k, v in acc.items():
                vals[k] = v
                g.write("%s,%s,ros3_integrate_kernel,%r\n" % (mech, k, v))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            json.dump({"mech": mech, "cells": cells[mech], "fetch_size_kib": vals["FETCH_SIZE"], "write_size_kib": vals["WRITE_SIZE"],
                       "bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                       "algorithmic_bytes_per_launch": cells[mech] * bench.ALG_BYTES[mech],
                       "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --mech %s --no-cpu-baseline --no-parity --no-extra --steps 1 --warmup 0`; "
                              "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), KiB -> bytes; see profiles/README.md" % mech,
                       "kernel": "ros3_integrate_kernel", "kernel_source_hash": bench.kernel_source_hash(), "round": int(tag[1:])},
                      open(os.path.join(out, "%s_traffic_%s.json" % (tag, mech)), "w"), indent=1)
f = find("tcc", "*counter_collection.csv")
if f:
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "ros3" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    if acc.get("TCC_HIT_sum", 0) + acc.get("TCC_MISS_sum", 0) > 0:
        open(os.path.join(out, tag + "_l2_hit_rate_tot.txt"), "w").write(
            "TCC_HIT_sum %.0f\nTCC_MISS_sum %.0f\nL2 hit rate %.4f  (tot, 100000 cells, one launch)\n"
            % (acc["TCC_HIT_sum"], acc["TCC_MISS_sum"], acc["TCC_HIT_sum"] / (acc["TCC_HIT_sum"] + acc["TCC_MISS_sum"])))
with open(os.path.join(out, tag + "_sq_counters_tot_4096cells.txt"), "w") as g:
    for d in sorted(glob.glob(os.path.join(out, "sq_*"))):
        f = find(os.path.basename(d), "*counter_collection.csv")
        if not f: continue
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "ros3" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items(): g.write("%s %f\n" % (k, v))
with open(os.path.join(out, tag + "_phase_profile.txt"), "w") as g:
    for l in open(os.path.join(out, "phases.log")):
        if "profile]" in l: g.write(l)
print("summaries in", out)
PY
ls $OUT | grep "^$TAG"
