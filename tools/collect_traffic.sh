#!/bin/bash
# Collects the memory-side traffic and the issue counters of the scoring kernel with rocprofv3 PMC counters in SEPARATE
# passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"; no trace domains next to
# --pmc but the kernel trace) plus one --kernel-trace --stats pass for the kernel's average duration, and writes
# gpurun_out/traffic_<workload>.json — the file bench.py's roofline record reads once it is copied to profiles/.
# The file records what it measured: the kernel's symbol as the trace lists it, its mean duration, the hash of the kernel
# sources (bench.py refuses a file whose hash or duration does not match the run) and the commit (.build_sha, written by
# tools/gpu.sh before the snapshot leaves; the box has no .git).
#     gpurun -- bash tools/collect_traffic.sh c3
set -e
W=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/traffic_$W.d
mkdir -p $R/gpurun_out
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HC_WORKLOAD_CACHE=${HC_WORKLOAD_CACHE:-/tmp/hcw}
B="python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none"
pass() { d=$1; shift; rocprofv3 "$@" --kernel-trace --output-format csv -d $O/$d -- $B > $O/$d.out 2> $O/$d.err || echo "pass $d failed" >&2; }
# the duration pass traces >= 30 launches of the kernel and the file keeps their MEDIAN (round 3 kept the mean of nine, one of them an outlier)
B_COUNTERS=$B
B="python3 $R/bench.py --workload $W --steps 30 --warmup 2 --no-cpu-baseline --no-stage --also none"
pass stats --stats
B=$B_COUNTERS
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass l2 --pmc TCC_HIT_sum TCC_MISS_sum
pass l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE
pass sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
pass lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$R")
import bench
out, names = {}, collections.Counter()
for d in sorted(glob.glob("$O/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "score_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names[r["Kernel_Name"]] += 1
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
dur, others = [], collections.defaultdict(list)
for d in sorted(glob.glob("$O/stats/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(d)):
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        (dur if "score_kernel" in r["Kernel_Name"] else others[r["Kernel_Name"]]).append(t)
bench_line = {}
try:
    bench_line = json.loads(open("$O/stats.out").read().strip().splitlines()[-1])
except Exception:
    pass
fetch_kb, write_kb = out.get("FETCH_SIZE", 0.0), out.get("WRITE_SIZE", 0.0)
sha = open("$R/.build_sha").read().strip() if os.path.exists("$R/.build_sha") else None
res = {"workload": "$W", "order": "sfo", "record_bytes": 16,
       "kernel": names.most_common(1)[0][0] if names else None,
       "kernel_ms_rocprof_median": sorted(dur)[len(dur) // 2] if dur else None, "kernel_ms_rocprof_min": min(dur) if dur else None,
       "kernel_ms_rocprof_avg": sum(dur) / len(dur) if dur else None, "kernel_launches_traced": len(dur),
       "other_kernels_ms_avg": {k.split("(")[0]: sum(v) / len(v) for k, v in others.items() if "bucket_perm" in k},
       "kernel_ms_hipevents_under_rocprof": (bench_line.get("roofline") or {}).get("kernel_ms"),
       "kernel_source_sha": bench.kernel_source_sha(), "git_sha": sha,
       "counters_per_launch": out, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; x1024 B per KB",
       "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
json.dump(res, open("$R/gpurun_out/traffic_$W.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "counters_per_launch"}))
PY
# the stats pass's own summary, for profiles/
cp $O/stats/*/*_kernel_stats.csv $R/gpurun_out/kernel_stats_$W.csv 2>/dev/null || true
