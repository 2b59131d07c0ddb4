#!/bin/bash
# Collects the HBM traffic of hc::score_kernel with rocprofv3 PMC counters in SEPARATE passes
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots") and
# writes profiles-ready summaries under gpurun_out/.  Run on the GPU box:
#     gpurun -- bash tools/collect_traffic.sh c2
set -e
W=${1:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/traffic_fetch $R/gpurun_out/traffic_write $R/gpurun_out/traffic_l2 $R/gpurun_out/traffic_l1 $R/gpurun_out/traffic_sq $R/gpurun_out/traffic_lds
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_write -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/traffic_l2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/traffic_l1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/traffic_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/traffic_lds -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$R/gpurun_out/traffic_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "score_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
fetch_kb, write_kb = out.get("FETCH_SIZE", 0.0), out.get("WRITE_SIZE", 0.0)
res = {"workload": "$W", "order": "sfo", "record_bytes": 16, "kernel": "hc::score_kernel", "counters_per_launch": out,
       "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md HBM section); WRITE_SIZE as read; x1024 B per KB",
       "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
json.dump(res, open("$R/gpurun_out/traffic_$W.json", "w"), indent=1)
print(json.dumps(res)[:600])
PY
