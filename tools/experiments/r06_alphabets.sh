#!/bin/bash
# Round 6: the scoring kernel at the north-star size (config 3: 500 000 pairs, 10^8 candidates) with the quality alphabets real reads
# have — which instantiation runs, its time per launch, and (for the workloads named in $PMC) the issue / LDS / memory counters.
#     tools/gpu.sh --timeout 2400 -- 'bash tools/experiments/r06_alphabets.sh "c3 c3q25 c3q35 c3q60 c3q35r c3q25r" "c3q35"'
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-"c3 c3q25 c3q35 c3q60"}
PMC=${2:-""}
TAG=${3:-r06_alphabets}
O=$R/gpurun_out/$TAG
mkdir -p $O
export HC_WORKLOAD_CACHE=/tmp/hcw
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-stage --also none > $O/$w.json 2> $O/$w.err || echo "$w failed" >&2
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$w", "ms_per_step", round(d["ms_per_step"], 4), "kernel_ms", round(r["kernel_ms"], 4), "pos/s %.3e" % r["kernel_positions_per_s"], r["kernel"], "parity", d.get("parity", {}).get("ok"))
except Exception as e:
    print("$w", "no line:", e)
PY
done
for w in $PMC; do
  B="python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none"
  pass() { d=$1; shift; rocprofv3 "$@" --kernel-trace --output-format csv -d $O/pmc_$w/$d -- $B > $O/pmc_$w.$d.out 2> $O/pmc_$w.$d.err || echo "pass $d failed" >&2; }
  pass fetch --pmc FETCH_SIZE
  pass write --pmc WRITE_SIZE
  pass l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE
  pass sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
  pass lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU
  python3 - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$O/pmc_$w/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "score_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
c = out
busy = {}
if c.get("GRBM_GUI_ACTIVE"):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    busy["kernel_cycles"] = cyc
    if c.get("TA_BUSY_avr"): busy["ta"] = c["TA_BUSY_avr"] / cyc
    if c.get("SQ_ACTIVE_INST_VALU"): busy["valu"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (4 * 256 * cyc)
    if c.get("SQ_LDS_IDX_ACTIVE"): busy["lds"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
if c.get("SQ_LDS_IDX_ACTIVE"): busy["lds_bank_conflict_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
busy["fabric_bytes"] = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
json.dump({"workload": "$w", "counters_per_launch": out, "busy": busy}, open("$O/pmc_$w.json", "w"), indent=1)
print("$w", json.dumps(busy))
PY
  rm -rf $O/pmc_$w
done
