#!/bin/bash
# One rocprofv3 PMC pass (issue counters) + one kernel-trace pass of the scoring kernel on a workload; prints the per-launch means.
#     gpurun -- bash tools/experiments/sq_pass.sh c3
W=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq_$W.d
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.out 2> $O/sq.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/l -- $B > $O/l.out 2> $O/l.err
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for d in glob.glob("$O/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(d)):
        if "score_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$W", {k: sum(v) / len(v) for k, v in sorted(agg.items())})
PY
