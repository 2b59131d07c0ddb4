#!/bin/bash
# Round 6 — LDS bank conflicts of the wide table's reads: the searched index assignment (hc_device.h: kWideRankLabel) against indices in byte
# order (HC_QIDX_ORDER=value), same kernel, same table layout.  c3q35, 3 steps, SQ counters only.
#     tools/gpu.sh --timeout 900 -- 'bash tools/experiments/r06_wide_labels_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_wide_labels
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HC_WORKLOAD_CACHE=/tmp/hcw
W=${1:-c3q35}
for order in searched value; do
  if [ $order = value ]; then export HC_QIDX_ORDER=value; else unset HC_QIDX_ORDER; fi
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/$order -- \
    python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none > $O/$order.out 2> $O/$order.err
  python3 $R/bench.py --workload $W --steps 30 --warmup 3 --no-cpu-baseline --no-stage --also none 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$order', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
python3 - <<PY
import csv, glob, collections
for order in ("searched", "value"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/%s/*/*_counter_collection.csv" % order):
        for r in csv.DictReader(open(f)):
            if "score_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    print(order, {k: "%.4g" % v for k, v in a.items()}, "conflict share %.3f" % (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]),
          "LDS cycles per LDS instruction %.2f" % (a["SQ_LDS_IDX_ACTIVE"] / a["SQ_INSTS_LDS"]))
PY
