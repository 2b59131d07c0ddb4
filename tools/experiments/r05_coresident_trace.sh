#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of a few tools/coresident.py scenarios: which kernel ran when, beside which.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a/trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in "$@"; do
  tag=$(echo $sc | tr ': ' '__')
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -- python3 $R/tools/coresident.py --steps 8 --only "$sc" > $O/$tag.out 2> $O/$tag.err
  f=$(ls $O/$tag/*/*_kernel_trace.csv | head -1)
  python3 - "$f" "$sc" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("score_kernel", "comm_", "sink_compact"))]
keep = keep[-40:]
t0 = int(keep[0]["Start_Timestamp"])
print("==", sys.argv[2])
for r in keep:
    nm = r["Kernel_Name"]
    nm = "SCORE" if "score_kernel" in nm else ("gate" if "gate" in nm else ("compact" if "compact" in nm else nm[:12]))
    print("%-8s start %9.1f us  dur %8.1f us  queue %s" % (nm, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY
done
