#!/bin/bash
# Round 6 — which allocations a stalled reads -> graph call waits in: HIP API trace (with arguments) over tools/experiments/r06_find_stall.py;
# every hipMalloc / hipFree / hipHostMalloc / hipHostFree of 32 MiB and more in time order with its duration, and every API call of 50 ms and more.
#     tools/gpu.sh --timeout 900 -- 'bash tools/experiments/r06_find_stall_trace.sh 4'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_find_stall
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export HC_WORKLOAD_CACHE=/tmp/hcw
T=/tmp/r06_find_stall_t
rm -rf $T
(cd $R && rocprofv3 --hip-trace --output-format json -d $T -- python3 $R/tools/experiments/r06_find_stall.py ${1:-4} > $O/run.out 2> $O/run.err)
tail -2 $O/run.out
python3 - > $O/allocations.txt <<PY
import json, glob
for f in glob.glob("$T/*/*results.json"):
    r = json.load(open(f))["rocprofiler-sdk-tool"][0]
    api = sorted(r["buffer_records"]["hip_api"], key=lambda x: x["start_timestamp"])
    t0 = api[0]["start_timestamp"]
    live = {}
    for x in api:
        a = {y["name"]: y["value"] for y in x.get("args", [])}
        d = (x["end_timestamp"] - x["start_timestamp"]) * 1e-6
        t = (x["start_timestamp"] - t0) * 1e-9
        names = set(a)
        what = None
        if names == {"ptr", "size"} or names == {"ptr", "size", "flags"}:
            sz = int(a["size"])
            what = ("alloc(flags)" if "flags" in a else "hipMalloc") + " %d MiB" % (sz >> 20)
            live[a["ptr"]] = sz
            if sz < (32 << 20) and d < 50:
                continue
        elif names == {"ptr"}:
            what = "free"
            if d < 50:
                continue
        elif d < 50:
            continue
        print("%9.3f s  %9.3f ms  %s %s" % (t, d, what or "api", "" if what else sorted(a)))
PY
cat $O/allocations.txt | tail -${2:-120}
rm -rf $T
