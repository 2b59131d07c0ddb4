#!/bin/bash
# Device time of the C3 stage per file, by kernel (rocprofv3 --kernel-trace --stats over four files).  Writes gpurun_out/r05_stage_kernels_${TAG}.txt + the stats csv.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
TAG=${TAG:-now}
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/stk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stk -- python3 tools/stage_profile.py --workload c3 --reps 4 > $O/r05_stage_kernels_${TAG}.json 2> $O/r05_stage_kernels_${TAG}.err
python3 tools/stage_profile.py --summarize /tmp/stk --reps 4 > $O/r05_stage_kernels_${TAG}.txt
cp $(find /tmp/stk -name "*kernel_stats.csv" | head -1) $O/r05_stage_kernel_stats_c3_${TAG}.csv
cat $O/r05_stage_kernels_${TAG}.txt
