#!/bin/bash
# Upper bound of what "text_parse_kernel writes no hc_line_rec"  can save: an ablation build in which the parse kernel
# leaves the parsed lines unwritten (the kept rows' lines are then garbage: timing only), A/B on ONE box against the product, device time of the
# C3 stage per file by kernel (rocprofv3 --kernel-trace --stats over four files), twice each, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/haploconduct_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=$(ls build/*.o build/host/*.o build/cli/*.o | grep -v hc_text_kernels.hip.o)
export HC_WORKLOAD_CACHE=/tmp/hcw
for A in 0 1 0 1; do
  /opt/rocm/bin/hipcc $FLAGS -DHC_TEXT_ABLATE_LINES=$A -x hip -c -o build/hc_text_kernels.hip.o hc_text_kernels.hip 2>/dev/null || { echo "compile failed"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhcedge.so build/hc_text_kernels.hip.o $OBJS
  (cd $R && TAG=nl$A bash tools/experiments/r04_stage_kernels.sh 2>&1 | grep -E "device time|text_parse|kept_scatter" | sed "s/^/no_lines=$A  /")
done
