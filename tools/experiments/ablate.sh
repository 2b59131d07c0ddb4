#!/bin/bash
# What each part of the cooperative scoring kernel costs: rebuilds hc_kernels.hip with HC_ABLATE = a bit mask of parts cut out
# (1 table reads, 2 row loads from memory, 4 the rows' passage through the LDS image; results are garbage) ON THE GPU BOX's
# scratch copy and times the C3 launch (round 6: 8 = no A rows fetched, 24 = B rows only with two steps in flight).  Prints one line per mask.   gpurun -- bash tools/experiments/ablate.sh c3 "0 1 2 4 3 7"
W=${1:-c3}
MASKS=${2:-"0 1 2 4 3 6 7"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/haploconduct_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=$(ls build/*.o build/host/*.o | grep -v hc_kernels.hip.o)
for A in $MASKS; do
  /opt/rocm/bin/hipcc $FLAGS -DHC_ABLATE=$A -x hip -c -o build/hc_kernels.hip.o hc_kernels.hip 2>$R/gpurun_out/ablate_cc_$A.err || { echo "mask $A: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhcedge.so build/hc_kernels.hip.o $OBJS
  (cd $R && HC_BENCH_ABLATION=1 HC_WORKLOAD_CACHE=/tmp/hcw timeout 600 python bench.py --workload $W --also none --no-stage --no-cpu-baseline --steps 10 2>$R/gpurun_out/ablate_$A.err | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$W HC_ABLATE=$A kernel_ms', round(d['roofline']['kernel_ms'],4))")
done
