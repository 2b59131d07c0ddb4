#!/bin/bash
# A/B on ONE box: the text parser's window with and without the two-word read-ahead (HC_TEXT_READ_AHEAD), device time of the C3 stage per
# (the HC_TEXT_READ_AHEAD switch was part of the withdrawn change: Cursor::load read word(a + 4), word(a + 8) as well, Cursor::next shifted
#  w <- w1 <- w2 <- word(at + 8); the product has the one-word window only — profiles/r06_stage_device_leg.md)
# file by kernel (rocprofv3 --kernel-trace --stats over four files), twice each, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/haploconduct_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=$(ls build/*.o build/host/*.o build/cli/*.o | grep -v hc_text_kernels.hip.o)
export HC_WORKLOAD_CACHE=/tmp/hcw
for A in 0 1 0 1; do
  /opt/rocm/bin/hipcc $FLAGS -DHC_TEXT_READ_AHEAD=$A -x hip -c -o build/hc_text_kernels.hip.o hc_text_kernels.hip 2>/dev/null || { echo "compile failed"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhcedge.so build/hc_text_kernels.hip.o $OBJS
  (cd $R && TAG=ra$A bash tools/experiments/r04_stage_kernels.sh 2>&1 | grep -E "device time|text_parse|text_lines" | sed "s/^/read_ahead=$A  /")
done
