#!/bin/bash
# The C3 stage under its pipeline knobs (blocks in flight, collectors, block size): when all blocks are consumed / the graph is resolved.
#     gpurun -- bash tools/experiments/stage_knobs.sh
run() { echo "== $*"; env "$@" HC_STAGE_TIMING=1 python tools/stage_bench.py --workload c3 --threads 32 --reps 3 --oracle-lines 1000 2>&1 | grep -E "all blocks scored|graph resolved at|producer:" | cut -c1-260; }
run HC_NOP=1
run HC_TEXT_DEPTH=16
run HC_COLLECTORS=6
run HC_COLLECTORS=8 HC_TEXT_DEPTH=16
run HC_TEXT_BLOCK=33554432
run HC_TEXT_BLOCK=8388608 HC_TEXT_DEPTH=16
