#!/bin/bash
# Round 6's long soak on the final build: every scenario bit-identical to its oracle / its pinned route, or the run fails.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r06_soak_long.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -E "passed|failed|error" | tail -2 >> $O; }
export HC_FUZZ_SEEDS=12000 HC_FUZZ_STAGE_SEEDS=1200 HC_FUZZ_FINDER_SEEDS=300 HC_FUZZ_FNO_SEEDS=2000 HC_FUZZ_BUCKET_SEEDS=800 HC_FUZZ_STORE_SEEDS=1500
run timeout 5000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage.py tests/test_gpu_overlap_finder.py tests/test_gpu_fno.py tests/test_gpu_c4_c5.py tests/test_gpu_stage_from_store.py -q -m gpu -k "fuzz or random_scenarios or under_add_duplicates or matching_on_the_device"
export HC_FUZZ_SEEDS=6000
HC_COOP_DMA_MIN=1 run timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
cat $O
