#!/bin/bash
# Round 6's closing fuzz soak on the device (every scenario bit-identical to its oracle / its pinned route, or the run fails).
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r06_soak.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -E "passed|failed|error" | tail -2 >> $O; }
export HC_FUZZ_SEEDS=3000 HC_FUZZ_STAGE_SEEDS=400 HC_FUZZ_FINDER_SEEDS=100 HC_FUZZ_FNO_SEEDS=1000 HC_FUZZ_BUCKET_SEEDS=200 HC_FUZZ_STORE_SEEDS=300
run timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage.py tests/test_gpu_overlap_finder.py tests/test_gpu_fno.py tests/test_gpu_c4_c5.py tests/test_gpu_stage_from_store.py -q -m gpu -k "fuzz or random_scenarios or under_add_duplicates or matching_on_the_device"
export HC_FUZZ_SEEDS=800
HC_COOP_DMA_MIN=1 run timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_COOP_DMA_MIN=1 HC_WIDE_DMA=0 run timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_COOP_DMA_MIN=1 HC_QIDX_ORDER=value run timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_WAVE_QUEUE=0 run timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_REGULAR_STORE=0 run timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_TEXT_BLOCK=65536 HC_TEXT_DEPTH=3 HC_FUZZ_STAGE_SEEDS=200 HC_FUZZ_STORE_SEEDS=200 run timeout 1200 python -m pytest tests/test_gpu_stage.py tests/test_gpu_stage_from_store.py -q -m gpu -k fuzz
cat $O
