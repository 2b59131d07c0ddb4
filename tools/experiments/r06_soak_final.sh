#!/bin/bash
# Round 6 — scorer soak on the round's FINAL kernel sources (after the 49..60-value encoding went in): the fuzz's read sets take all five
# encodings (K in 1..70); default launches, the LDS-DMA forms forced, the wide tables register-staged, quality indices in byte order.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r06_soak_final.txt
: > $O
run() { echo "== $1" >> $O; shift; ( "$@" ) 2>&1 | grep -E "passed|failed|error" | tail -2 >> $O; }
export HC_FUZZ_SEEDS=12000
run "default launches" timeout 3000 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
export HC_FUZZ_SEEDS=6000
HC_COOP_DMA_MIN=1 run "HC_COOP_DMA_MIN=1 (every launch through the LDS-DMA forms)" timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_COOP_DMA_MIN=1 HC_WIDE_DMA=0 run "HC_COOP_DMA_MIN=1 HC_WIDE_DMA=0 (wide tables register-staged)" timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_COOP_DMA_MIN=1 HC_QIDX_ORDER=value run "HC_COOP_DMA_MIN=1 HC_QIDX_ORDER=value (quality indices in byte order)" timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz
HC_FUZZ_STAGE_SEEDS=600 HC_FUZZ_STORE_SEEDS=600 run "the stage and the device-resident stage a" timeout 3000 python -m pytest tests/test_gpu_stage.py tests/test_gpu_stage_from_store.py -q -m gpu -k fuzz
cat $O
