#!/bin/bash
# The wave queue's LOCAL form (tickets in LDS, equal ranges per workgroup; HC_WAVE_QUEUE=3) against the static grid (0) and the global queue (2)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
B="python3 bench.py --also none --no-stage --no-cpu-baseline --steps 30 --warmup 3"
line() { python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(sys.argv[1], '%.4f ms' % d['roofline']['kernel_ms'], 'step %.4f' % d['ms_per_step'], 'edges', d['edges'], flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
" "$1"; }
{
for w in c2 c3-lite c3; do
  HC_WAVE_QUEUE=0 $B --workload $w 2>/dev/null | line "$w static grid"
  HC_WAVE_QUEUE=2 HC_WAVE_QUEUE_STEPS=8 $B --workload $w 2>/dev/null | line "$w global queue steps=8"
  for st in 1 2 4 8; do HC_WAVE_QUEUE=3 HC_WAVE_QUEUE_STEPS=$st $B --workload $w 2>/dev/null | line "$w local tickets steps=$st"; done
done
HC_WAVE_QUEUE=3 HC_COOP_DMA_MIN=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
} > $O/r04_wq_local.txt 2>&1
cat $O/r04_wq_local.txt
