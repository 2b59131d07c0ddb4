#!/bin/bash
# Tickets in the register-staged cooperative launches (small launches, wide 8-bit and 16-bit symbol tables): against the static grid, and against
# the LDS-DMA form where both exist
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
B="python3 bench.py --also none --no-stage --no-cpu-baseline --steps 30 --warmup 3"
line() { python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(sys.argv[1], '%.4f ms' % d['roofline']['kernel_ms'], 'step %.4f' % d['ms_per_step'], d['roofline']['kernel'].split('<')[1], flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
" "$1"; }
{
for w in c2-100k c2-small c2-mid; do
  $B --workload $w 2>/dev/null | line "$w register-staged + tickets"
  HC_WAVE_QUEUE=0 $B --workload $w 2>/dev/null | line "$w register-staged, static grid"
  HC_COOP_DMA_MIN=1 $B --workload $w 2>/dev/null | line "$w LDS-DMA + tickets"
done
for k in 35 60 25; do
  HC_C4_K=$k $B --workload c4 2>/dev/null | line "c4 K=$k tickets"
  HC_C4_K=$k HC_WAVE_QUEUE=0 $B --workload c4 2>/dev/null | line "c4 K=$k static grid"
done
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py tests/test_gpu_golden_and_properties.py tests/test_gpu_c4_c5.py tests/test_gpu_stage.py -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -4
} > $O/r04_tickets_reg.txt 2>&1
cat $O/r04_tickets_reg.txt
