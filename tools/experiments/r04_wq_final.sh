#!/bin/bash
# The round's final scoring kernel (LDS tickets + two explicit v_bitop3 per four positions): parity, kernel ms by size, and where the LDS-DMA form
# should start (HC_COOP_DMA_MIN) now that its waves balance themselves
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py tests/test_gpu_golden_and_properties.py tests/test_gpu_dispatch.py tests/test_gpu_c4_c5.py -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -4 > $O/r04_wq_final_tests.txt
HC_COOP_DMA_MIN=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py tests/test_gpu_stage.py -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -4 >> $O/r04_wq_final_tests.txt
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
for w in c3 c3-lite c2; do
  $B --workload $w 2>/dev/null | line "$w tickets"
  HC_WAVE_QUEUE=0 $B --workload $w 2>/dev/null | line "$w static grid"
done
for w in c2-100k c2-small c2-mid; do
  $B --workload $w 2>/dev/null | line "$w default (register-staged below 500000)"
  HC_COOP_DMA_MIN=1 $B --workload $w 2>/dev/null | line "$w LDS-DMA + tickets"
done
} > $O/r04_wq_final.txt 2>&1
cat $O/r04_wq_final_tests.txt $O/r04_wq_final.txt
