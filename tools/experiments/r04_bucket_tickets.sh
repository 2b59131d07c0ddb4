#!/bin/bash
# Bucketed (length-sorted) launches by ticket (HC_BUCKET_TICKETS=1: no global queue atomic, no barrier per piece) against the workgroup queue
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
for round in 1 2; do for w in c5 c5m c5t; do
  $B --workload $w 2>/dev/null | line "$w workgroup queue"
  HC_BUCKET_TICKETS=1 $B --workload $w 2>/dev/null | line "$w tickets"
done; done
HC_BUCKET_TICKETS=1 python3 -m pytest tests/test_gpu_c4_c5.py tests/test_gpu_row_sink.py tests/test_gpu_dispatch.py -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -4
} > $O/r04_bucket_tickets.txt 2>&1
cat $O/r04_bucket_tickets.txt
