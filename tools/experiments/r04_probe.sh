#!/bin/bash
# Round-4 measurement session on one MI355X: (A) grid shape of small launches, (B) kernel dispatch by read-set shape, (C) where the stage's
# time goes, (D) a rank's step at the strong split's shard sizes.  Writes gpurun_out/r04_probe_*.txt.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
mkdir -p $O
B="python3 bench.py --also none --no-stage --no-cpu-baseline --steps 30 --warmup 3"
line() { python3 -c "
import sys, json
name = sys.argv[1]
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(name, '%.4f ms' % d['roofline']['kernel_ms'], 'step %.4f' % d['ms_per_step'], d['roofline']['kernel'], flush=True)
except Exception as e:
    print(name, 'FAILED', e, flush=True)
" "$1"; }
{
echo "# (A) C2 (2*10^6 candidates): workgroups per resident slot"
for m in 1 2 4 8 16; do HC_GRID_MULT=$m $B --workload c2 2>/dev/null | line "c2 dma grid_mult=$m"; done
for m in 1 2 4; do HC_COOP_DMA=0 HC_GRID_MULT=$m $B --workload c2 2>/dev/null | line "c2 register-staged grid_mult=$m"; done
} > $O/r04_probe_grid.txt 2>&1
{
echo "# (B) kernel by read-set shape, 2*10^6 s-s candidates each: cooperative (default) / per lane G=2 / G=4 / per lane + block balancing / bucketed cooperative"
for w in c1s c5s c5t c5m c4; do
  $B --workload $w 2>/dev/null | line "$w default"
  HC_FETCH_GROUP=2 $B --workload $w 2>/dev/null | line "$w per-lane G=2"
  HC_FETCH_GROUP=4 $B --workload $w 2>/dev/null | line "$w per-lane G=4"
  HC_FETCH_GROUP=2 HC_BALANCE=1 $B --workload $w 2>/dev/null | line "$w per-lane G=2 BAL"
  HC_BALANCE=1 $B --workload $w 2>/dev/null | line "$w bucketed"
  HC_BALANCE=0 $B --workload $w 2>/dev/null | line "$w plain cooperative"
done
} > $O/r04_probe_dispatch.txt 2>&1
HC_STAGE_TIMING=1 python3 bench.py --workload c3 --also none --no-cpu-baseline --steps 3 --warmup 1 > $O/r04_probe_stage.json 2> $O/r04_probe_stage.err
python3 tools/predict_scaling.py --steps 20 > $O/r04_predicted_scaling.json 2> $O/r04_predicted_scaling.err
tail -5 $O/r04_probe_grid.txt $O/r04_probe_dispatch.txt
