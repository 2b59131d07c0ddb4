#!/bin/bash
# A/B of the wave queue (HC_WAVE_QUEUE=0|1) on the LDS-DMA form: kernel ms at C2 / C3-lite / C3, items of 1..8 steps
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py tests/test_gpu_golden_and_properties.py tests/test_gpu_reference_patch.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r04_wq_tests.txt
HC_WAVE_QUEUE=2 HC_COOP_DMA_MIN=1 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_row_sink.py tests/test_gpu_golden_and_properties.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r04_wq_tests_forced.txt
python3 -m pytest tests/test_gpu_stage.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r04_stage_tests.txt
B="python3 bench.py --also none --no-stage --no-cpu-baseline --steps 30 --warmup 3"
line() { python3 -c "
import sys, json
name = sys.argv[1]
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(name, '%.4f ms' % d['roofline']['kernel_ms'], 'step %.4f' % d['ms_per_step'], 'edges', d['edges'], flush=True)
except Exception as e:
    print(name, 'FAILED', e, flush=True)
" "$1"; }
{
for w in c2 c3-lite c3; do
  HC_WAVE_QUEUE=0 $B --workload $w 2>/dev/null | line "$w static grid"
  HC_WAVE_QUEUE=2 $B --workload $w 2>/dev/null | line "$w wave queue (auto steps)"
  for st in 2 4 8; do HC_WAVE_QUEUE=2 HC_WAVE_QUEUE_STEPS=$st $B --workload $w 2>/dev/null | line "$w wave queue steps=$st"; done
done
} > $O/r04_wq.txt 2>&1
# the stage's tail: the resolved graph fetched whole and then adopted (round 3) against fetched in pieces with the host adopting behind the copy
for pb in 0 33554432; do
  HC_STAGE_TIMING=1 HC_FETCH_PIECE_BYTES=$pb python3 bench.py --workload c3 --also none --no-cpu-baseline --steps 3 --warmup 1 > $O/r04_stage_pieces_$pb.json 2> $O/r04_stage_pieces_$pb.err
done
grep -h "device resolve\|took" $O/r04_stage_pieces_0.err | sed 's/^/whole:  /' > $O/r04_stage_pieces.txt
grep -h "device resolve\|took" $O/r04_stage_pieces_33554432.err | sed 's/^/pieces: /' >> $O/r04_stage_pieces.txt
{
echo "# quality-trimmed PAIRS (mates of 60..150 bp, 1.1 * 10^6 p-p candidates): cooperative / per lane / bucketed"
w=c2t
$B --workload $w 2>/dev/null | line "$w default"
HC_FETCH_GROUP=2 $B --workload $w 2>/dev/null | line "$w per-lane G=2"
HC_FETCH_GROUP=4 $B --workload $w 2>/dev/null | line "$w per-lane G=4"
HC_BALANCE=1 $B --workload $w 2>/dev/null | line "$w bucketed"
HC_BALANCE=0 $B --workload $w 2>/dev/null | line "$w plain cooperative"
} > $O/r04_dispatch_pairs.txt 2>&1
python3 tools/c1_process.py --reps 5 > $O/r04_c1_process.json 2> $O/r04_c1_process.err
cat $O/r04_wq_tests.txt $O/r04_wq_tests_forced.txt $O/r04_stage_tests.txt $O/r04_wq.txt $O/r04_stage_pieces.txt $O/r04_dispatch_pairs.txt
