#!/bin/bash
# The round's closing run on one GPU box: the GPU suite, the bench lines (roofline from the committed PMC files when they still
# describe the kernel sources), reads -> graph, the finder and find-next-overlaps.  Everything lands in gpurun_out/.
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests.log 2>&1; grep -E "passed|failed" gpurun_out/r03_gputests.log | tail -2
python bench.py > gpurun_out/r03_bench_default.json 2>gpurun_out/r03_bench_default.err
python bench.py --workload c5 --also none --no-cpu-baseline > gpurun_out/r03_bench_c5.json 2>/dev/null
python bench.py --workload c4 --also none --no-cpu-baseline > gpurun_out/r03_bench_c4.json 2>/dev/null
HC_STAGE_TIMING=1 HC_SFO_TIMING=1 HC_FIND_TIMING=1 python tools/reads_to_graph.py --workload c3 --one-call > gpurun_out/r2g_one.jsonl 2> gpurun_out/r2g_one.err
python tools/reads_to_graph.py --workload c3 > gpurun_out/r2g_file.jsonl 2>/dev/null
python tools/finder_bench.py --workload c3 > gpurun_out/finder_c3.jsonl 2>/dev/null
python tools/fno_bench.py > gpurun_out/fno_bench.jsonl 2>/dev/null
python -c "
import json
for f in ('r03_bench_default','r03_bench_c5','r03_bench_c4'):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']; print(f, d['value'], r['kernel_ms'], r['bound'], r['frac'], r['frac_encoded'], r.get('traffic_note'))
d=json.load(open('gpurun_out/r03_bench_default.json')); s=d['stage_end_to_end']; print('stage', s['median']['construct_edges_sorted_s'], s['median']['open_s'])
print(open('gpurun_out/r2g_one.jsonl').read()[300:900])
print(open('gpurun_out/finder_c3.jsonl').read()[200:420])
print(open('gpurun_out/fno_bench.jsonl').read()[:700])
"
