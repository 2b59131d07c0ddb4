timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests.log 2>&1; grep -E "passed|failed" gpurun_out/r03_gputests.log | tail -2
for w in c3 c2 c5 c4; do bash tools/collect_traffic.sh $w | cut -c1-100; cp gpurun_out/traffic_$w.json profiles/; done
python bench.py > gpurun_out/r03_bench_default.json 2>gpurun_out/r03_bench_default.err
python bench.py --workload c5 --also none --no-cpu-baseline > gpurun_out/r03_bench_c5.json 2>/dev/null
python bench.py --workload c4 --also none --no-cpu-baseline > gpurun_out/r03_bench_c4.json 2>/dev/null
python -c "
import json
for f in ('r03_bench_default','r03_bench_c5','r03_bench_c4'):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d['roofline']; print(f, d['value'], r['kernel_ms'], r['bound'], r['frac'], r['frac_encoded'], r.get('traffic_note'))
d=json.load(open('gpurun_out/r03_bench_default.json')); s=d['stage_end_to_end']; print('stage', s['median']['construct_edges_sorted_s'])
"
