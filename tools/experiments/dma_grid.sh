B="python bench.py --also none --no-stage --no-cpu-baseline --steps 20"
for w in c3-lite c3; do for g in 16 32 64 256 100000; do HC_GRID_MULT=$g timeout 400 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w HC_GRID_MULT=$g', round(d['roofline']['kernel_ms'],4))"; done; done
