B="python bench.py --also none --no-stage --no-cpu-baseline --steps 30 --workload c5"
for d in 0 1; do HC_COOP_DEPTH=$d timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 HC_COOP_DEPTH=$d', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])"; done
