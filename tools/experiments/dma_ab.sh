B="python bench.py --also none --no-stage --no-cpu-baseline --steps 20"
for w in c2-small c2 c3-lite c3; do for v in 0 1; do HC_COOP_DMA=$v timeout 400 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w HC_COOP_DMA=$v', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])"; done; done
