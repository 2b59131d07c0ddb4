B="python bench.py --also none --no-stage --no-cpu-baseline --steps 20"
for w in c5 c3-lite; do for k in 0 3 2 1; do
  HC_COOP_WG_PER_CU=$k timeout 300 $B --workload $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w per_cu=$k', round(d['roofline']['kernel_ms'],4))"
done; done
