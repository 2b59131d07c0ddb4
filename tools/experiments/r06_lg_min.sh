#!/bin/bash
# Round 6 — do the smallest quality alphabets (K <= 6: the sparse 16 KiB table, LG = 3) run faster through the dense 4 KiB planes (LG = 4)?
# Patches lut_lg() in the GPU BOX's scratch copy of hc_device.h (the repository's file stays as it is: the PMC files are keyed by its hash),
# rebuilds with -DHC_LG_MIN=3 / 4 and times the workloads (in-run parity stays on: both are products).
# Result (one box, two rounds): c3 6.69 / 6.60 ms with LG = 3 against 6.57 / 6.53 with LG = 4, c3q4 6.74 / 6.74 against 6.57 / 6.59 — the dense
# planes are 1 - 2 % faster on 10^8 candidates — but c2 0.156 / 0.158 against 0.162 / 0.165 and c5 0.604 / 0.607 against 0.623 / 0.622: 3 - 5 %
# slower on 2 * 10^6.  Not switched.
#     tools/gpu.sh --timeout 1500 -- 'bash tools/experiments/r06_lg_min.sh "c3 c2 c3q4 c5"'
WL=${1:-"c3 c2 c3q4"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export HC_WORKLOAD_CACHE=/tmp/hcw
cd $R/haploconduct_amd/csrc
python3 - <<PY
p = "hc_device.h"
s = open(p).read()
old = "__host__ __device__ inline uint32_t lut_lg(uint32_t K) { return K + 2 <= 8 ? 3u :"
new = "#ifndef HC_LG_MIN\n#define HC_LG_MIN 3\n#endif\n__host__ __device__ inline uint32_t lut_lg(uint32_t K) { return K + 2 <= 8 && HC_LG_MIN <= 3 ? 3u :"
assert old in s
open(p, "w").write(s.replace(old, new))
PY
for round in 1 2; do
for M in 3 4; do
  make -s clean >/dev/null 2>&1
  make -s -j16 EXTRA="-DHC_LG_MIN=$M" >/dev/null 2>$R/gpurun_out/lg_min_cc_$M.err || { echo "HC_LG_MIN=$M: build failed"; tail -3 $R/gpurun_out/lg_min_cc_$M.err; continue; }
  for w in $WL; do
    (cd $R && python3 bench.py --workload $w --also none --no-stage --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HC_LG_MIN=$M', '$w', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'][:60], d['parity']['parity'][:9])")
  done
done
done
