#!/bin/bash
# Round 6 — where the wide-table kernel's LDS cycles go: the product against a build whose table reads are replaced by arithmetic
# (HC_ABLATE=1; results are garbage, timing and counters only).  c3q35, LDS counters + hipEvents time.
#     tools/gpu.sh --timeout 1200 -- 'bash tools/experiments/r06_wide_lds_split.sh'
W=${1:-c3q35}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_wide_lds_split
rm -rf $O; mkdir -p $O
export HC_WORKLOAD_CACHE=/tmp/hcw
cd $R/haploconduct_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function"
OBJS=$(ls build/*.o build/host/*.o | grep -v hc_kernels.hip.o)
for A in 0 1; do
  /opt/rocm/bin/hipcc $FLAGS -DHC_ABLATE=$A -x hip -c -o build/hc_kernels.hip.o hc_kernels.hip 2>$O/cc_$A.err || { echo "mask $A: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhcedge.so build/hc_kernels.hip.o $OBJS
  (cd /tmp && TMPDIR=/tmp HC_BENCH_ABLATION=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a$A -- \
    python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-stage --also none > $O/a$A.out 2> $O/a$A.err)
  (cd $R && HC_BENCH_ABLATION=1 python3 bench.py --workload $W --also none --no-stage --no-cpu-baseline --steps 20 2>$O/t$A.err | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W HC_ABLATE=$A kernel_ms', round(d['roofline']['kernel_ms'],4))")
done
python3 - <<PY
import csv, glob, collections
for A in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/a%d/*/*_counter_collection.csv" % A):
        for r in csv.DictReader(open(f)):
            if "score_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a = {k: sum(v) / len(v) for k, v in agg.items()}
    if a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        print("HC_ABLATE=%d" % A, {k: "%.4g" % v for k, v in a.items()}, "LDS busy %.3f VALU busy %.3f" % (a["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, a["SQ_ACTIVE_INST_VALU"] / 1024 / cyc))
PY
