#!/bin/bash
# A/B of two builds of libhcedge.so (build/ab/libhcedge_old.so, libhcedge_new.so) on the kernel time of C3 / C2: alternating, three rounds
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
B="python3 bench.py --also none --no-stage --no-cpu-baseline --steps 30 --warmup 3"
line() { python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(sys.argv[1], '%.4f ms' % d['roofline']['kernel_ms'], 'step %.4f' % d['ms_per_step'], flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e, flush=True)
" "$1"; }
cp haploconduct_amd/csrc/libhcedge.so /tmp/libhcedge_keep.so
{
for round in 1 2 3; do for v in old new; do
  cp build/ab/libhcedge_$v.so haploconduct_amd/csrc/libhcedge.so
  $B --workload c3 2>/dev/null | line "c3 $v"
  $B --workload c2 2>/dev/null | line "c2 $v"
done; done
} > $O/r04_ab_so.txt 2>&1
cp /tmp/libhcedge_keep.so haploconduct_amd/csrc/libhcedge.so
cat $O/r04_ab_so.txt
