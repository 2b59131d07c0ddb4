#!/bin/bash
# A/B of two builds of libhcedge.so (build/ab/libhcedge_old.so, libhcedge_new.so) on the C3 STAGE (four files per run): alternating, three rounds, one box
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
cp haploconduct_amd/csrc/libhcedge.so /tmp/libhcedge_keep.so
{
for round in 1 2 3; do for v in old new; do
  cp build/ab/libhcedge_$v.so haploconduct_amd/csrc/libhcedge.so
  HC_STAGE_TIMING=1 python3 tools/stage_profile.py --workload c3 --reps 4 2> $O/r04_ab_stage.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['stage']
print(sys.argv[1], 'construct_edges_sorted_s:', ' '.join('%.4f' % r['construct_edges_sorted_s'] for r in d['runs']), 'median %.4f' % d['median']['construct_edges_sorted_s'], flush=True)
" "$v"
  grep -E "all blocks scored" $O/r04_ab_stage.err | tr '\n' ' '; echo
done; done
} > $O/r04_ab_stage.txt 2>&1
cp /tmp/libhcedge_keep.so haploconduct_amd/csrc/libhcedge.so
cat $O/r04_ab_stage.txt
