#!/bin/bash
# Text block size of the C3 stage: 16 MiB (default) against 32 MiB (and 24): the scoring kernel's 64-candidate items per wave are 1.5 a
# block at 16 MiB (two steps for half the waves) and 3.0 at 32 MiB, and every per-block launch is paid half as often.  Stage time (four
# files, alternating, three rounds) and, under rocprofv3, the device's time per file.  Writes gpurun_out/r04_block_size.txt.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() {
  env "$@" HC_STAGE_TIMING=1 python3 tools/stage_profile.py --workload c3 --reps 4 2> $O/r04_block_size.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['stage']
print(sys.argv[1], 'construct_edges_sorted_s:', ' '.join('%.4f' % r['construct_edges_sorted_s'] for r in d['runs']), 'median %.4f' % d['median']['construct_edges_sorted_s'], flush=True)
" "$*"
}
{
for round in 1 2 3; do
  run HC_X=default
  run HC_TEXT_BLOCK=33554432
  run HC_TEXT_BLOCK=33554432 HC_TEXT_DEPTH=6
  run HC_TEXT_BLOCK=25165824
done
for cfg in "HC_X=default" "HC_TEXT_BLOCK=33554432"; do
  rm -rf /tmp/stk
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stk -- python3 tools/stage_profile.py --workload c3 --reps 4 > /dev/null 2>&1
  echo "== $cfg"; python3 tools/stage_profile.py --summarize /tmp/stk --reps 4 | head -14
done
} > $O/r04_block_size.txt 2>&1
cat $O/r04_block_size.txt
