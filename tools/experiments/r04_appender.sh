#!/bin/bash
# The device's copy of a block's admitted records: queued to a thread of its own (default) against issued from the collectors' in-order
# half (HC_APPEND_INLINE=1, the form until round 4; the knob lived from commit 'find-next-overlaps under --add_duplicates ...' until the
# clean-up after this measurement: check that commit out to run the A side again).  C3 stage, four runs each, twice.  Writes gpurun_out/r04_appender.txt.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
mkdir -p $O
run() {
  env "$@" HC_STAGE_TIMING=1 python3 bench.py --workload c3 --also none --no-cpu-baseline --steps 3 --warmup 1 2> $O/r04_appender.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['stage_end_to_end']
print(sys.argv[1], 'construct_edges_sorted_s:', ' '.join('%.4f' % r['construct_edges_sorted_s'] for r in d['runs']), 'median %.4f' % d['median']['construct_edges_sorted_s'], flush=True)
" "$*"
  grep -E "device-parsed pipeline|all blocks scored" $O/r04_appender.err | tail -8
}
{
run HC_X=0
run HC_APPEND_INLINE=1
run HC_X=0
run HC_APPEND_INLINE=1
} > $O/r04_appender.txt 2>&1
cat $O/r04_appender.txt
