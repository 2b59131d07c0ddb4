#!/bin/bash
# Is the link busy while the C3 stage runs?  rocprofv3 --memory-copy-trace --kernel-trace over the bench's four stage runs; the copies
# of >= 4 MiB host -> device are the blocks' text: per run (a gap of > 20 ms separates runs) the span from the first copy's start to the
# last one's end, the summed copy time, the rate inside the copies and the gaps between consecutive copies.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/h2d_tl
rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d /tmp/h2d_tl -- python3 bench.py --workload c3 --also none --no-cpu-baseline --steps 3 --warmup 1 > $O/r04_h2d_timeline.json 2> $O/r04_h2d_timeline.err
python3 - > $O/r04_h2d_timeline.txt <<'PY'
import csv, glob, statistics as st
f = glob.glob('/tmp/h2d_tl/**/*memory_copy_trace.csv', recursive=True)
rows = []
for p in f:
    for r in csv.DictReader(open(p)):
        rows.append(r)
print('copy rows', len(rows), 'columns', list(rows[0].keys()) if rows else None)
def num(r, *names):
    for n in names:
        if n in r: return int(r[n])
    raise KeyError(names)
big = []
for r in rows:
    d = r.get('Direction', r.get('Kind', ''))
    s, e = num(r, 'Start_Timestamp'), num(r, 'End_Timestamp')
    b = int(r.get('Bytes', r.get('Size', 0)) or 0)
    if 'HOST_TO_DEVICE' in d.upper().replace(' ', '_') and (b >= (4 << 20) or b == 0): big.append((s, e, b))
big.sort()
print('text copies', len(big))
runs, cur = [], []
for c in big:
    if cur and c[0] - cur[-1][1] > 20_000_000: runs.append(cur); cur = []
    cur.append(c)
if cur: runs.append(cur)
for k, run in enumerate(runs):
    if len(run) < 50: continue
    span = (run[-1][1] - run[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in run) / 1e6
    by = sum(b for _, _, b in run)
    # union of intervals (two copy streams may overlap)
    u, last = 0, None
    for s, e, _ in run:
        if last is None or s > last: u += e - s; last = e
        elif e > last: u += e - last; last = e
    gaps = [max(0, run[i + 1][0] - max(x[1] for x in run[:i + 1][-2:])) / 1e3 for i in range(len(run) - 1)]
    durs = [(e - s) / 1e3 for s, e, _ in run]
    print('run %d: %d copies, %.1f MB, span %.1f ms, summed copy time %.1f ms, link busy (union) %.1f ms, rate inside a copy %.1f GB/s (median copy %.0f us, p90 %.0f), '
          'gaps between copies: median %.0f us, p90 %.0f, max %.0f, idle total %.1f ms' % (
              k, len(run), by / 1e6, span, busy, u / 1e6, by / max(1, busy) / 1e6, st.median(durs), sorted(durs)[int(.9 * len(durs))],
              st.median(gaps), sorted(gaps)[int(.9 * len(gaps))], max(gaps), span - u / 1e6))
PY
cat $O/r04_h2d_timeline.txt
