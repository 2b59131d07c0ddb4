#!/bin/bash
# Wrapper around gpurun: records the commit the snapshot is taken from in .build_sha (the GPU box has no .git; the PMC
# collection writes it into the files bench.py later checks), then hands everything to gpurun.
#     tools/gpu.sh --timeout 900 -- 'bash tools/collect_traffic.sh c3'
cd "$(dirname "$0")/.."
sha=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . ':!PROGRESS.jsonl' || sha="$sha-dirty"
echo "$sha" > .build_sha
exec /usr/local/graft/bin/gpurun "$@"
