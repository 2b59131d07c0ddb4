#!/usr/bin/env python3
"""The few figures of a bench.py line one looks at first.  usage: summarize_bench.py <file with the JSON line last>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")})
print({k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "kernel_ms", "busiest_unit")})
if "also" in d and "c2" in d["also"]:
    print("c2 ms_per_step", d["also"]["c2"]["ms_per_step"])
if "stage_end_to_end" in d:
    s = d["stage_end_to_end"]
    print("stage median", round(s["median"]["construct_edges_sorted_s"], 4), [round(x["construct_edges_sorted_s"], 4) for x in s["runs"]])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu_baseline", c.get("value"), c.get("kind"), c.get("cores"), "stage:", (c.get("stage") or {}).get("value"))
print("parity", (d.get("parity") or {}).get("digest_matches_untimed_launch"), (d.get("parity") or {}).get("parity_checked_records"))
