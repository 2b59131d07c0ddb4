#!/usr/bin/env python3
"""Soak of the two device forms added in round 2 against their pinned host forms, many random inputs:
find-next-overlaps (HC_FNO=device vs the sequential oracle) and hc_found_to_overlaps (vs hc_sfo_records_to_overlaps).
Prints one JSON line; exit code 1 on the first difference."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import haploconduct_amd as hc  # noqa: E402
from haploconduct_amd import fno as F, host  # noqa: E402
from tests import _fno as T  # noqa: E402
from tests.test_gpu_overlap_finder import make_reads  # noqa: E402


def main():
    n_fno = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_sfo = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    olib = T.load_oracle()
    t0 = time.time()
    os.environ["HC_FNO"] = "device"
    lines = 0
    for seed in range(1000, 1000 + n_fno):
        rng = np.random.default_rng(seed)
        flags = [F.RESOLVE_ORIENTATIONS, F.RESOLVE_ORIENTATIONS | F.NO_INCLUSIONS, F.RESOLVE_ORIENTATIONS | F.OPTIMIZE, 0, F.NO_INCLUSIONS][seed % 5]
        inp = T.fno1_scenario(seed, n_nodes=int(rng.integers(20, 400)), n_srs=int(rng.integers(5, 120)), n_edges=int(rng.integers(50, 3000)),
                              with_extras=bool(seed % 2), flags=flags, paired_frac=float(rng.choice([0.0, 0.3, 0.7, 1.0])), n_threads=int(rng.choice([1, 0, 5])))
        try:
            want, wc = T.oracle_fno1(olib, inp)
        except T.OracleAbort:
            try:
                F.find_next_overlaps(inp)
            except hc.HcError:
                continue
            print(json.dumps({"fno_seed": seed, "problem": "the oracle stops, the device form does not"}))
            return 1
        got, gc = F.find_next_overlaps(inp)
        if not F.last_on_device or got != want or gc != wc:
            print(json.dumps({"fno_seed": seed, "problem": "device form differs", "on_device": F.last_on_device}))
            return 1
        lines += gc["n_lines"]
    os.environ.pop("HC_FNO", None)
    t1 = time.time()
    d = tempfile.mkdtemp(prefix="hcsoak_") + "/"
    sfo_lines = 0
    for seed in range(2000, 2000 + n_sfo):
        rng = np.random.default_rng(seed)
        ns, npair = int(rng.integers(0, 500)), int(rng.integers(0, 600))
        if ns + npair < 50:
            npair += 100
        err = float(rng.choice([0.0, 0.01, 0.02]))
        reads = make_reads(seed, n_single=ns, n_pair=npair, glen=int(rng.integers(800, 4000)), lo=int(rng.integers(80, 140)), hi=int(rng.integers(150, 320)),
                           err=err, n_rate=float(rng.choice([0.0, 0.002])), rc_frac=float(rng.choice([0.0, 0.5])), repeat=bool(seed % 3 == 0))
        with hc.EdgeScorer(hc.Settings()) as sc:
            sc.set_reads(reads)
            recs = sc.find_overlaps(0.0 if err == 0.0 else 0.03, 60 if err == 0.0 else 70)
            want_n = host.sfo_records_to_overlaps(recs, d + "want.txt", ns, npair)
            os.environ["HC_SFO_CHUNK"] = str(int(rng.choice([1, 2, 5, 64, 1000, 1 << 21])))
            os.environ["HC_SFO_BUCKETS"] = str(int(rng.choice([1, 2, 9, 100])))
            try:
                got_n = sc.found_to_overlaps(d + "got.txt", ns, npair)
            finally:
                os.environ.pop("HC_SFO_CHUNK", None)
                os.environ.pop("HC_SFO_BUCKETS", None)
        if got_n != want_n or open(d + "got.txt", "rb").read() != open(d + "want.txt", "rb").read():
            print(json.dumps({"sfo_seed": seed, "problem": "hc_found_to_overlaps differs", "records": int(recs.size)}))
            return 1
        sfo_lines += want_n
    print(json.dumps({"fno_scenarios": n_fno, "fno_lines": int(lines), "fno_s": round(t1 - t0, 1), "sfo_read_sets": n_sfo, "overlap_lines": int(sfo_lines),
                      "sfo_s": round(time.time() - t1, 1), "differences": 0}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
