#!/usr/bin/env python3
"""Generates tests/golden/*.json by EXECUTING GENUINE REFERENCE CODE in the build
container (needs /root/reference and `make -C oracle ref`):

  ref_headers.json   from oracle/_ref/libhcref_headers.so — the reference's own Boost-free
                     headers Overlap.h / Types.h / Read.h / Edge.h compiled where they lie.
  ref_fragment.json  from oracle/_ref/libhcref_fragment.so — FRAGMENT PROBE: lines 26-139 of
                     src/EdgeCalculator.cpp (score, phred_to_prob, overlap_score) piped verbatim
                     into the compiler; see oracle/ref_fragment_prelude.inc for what that is and is not.

The full reference (EdgeCalculator.cpp as a TU, compute_overlap, process_overlaps,
construct_edges) cannot be built here (Boost absent), so no vectors exist for those.
Doubles are stored as C99 hex strings so the fixtures pin bit patterns.
Inputs are generated with seeded RNGs; this script is the only thing needed to
regenerate the fixtures:   python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
H = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_headers.so"))
F = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhcref_fragment.so"))


class ref_overlap_out(C.Structure):
    _fields_ = [("id1", C.c_ulong), ("id2", C.c_ulong), ("pos1", C.c_int), ("pos2", C.c_int), ("ord", C.c_char),
                ("ori1", C.c_char), ("ori2", C.c_char), ("type1", C.c_char), ("type2", C.c_char),
                ("perc", C.c_uint), ("len1", C.c_uint), ("len2", C.c_uint), ("line", C.c_char * 256)]


class ref_edge_io(C.Structure):
    _fields_ = [("score", C.c_double), ("mismatch", C.c_double), ("pos1", C.c_int), ("pos2", C.c_int),
                ("pos3", C.c_int), ("pos4", C.c_int), ("ori1", C.c_int), ("ori2", C.c_int), ("ord", C.c_char),
                ("v1", C.c_ulong), ("v2", C.c_ulong), ("perc", C.c_int), ("len0", C.c_int), ("len1", C.c_int),
                ("len2", C.c_int), ("read1_is_a", C.c_int)]


H.ref_overlap_parse.argtypes = [C.POINTER(C.c_char_p), C.POINTER(ref_overlap_out)]
H.ref_build_rev_comp.argtypes = [C.c_char_p, C.c_char_p]
H.ref_str_to_read_id.restype = C.c_ulong
H.ref_str_to_read_id.argtypes = [C.c_char_p]
H.ref_read_get.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
H.ref_edge_build.argtypes = [C.POINTER(ref_edge_io), C.c_int]
F.frag_phred_to_prob.restype = C.c_double
F.frag_phred_to_prob.argtypes = [C.c_int]
F.frag_score.restype = C.c_double
F.frag_score.argtypes = [C.c_char, C.c_char, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
F.frag_overlap_score.restype = C.c_double
F.frag_overlap_score.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.c_double,
                                 C.POINTER(C.c_double)]


def headers_vectors():
    out = {"provenance": "reference headers Overlap.h/Types.h/Read.h/Edge.h executed via oracle/_ref/libhcref_headers.so"}
    # --- Overlap(std::vector<std::string>) ---
    cases = [
        ["12", "345", "5", "-", "-", "+", "+", "93", "-", "140", "-", "s", "s"],
        ["0x10", "017", "0", "-", "-", "-", "+", "100", "-", "150", "-", "s", "s"],
        ["7", "9", "12", "30", "1", "+", "-", "80", "70", "120", "105", "p", "p"],
        ["7", "9", "12", "30", "2", "-", "-", "80", "71", "120", "105", "p", "p"],
        ["3", "4", "10", "20", "-", "+", "+", "66", "50", "99", "75", "s", "p"],
        ["4", "3", "10", "20", "-", "+", "-", "66", "0", "99", "75", "p", "s"],
        ["abc", "5x", " 17", "-", " -", "+ ", " -", "55abc", "-", "88 ", "-", "s ", " s"],
        ["18446744073709551615", "1", "0", "0", "-", "+", "+", "1", "0", "1", "0", "s", "s"],
        ["5", "6", "3", "-", "-", "+", "+", "49", "51", "77", "33", "s", "s"],
        ["5", "6", "3", "0", "-", "+", "+", "49", "51", "77", "33", "s", "s"],
        ["21", "22", "0", "0", "1", "+", "+", "100", "100", "150", "150", "p", "p"],
        ["21", "22", "149", "149", "2", "-", "+", "1", "2", "1", "1", "p", "p"],
    ]
    rng = random.Random(7)
    for _ in range(60):
        pp = rng.random() < 0.5
        t1, t2 = ("p", "p") if pp else rng.choice([("s", "s"), ("s", "p"), ("p", "s")])
        ss = t1 == "s" and t2 == "s"
        cases.append([str(rng.randrange(10**6)), str(rng.randrange(10**6)), str(rng.randrange(500)),
                      "-" if ss else str(rng.randrange(500)), rng.choice("12") if (t1 == "p" and t2 == "p") else "-",
                      rng.choice("+-"), rng.choice("+-"), str(rng.randrange(101)), "-" if ss else str(rng.randrange(101)),
                      str(rng.randrange(1, 600)), "-" if ss else str(rng.randrange(600)), t1, t2])
    ov = []
    for f in cases:
        arr = (C.c_char_p * 13)(*[x.encode() for x in f])
        o = ref_overlap_out()
        H.ref_overlap_parse(arr, C.byref(o))
        ov.append({"fields": f, "id1": o.id1, "id2": o.id2, "pos1": o.pos1, "pos2": o.pos2, "ord": o.ord.decode(),
                   "ori1": o.ori1.decode(), "ori2": o.ori2.decode(), "type1": o.type1.decode(),
                   "type2": o.type2.decode(), "perc": o.perc, "len1": o.len1, "len2": o.len2,
                   "line": o.line.decode()})
    out["overlap_parse"] = ov
    # --- build_rev_comp / str_to_read_id ---
    rc = []
    for n in [1, 2, 5, 31, 150, 333]:
        s = "".join(rng.choice("ACGTN") for _ in range(n))
        buf = C.create_string_buffer(n + 1)
        H.ref_build_rev_comp(s.encode(), buf)
        rc.append({"seq": s, "rev_comp": buf.value.decode()})
    out["rev_comp"] = rc
    out["read_id"] = [{"s": s, "id": H.ref_str_to_read_id(s.encode())}
                      for s in ["0", "42", "0x1F", "010", "12abc", "abc", "", " 9", "-1", "99999999999"]]
    # --- Read getters ---
    rg = []
    s1, s2, p1, p2 = "ACGTTGCANN", "GGGTAC", "ABCDEFGHIJ", "KLMNOP"
    for paired in (0, 1):
        for which in range(4):
            for i in ((1, 2) if paired else (0,)):
                buf = C.create_string_buffer(64)
                H.ref_read_get(paired, s1.encode(), (s2 if paired else "").encode(), p1.encode(),
                               (p2 if paired else "").encode(), which, i, buf)
                rg.append({"paired": paired, "which": which, "i": i, "out": buf.value.decode()})
    out["read_get"] = {"seq1": s1, "seq2": s2, "phred1": p1, "phred2": p2, "cases": rg}
    # --- Edge build / swap_reads ---
    ed = []
    for k in range(40):
        pos1 = 0 if k % 2 == 0 else rng.randrange(1, 100)
        v1, v2 = rng.randrange(1000), rng.randrange(1000)
        if v1 == v2:
            v2 += 1
        do_swap = 1 if (pos1 == 0 and v1 > v2) else 0
        io = ref_edge_io(rng.random(), rng.choice([0.0, 0.01, 0.5, 1.0]), pos1, rng.randrange(100),
                         rng.randrange(-100, 100), rng.randrange(-100, 100), rng.randrange(2), rng.randrange(2),
                         rng.choice(b"-12".decode()).encode(), v1, v2, rng.randrange(101), 0, rng.randrange(1, 300),
                         rng.randrange(0, 300), 0)
        inp = {k2: (getattr(io, k2).decode() if k2 == "ord" else getattr(io, k2))
               for k2, _ in ref_edge_io._fields_ if k2 not in ("len0", "read1_is_a")}
        inp["score"], inp["mismatch"] = io.score.hex(), io.mismatch.hex()
        H.ref_edge_build(C.byref(io), do_swap)
        res = {k2: (getattr(io, k2).decode() if k2 == "ord" else getattr(io, k2)) for k2, _ in ref_edge_io._fields_}
        res["score"], res["mismatch"] = io.score.hex(), io.mismatch.hex()
        ed.append({"in": inp, "do_swap": do_swap, "out": res})
    out["edge"] = ed
    return out


def fragment_vectors():
    out = {"provenance": "FRAGMENT PROBE: src/EdgeCalculator.cpp:26-139 piped verbatim, oracle/_ref/libhcref_fragment.so; "
                         "glibc " + os.confstr("CS_GNU_LIBC_VERSION")}
    out["phred_to_prob"] = [{"phred": q, "p": F.frag_phred_to_prob(q).hex()} for q in range(0, 95)]
    sc = []
    quals = [0, 1, 2, 5, 12, 20, 30, 37, 40, 41, 60, 93, 94]
    for q1 in quals:
        for q2 in quals:
            p1, p2 = F.frag_phred_to_prob(q1), F.frag_phred_to_prob(q2)
            for nt1, nt2 in (("A", "A"), ("A", "C"), ("G", "T"), ("N", "A"), ("C", "N")):
                for ms in (0.0, 0.3):
                    mm = C.c_int(3)
                    v = F.frag_score(nt1.encode(), nt2.encode(), p1, p2, ms, C.byref(mm))
                    sc.append({"nt1": nt1, "nt2": nt2, "q1": q1, "q2": q2, "mismatch": ms, "value": v.hex(), "mm": mm.value})
    out["score"] = sc
    rng = random.Random(11)
    qsets = ["!#+5?FGIJ", "5?FGGGIII", "".join(chr(33 + q) for q in range(0, 61)), "II", "!I", "~}|I5"]
    ov = []
    for k in range(260):
        n1, n2 = rng.randrange(20, 420), rng.randrange(20, 420)
        qs = rng.choice(qsets)
        s1 = [rng.choice("ACGT") for _ in range(n1)]
        pos = rng.randrange(0, n1 + (5 if k % 17 == 0 else 0))
        # make seq2 a noisy copy of the suffix of seq1 so that scores are not all tiny
        s2 = []
        for i in range(n2):
            if pos + i < n1 and rng.random() > (0.02 if k % 3 else 0.0):
                s2.append(s1[pos + i])
            else:
                s2.append(rng.choice("ACGT"))
        nrate = rng.choice([0.0, 0.0, 0.01, 0.3, 1.0 if k % 41 == 0 else 0.0])
        s1 = [("N" if rng.random() < nrate else c) for c in s1]
        s2 = [("N" if rng.random() < nrate else c) for c in s2]
        q1 = "".join(rng.choice(qs) for _ in range(n1))
        q2 = "".join(rng.choice(qs) for _ in range(n2))
        mrl = rng.choice([0, 0, 0, 50, 300])
        ms = rng.choice([0.0, 0.0, 1e-5, 0.01, 0.4])
        mr = C.c_double(-7.0)
        v = F.frag_overlap_score("".join(s1).encode(), "".join(s2).encode(), q1.encode(), q2.encode(), pos, mrl, ms,
                                 C.byref(mr))
        ov.append({"seq1": "".join(s1), "seq2": "".join(s2), "q1": q1, "q2": q2, "pos": pos, "min_read_len": mrl,
                   "mismatch": ms, "score": v.hex(), "mismatch_rate": mr.value.hex()})
    out["overlap_score"] = ov
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "ref_headers.json"), "w") as f:
        json.dump(headers_vectors(), f, indent=0)
    with open(os.path.join(HERE, "ref_fragment.json"), "w") as f:
        json.dump(fragment_vectors(), f, indent=0)
    print("golden fixtures written")
