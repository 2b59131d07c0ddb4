"""CPU ORACLE (test infrastructure, NOT product code) for the SFO ingest, SURVEY.md §8(f2):
a Python-3 restatement of the reference's scripts/sfo2overlaps.py — rust-overlaps' 8-column SFO lines
(idA idB ori OHA OHB OLA OLB K) to SAVAGE's 13-column overlaps file.  Pinned against vectors produced
by the reference script itself (tests/golden/sfo/, made by tests/golden/make_golden_sfo.py).
Every function cites the reference lines it follows."""
import math


def _round_half_away(x):  # Python 2 round(), sfo2overlaps.py:189
    f = math.floor(abs(x))
    r = f + 1.0 if abs(x) - f >= 0.5 else float(f)
    return r if x >= 0 else -r


def get_original_id(sfo_id, num_singles, num_pairs):  # sfo2overlaps.py:136-147
    if num_pairs == 0:
        return sfo_id
    assert 0 <= sfo_id < num_singles + 2 * num_pairs
    return sfo_id if sfo_id < num_singles + num_pairs else sfo_id - num_pairs


def is_paired(rid, num_singles, num_pairs):  # sfo2overlaps.py:124-134
    if num_pairs == 0:
        return False
    assert 0 <= rid < num_singles + num_pairs
    return rid >= num_singles


def flip(t):  # flip_N / flip_I, sfo2overlaps.py:112-122
    if t[2] == "I":
        return [t[1], t[0], t[2], t[4], t[3], t[6], t[5], t[7]]
    return [t[1], t[0], t[2], str(-int(t[3])), str(-int(t[4])), t[6], t[5], t[7]]


def get_s_s_overlap(l):  # sfo2overlaps.py:150-200; l = [idA, idB, sfo_idA, sfo_idB, ori, OHA, OHB, OLA, OLB, K]
    ida, idb = l[0], l[1]
    oha, ohb, ola, olb = int(l[5]), int(l[6]), int(l[7]), int(l[8])
    ori = "+" if l[4] == "N" else "-"
    ovlen = min(ola, olb)
    if oha >= 0:  # read A is first
        la, lb = (ola + oha, olb + ohb) if ohb >= 0 else (ola + oha - ohb, olb)
        id1, id2, pos1, ori1, ori2 = ida, idb, str(oha), "+", ori
    else:         # read B is first
        la, lb = (ola, -oha + olb + ohb) if ohb >= 0 else (ola - ohb, -oha + olb)
        id1, id2, pos1, ori1, ori2 = idb, ida, str(-oha), ori, "+"
    minlen = min(la, lb)
    perc = min(_round_half_away(100 * ovlen / minlen), 100)
    assert minlen > 0
    return [id1, id2, pos1, "-", "-", ori1, ori2, "{:.0f}".format(perc), "-", str(ovlen), "-", "s", "s"]


def merge_overlaps(o1, o2, type1, type2):  # sfo2overlaps.py:312-329
    o = list(o1)
    o[11], o[12] = type1, type2
    if type1 == "p" and type2 == "p":
        if o1[0] != o2[0]:
            assert o1[0] == o2[1]
            o[4] = "2"
        else:
            o[4] = "1"
    o[3], o[8], o[10] = o2[2], o2[7], o2[9]
    return o


def find_paired_overlap(c1, c2, type_a, type_b):  # sfo2overlaps.py:221-310
    if c1[4] != c2[4]:
        return []
    a1, b1, a2, b2 = int(c1[2]), int(c1[3]), int(c2[2]), int(c2[3])
    normal = c1[4] == "N"
    inv = c1[4] == "I"
    first = None  # which candidate gives overlap1
    if type_a and type_b:
        if normal:
            first = 1 if (a1 < a2 and b1 < b2) else (2 if (a1 > a2 and b1 > b2) else None)
        elif inv:
            first = 1 if (a1 < a2 and b1 > b2) else (2 if (a1 > a2 and b1 < b2) else None)
    else:
        p1, p2 = int(c1[5]), int(c2[5])
        k1, k2 = (a1, a2) if (type_a and not type_b) else (b1, b2)
        if normal:
            first = 1 if (k1 < k2 and p1 < p2) else (2 if (k1 > k2 and p1 > p2) else None)
        elif inv:
            first = 2 if (k1 < k2 and p1 > p2) else (1 if (k1 > k2 and p1 < p2) else None)
    if first is None:
        return []
    o1, o2 = (get_s_s_overlap(c1), get_s_s_overlap(c2)) if first == 1 else (get_s_s_overlap(c2), get_s_s_overlap(c1))
    if o1[0] == c1[0]:
        assert o1[1] == c1[1]
        t1, t2 = ("p" if type_a else "s"), ("p" if type_b else "s")
    else:
        assert o1[1] == c1[0] and o1[0] == c1[1]
        t1, t2 = ("p" if type_b else "s"), ("p" if type_a else "s")
    return merge_overlaps(o1, o2, t1, t2)


def sfo2overlaps(sfo_lines, num_singles, num_pairs):
    """sfo_lines: iterable of text lines (with or without the newline).  Returns the output lines (no newline)."""
    tmp = []
    for line in sfo_lines:  # sfo2overlaps.py:31-50
        raw = line if line.endswith("\n") else line + "\n"
        f = raw.strip("\n").split()
        assert len(f) == 8
        ida, idb = int(f[0]), int(f[1])
        na, nb = get_original_id(ida, num_singles, num_pairs), get_original_id(idb, num_singles, num_pairs)
        if na > nb:
            tmp.append(f"{nb}\t{na}\t" + "\t".join(flip(f)) + "\n")
        else:
            tmp.append(f"{na}\t{nb}\t" + raw)
    # sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq  under LC_ALL=C (sfo2overlaps.py:53): numeric keys, then the whole line bytewise
    def key(t):
        f = t.split()
        return (int(f[0]), int(f[1]), int(f[2]), int(f[3]), t.encode())
    tmp.sort(key=key)
    uniq = [t for i, t in enumerate(tmp) if i == 0 or t != tmp[i - 1]]
    out, cands = [], []
    for t in uniq:  # sfo2overlaps.py:63-103
        l = t.strip("\n").split()
        assert len(l) == 10
        ida, idb = int(l[0]), int(l[1])
        if ida == idb:
            continue
        pa, pb = is_paired(ida, num_singles, num_pairs), is_paired(idb, num_singles, num_pairs)
        if not pa and not pb:
            out.append("\t".join(get_s_s_overlap(l)))
            continue
        if cands and cands[0][0:2] != [str(ida), str(idb)]:
            ca, cb = int(cands[0][0]), int(cands[0][1])
            pca, pcb = is_paired(ca, num_singles, num_pairs), is_paired(cb, num_singles, num_pairs)
            # NB: the reference passes the types of the CURRENT line to match_candidates (sfo2overlaps.py:94)
            if len(cands) >= 2:
                for i in range(len(cands)):
                    for j in range(i + 1, len(cands)):
                        o = find_paired_overlap(cands[i], cands[j], pa, pb)
                        if o:
                            out.append("\t".join(o))
            cands = []
            del pca, pcb
        cands.append(l)
    # the last group is never flushed by the reference (no match_candidates call after the loop)
    return [o for i, o in enumerate(out) if i == 0 or o != out[i - 1]]  # `uniq`, sfo2overlaps.py:107
