"""numpy views of the C-ABI records (include/hcedge.h)."""
from dataclasses import dataclass

import numpy as np

from ._native import hc_settings

# hc_overlap_rec, 32 bytes
OVERLAP_DTYPE = np.dtype(
    [
        ("read1", "<u4"), ("read2", "<u4"), ("pos1", "<u4"), ("pos2", "<u4"),
        ("ori1", "u1"), ("ori2", "u1"), ("ord", "u1"), ("flags", "u1"),
        ("len1", "<u4"), ("len2", "<u4"), ("perc", "<u4"),
    ],
    align=False,
)
assert OVERLAP_DTYPE.itemsize == 32

# hc_cand_rec, 16 bytes: what the device reads of a candidate (the form that crosses PCIe in the stage)
CAND_DTYPE = np.dtype([("read1", "<u4"), ("read2", "<u4"), ("pos1_bits", "<u4"), ("pos2_bits", "<u4")], align=False)
assert CAND_DTYPE.itemsize == 16
REC_FULL, REC_COMPACT = 0, 1

# hc_gather_row, 32 bytes: a non-dropped result record tagged with its candidate index
ROW_DTYPE = np.dtype([("index", "<u8"), ("x1", "<f8"), ("x2", "<f8"), ("mm", "<u4"), ("n_cls", "<u4")], align=False)
assert ROW_DTYPE.itemsize == 32

# hc_line_rec, 48 bytes: one parsed line of the overlaps file; hc_text_row = row + line; hc_text_reject = line number + line
LINE_DTYPE = np.dtype([("id1", "<u8"), ("id2", "<u8"), ("pos1", "<u4"), ("pos2", "<u4"), ("perc1", "<u4"), ("perc2", "<u4"), ("len1", "<u4"),
                       ("len2", "<u4"), ("ord", "u1"), ("ori1", "u1"), ("ori2", "u1"), ("type1", "u1"), ("type2", "u1"), ("pad", "u1", (3,))], align=False)
assert LINE_DTYPE.itemsize == 48
TEXT_ROW_DTYPE = np.dtype([("row", ROW_DTYPE), ("line", LINE_DTYPE)], align=False)
assert TEXT_ROW_DTYPE.itemsize == 80
TEXT_REJECT_DTYPE = np.dtype([("line_index", "<u4"), ("pad", "<u4"), ("line", LINE_DTYPE)], align=False)
assert TEXT_REJECT_DTYPE.itemsize == 56

# hc_admit_rec, 48 bytes: an admitted candidate handed to the device's duplicate resolution
ADMIT_DTYPE = np.dtype([("score", "<f8"), ("read1", "<u4"), ("read2", "<u4"), ("pos1", "<u4"), ("pos2", "<u4"), ("mm", "<u4"), ("n", "<u4"),
                        ("len1", "<u4"), ("len2", "<u4"), ("perc", "<u4"), ("ori1", "u1"), ("ori2", "u1"), ("ord", "u1"), ("pad", "u1")],
                       align=False)
assert ADMIT_DTYPE.itemsize == 48

# hc_result_rec, 24 bytes
RESULT_DTYPE = np.dtype([("x1", "<f8"), ("x2", "<f8"), ("mm", "<u4"), ("n_cls", "<u4")], align=False)
assert RESULT_DTYPE.itemsize == 24

CLS_DROP, CLS_NONEDGE, CLS_EDGE, CLS_EDGE_MC, CLS_AMBIG, CLS_ERROR = 0, 1, 2, 3, 4, 7
CLS_NAMES = {0: "drop", 1: "nonedge", 2: "edge", 3: "edge_mc", 4: "ambig", 7: "error"}

FLAG_ADD_DUPLICATES = 0x1
FLAG_RESOLVE_ORIENTATIONS = 0x2
FLAG_IGNORE_INCLUSIONS = 0x4
FLAG_RELAX_PE_EDGES = 0x8
FLAG_ALLOW_SPACES = 0x10
FLAG_VERBOSE = 0x20


@dataclass
class Settings:
    """The ProgramSettings fields the hot path reads (reference src/Types.h:19-67),
    with the defaults of src/ViralQuasispecies.cpp:49-99."""

    edge_threshold: float = 0.99
    ov_threshold: float = 0.9
    merge_contigs: float = 0.0
    mismatch: float = 0.0
    min_read_len: int = 0
    min_overlap_len: int = 150
    min_overlap_perc: int = 0
    flags: int = FLAG_RESOLVE_ORIENTATIONS
    max_overlaps: int = 100000000
    device: int = 0
    n_threads: int = 1
    device_mask: int = 0

    def to_c(self):
        return hc_settings(
            self.edge_threshold, self.ov_threshold, self.merge_contigs, self.mismatch, self.min_read_len,
            self.min_overlap_len, self.min_overlap_perc, self.flags, self.max_overlaps, self.device, self.n_threads,
            self.device_mask, 0,
        )


def result_n(res):
    return res["n_cls"] & 0x0FFFFFFF


def result_cls(res):
    return res["n_cls"] >> 28


# hc_sfo_rec (include/hcedge.h): one suffix-prefix overlap as rust-overlaps reports it
SFO_DTYPE = np.dtype([("idA", "<u4"), ("idB", "<u4"), ("OHA", "<i4"), ("OHB", "<i4"), ("OLA", "<u4"), ("OLB", "<u4"), ("K", "<u4"),
                      ("inverted", "<u4")])
assert SFO_DTYPE.itemsize == 32
FIND_REVERSALS, FIND_INCLUSIONS = 1, 2
