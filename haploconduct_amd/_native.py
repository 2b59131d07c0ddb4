"""ctypes bindings for libhcedge.so (include/hcedge.h).  Fails loudly if the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "csrc", "libhcedge.so")


class HcError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        try:
            msg = lib.hc_strerror(status).decode()
            detail = lib.hc_last_error().decode()
        except Exception:  # pragma: no cover
            msg, detail = "?", ""
        super().__init__(f"{where}: {msg} ({status}) {detail}".strip())


if not os.path.exists(lib_path):
    raise ImportError(
        f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C haploconduct_amd/csrc`.  haploconduct_amd has no CPU fallback."
    )

# PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in
# one process cannot both own the GPU, so when torch is installed let it load its runtime
# first: libhcedge.so (linked against libamdhip64.so.7 by SONAME) then binds to that copy.
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:  # torch is optional for the C ABI itself
    pass

lib = C.CDLL(lib_path)


class hc_settings(C.Structure):
    _fields_ = [
        ("edge_threshold", C.c_double),
        ("ov_threshold", C.c_double),
        ("merge_contigs", C.c_double),
        ("mismatch", C.c_double),
        ("min_read_len", C.c_uint32),
        ("min_overlap_len", C.c_uint32),
        ("min_overlap_perc", C.c_uint32),
        ("flags", C.c_uint32),
        ("max_overlaps", C.c_uint64),
        ("device", C.c_int32),
        ("n_threads", C.c_uint32),
        ("device_mask", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class hc_text_result(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_lines", "lines_read", "needs_host", "n_nonplain", "n_unknown_id", "self_overlaps", "silently_dropped",
                                          "prefilter_rejected", "scored")] + \
               [("rows", C.c_void_p), ("n_rows", C.c_uint64), ("rejected", C.c_void_p), ("n_rejected", C.c_uint64),
                ("nonplain", C.c_void_p), ("n_nonplain_listed", C.c_uint64)]


class hc_graph_counts(C.Structure):
    _fields_ = [("n_admitted", C.c_uint64), ("n_edges", C.c_uint64), ("inclusion_count", C.c_uint64), ("dup_count", C.c_uint64),
                ("n_tied_lists", C.c_uint64), ("first_bad", C.c_int64)]


_vp = C.c_void_p
_sig = {
    "hc_version": (C.c_char_p, []),
    "hc_strerror": (C.c_char_p, [C.c_int]),
    "hc_last_error": (C.c_char_p, []),
    "hc_device_count": (C.c_int, []),
    "hc_create": (C.c_int, [C.POINTER(_vp), C.POINTER(hc_settings)]),
    "hc_destroy": (C.c_int, [_vp]),
    "hc_set_reads": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32]),
    "hc_score_batch": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "hc_score_batch_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "hc_synchronize": (C.c_int, [_vp]),
    "hc_compact_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "hc_find_overlaps": (C.c_int, [_vp, C.c_double, C.c_uint32, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_found_to_overlaps": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_set_found_records": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hc_set_found_from_sfo_text": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_found_to_lines_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "hc_found_lines_fetch": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "hc_score_pack_device": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, _vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "hc_narrow_payload_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "hc_pack_cands": (None, [_vp, C.c_uint64, _vp]),
    "hc_score_cands": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "hc_score_cands_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "hc_block_create": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "hc_block_submit": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64]),
    "hc_block_wait": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
    "hc_block_destroy": (C.c_int, [_vp]),
    "hc_text_set_ids": (C.c_int, [_vp, _vp, C.c_uint32]),
    "hc_textblock_create": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "hc_textblock_buffer": (_vp, [_vp]),
    "hc_textblock_submit": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64]),
    "hc_textblock_submit_from": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, _vp, C.c_uint64]),
    "hc_linechain_create": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "hc_linechain_destroy": (C.c_int, [_vp]),
    "hc_textblock_wait": (C.c_int, [_vp, C.POINTER(hc_text_result)]),
    "hc_textblock_list_nonplain": (C.c_int, [_vp, C.c_uint32]),
    "hc_textblock_destroy": (C.c_int, [_vp]),
    "hc_textblock_regrown": (C.c_uint64, [_vp]),
    "hc_textblock_reserve_rows": (C.c_int, [_vp, C.c_uint64]),
    "hc_graph_begin": (C.c_int, [_vp]),
    "hc_graph_append": (C.c_int, [_vp, _vp, C.c_uint64]),
    "hc_graph_resolve": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_uint32, C.POINTER(hc_graph_counts)]),
    "hc_graph_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hc_reset": (C.c_int, [_vp, _vp]),
    "hc_set_comm_reserve": (C.c_int, [_vp, C.c_uint32]),
    "hc_comm_gate_device": (C.c_int, [_vp, _vp, C.c_uint32]),
    "hc_compact_pack_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "hc_pack_rows_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, C.c_uint64, _vp, _vp]),
    "hc_score_batch_compact": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "hc_set_reorder": (C.c_int, [_vp, C.c_int]),
    "hc_host_alloc": (C.c_int, [_vp, C.POINTER(_vp), C.c_uint64]),
    "hc_host_free": (C.c_int, [_vp, _vp]),
    "hc_time_score_kernel": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, _vp, C.c_int, C.POINTER(C.c_float)]),
    "hc_count_positions_device": (C.c_int, [_vp, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hc_finalize": (C.c_int, [C.POINTER(hc_settings), _vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "hc_finalize_batch": (C.c_int, [C.POINTER(hc_settings), _vp, C.c_uint64, _vp, _vp, _vp]),
    "hc_get_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)] + [C.POINTER(C.c_double)] * 4),
    "hc_get_kernel_info": (C.c_int, [_vp, C.c_char_p, C.c_uint32]),
    "hc_get_kernel_info_for": (C.c_int, [_vp, C.c_uint64, C.c_char_p, C.c_uint32]),
    "hc_dev_radix_sort": (C.c_int, [C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint64, C.c_int, C.c_int]),
    "hc_dev_exclusive_sum": (C.c_int, [C.c_uint32, _vp, _vp, C.c_uint64]),
    "hc_dev_select_flagged": (C.c_int, [_vp, C.c_uint64, _vp, C.POINTER(C.c_uint64)]),
    "hc_dev_unique_u64": (C.c_int, [_vp, C.c_uint64, _vp, C.POINTER(C.c_uint64)]),
}
for _name, (_res, _args) in _sig.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def check(status, where=""):
    if status != 0:
        raise HcError(status, where)


def device_count():
    return int(lib.hc_device_count())


def version():
    return lib.hc_version().decode()
