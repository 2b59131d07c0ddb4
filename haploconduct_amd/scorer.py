"""EdgeScorer — Python face of the hc_ctx C-ABI (include/hcedge.h).

It plays the role of the reference's ``EdgeCalculator`` object for the scoring half
of the path (reference src/EdgeCalculator.h:25-64): constructed from the settings,
given the reads once, then fed batches of candidate overlaps."""
import ctypes as C

import numpy as np

from . import _native as N
from .records import ADMIT_DTYPE, CAND_DTYPE, OVERLAP_DTYPE, REC_COMPACT, REC_FULL, RESULT_DTYPE, ROW_DTYPE, TEXT_REJECT_DTYPE, TEXT_ROW_DTYPE, Settings


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class EdgeScorer:
    def __init__(self, settings: Settings = None):
        self.settings = settings or Settings()
        self._cs = self.settings.to_c()
        self._ctx = C.c_void_p()
        N.check(N.lib.hc_create(C.byref(self._ctx), C.byref(self._cs)), "hc_create")
        self._reads = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            N.lib.hc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- reads --------------------------------------------------------------
    def set_reads(self, reads):
        self._reads = reads
        N.check(
            N.lib.hc_set_reads(self._ctx, _ptr(reads.bases), _ptr(reads.quals), _ptr(reads.seq_off),
                               _ptr(reads.read_first_seq), reads.n_reads),
            "hc_set_reads",
        )

    def info(self):
        k, sb = C.c_uint32(), C.c_uint64()
        d = [C.c_double() for _ in range(4)]
        N.check(N.lib.hc_get_info(self._ctx, C.byref(k), C.byref(sb), *[C.byref(x) for x in d]), "hc_get_info")
        return {"qual_alphabet": k.value, "store_bytes": sb.value, "x_edge": (d[0].value, d[1].value),
                "x_ov": (d[2].value, d[3].value)}

    def kernel_info(self, n=0):
        """hc_get_kernel_info[_for]: the scoring kernel chosen for the read set (n > 0: for a launch of n candidates), as text."""
        buf = C.create_string_buffer(768)
        N.check(N.lib.hc_get_kernel_info_for(self._ctx, int(n), buf, 768), "hc_get_kernel_info_for")
        return buf.value.decode()

    def set_reorder(self, mode):
        """0 never, 1 always, 2 auto (hc_set_reorder)."""
        N.check(N.lib.hc_set_reorder(self._ctx, int(mode)), "hc_set_reorder")

    # -- scoring ------------------------------------------------------------
    def score_batch(self, overlaps):
        """Host buffers in, host buffers out (hc_score_batch)."""
        ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE)
        out = np.empty(ov.shape[0], dtype=RESULT_DTYPE)
        N.check(N.lib.hc_score_batch(self._ctx, _ptr(ov), ov.shape[0], _ptr(out)), "hc_score_batch")
        return out

    @staticmethod
    def pack_cands(overlaps):
        """hc_pack_cands: hc_overlap_rec -> hc_cand_rec (16 bytes: what the device reads)."""
        ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE)
        out = np.empty(ov.shape[0], dtype=CAND_DTYPE)
        N.lib.hc_pack_cands(_ptr(ov), ov.shape[0], _ptr(out))
        return out

    def score_cands(self, cands):
        """Compact records in, result records out (hc_score_cands; host buffers)."""
        cd = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        out = np.empty(cd.shape[0], dtype=RESULT_DTYPE)
        N.check(N.lib.hc_score_cands(self._ctx, _ptr(cd), cd.shape[0], _ptr(out)), "hc_score_cands")
        return out

    def score_cands_device(self, d_in_ptr, n, d_out_ptr, stream=None):
        N.check(N.lib.hc_score_cands_device(self._ctx, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr), C.c_void_p(stream or 0)),
                "hc_score_cands_device")

    def score_blocks(self, cands, block=250000, in_flight=2):
        """The stage's device leg on a whole candidate array: hc_block_submit / hc_block_wait over blocks of `block`
        compact records, `in_flight` block objects; returns the non-dropped rows of all blocks in index order."""
        cd = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        n = cd.shape[0]
        blocks = []
        for _ in range(in_flight):
            b = C.c_void_p()
            N.check(N.lib.hc_block_create(self._ctx, max(1, min(block, max(n, 1))), C.byref(b)), "hc_block_create")
            blocks.append(b)
        out, pending = [], []

        def wait(b):
            rows, k = C.c_void_p(), C.c_uint64()
            N.check(N.lib.hc_block_wait(b, C.byref(rows), C.byref(k)), "hc_block_wait")
            if k.value:
                out.append(np.frombuffer((C.c_char * (k.value * 32)).from_address(rows.value), dtype=ROW_DTYPE).copy())

        try:
            for i, at in enumerate(range(0, n, block)):
                b = blocks[i % in_flight]
                if len(pending) == in_flight:
                    wait(pending.pop(0))
                m = min(block, n - at)
                N.check(N.lib.hc_block_submit(b, C.c_void_p(cd.ctypes.data + at * 16), m, at), "hc_block_submit")
                pending.append(b)
            while pending:
                wait(pending.pop(0))
        finally:
            for b in blocks:
                N.lib.hc_block_destroy(b)
        return np.concatenate(out) if out else np.zeros(0, ROW_DTYPE)

    def set_ids(self, read_ids):
        """hc_text_set_ids: the id -> read index table of the device's text parser."""
        ids = np.ascontiguousarray(read_ids, dtype=np.uint64)
        N.check(N.lib.hc_text_set_ids(self._ctx, _ptr(ids), ids.shape[0]), "hc_text_set_ids")

    def score_text(self, text, block_bytes=1 << 20, first_line_no=0, chained=False, reserve_rows=0):
        """The overlaps file's TEXT through hc_textblock_submit / hc_textblock_wait, block by block (cut at line ends).
        Returns a list with one dict per block: the hc_text_result fields, rows / rejected as numpy copies.
        chained: hc_textblock_submit_from — the text goes to the device straight from `text`, two blocks in flight, line
        numbers through an hc_linechain (first_line_no must be 0).  reserve_rows: hc_textblock_reserve_rows before the first submit;
        every dict carries "regrown" = hc_textblock_regrown of the block so far."""
        raw = text if isinstance(text, bytes) else text.encode()
        if chained:
            return self._score_text_chained(raw, block_bytes)
        b = C.c_void_p()
        N.check(N.lib.hc_textblock_create(self._ctx, max(block_bytes, 64), C.byref(b)), "hc_textblock_create")
        out, at, line_no, base = [], 0, first_line_no, 0
        try:
            if reserve_rows:
                N.check(N.lib.hc_textblock_reserve_rows(b, reserve_rows), "hc_textblock_reserve_rows")
            buf = N.lib.hc_textblock_buffer(b)
            while at < len(raw):
                end = min(len(raw), at + block_bytes)
                if end < len(raw):
                    nl = raw.rfind(b"\n", at, end)
                    if nl < 0:
                        raise ValueError("a line longer than the block")
                    end = nl + 1
                C.memmove(buf, raw[at:end], end - at)
                N.check(N.lib.hc_textblock_submit(b, end - at, line_no, base), "hc_textblock_submit")
                r = N.hc_text_result()
                N.check(N.lib.hc_textblock_wait(b, C.byref(r)), "hc_textblock_wait")
                d = {k: getattr(r, k) for k, _ in r._fields_ if k not in ("rows", "rejected")}
                d["rows"] = np.frombuffer((C.c_char * (r.n_rows * 80)).from_address(r.rows), dtype=TEXT_ROW_DTYPE).copy() if r.n_rows else np.zeros(0, TEXT_ROW_DTYPE)
                d["rejected"] = (np.frombuffer((C.c_char * (r.n_rejected * 56)).from_address(r.rejected), dtype=TEXT_REJECT_DTYPE).copy()
                                 if r.n_rejected else np.zeros(0, TEXT_REJECT_DTYPE))
                d["base"], d["bytes"] = base, (at, end)
                d["regrown"] = int(N.lib.hc_textblock_regrown(b))
                out.append(d)
                line_no += r.n_lines
                base += r.n_lines
                at = end
        finally:
            N.lib.hc_textblock_destroy(b)
        return out

    def _score_text_chained(self, raw, block_bytes):
        cuts, at = [], 0
        while at < len(raw):
            end = min(len(raw), at + block_bytes)
            if end < len(raw):
                nl = raw.rfind(b"\n", at, end)
                if nl < 0:
                    raise ValueError("a line longer than the block")
                end = nl + 1
            cuts.append((at, end))
            at = end
        src = np.frombuffer(raw, np.uint8)  # pageable host memory, read in place
        blocks = [C.c_void_p(), C.c_void_p()]
        chain = C.c_void_p()
        for b in blocks:
            N.check(N.lib.hc_textblock_create(self._ctx, max(block_bytes, 64), C.byref(b)), "hc_textblock_create")
        N.check(N.lib.hc_linechain_create(self._ctx, len(cuts), C.byref(chain)), "hc_linechain_create")
        out, base = [], 0

        def collect(k):
            nonlocal base
            r = N.hc_text_result()
            N.check(N.lib.hc_textblock_wait(blocks[k % 2], C.byref(r)), "hc_textblock_wait")
            d = {f: getattr(r, f) for f, _ in r._fields_ if f not in ("rows", "rejected")}
            d["rows"] = np.frombuffer((C.c_char * (r.n_rows * 80)).from_address(r.rows), dtype=TEXT_ROW_DTYPE).copy() if r.n_rows else np.zeros(0, TEXT_ROW_DTYPE)
            d["rejected"] = (np.frombuffer((C.c_char * (r.n_rejected * 56)).from_address(r.rejected), dtype=TEXT_REJECT_DTYPE).copy()
                             if r.n_rejected else np.zeros(0, TEXT_REJECT_DTYPE))
            d["base"], d["bytes"] = base, cuts[k]
            base += r.n_lines
            out.append(d)

        try:
            for k, (a, e) in enumerate(cuts):
                if k >= 2:
                    collect(k - 2)
                N.check(N.lib.hc_textblock_submit_from(blocks[k % 2], src.ctypes.data + a, e - a, chain, k, blocks[(k - 1) % 2] if k else None, 0),
                        "hc_textblock_submit_from")
            for k in range(max(0, len(cuts) - 2), len(cuts)):
                collect(k)
        finally:
            for b in blocks:
                N.lib.hc_textblock_destroy(b)
            N.lib.hc_linechain_destroy(chain)
        return out

    def graph_resolve(self, admitted, n_vertices, vertex_of_read=None, sorted_order=False, pieces=0):
        """hc_graph_resolve + hc_graph_fetch: duplicate resolution and adjacency lists on the device.  Returns a dict
        with counts, edges (EDGE_DTYPE), out_off, in_nodes, in_off, seq, inclusions, tied_vertices.
        pieces > 0: hand the records over with hc_graph_begin / hc_graph_append in that many pieces first."""
        from .host import EDGE_DTYPE

        adm = np.ascontiguousarray(admitted, dtype=ADMIT_DTYPE)
        gc = N.hc_graph_counts()
        vtx = None if vertex_of_read is None else np.ascontiguousarray(vertex_of_read, dtype=np.uint32)
        src = _ptr(adm)
        if pieces > 0:
            N.check(N.lib.hc_graph_begin(self._ctx), "hc_graph_begin")
            cuts = [adm.shape[0] * k // pieces for k in range(pieces + 1)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                N.check(N.lib.hc_graph_append(self._ctx, C.c_void_p(adm.ctypes.data + a * 48), b - a), "hc_graph_append")
            src = None
        N.check(N.lib.hc_graph_resolve(self._ctx, src, adm.shape[0], n_vertices, None if vtx is None else _ptr(vtx),
                                       1 if sorted_order else 0, C.byref(gc)), "hc_graph_resolve")
        res = {"counts": {k: getattr(gc, k) for k, _ in gc._fields_}}
        if gc.first_bad >= 0:
            return res
        E = int(gc.n_edges)
        edges = np.zeros(E, EDGE_DTYPE)
        out_off, in_off = np.zeros(n_vertices + 1, np.uint64), np.zeros(n_vertices + 1, np.uint64)
        in_nodes, seq = np.zeros(E, np.uint32), np.zeros(E, np.uint32)
        incl = np.zeros(n_vertices, np.uint8)
        tied = np.zeros(int(gc.n_tied_lists), np.uint32)
        N.check(N.lib.hc_graph_fetch(self._ctx, _ptr(edges), _ptr(out_off), _ptr(in_nodes), _ptr(in_off), _ptr(seq), _ptr(incl),
                                     _ptr(tied) if tied.size else None), "hc_graph_fetch")
        res.update(edges=edges, out_off=out_off, in_nodes=in_nodes, in_off=in_off, seq=seq, inclusions=incl, tied_vertices=tied)
        return res

    def score_batch_compact(self, overlaps):
        """hc_score_batch_compact: (indices, records) of the non-DROP candidates only."""
        ov = np.ascontiguousarray(overlaps, dtype=OVERLAP_DTYPE)
        n = ov.shape[0]
        idx = np.empty(n, dtype=np.uint32)
        res = np.empty(n, dtype=RESULT_DTYPE)
        k = C.c_uint64()
        N.check(N.lib.hc_score_batch_compact(self._ctx, _ptr(ov), n, _ptr(idx), _ptr(res), n, C.byref(k)),
                "hc_score_batch_compact")
        return idx[: k.value].copy(), res[: k.value].copy()

    def compact_device(self, d_results_ptr, n, d_indices_ptr, d_count_ptr, stream=None):
        N.check(N.lib.hc_compact_device(self._ctx, C.c_void_p(d_results_ptr), n, C.c_void_p(d_indices_ptr),
                                        C.c_void_p(d_count_ptr), C.c_void_p(stream or 0)), "hc_compact_device")

    def find_overlaps(self, err_rate, min_overlap, reversals=True, inclusions=True, count_only=False):
        """hc_find_overlaps: all suffix-prefix overlaps / inclusions between the stored sequences (SFO records).
        count_only: compute on the device (always afresh) and return the number of records."""
        from .records import FIND_INCLUSIONS, FIND_REVERSALS, SFO_DTYPE

        flags = (FIND_REVERSALS if reversals else 0) | (FIND_INCLUSIONS if inclusions else 0)
        n = C.c_uint64()
        N.check(N.lib.hc_find_overlaps(self._ctx, err_rate, min_overlap, flags | 4, None, 0, C.byref(n)), "hc_find_overlaps")
        if count_only:
            return int(n.value)
        out = np.zeros(n.value, SFO_DTYPE)
        if n.value:
            N.check(N.lib.hc_find_overlaps(self._ctx, err_rate, min_overlap, flags, out.ctypes.data, out.size, C.byref(n)), "hc_find_overlaps")
        return out[: n.value]

    def found_to_overlaps(self, out_path, num_singles, num_pairs):
        """hc_found_to_overlaps: the SFO ingest (scripts/sfo2overlaps.py) straight from the records the last find_overlaps
        left on the device — flip and sort there, matching on the host threads.  Returns the number of overlap lines."""
        n = C.c_uint64()
        N.check(N.lib.hc_found_to_overlaps(self._ctx, str(out_path).encode(), int(num_singles), int(num_pairs), C.byref(n)), "hc_found_to_overlaps")
        return int(n.value)

    def set_found_records(self, recs):
        """hc_set_found_records: SFO records from elsewhere (records.SFO_DTYPE) in the place of the finder's."""
        recs = np.ascontiguousarray(recs)
        N.check(N.lib.hc_set_found_records(self._ctx, _ptr(recs), recs.size), "hc_set_found_records")

    def set_found_from_sfo_text(self, text):
        """hc_set_found_from_sfo_text: the SFO file's text (bytes) read on the device into the finder's place.  Returns the number of records;
        raises HcError (HC_ERR_NOT_ON_DEVICE) for a text that is not canonical."""
        n = C.c_uint64()
        N.check(N.lib.hc_set_found_from_sfo_text(self._ctx, text, len(text), C.byref(n)), "hc_set_found_from_sfo_text")
        return int(n.value)

    def found_to_lines(self, num_singles, num_pairs):
        """hc_found_to_lines_device + hc_found_lines_fetch: the SFO ingest entirely on the device; the overlaps file's lines as records
        (records.LINE_DTYPE), in file order.  Raises RuntimeError("... not on the device ...") where the device cannot decide."""
        from .records import LINE_DTYPE

        p, n = C.c_void_p(), C.c_uint64()
        N.check(N.lib.hc_found_to_lines_device(self._ctx, int(num_singles), int(num_pairs), C.byref(p), C.byref(n)), "hc_found_to_lines_device")
        out = np.zeros(n.value, dtype=LINE_DTYPE)
        N.check(N.lib.hc_found_lines_fetch(self._ctx, p, n.value, _ptr(out)), "hc_found_lines_fetch")
        return out

    def score_pack_device(self, d_in_ptr, n, d_out_ptr, cap, base_index, d_payload_ptr, stream=None, fmt=REC_FULL):
        """hc_score_pack_device: scoring + the collection payload in one kernel (rows unordered, count in row 0)."""
        N.check(N.lib.hc_score_pack_device(self._ctx, fmt, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr), cap, base_index,
                                           C.c_void_p(d_payload_ptr), C.c_void_p(stream or 0)), "hc_score_pack_device")
        return True

    def narrow_payload_device(self, d_payload_ptr, cap, d_payload24_ptr, stream=None):
        """hc_narrow_payload_device: a (cap + 1)-row payload of 32-byte rows -> 24-byte rows (x1, x2, index | mm << 32 | n << 46 | class << 60)."""
        N.check(N.lib.hc_narrow_payload_device(self._ctx, C.c_void_p(d_payload_ptr), cap, C.c_void_p(d_payload24_ptr), C.c_void_p(stream or 0)),
                "hc_narrow_payload_device")

    def set_comm_reserve(self, cus):
        """hc_set_comm_reserve: launches leave `cus` CUs to the kernels of the multi-GPU exchange (0: none)."""
        N.check(N.lib.hc_set_comm_reserve(self._ctx, int(cus)), "hc_set_comm_reserve")

    def comm_gate_device(self, stream, timeout_us=0):
        """hc_comm_gate_device: `stream` continues once the workgroups of the score_pack_device launches enqueued so far have started."""
        N.check(N.lib.hc_comm_gate_device(self._ctx, C.c_void_p(stream or 0), int(timeout_us)), "hc_comm_gate_device")

    def compact_pack_device(self, d_results_ptr, n, d_indices_ptr, d_count_ptr, cap, base_index, d_payload_ptr, stream=None):
        """hc_compact_pack_device: compaction + pack, the count in row 0 of the (cap + 1)-row payload."""
        N.check(N.lib.hc_compact_pack_device(self._ctx, C.c_void_p(d_results_ptr), n, C.c_void_p(d_indices_ptr), C.c_void_p(d_count_ptr),
                                             cap, base_index, C.c_void_p(d_payload_ptr), C.c_void_p(stream or 0)), "hc_compact_pack_device")

    def pack_rows_device(self, d_results_ptr, d_indices_ptr, d_count_ptr, cap, base_index, d_rows_ptr, stream=None):
        """hc_pack_rows_device: compacted records -> 32-byte rows tagged with their global candidate index."""
        N.check(N.lib.hc_pack_rows_device(self._ctx, C.c_void_p(d_results_ptr), C.c_void_p(d_indices_ptr), C.c_void_p(d_count_ptr),
                                          cap, base_index, C.c_void_p(d_rows_ptr), C.c_void_p(stream or 0)), "hc_pack_rows_device")

    def score_batch_device(self, d_in_ptr, n, d_out_ptr, stream=None):
        """Device pointers (ints, e.g. torch tensor .data_ptr()); asynchronous."""
        N.check(N.lib.hc_score_batch_device(self._ctx, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr),
                                            C.c_void_p(stream or 0)), "hc_score_batch_device")

    def synchronize(self):
        N.check(N.lib.hc_synchronize(self._ctx), "hc_synchronize")

    def time_kernel(self, d_in_ptr, n, d_out_ptr, iters, fmt=REC_FULL):
        ms = C.c_float()
        N.check(N.lib.hc_time_score_kernel(self._ctx, fmt, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr), iters,
                                           C.byref(ms)), "hc_time_score_kernel")
        return float(ms.value)

    def count_positions_device(self, d_in_ptr, n, fmt=REC_FULL):
        a, b = C.c_uint64(), C.c_uint64()
        N.check(N.lib.hc_count_positions_device(self._ctx, fmt, C.c_void_p(d_in_ptr), n, C.byref(a), C.byref(b)),
                "hc_count_positions_device")
        return int(a.value), int(b.value)

    def finalize(self, results, allow_errors=False):
        """Host libm finalisation -> (score, mismatch_rate, cls) arrays (hc_finalize_batch)."""
        res = np.ascontiguousarray(results, dtype=RESULT_DTYPE)
        n = res.shape[0]
        score = np.empty(n, np.float64)
        mrate = np.empty(n, np.float64)
        cls = np.empty(n, np.uint32)
        st = N.lib.hc_finalize_batch(C.byref(self._cs), _ptr(res), n, _ptr(score), _ptr(mrate), _ptr(cls))
        if st != 0 and not (allow_errors and st == -10):
            raise N.HcError(st, "hc_finalize_batch")
        return score, mrate, cls
