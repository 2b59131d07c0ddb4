"""Multi-GPU sharding of the edge-calculation path.

Candidates are independent given the read store, so the path shards with NO data-path
collective: rank r scores the contiguous slice shard_range(n, r, world) of the candidate
array against a replicated read store.  The only exchange is the collection of the non-dropped
records, once per batch (PayloadGather / StreamedGather below): either ONE all-gather of a
fixed-capacity payload whose row 0 is the count ("ring"), or the all-gather-v proper — the
counts by one small all-gather, then grouped per-peer send / recv of exactly the rows
("direct": every pair of ranks on its own xGMI link), or the same towards ONE rank only ("root":
the inserting host is the only consumer of the set — SURVEY.md §8(e): "rank 0 (or all ranks)").
Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests.  Records travel as int64 rows [global_index, x1_bits, x2_bits, mm | n_cls << 32] (32 bytes)
or, when every index is below 2^32 and no overlap longer than 16 383 positions (round 6), as
[x1_bits, x2_bits, index | mm << 32 | n << 46 | class << 60] (24 bytes: three quarters of the bytes
over every link); collect() always returns the 32-byte form."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced, order-preserving partition: rank r gets [lo, hi)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def admitted_rows(results_i64, global_offset):
    """results_i64: [n, 3] int64 view of hc_result_rec rows on any device -> [k, 4] rows of admitted records."""
    cls = (results_i64[:, 2] >> 60) & 0xF
    idx = torch.nonzero((cls >= 2) & (cls <= 4)).squeeze(1)
    return torch.cat([(idx + global_offset).unsqueeze(1), results_i64[idx]], dim=1).contiguous()


def all_gather_v(rows, group=None):
    """rows: [k_r, C] int64 on this rank -> [sum k_r, C] on every rank, in rank order."""
    world = dist.get_world_size(group)
    cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    kmax = max(counts) if counts else 0
    pad = torch.zeros((kmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0), counts


def gather_admitted(results_np_or_tensor, global_offset, group=None, device=None):
    """Convenience: hc_result_rec array (numpy structured or uint8/int64 tensor) of this rank's shard ->
    all ranks' admitted rows, ordered by global candidate index."""
    if isinstance(results_np_or_tensor, np.ndarray):
        t = torch.from_numpy(results_np_or_tensor.view(np.int64).reshape(-1, 3).copy())
        if device is not None:
            t = t.to(device)
    else:
        t = results_np_or_tensor.view(torch.int64).view(-1, 3)
    rows, counts = all_gather_v(admitted_rows(t, global_offset), group)
    return rows, counts


def narrow_rows(rows4):
    """[k, 4] int64 rows (32-byte form) -> [k, 3] (24-byte form); the caller has checked that they fit (rows_fit_narrow)."""
    w = rows4[:, 3]
    mm, n_cls = w & 0xFFFFFFFF, (w >> 32) & 0xFFFFFFFF
    n, cls = n_cls & 0x0FFFFFFF, (n_cls >> 28) & 0xF
    out = torch.empty((rows4.shape[0], 3), dtype=torch.int64, device=rows4.device)
    out[:, 0], out[:, 1] = rows4[:, 1], rows4[:, 2]
    out[:, 2] = (rows4[:, 0] & 0xFFFFFFFF) | ((mm & 0x3FFF) << 32) | ((n & 0x3FFF) << 46) | (cls << 60)
    return out


def widen_rows(rows3):
    """[k, 3] int64 rows (24-byte form) -> [k, 4] (32-byte form): what hc_narrow_payload_device / narrow_rows packed, unpacked."""
    w = rows3[:, 2]
    out = torch.empty((rows3.shape[0], 4), dtype=torch.int64, device=rows3.device)
    out[:, 0] = w & 0xFFFFFFFF
    out[:, 1], out[:, 2] = rows3[:, 0], rows3[:, 1]
    out[:, 3] = ((w >> 32) & 0x3FFF) | ((((w >> 46) & 0x3FFF) | (((w >> 60) & 0xF) << 28)) << 32)
    return out


def rows_fit_narrow(n_job, longest_sequence):
    """Whether a job's rows fit the 24-byte form: every global index below 2^32, every count of overlapped positions below 2^14."""
    return int(n_job) < (1 << 32) and int(longest_sequence) + 16 < (1 << 14)


def pack_payload(results, base_index, cap_rows, shuffle_seed=None, width=4):
    """The collection payload of one batch, built on the host (numpy) — the layout `hc_score_pack_device` /
    `hc_compact_pack_device` write on the device: int64 rows [index, x1 bits, x2 bits, mm | n_cls << 32], row 0 =
    [count, 0, 0, 0], then the records whose class is not DROP.  shuffle_seed: emit the rows in a random order,
    as the fused scoring kernel does.  Used where no device is involved (host callers, the gloo tests)."""
    res = np.asarray(results)
    cls = res["n_cls"] >> 28
    kept = np.nonzero(cls != 0)[0]
    if shuffle_seed is not None:
        kept = kept[np.random.default_rng(shuffle_seed).permutation(kept.size)]
    out = np.zeros((cap_rows + 1, 4), np.int64)
    out[0, 0] = kept.size
    k = kept[:cap_rows]
    out[1:1 + k.size, 0] = base_index + k
    out[1:1 + k.size, 1] = res["x1"][k].view(np.int64)
    out[1:1 + k.size, 2] = res["x2"][k].view(np.int64)
    out[1:1 + k.size, 3] = res["mm"][k].astype(np.int64) | (res["n_cls"][k].astype(np.int64) << 32)
    out = torch.from_numpy(out)
    if width == 3:  # the layout hc_narrow_payload_device writes: row 0 = [count, rows that did not fit, 0]
        narrow = torch.zeros((cap_rows + 1, 3), dtype=torch.int64)
        narrow[0, 0] = int(kept.size)
        narrow[1:1 + k.size] = narrow_rows(out[1:1 + k.size])
        return narrow
    return out


GATHER_MODES = ("ring", "direct", "root")


class PayloadGather:
    """The all-gather-v of the collection payloads, on any backend (nccl = RCCL, gloo).  Every rank contributes a payload whose row 0 is
    its count; `depth` buffer sets in turn; everything asynchronous on a side stream; `collect` (host side) waits, checks the capacity,
    trims by the counts, and sorts the rows by global index when their producer wrote them in no particular order.  Two forms of the
    exchange, selected by `mode`, bit-identical in what every rank ends up with:

      "ring"    ONE all-gather of the fixed-capacity payload (cap + 1 rows per rank whatever the counts): the library's own ring / tree
                over the links; nothing of it ever waits for the host.
      "direct"  the all-gather-V proper: the counts by one small all-gather, then the payload by grouped per-peer send / recv of exactly
                1 + count rows (batch_isend_irecv = one ncclGroup: every pair of ranks uses its own xGMI link, nothing is forwarded).
                The sizes of the send / recv calls must be known on the host: the host waits for the counts.  With lag = True (what
                StreamedGather.score_step does) the exchange of batch i is issued right behind the launch of batch i + 1's kernel, so
                the host reads the counts of batch i while the device runs batch i + 1: no bubble on the device.
      "root"    as "direct", but the rows travel to rank `root` only (the rank whose host inserts the edges): every other rank sends its
                1 + count rows over one link and receives nothing — a seventh of the bytes "direct" moves at world size 8, none of them
                into the memory of a rank that is scoring.  collect() returns the rows on the root and None elsewhere.
    width: 4 = 32-byte rows, 3 = 24-byte rows (module docstring); collect() widens.

    Timing: every batch's collective(s) are bracketed by events on the side stream (`gather_ms()`: mean duration since `reset_timings()`)."""

    def __init__(self, cap_rows, group=None, depth=2, device=None, mode="ring", lag=False, width=4, root=0):
        if mode not in GATHER_MODES:
            raise ValueError(f"gather mode {mode!r}: one of {GATHER_MODES}")
        if width not in (3, 4):
            raise ValueError("row width: 4 (32-byte rows) or 3 (24-byte rows)")
        self.cap, self.group, self.depth, self.mode, self.lag, self.width, self.root = int(cap_rows), group, depth, mode, lag, int(width), int(root)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.staged = self.cuda and dist.get_backend(group) == "gloo"  # test runs on a one-GPU box: collectives staged through the host
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        rows = self.cap + 1
        ev = (lambda: torch.cuda.Event()) if self.cuda else (lambda: None)
        self.bufs = [{"payload": torch.zeros((rows, self.width), dtype=torch.int64, device=self.device),
                      "all": torch.zeros((self.world * rows, self.width), dtype=torch.int64, device=self.device),
                      "counts_dev": torch.zeros(self.world, dtype=torch.int64, device=self.device),
                      "counts_host": torch.zeros(self.world, dtype=torch.int64, pin_memory=self.cuda),
                      "scored": ev(), "packed": ev(), "counted": ev(), "done": ev(),
                      "busy": False, "pending": False, "unordered": False} for _ in range(depth)]
        self.i = 0
        self._pending = []   # submitted batches whose exchange has not been issued yet (lag mode: until the next flush())
        self._timed = []     # (start event, end event) pairs on the side stream / (seconds,) on CPU
        self.steps_timed = 0

    # -- timing ---------------------------------------------------------------------------------------------------------------------
    def reset_timings(self):
        self._timed = []

    def _bracket(self):
        if self.cuda:
            return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        return None, None

    def gather_ms(self):
        """Mean duration of a batch's collective(s) since reset_timings(), in ms (events on the side stream; call after finish())."""
        per_batch = {}
        for key, a, b in self._timed[-4096:]:
            per_batch[key] = per_batch.get(key, 0.0) + (a.elapsed_time(b) if self.cuda else (b - a) * 1e3)
        return sum(per_batch.values()) / len(per_batch) if per_batch else 0.0

    # -- the two forms of the exchange ----------------------------------------------------------------------------------------------
    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _ring(self, b):
        """One all-gather of b["payload"] into b["all"] (current stream = the side stream on CUDA)."""
        if self.staged:
            torch.cuda.current_stream().synchronize()
            host_all = torch.empty(b["all"].shape, dtype=b["all"].dtype)
            dist.all_gather_into_tensor(host_all, b["payload"].cpu(), group=self.group)
            b["all"].copy_(host_all)
            return []
        return [dist.all_gather_into_tensor(b["all"], b["payload"], group=self.group, async_op=True)]

    def _counts(self, b):
        """direct, first half: every rank's count (row 0 of its payload) to every rank, and on to the host (which waits for them)."""
        mine = b["payload"][0, :1]
        if self.staged:
            torch.cuda.current_stream().synchronize()
            dist.all_gather_into_tensor(b["counts_host"], mine.cpu(), group=self.group)
        elif not self.cuda:
            dist.all_gather_into_tensor(b["counts_host"], mine.contiguous(), group=self.group)
        else:
            w = dist.all_gather_into_tensor(b["counts_dev"], mine, group=self.group, async_op=True)
            w.wait()  # the side stream waits, not the host
            b["counts_host"].copy_(b["counts_dev"], non_blocking=True)
            b["counted"].record(torch.cuda.current_stream())
            b["counted"].synchronize()  # the host needs the sizes of the send / recv calls
        b["counts"] = [int(c) for c in b["counts_host"].tolist()]
        return []

    def _exchange(self, b):
        """direct, second half: 1 + count rows to and from every peer, each pair of ranks on its own."""
        counts = b["counts"]
        if max(counts) > self.cap:  # every rank sees the same counts: every rank stops here, nothing has been posted
            b["overflow"] = max(counts)
            return []
        rows = self.cap + 1
        mine = 1 + counts[self.rank]
        src = b["payload"][:mine]
        # who sends to whom: "direct" every rank to every other; "root" every rank to the root only
        to = [p for p in range(self.world) if p != self.rank and (self.mode == "direct" or p == self.root)]
        frm = [p for p in range(self.world) if p != self.rank and (self.mode == "direct" or self.rank == self.root)]
        if self.staged or not self.cuda:
            host_src = src.cpu() if self.cuda else src
            host_all = torch.zeros(b["all"].shape, dtype=b["all"].dtype) if self.cuda else b["all"]
            host_all[self.rank * rows: self.rank * rows + mine] = host_src
            ops = [dist.P2POp(dist.isend, host_src, self._peer(p), self.group) for p in to]
            ops += [dist.P2POp(dist.irecv, host_all[p * rows: p * rows + 1 + counts[p]], self._peer(p), self.group) for p in frm]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if self.cuda:
                b["all"].copy_(host_all)
            return []
        b["all"][self.rank * rows: self.rank * rows + mine].copy_(src, non_blocking=True)
        ops = [dist.P2POp(dist.isend, src, self._peer(p), self.group) for p in to]
        ops += [dist.P2POp(dist.irecv, b["all"][p * rows: p * rows + 1 + counts[p]], self._peer(p), self.group) for p in frm]
        return dist.batch_isend_irecv(ops) if ops else []

    def _run_timed(self, b, key, fn):
        """fn(b) on the side stream between two events; the side stream waits for the works fn returns."""
        if self.cuda:
            with torch.cuda.stream(self.side):
                t0, t1 = self._bracket()
                t0.record(self.side)
                for w in fn(b) or []:
                    w.wait()
                t1.record(self.side)
            self._timed.append((key, t0, t1))
            if len(self._timed) > 8192:  # a long run keeps the latest few thousand brackets
                del self._timed[:4096]
        else:
            import time

            t0 = time.perf_counter()
            for w in fn(b) or []:
                w.wait()
            self._timed.append((key, t0, time.perf_counter()))

    def _issue(self, b, gate=None):
        """The exchange of one submitted batch: on the side stream, behind the work that wrote its payload.  gate: called with the side
        stream current in front of the collective(s) (StreamedGather: hc_comm_gate_device)."""
        key, on_side = b["key"], b.pop("on_side", None)
        if self.cuda:
            self.side.wait_event(b["scored"])
            with torch.cuda.stream(self.side):
                if on_side:
                    on_side()
                if gate:
                    gate()
        elif on_side:
            on_side()
        if self.mode == "ring":
            self._run_timed(b, key, self._ring)
        else:  # "direct", "root": the counts, then the rows
            self._run_timed(b, key, self._counts)
            self._run_timed(b, key, self._exchange)
        if self.cuda:
            b["done"].record(self.side)
        b["pending"] = False

    def flush(self, gate=None):
        """lag mode: the exchanges of the batches submitted so far.  StreamedGather calls it right behind the NEXT batch's kernel launch (with
        the gate that waits for that kernel's workgroups to have started); collect / finish / a buffer set's reuse call it without one."""
        while self._pending:
            self._issue(self._pending.pop(0), gate)

    # -- the batch protocol ---------------------------------------------------------------------------------------------------------
    def next_buffers(self):
        """The buffer set of the next batch; its payload may be written once the exchange that last used it is done
        (on CUDA the current stream is made to wait for it, on CPU it has completed already)."""
        b = self.bufs[self.i % self.depth]
        self.i += 1
        if b["pending"]:
            self.flush()
        if b["busy"]:
            if self.cuda:
                torch.cuda.current_stream().wait_event(b["done"])
            b["busy"] = False
        b.pop("overflow", None)
        b.pop("counts", None)
        return b

    def submit(self, b, on_side=None):
        """Hand over b["payload"] (written by work already enqueued on the current stream, or by on_side(), which is called with the side
        stream current, behind that work).  lag = False: the exchange is issued here.  lag = True: at the next flush()."""
        b["key"] = self.steps_timed
        self.steps_timed += 1
        if self.cuda:
            b["scored"].record(torch.cuda.current_stream())
        b["busy"], b["pending"], b["on_side"] = True, True, on_side
        self._pending.append(b)
        if not self.lag:
            self.flush()
        return b

    def _wait(self, b):
        if b["pending"]:
            self.flush()
        if self.cuda and b["busy"]:
            b["done"].synchronize()

    def collect(self, b):
        """Host side of one batch: (rows [sum k_r, 4] int64 ordered by global index, counts per rank)."""
        self._wait(b)
        rows = self.cap + 1
        if b.get("overflow"):
            raise OverflowError(f"a rank produced {b['overflow']} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        if self.mode == "root" and self.rank != self.root:  # this rank sent its rows and holds nobody else's
            if self.width == 3 and int(b["payload"][0, 1]) != 0:
                raise OverflowError(f"{int(b['payload'][0, 1])} rows of this rank do not fit the 24-byte form (index >= 2^32 or more than 16 383 positions)")
            return None, list(b["counts"])
        heads = [b["all"][r * rows] for r in range(self.world)]
        counts = [int(h[0]) for h in heads]
        if max(counts) > self.cap:
            raise OverflowError(f"a rank produced {max(counts)} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        if self.mode != "ring" and counts != b.get("counts"):
            raise RuntimeError(f"{self.mode} all-gather-v: the counts that travelled with the payloads {counts} are not the gathered counts {b.get('counts')}")
        if self.width == 3 and any(int(h[1]) != 0 for h in heads):
            raise OverflowError("rows that do not fit the 24-byte form (index >= 2^32 or more than 16 383 overlapped positions): exchange 32-byte rows (width=4)")
        out = torch.cat([b["all"][r * rows + 1: r * rows + 1 + counts[r]] for r in range(self.world)], dim=0)
        if self.width == 3:
            out = widen_rows(out)
        if b.get("unordered"):  # rows appended by the scoring kernel arrive in any order; they carry their index
            out = out[torch.argsort(out[:, 0])]
        return out, counts

    def finish(self):
        self.flush()
        if self.cuda:
            self.side.synchronize()
            torch.cuda.current_stream().synchronize()
        for b in self.bufs:
            b["busy"] = False


class StreamedGather(PayloadGather):
    """PayloadGather fed by the device — what `bench.py --gpus N` runs: no host round trip per batch.

    score_step: `hc_score_pack_device` — the scoring kernel's row-appending twin writes the payload itself (rows
    unordered); the exchange goes out on a side stream right behind the launch of the NEXT batch's kernel, which it overlaps.
    step: for results that exist already: `hc_compact_pack_device` (the library's own ordered selection + pack kernel, rows ordered) on the
    side stream; call before_write(results) before overwriting a results tensor that a step may still be reading.
    """

    def __init__(self, scorer, n_local, base_index, cap_rows, group=None, depth=2, rec_fmt=0, mode="ring", reserve_cus=0, narrow=False, root=0):
        super().__init__(cap_rows, group, depth, torch.device("cuda", torch.cuda.current_device()), mode, lag=True, width=3 if narrow else 4, root=root)
        # narrow: the kernels write 32-byte rows into a payload of their own; hc_narrow_payload_device (on the side stream, in front of the
        # exchange) makes the 24-byte rows that travel.  The caller decides per job (rows_fit_narrow); a row that does not fit is counted
        # in the payload's header and collect() raises.
        if narrow:
            for b in self.bufs:
                b["payload32"] = torch.zeros((self.cap + 1, 4), dtype=torch.int64, device=self.device)
        self.sc, self.n, self.base, self.fmt = scorer, int(n_local), int(base_index), int(rec_fmt)  # rec_fmt: records.REC_FULL / REC_COMPACT
        # reserve_cus > 0: the scoring launches leave that many CUs to the collective library's kernels, and every exchange waits (a gate
        # kernel on the side stream) until the scoring kernel it runs beside has taken its CUs: hc_set_comm_reserve / hc_comm_gate_device
        self.reserve = int(reserve_cus)
        self.sc.set_comm_reserve(self.reserve)
        # zero-initialised: entries beyond the count of a batch are stale but always valid indices
        self.idx = torch.zeros(max(self.n, 1), dtype=torch.int32, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.packed = {}  # data_ptr of a results tensor -> event: its rows have been packed, it may be overwritten

    def score_step(self, d_in_ptr, d_results):
        """Score this rank's n candidate records (device pointer) into d_results and collect the batch."""
        b = self.next_buffers()
        wide = b.get("payload32", b["payload"])
        b["unordered"] = self.sc.score_pack_device(d_in_ptr, self.n, d_results.data_ptr(), self.cap, self.base, wide.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream, self.fmt)
        # the exchange of the batch BEFORE this one goes out now: it runs beside this batch's kernel, which is in the stream already
        self.flush(gate=(lambda: self.sc.comm_gate_device(self.side.cuda_stream)) if self.reserve else None)
        return self.submit(b, on_side=(lambda: self._narrow(b)) if self.width == 3 else None)

    def _narrow(self, b):
        """32-byte rows -> the 24-byte rows that travel (side stream, behind the work that wrote them)."""
        self.sc.narrow_payload_device(b["payload32"].data_ptr(), self.cap, b["payload"].data_ptr(), self.side.cuda_stream)

    def close(self):
        """The scorer goes back to launches over the whole device (round-5 advisor: a reserve left behind made later kernel timings
        reduced-CU figures)."""
        self.finish()
        if self.reserve:
            self.sc.set_comm_reserve(0)
            self.reserve = 0

    def before_write(self, d_results):
        """The current stream waits until the batch that last used `d_results` has been read out of it."""
        ev = self.packed.get(d_results.data_ptr())
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def step(self, d_results):
        """Collect a batch whose n result records were written by work already enqueued on the current stream."""
        b = self.next_buffers()
        b["unordered"] = False

        def pack():  # (self.idx / self.count are shared: consecutive batches are ordered on the side stream)
            self.sc.compact_pack_device(d_results.data_ptr(), self.n, self.idx.data_ptr(), self.count.data_ptr(), self.cap, self.base,
                                        b.get("payload32", b["payload"]).data_ptr(), self.side.cuda_stream)
            if self.width == 3:
                self._narrow(b)
            b["packed"].record(self.side)
            self.packed[d_results.data_ptr()] = b["packed"]

        self.submit(b, on_side=pack)
        self.flush()  # (results that exist already: nothing to run beside, the pack and the exchange go out at once)
        return b
