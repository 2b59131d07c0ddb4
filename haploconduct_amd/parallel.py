"""Multi-GPU sharding of the edge-calculation path.

Candidates are independent given the read store, so the path shards with NO data-path
collective: rank r scores the contiguous slice shard_range(n, r, world) of the candidate
array against a replicated read store.  The only exchange is the collection of the non-dropped
records, once per batch (PayloadGather / StreamedGather below): either ONE all-gather of a
fixed-capacity payload whose row 0 is the count ("ring"), or the all-gather-v proper — the
counts by one small all-gather, then grouped per-peer send / recv of exactly the rows
("direct": every pair of ranks on its own xGMI link).  Works on any torch.distributed backend:
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Records travel as int64 rows
[global_index, x1_bits, x2_bits, mm | n_cls << 32]."""
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


def pack_payload(results, base_index, cap_rows, shuffle_seed=None):
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
    return torch.from_numpy(out)


GATHER_MODES = ("ring", "direct")


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

    Timing: every batch's collective(s) are bracketed by events on the side stream (`gather_ms()`: mean duration since `reset_timings()`)."""

    def __init__(self, cap_rows, group=None, depth=2, device=None, mode="ring", lag=False):
        if mode not in GATHER_MODES:
            raise ValueError(f"gather mode {mode!r}: one of {GATHER_MODES}")
        self.cap, self.group, self.depth, self.mode, self.lag = int(cap_rows), group, depth, mode, lag
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device if device is not None else torch.device("cpu")
        self.cuda = self.device.type == "cuda"
        self.staged = self.cuda and dist.get_backend(group) == "gloo"  # test runs on a one-GPU box: collectives staged through the host
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        rows = self.cap + 1
        ev = (lambda: torch.cuda.Event()) if self.cuda else (lambda: None)
        self.bufs = [{"payload": torch.zeros((rows, 4), dtype=torch.int64, device=self.device),
                      "all": torch.zeros((self.world * rows, 4), dtype=torch.int64, device=self.device),
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
        if self.staged or not self.cuda:
            host_src = src.cpu() if self.cuda else src
            host_all = torch.zeros(b["all"].shape, dtype=b["all"].dtype) if self.cuda else b["all"]
            host_all[self.rank * rows: self.rank * rows + mine] = host_src
            ops = []
            for p in range(self.world):
                if p != self.rank:
                    ops.append(dist.P2POp(dist.isend, host_src, self._peer(p), self.group))
                    ops.append(dist.P2POp(dist.irecv, host_all[p * rows: p * rows + 1 + counts[p]], self._peer(p), self.group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if self.cuda:
                b["all"].copy_(host_all)
            return []
        b["all"][self.rank * rows: self.rank * rows + mine].copy_(src, non_blocking=True)
        ops = []
        for p in range(self.world):
            if p != self.rank:
                ops.append(dist.P2POp(dist.isend, src, self._peer(p), self.group))
                ops.append(dist.P2POp(dist.irecv, b["all"][p * rows: p * rows + 1 + counts[p]], self._peer(p), self.group))
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
        else:
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
        counts = [int(b["all"][r * rows, 0]) for r in range(self.world)]
        if max(counts) > self.cap:
            raise OverflowError(f"a rank produced {max(counts)} records, capacity is {self.cap}: rerun the batch with a larger cap_rows")
        if self.mode == "direct" and counts != b.get("counts"):
            raise RuntimeError(f"direct all-gather-v: the counts that travelled with the payloads {counts} are not the gathered counts {b.get('counts')}")
        out = torch.cat([b["all"][r * rows + 1: r * rows + 1 + counts[r]] for r in range(self.world)], dim=0)
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

    def __init__(self, scorer, n_local, base_index, cap_rows, group=None, depth=2, rec_fmt=0, mode="ring", reserve_cus=0):
        super().__init__(cap_rows, group, depth, torch.device("cuda", torch.cuda.current_device()), mode, lag=True)
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
        b["unordered"] = self.sc.score_pack_device(d_in_ptr, self.n, d_results.data_ptr(), self.cap, self.base, b["payload"].data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream, self.fmt)
        # the exchange of the batch BEFORE this one goes out now: it runs beside this batch's kernel, which is in the stream already
        self.flush(gate=(lambda: self.sc.comm_gate_device(self.side.cuda_stream)) if self.reserve else None)
        return self.submit(b)

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
                                        b["payload"].data_ptr(), self.side.cuda_stream)
            b["packed"].record(self.side)
            self.packed[d_results.data_ptr()] = b["packed"]

        self.submit(b, on_side=pack)
        self.flush()  # (results that exist already: nothing to run beside, the pack and the exchange go out at once)
        return b
