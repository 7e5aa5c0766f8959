"""Multi-GPU use of the column DP: images are independent, so a batch is sharded over ranks
(one process per GPU) with no data-path collective; the only exchange is the final gather of the
stixel outputs to one rank (RCCL over xGMI on GPUs, gloo in the CPU tests): fixed-stride Section
tensors (gather_sections / PipelinedGather) or, ~10 x fewer bytes, per-column counts + the used
sections only (gather_compact / PipelinedCompactGather)."""
import inspect

import torch
import torch.distributed as dist

# `group_dst=` / `group_peer=` (ranks of the group instead of global ranks) exist since torch 2.6; older
# versions get the same call through the global rank of the group member
_HAS_GROUP_RANKS = "group_dst" in inspect.signature(dist.gather).parameters


def _global_rank(group, r):
    return r if group is None else dist.get_global_rank(group, r)


def _gather_to(tensor, gather_list, dst, group, async_op=False):
    """dist.gather to the GROUP rank `dst`."""
    if _HAS_GROUP_RANKS:
        return dist.gather(tensor, gather_list=gather_list, group=group, group_dst=dst, async_op=async_op)
    return dist.gather(tensor, gather_list=gather_list, group=group, dst=_global_rank(group, dst),
                       async_op=async_op)


def _p2p(op, tensor, peer, group):
    """dist.P2POp with the GROUP rank `peer`."""
    if _HAS_GROUP_RANKS:
        return dist.P2POp(op, tensor, group=group, group_peer=peer)
    return dist.P2POp(op, tensor, _global_rank(group, peer), group=group)


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_sections(local: torch.Tensor, gathered, dst: int = 0, group=None):
    """Gathers every rank's fixed-stride Section tensor ([B][C][S][8] int32) on `dst` (a rank of
    `group`, like everywhere in this module).

    `gathered` is a list of world_size tensors on `dst` and None elsewhere."""
    rank = dist.get_rank(group)
    _gather_to(local, gathered if rank == dst else None, dst, group)
    return gathered


def gather_variable(local: torch.Tensor, dst: int = 0, group=None):
    """Gather of shards whose first dimension differs per rank (uneven batch split): sizes are
    exchanged first, shards are padded to the largest one, trimmed again on `dst`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v) for v in torch.cat(sizes).cpu().tolist()]
    m = max(sizes)
    padded = local
    if local.shape[0] < m:
        pad = torch.zeros((m - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        padded = torch.cat([local, pad], dim=0)
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    _gather_to(padded.contiguous(), out, dst, group)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


class PipelinedGather:
    """Final gather of per-step outputs to `dst`, overlapped with the next step's compute.

    The column DP has no data-path collective; only its outputs travel.  Step k writes its
    Section tensor into buffer k % depth, the gather of that buffer is issued asynchronously (on
    the communication stream of the backend) and step k+1 computes into the other buffer meanwhile.
    `flush()` waits for everything that is still in flight."""

    def __init__(self, like: torch.Tensor, depth: int = 2, dst: int = 0, group=None):
        self.dst, self.group, self.depth = dst, group, depth
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.buffers = [torch.empty_like(like) for _ in range(depth)]
        self.gathered = None
        if self.rank == dst:
            self.gathered = [[torch.empty_like(like) for _ in range(self.world)]
                             for _ in range(depth)]
        self.inflight = [None] * depth
        self.step = 0

    def next_buffer(self) -> torch.Tensor:
        """Output buffer of the coming step; blocks (stream-wise) until its last gather is done."""
        slot = self.step % self.depth
        if self.inflight[slot] is not None:
            self.inflight[slot].wait()
            self.inflight[slot] = None
        return self.buffers[slot]

    def submit(self):
        """Issues the gather of the buffer handed out by the last next_buffer() call."""
        slot = self.step % self.depth
        self.inflight[slot] = _gather_to(
            self.buffers[slot], self.gathered[slot] if self.rank == self.dst else None, self.dst, self.group,
            async_op=True)
        self.step += 1
        return slot

    def last_local(self) -> torch.Tensor:
        """This rank's output buffer of the most recent step (valid after flush())."""
        return self.buffers[(self.step - 1) % self.depth]

    def last_gathered(self):
        """On `dst` after flush(): the list of all ranks' outputs of the most recent step."""
        return None if self.gathered is None else self.gathered[(self.step - 1) % self.depth]

    def flush(self):
        for i, w in enumerate(self.inflight):
            if w is not None:
                w.wait()
                self.inflight[i] = None

    def check_last(self, max_sections=None):
        """On `dst` after flush(): is rank dst's gathered copy its own output?"""
        got = self.last_gathered()
        same = bool(torch.equal(got[self.dst], self.last_local()))
        return {"kind": "fixed", "tensors": len(got), "bytes_per_rank": got[0].numel() * 4,
                "rank0_copy_equals_local": same}

    def stats(self):
        b = self.buffers[0].numel() * 4
        return {"kind": "fixed", "bytes_per_rank_per_step": b, "fixed_stride_bytes_per_rank_per_step": b}


# ---------------------------------------------------------------------------------------------
# Compacted gather: per-column counts + the used sections only
# ---------------------------------------------------------------------------------------------

def pack_sections(sections: torch.Tensor):
    """[..., C, S, 8] int32 fixed-stride Section tensor -> (counts [n_columns] int32, packed [N, 8]
    int32): the sections in front of each column's terminator (type == -1), in (image, column,
    section) order.  Device tensors go through the HIP kernels of the core (is_pack_sections: no
    host round trip except the caller's read of N); CPU tensors -- the gloo tests -- through torch."""
    S = sections.shape[-2]
    flat = sections.reshape(-1, S, 8)
    n_columns = flat.shape[0]
    if flat.is_cuda:
        from . import core
        counts = torch.empty(n_columns, dtype=torch.int32, device=flat.device)
        offsets = torch.empty(n_columns + 1, dtype=torch.int32, device=flat.device)
        packed = torch.empty((n_columns * (S - 1), 8), dtype=torch.int32, device=flat.device)
        flat = flat.contiguous()
        core.pack_sections_ptr(flat.data_ptr(), n_columns, S, counts.data_ptr(), offsets.data_ptr(),
                               packed.data_ptr(), torch.cuda.current_stream(flat.device).cuda_stream)
        return counts, packed[: int(offsets[-1].item())]
    term = flat[:, :, 0] == -1
    counts = torch.where(term.any(dim=1), term.to(torch.int32).argmax(dim=1),
                         torch.full((n_columns,), S - 1, dtype=torch.int64)).to(torch.int32)
    mask = torch.arange(S)[None, :] < counts[:, None]
    return counts, flat[mask]


def unpack_sections(counts: torch.Tensor, packed: torch.Tensor, max_sections: int) -> torch.Tensor:
    """Inverse of pack_sections: [n_columns][S][8] with a terminator behind each column's sections
    and zeros behind the terminator."""
    n_columns, S = counts.numel(), max_sections
    if counts.is_cuda:
        from . import core
        out = torch.zeros((n_columns, S, 8), dtype=torch.int32, device=counts.device)
        offsets = torch.empty(n_columns + 1, dtype=torch.int32, device=counts.device)
        buf = packed.contiguous() if packed.numel() else torch.zeros((1, 8), dtype=torch.int32,
                                                                      device=counts.device)
        core.unpack_sections_ptr(counts.data_ptr(), offsets.data_ptr(), buf.data_ptr(), n_columns, S,
                                 out.data_ptr(), torch.cuda.current_stream(counts.device).cuda_stream)
        return out
    out = torch.zeros((n_columns, S, 8), dtype=torch.int32)
    mask = torch.arange(S)[None, :] < counts[:, None]
    out[mask] = packed
    out[torch.arange(n_columns), counts.long(), 0] = -1
    return out


def _post_compact(counts, packed, sizes, dst, group, recv=None):
    """The point-to-point part of the compacted gather: counts + packed sections of every rank to
    the GROUP rank `dst`, sized by `sizes` = [(n_columns_r, n_sections_r)] of all group ranks (known
    on every rank).  `recv(r, n_columns, n_sections)` (optional) returns the landing tensors of
    rank r.  Returns (out, works): on `dst` the list of (counts_r, packed_r) per group rank (its
    own: the inputs), None elsewhere; `works`: the asynchronous handles in flight."""
    rank = dist.get_rank(group)
    dev = counts.device
    ops, out = [], None
    if rank == dst:
        out = []
        for r, (nc, n) in enumerate(sizes):
            if r == dst:
                out.append((counts, packed))
                continue
            if recv is not None:
                c_r, p_r = recv(r, nc, n)
            else:
                c_r = torch.empty(nc, dtype=torch.int32, device=dev)
                p_r = torch.empty((n, 8), dtype=torch.int32, device=dev)
            # group_peer: ranks of `group` (with a sub-group they differ from the global ranks)
            ops.append(_p2p(dist.irecv, c_r, r, group))
            if n > 0:
                ops.append(_p2p(dist.irecv, p_r, r, group))
            out.append((c_r, p_r))
    else:
        ops.append(_p2p(dist.isend, counts, dst, group))
        if packed.shape[0] > 0:
            ops.append(_p2p(dist.isend, packed, dst, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return out, works


def _exchange_compact(counts, packed, dst, group, recv=None):
    """Sizes by all_gather (ONE device-to-host copy of all of them), then _post_compact.  `dst` is
    a rank of `group`.  Returns (out, works, sizes)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([counts.numel(), packed.shape[0]], dtype=torch.int64, device=counts.device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes = [tuple(int(v) for v in row) for row in torch.stack(sizes).cpu().tolist()]
    out, works = _post_compact(counts, packed, sizes, dst, group, recv)
    return out, works, sizes


def gather_compact(sections: torch.Tensor, dst: int = 0, group=None):
    """Compacted gather of every rank's Section tensor ([n_r][C][S][8] int32, n_r may differ per
    rank) on the GROUP rank `dst`: returns there the list of per-rank (counts, packed) pairs (see
    pack_sections), None on the other ranks.  About 10 x fewer bytes than the fixed-stride gather:
    10-40 of the 200 slots of a column are used."""
    counts, packed = pack_sections(sections)
    out, works, _ = _exchange_compact(counts, packed, dst, group)
    for w in works:
        w.wait()
    return out


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class PipelinedCompactGather:
    """PipelinedGather with the compacted payload, without a host wait on recent GPU work.

    Step k computes into buffer k % depth; its pack kernels are queued behind it and the two
    sizes of the payload (columns, sections) are all-gathered ASYNCHRONOUSLY, on the device, on a
    side stream that waits only for the pack.  The point-to-point transfers of step k are posted
    `lag` (= 2) steps later, with the exact sizes of every rank: the host reads the gathered sizes
    of a step whose kernels ran two steps ago (one small copy on the side stream, nothing of the
    compute stream in front of it), so it never waits for work the GPU has not finished long ago,
    never leaves the GPU without queued work, and no size is ever guessed (no truncation, no
    worst-case transfer).  The transfer of step k overlaps the compute of steps k+2, k+3; its
    buffers are reused at step k + depth (depth >= lag + 1; with lag + 1 the step that reuses a slot
    waits for a transfer posted one step earlier, lag + 2 and more leave it a whole step).  `dst` is
    a rank of `group`.  The
    landing buffers on `dst` grow on demand from what really arrives (not depth x ranks x the
    worst case).  `flush()` finishes what is still pending."""

    def __init__(self, like: torch.Tensor, core=None, depth: int = 4, dst: int = 0, group=None, lag: int = 2):
        if depth < lag + 1:
            raise ValueError("PipelinedCompactGather: depth must be at least lag + 1")
        self.dst, self.group, self.depth, self.lag = dst, group, depth, lag
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.S = like.shape[-2]
        self.n_columns = like.numel() // (self.S * 8)
        dev = like.device
        self.dev = dev
        self.cuda = dev.type == "cuda"
        cap = self.n_columns * (self.S - 1)
        self.buffers = [torch.empty_like(like) for _ in range(depth)]
        self.counts = [torch.empty(self.n_columns, dtype=torch.int32, device=dev) for _ in range(depth)]
        self.offsets = [torch.empty(self.n_columns + 1, dtype=torch.int32, device=dev) for _ in range(depth)]
        self.packed = [torch.empty((cap, 8), dtype=torch.int32, device=dev) for _ in range(depth)]
        self.size_mine = [torch.tensor([self.n_columns, 0], dtype=torch.int64, device=dev) for _ in range(depth)]
        self.size_all = [[torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.world)]
                         for _ in range(depth)]
        self.size_work = [None] * depth
        self.events = [torch.cuda.Event() if self.cuda else None for _ in range(depth)]
        self.comm = torch.cuda.Stream(dev) if self.cuda else None
        self.landing = [dict() for _ in range(depth)]   # slot -> {rank: (counts, packed)} grown on demand
        self.retired = [[] for _ in range(depth)]       # outgrown landing buffers, kept until the slot's next use
        self.works = [[] for _ in range(depth)]
        self.result = [None] * depth
        self.sizes = [None] * depth
        self.pending = []          # slots whose transfers are not posted yet, oldest first
        self.step = 0
        self.host_seconds = {"post": 0.0, "wait": 0.0}   # host time inside the size read / the waits

    # ---- landing buffers of rank r in a slot: reused, grown by half when a payload outgrows them
    def _landing(self, slot, r, nc, n):
        """Allocated on the CONSUMER's stream (the stream that was current when the transfers are posted), not
        on the communication stream this runs under: the caching allocator ties a block to the stream it was
        allocated on, and the results are read on the compute stream; the communication stream's use is
        declared with record_stream.  An outgrown buffer stays alive until the slot is used again, so a reader
        of the previous result that is still queued never sees its block reused."""
        c_r, p_r = self.landing[slot].get(r, (None, None))
        grow_c = c_r is None or c_r.numel() < nc
        grow_p = p_r is None or p_r.shape[0] < n
        if grow_c or grow_p:
            self.retired[slot].append((c_r, p_r))
            ctx = torch.cuda.stream(self._consumer) if self.cuda else _NullCtx()
            with ctx:
                if grow_c:
                    c_r = torch.empty(nc, dtype=torch.int32, device=self.dev)
                if grow_p:
                    p_r = torch.empty((max(n, int(1.5 * (0 if p_r is None else p_r.shape[0]))), 8),
                                      dtype=torch.int32, device=self.dev)
            if self.cuda:
                c_r.record_stream(self.comm)
                p_r.record_stream(self.comm)
        self.landing[slot][r] = (c_r, p_r)
        return c_r[:nc], p_r[:n]

    def next_buffer(self) -> torch.Tensor:
        slot = self.step % self.depth
        self._finish(slot)   # the slot's packed buffers may still be on the wire
        return self.buffers[slot]

    def _finish(self, slot):
        import time
        self.retired[slot].clear()           # (the result of the slot's previous use is out of reach by now)
        while slot in self.pending:          # (only when fewer than `lag` steps follow: flush, tiny runs)
            self._post(self.pending[0])
        t0 = time.perf_counter()
        for w in self.works[slot]:
            w.wait()
        self.works[slot] = []
        self.host_seconds["wait"] += time.perf_counter() - t0

    def submit(self):
        slot = self.step % self.depth
        if self.cuda:
            from . import core
            stream = torch.cuda.current_stream(self.dev)
            core.pack_sections_ptr(self.buffers[slot].data_ptr(), self.n_columns, self.S,
                                   self.counts[slot].data_ptr(), self.offsets[slot].data_ptr(),
                                   self.packed[slot].data_ptr(), stream.cuda_stream)
            self.size_mine[slot][1:2].copy_(self.offsets[slot][-1:])       # (device to device)
            self.events[slot].record(stream)
            self.comm.wait_event(self.events[slot])
            with torch.cuda.stream(self.comm):
                self.size_work[slot] = dist.all_gather(self.size_all[slot], self.size_mine[slot],
                                                       group=self.group, async_op=True)
        else:
            c, p = pack_sections(self.buffers[slot])
            self.counts[slot].copy_(c)
            self.packed[slot][: p.shape[0]].copy_(p)
            self.size_mine[slot][1] = p.shape[0]
            self.size_work[slot] = dist.all_gather(self.size_all[slot], self.size_mine[slot],
                                                   group=self.group, async_op=True)
        self.pending.append(slot)
        self.step += 1
        while len(self.pending) > self.lag:   # the transfers of the step `lag` steps back
            self._post(self.pending[0])
        return slot

    def _post(self, slot):
        import time
        t0 = time.perf_counter()
        self.pending.remove(slot)

        def go():
            self.size_work[slot].wait()
            sizes = [tuple(int(v) for v in row) for row in torch.stack(self.size_all[slot]).cpu().tolist()]
            n = sizes[self.rank][1]
            out, works = _post_compact(self.counts[slot], self.packed[slot][:n], sizes, self.dst, self.group,
                                       lambda r, nc, m: self._landing(slot, r, nc, m))
            return out, works, sizes
        if self.cuda:
            self._consumer = torch.cuda.current_stream(self.dev)
            with torch.cuda.stream(self.comm):   # nothing of the compute stream in front of the size copy
                out, works, sizes = go()
        else:
            out, works, sizes = go()
        self.works[slot], self.result[slot], self.sizes[slot] = works, out, sizes
        self.host_seconds["post"] += time.perf_counter() - t0

    def flush(self):
        while self.pending:
            self._post(self.pending[0])
        for slot in range(self.depth):
            for w in self.works[slot]:
                w.wait()
            self.works[slot] = []
        if self.comm is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.comm)

    def last_local(self) -> torch.Tensor:
        return self.buffers[(self.step - 1) % self.depth]

    def last_gathered(self):
        """On `dst` after flush(): [(counts_r, packed_r)] of the most recent step."""
        return self.result[(self.step - 1) % self.depth]

    def check_last(self, max_sections=None):
        """On `dst` after flush(): every rank's payload is consistent (its counts sum to its
        packed length) and rank dst's own unpacks to its fixed-stride output."""
        got = self.last_gathered()
        ok = all(int(c.sum().item()) == p.shape[0] for c, p in got)
        c0, p0 = got[self.dst]
        local = self.last_local().reshape(-1, self.S, 8)
        back = unpack_sections(c0, p0, self.S)
        idx = torch.arange(self.S, device=local.device)[None, :]
        body = idx < c0[:, None]
        same = bool(torch.equal(back[body], local[body])) and \
            bool((local[..., 0][idx == c0[:, None]] == -1).all())
        return {"kind": "compact", "ranks": len(got), "payload_consistent": ok,
                "rank0_copy_equals_local": bool(ok and same),
                "sections_per_rank": [int(p.shape[0]) for _, p in got]}

    def stats(self):
        slot = (self.step - 1) % self.depth
        sizes = self.sizes[slot] or []
        fixed = self.buffers[0].numel() * 4
        per_rank = [4 * nc + 32 * n for nc, n in sizes]
        landing = sum(c.numel() * 4 + p.numel() * 4 for d in self.landing for c, p in d.values())
        return {"kind": "compact", "bytes_per_rank_per_step": per_rank,
                "fixed_stride_bytes_per_rank_per_step": fixed,
                "ratio_vs_fixed": (sum(per_rank) / len(per_rank) / fixed) if per_rank else None,
                "lag_steps": self.lag, "depth": self.depth, "landing_buffer_bytes_on_dst": landing,
                "host_seconds_in_size_read_and_posting": self.host_seconds["post"],
                "host_seconds_in_waits": self.host_seconds["wait"],
                "host_seconds_note": "the size read of step k - lag blocks the host until the device has "
                                     "packed that step -- with steps k - lag + 1 .. k already queued behind it: "
                                     "back-pressure of a host that runs `lag` steps ahead, not a device stall "
                                     "(see exposed_ms_per_step)"}
