"""Multi-GPU use of the column DP: images are independent, so a batch is sharded over ranks
(one process per GPU) with no data-path collective; the only exchange is the final gather of the
stixel outputs to one rank (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_sections(local: torch.Tensor, gathered, dst: int = 0, group=None):
    """Gathers every rank's fixed-stride Section tensor ([B][C][S][8] int32) on `dst`.

    `gathered` is a list of world_size tensors on `dst` and None elsewhere."""
    rank = dist.get_rank(group)
    dist.gather(local, gather_list=gathered if rank == dst else None, dst=dst, group=group)
    return gathered


def gather_variable(local: torch.Tensor, dst: int = 0, group=None):
    """Gather of shards whose first dimension differs per rank (uneven batch split): sizes are
    exchanged first, shards are padded to the largest one, trimmed again on `dst`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    padded = local
    if local.shape[0] < m:
        pad = torch.zeros((m - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        padded = torch.cat([local, pad], dim=0)
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded.contiguous(), gather_list=out, dst=dst, group=group)
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
        self.inflight[slot] = dist.gather(
            self.buffers[slot], gather_list=self.gathered[slot] if self.rank == self.dst else None,
            dst=self.dst, group=self.group, async_op=True)
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
