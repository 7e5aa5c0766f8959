"""The N>1 path on CPU: world_size 2 over gloo.  Each rank runs its shard of a batch (here with
the oracle standing in for the device step -- the sharding / gather logic under test is the
product's instance_stixels_amd.parallel) and rank 0 must end up with exactly the single-process
result, in image order."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_stixels_amd.parallel import (shard_range, gather_variable, gather_sections,
                                               PipelinedGather)
    case = helpers.build_case("drn_d_22_unary", 64, 64, 32, seed=21, n_images=n_images)
    lo, hi = shard_range(n_images, rank, world)
    local = np.stack([helpers.run_oracle(case, image=i)["sections"] for i in range(lo, hi)])
    t = torch.from_numpy(local.view(np.int32).reshape(hi - lo, local.shape[1], local.shape[2], 8))
    full = gather_variable(t, dst=0)
    # fixed-stride gather (what bench.py uses when every rank has the same batch)
    fixed = t[:1].contiguous()
    lst = [torch.empty_like(fixed) for _ in range(world)] if rank == 0 else None
    gather_sections(fixed, lst, dst=0)
    # pipelined (double-buffered, asynchronous) gather of 3 consecutive "steps": step k sends
    # `fixed + k`; slots alternate 0, 1, 0, so slot 0 ends with step 2 and slot 1 with step 1
    pipe = PipelinedGather(fixed, depth=2, dst=0)
    slots = []
    for k in range(3):
        buf = pipe.next_buffer()
        buf.copy_(fixed + k)
        slots.append(pipe.submit())
    pipe.flush()
    assert slots == [0, 1, 0]
    if rank == 0:
        firsts = [shard_range(n_images, r, world)[0] for r in range(world)]
        for r, f in enumerate(firsts):
            assert torch.equal(pipe.gathered[0][r][0], full[f] + 2)
            assert torch.equal(pipe.gathered[1][r][0], full[f] + 1)
    if rank == 0:
        np.save(out_path, full.numpy())
        firsts = [shard_range(n_images, r, world)[0] for r in range(world)]
        for r, f in enumerate(firsts):
            assert torch.equal(lst[r][0], full[f])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from instance_stixels_amd.parallel import shard_range
    for n in (1, 5, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_equals_single_process(tmp_path):
    n_images, world = 5, 2                       # uneven split: 3 + 2
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_images, out), nprocs=world, join=True)
    got = np.load(out)
    case = helpers.build_case("drn_d_22_unary", 64, 64, 32, seed=21, n_images=n_images)
    want = np.stack([helpers.run_oracle(case, image=i)["sections"] for i in range(n_images)])
    assert np.array_equal(got.reshape(-1), want.view(np.int32).reshape(-1))
