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


def _sections_with_empty_columns(n_images):
    """Oracle sections of a small case with two columns emptied (terminator first): the compacted
    gather must carry zero-length columns."""
    case = helpers.build_case("drn_d_22_unary", 64, 64, 32, seed=21, n_images=n_images)
    sec = np.stack([helpers.run_oracle(case, image=i)["sections"] for i in range(n_images)])
    t = torch.from_numpy(sec.view(np.int32).reshape(n_images, sec.shape[1], sec.shape[2], 8).copy())
    t[0, 1, 0, 0] = -1
    t[n_images - 1, 3, 0, 0] = -1
    return t


def _worker_compact(rank, world, port, n_images, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_stixels_amd.parallel import (shard_range, gather_compact, unpack_sections,
                                               pack_sections, PipelinedCompactGather)
    full = _sections_with_empty_columns(n_images)
    S = full.shape[2]
    lo, hi = shard_range(n_images, rank, world)            # uneven: 3 + 2 images
    got = gather_compact(full[lo:hi].contiguous(), dst=0)
    if rank == 0:
        assert len(got) == world
        back = torch.cat([unpack_sections(c, p, S) for c, p in got], dim=0)
        np.save(out_path, back.numpy())
        n_sent = sum(4 * c.numel() + 32 * p.shape[0] for c, p in got)
        assert n_sent < 0.5 * full.numel() * 4             # far fewer bytes than the fixed stride
    else:
        assert got is None
    # pipelined: three steps, step k sends the first image of the shard with vB fields + k
    like = full[lo:lo + 1].contiguous()
    pipe = PipelinedCompactGather(like, depth=2, dst=0)
    for k in range(3):
        buf = pipe.next_buffer()
        buf.copy_(like)
        buf[..., 1] += k
        pipe.submit()
    pipe.flush()
    if rank == 0:
        res = pipe.last_gathered()
        firsts = [shard_range(n_images, r, world)[0] for r in range(world)]
        for r, f in enumerate(firsts):
            want = full[f:f + 1].clone()
            want[..., 1] += 2
            wc, wp = pack_sections(want)
            assert torch.equal(res[r][0], wc) and torch.equal(res[r][1], wp), r
        chk = pipe.check_last()
        assert chk["payload_consistent"] and chk["rank0_copy_equals_local"]
        st = pipe.stats()
        assert st["ratio_vs_fixed"] < 0.5
    dist.barrier()
    dist.destroy_process_group()


def test_pack_unpack_sections_round_trip():
    from instance_stixels_amd.parallel import pack_sections, unpack_sections
    t = _sections_with_empty_columns(2)
    S = t.shape[2]
    counts, packed = pack_sections(t)
    assert counts.numel() == t.shape[0] * t.shape[1] and int(counts.sum()) == packed.shape[0]
    assert counts[1] == 0 and counts[t.shape[1] + 3] == 0           # the emptied columns
    back = unpack_sections(counts, packed, S).reshape(t.shape)
    flat, bflat = t.reshape(-1, S, 8), back.reshape(-1, S, 8)
    for c in range(flat.shape[0]):
        n = int(counts[c])
        assert torch.equal(flat[c, :n], bflat[c, :n]) and bflat[c, n, 0] == -1 and flat[c, n, 0] == -1
    c2, p2 = pack_sections(back)                                    # idempotent
    assert torch.equal(c2, counts) and torch.equal(p2, packed)


def test_two_rank_compact_gather_equals_single_process(tmp_path):
    n_images, world = 5, 2
    out = str(tmp_path / "compact.npy")
    mp.spawn(_worker_compact, args=(world, _free_port(), n_images, out), nprocs=world, join=True)
    got = torch.from_numpy(np.load(out))
    from instance_stixels_amd.parallel import pack_sections, unpack_sections
    full = _sections_with_empty_columns(n_images)
    c, p = pack_sections(full)
    want = unpack_sections(c, p, full.shape[2])
    assert torch.equal(got, want)
