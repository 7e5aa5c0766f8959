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
    firsts = [shard_range(n_images, r, world)[0] for r in range(world)]
    for depth, lag, n_steps in ((4, 2, 7), (3, 2, 5), (2, 1, 3), (4, 2, 1)):
        # the transfers of step k are posted `lag` steps later with the exact sizes of every rank;
        # payload sizes change from step to step (columns are emptied): the landing buffers grow
        pipe = PipelinedCompactGather(like, depth=depth, dst=0, lag=lag)
        seen = {}
        for k in range(n_steps):
            buf = pipe.next_buffer()
            buf.copy_(like)
            buf[..., 1] += k
            if k % 2 == 1:
                buf[0, :: (k + 1), 0, 0] = -1          # fewer sections in the odd steps
            pipe.submit()
            if rank == 0:
                for slot in range(depth):              # whatever has been posted so far is consistent
                    if pipe.result[slot] is not None and not pipe.works[slot]:
                        seen[slot] = True
        pipe.flush()
        if rank == 0:
            res = pipe.last_gathered()
            k = n_steps - 1
            for r, f in enumerate(firsts):
                want = full[f:f + 1].clone()
                want[..., 1] += k
                if k % 2 == 1:
                    want[0, :: (k + 1), 0, 0] = -1
                wc, wp = pack_sections(want)
                assert torch.equal(res[r][0], wc) and torch.equal(res[r][1], wp), (depth, lag, r)
            chk = pipe.check_last()
            assert chk["payload_consistent"] and chk["rank0_copy_equals_local"]
            st = pipe.stats()
            assert st["ratio_vs_fixed"] < 0.5 and st["lag_steps"] == lag
            assert st["landing_buffer_bytes_on_dst"] < 0.8 * (world - 1) * depth * like.numel() * 4
    dist.barrier()
    dist.destroy_process_group()


def _worker_subgroup(rank, world, port, out_path):
    """Three processes, the gather inside the sub-group of global ranks [1, 2] (group ranks 0, 1):
    group ranks differ from global ranks, `dst` is a GROUP rank everywhere in parallel.py."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_stixels_amd.parallel import (gather_compact, gather_sections, gather_variable, unpack_sections,
                                               PipelinedCompactGather, PipelinedGather, pack_sections)
    group = dist.new_group([1, 2])
    full = _sections_with_empty_columns(2)
    S = full.shape[2]
    if rank in (1, 2):
        grank = dist.get_rank(group)                   # global 1 -> 0, global 2 -> 1
        assert grank == rank - 1
        mine = full[grank:grank + 1].contiguous()
        dst = 1                                        # group rank 1 = global rank 2
        got = gather_compact(mine, dst=dst, group=group)
        lst = [torch.empty_like(mine) for _ in range(2)] if grank == dst else None
        gather_sections(mine, lst, dst=dst, group=group)
        var = gather_variable(mine, dst=dst, group=group)
        pipe = PipelinedCompactGather(mine, dst=dst, group=group)
        fpipe = PipelinedGather(mine, depth=2, dst=dst, group=group)
        for k in range(5):
            for pp in (pipe, fpipe):
                b = pp.next_buffer()
                b.copy_(mine)
                b[..., 2] += k
                pp.submit()
        pipe.flush(); fpipe.flush()
        if grank == dst:
            back = torch.cat([unpack_sections(c, p, S) for c, p in got], dim=0)
            wc, wp = pack_sections(full)
            assert torch.equal(back, unpack_sections(wc, wp, S))
            assert torch.equal(torch.cat(lst), full) and torch.equal(var, full)
            res = pipe.last_gathered()
            for r in range(2):
                want = full[r:r + 1].clone()
                want[..., 2] += 4
                c_, p_ = pack_sections(want)
                assert torch.equal(res[r][0], c_) and torch.equal(res[r][1], p_)
                assert torch.equal(fpipe.last_gathered()[r], want)
            np.save(out_path, back.numpy())
        else:
            assert got is None and var is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_inside_a_subgroup_uses_group_ranks(tmp_path):
    out = str(tmp_path / "sub.npy")
    mp.spawn(_worker_subgroup, args=(3, _free_port(), out), nprocs=3, join=True)
    assert os.path.exists(out)


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
