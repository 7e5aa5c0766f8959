"""Child process (one RANK) of test_capi_gather_two_ranks_on_one_gpu: never imported by pytest.

    capi_gather_ranks_child.py <rank> <nranks> <id file>

nranks processes share cuda:0; the communicator comes from tests/mock_rccl (IS_RCCL_LIB, set by the parent): the
C-ABI gather runs its REAL code with more than one rank -- the order of the collectives on every rank, counts and
offsets, the uneven-shard path, dst's go-ahead.  Per model (unary / pairwise), five frames split 3 + 2 (nranks = 2):
  1. is_pack_sections -> is_gather_sections -> is_unpack_sections on raw buffers, dst = 0 and dst = 1: dst holds the
     fixed-stride output of ALL frames (it computes every frame itself to compare);
  2. an undersized landing buffer on dst: EVERY rank gets IS_ENOMEM, and the next collective still matches;
  3. equal shards (2 + 2): the ncclGather path of the per-column counts;
  4. Stixels::ComputeBatchGather on every rank; dst's frames equal ComputeBatch of all frames and the oracle.
Prints RANK_OK <rank>."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import helpers  # noqa: E402
from instance_stixels_amd import core, host  # noqa: E402
from instance_stixels_amd.config import SECTION_DTYPE  # noqa: E402


def main():
    rank, nranks, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert nranks == 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if rank == 0:
        uid = core.comm_unique_id()
        with open(id_file + ".tmp", "wb") as fh:
            fh.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            assert time.time() - t0 < 120, "rank 0 never wrote the communicator id"
            time.sleep(0.01)
        uid = open(id_file, "rb").read()
    comm = core.comm_init_rank(nranks, uid, rank)
    stream = torch.cuda.current_stream(dev).cuda_stream

    for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
        case = helpers.build_case(preset, 128, 256, 32, seed=9, n_images=5)
        cfg, p = case["cfg"], case["params"]
        C, S = p.cols, p.max_sections
        full = helpers.run_core(case, want_tables=False)["sections"]              # every frame, computed here
        for shards in ([3, 2], [2, 2]):
            lo = sum(shards[:rank])
            mine_idx = list(range(lo, lo + shards[rank]))
            n_all = sum(shards)
            mine = torch.from_numpy(full[mine_idx].view(np.int32).reshape(len(mine_idx), C, S, 8).copy()).to(dev)
            ncol = len(mine_idx) * C
            counts = torch.empty(ncol, dtype=torch.int32, device=dev)
            offsets = torch.empty(ncol + 1, dtype=torch.int32, device=dev)
            packed = torch.empty((ncol * (S - 1), 8), dtype=torch.int32, device=dev)
            core.pack_sections_ptr(mine.data_ptr(), ncol, S, counts.data_ptr(), offsets.data_ptr(), packed.data_ptr(),
                                   stream)
            my_total = int(offsets[-1].item())
            columns = [s * C for s in shards]
            all_cols = sum(columns)
            for dst in (0, 1):
                cap = all_cols * (S - 1)
                all_counts = torch.zeros(all_cols, dtype=torch.int32, device=dev)
                all_packed = torch.zeros((cap, 8), dtype=torch.int32, device=dev)
                # ---- 2. too small on dst: refused on EVERY rank (the go-ahead is broadcast), nothing posted
                try:
                    core.gather_sections_ptr(comm, dst, columns, counts.data_ptr(), offsets.data_ptr(),
                                             packed.data_ptr(), all_counts.data_ptr(), all_packed.data_ptr(), 7, stream)
                except core.CoreError as e:
                    assert "cap_sections" in str(e), e
                else:
                    raise AssertionError(f"rank {rank}: an undersized landing buffer on dst must be refused everywhere")
                # ---- 1. / 3. the gather proper (the collective sequence still matches after the refusal)
                totals = core.gather_sections_ptr(comm, dst, columns, counts.data_ptr(), offsets.data_ptr(),
                                                  packed.data_ptr(), all_counts.data_ptr(), all_packed.data_ptr(), cap,
                                                  stream)
                torch.cuda.synchronize(dev)
                assert int(totals[rank]) == my_total, (totals, my_total)
                if rank == dst:
                    back = torch.zeros((all_cols, S, 8), dtype=torch.int32, device=dev)
                    off2 = torch.empty(all_cols + 1, dtype=torch.int32, device=dev)
                    core.unpack_sections_ptr(all_counts.data_ptr(), off2.data_ptr(), all_packed.data_ptr(), all_cols, S,
                                             back.data_ptr(), stream)
                    torch.cuda.synchronize(dev)
                    back = back.cpu().numpy().view(SECTION_DTYPE).reshape(n_all, C, S)
                    assert int(totals.sum()) == int(off2[-1].item())
                    for i in range(n_all):
                        assert helpers.sections_equal(full[i], back[i]), (preset, shards, dst, i)

        # ---- 4. the host class, 3 + 2 frames, dst = 1 (not the rank that made the id)
        shards, dst = [3, 2], 1
        lo = sum(shards[:rank])
        idx = list(range(lo, lo + shards[rank]))
        st = host.Stixels()
        st.SetConfig(cfg)
        st.Initialize(max_batch=5)
        big = torch.from_numpy(case["disparity"]).to(dev)
        seg = torch.from_numpy(case["segmentation"]).to(dev)
        road_all = [(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground) for f in case["frames"]]
        my_big, my_seg = big[idx].contiguous(), seg[idx].contiguous()   # (kept alive: the call takes raw pointers)
        got = st.ComputeBatchGather(cfg.pairwise, my_big.data_ptr(), my_seg.data_ptr(),
                                    [road_all[i] for i in idx], comm, dst, shards,
                                    road_all=road_all if rank == dst else None)
        if rank == dst:
            ref, _ = st.ComputeBatch(cfg.pairwise, big.data_ptr(), seg.data_ptr(), road_all, with_instances=False)
            assert len(got) == 5
            for i in range(5):
                assert helpers.sections_equal(ref[i].sections, got[i].sections), (preset, i)
                assert got[i].vhor == ref[i].vhor
                o = helpers.run_oracle(case, image=i)
                assert helpers.sections_equal(o["sections"], got[i].sections), (preset, i)
        else:
            assert got == []
        st.Finish()
    core.comm_destroy(comm)
    print(f"RANK_OK {rank}", flush=True)


if __name__ == "__main__":
    main()
