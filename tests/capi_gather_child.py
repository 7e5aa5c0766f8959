"""Child process of test_capi_rccl_gather_one_rank (never imported by pytest: no test_ prefix).

One rank, a real RCCL communicator created through the C ABI (is_comm_unique_id / is_comm_init_rank):
  1. is_pack_sections -> is_gather_sections -> is_unpack_sections on raw device buffers, compared with the
     fixed-stride Section output of the same is_compute call;
  2. an undersized landing buffer: IS_ENOMEM, nothing transferred, the totals reported;
  3. Stixels::ComputeBatchGather against Stixels::ComputeBatch and the oracle.
Prints GATHER_OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import helpers  # noqa: E402
from instance_stixels_amd import core, host  # noqa: E402
from instance_stixels_amd.config import SECTION_DTYPE  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    uid = core.comm_unique_id()
    comm = core.comm_init_rank(1, uid, 0)
    for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
        case = helpers.build_case(preset, 128, 256, 32, seed=5, n_images=3)
        cfg, p = case["cfg"], case["params"]
        C, S, n = p.cols, p.max_sections, 3
        got = helpers.run_core(case, want_tables=False)
        fixed = torch.from_numpy(got["sections"].view(np.int32).reshape(n, C, S, 8).copy()).to(dev)
        ncol = n * C
        counts = torch.empty(ncol, dtype=torch.int32, device=dev)
        offsets = torch.empty(ncol + 1, dtype=torch.int32, device=dev)
        packed = torch.empty((ncol * (S - 1), 8), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        core.pack_sections_ptr(fixed.data_ptr(), ncol, S, counts.data_ptr(), offsets.data_ptr(), packed.data_ptr(),
                               stream)
        total = int(offsets[-1].item())
        all_counts = torch.zeros(ncol, dtype=torch.int32, device=dev)
        all_packed = torch.zeros((total, 8), dtype=torch.int32, device=dev)
        totals = core.gather_sections_ptr(comm, 0, [ncol], counts.data_ptr(), offsets.data_ptr(), packed.data_ptr(),
                                          all_counts.data_ptr(), all_packed.data_ptr(), total, stream)
        torch.cuda.synchronize(dev)
        assert totals.tolist() == [total], (totals, total)
        assert torch.equal(all_counts, counts) and torch.equal(all_packed, packed[:total])
        back = torch.zeros((ncol, S, 8), dtype=torch.int32, device=dev)
        off2 = torch.empty(ncol + 1, dtype=torch.int32, device=dev)
        core.unpack_sections_ptr(all_counts.data_ptr(), off2.data_ptr(), all_packed.data_ptr(), ncol, S,
                                 back.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        back = back.cpu().numpy().view(SECTION_DTYPE).reshape(n, C, S)
        for i in range(n):
            assert helpers.sections_equal(got["sections"][i], back[i]), (preset, i)
        # an undersized landing buffer: refused on every rank, with the sizes
        try:
            core.gather_sections_ptr(comm, 0, [ncol], counts.data_ptr(), offsets.data_ptr(), packed.data_ptr(),
                                     all_counts.data_ptr(), all_packed.data_ptr(), total - 1, stream)
        except core.CoreError as e:
            assert "cap_sections" in str(e), e
        else:
            raise AssertionError("an undersized landing buffer must be refused")

        # ---- the host class: ComputeBatchGather == ComputeBatch == oracle
        st = host.Stixels()
        st.SetConfig(cfg)
        st.Initialize(max_batch=n)
        big = torch.from_numpy(case["disparity"]).to(dev)
        seg = torch.from_numpy(case["segmentation"]).to(dev)
        road = [(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground) for f in case["frames"]]
        ref, _ = st.ComputeBatch(cfg.pairwise, big.data_ptr(), seg.data_ptr(), road, with_instances=False)
        gathered = st.ComputeBatchGather(cfg.pairwise, big.data_ptr(), seg.data_ptr(), road, comm, 0, [n],
                                         road_all=road)
        st.Finish()
        assert len(gathered) == n
        for i in range(n):
            assert helpers.sections_equal(ref[i].sections, gathered[i].sections), (preset, i)
            assert gathered[i].vhor == ref[i].vhor and gathered[i].alpha_ground == ref[i].alpha_ground
            o = helpers.run_oracle(case, image=i)
            assert helpers.sections_equal(o["sections"], gathered[i].sections), (preset, i)
    core.comm_destroy(comm)
    print("GATHER_OK", flush=True)


if __name__ == "__main__":
    main()
