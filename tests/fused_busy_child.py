"""Child process of tests/test_fused_handover_gpu.py: its own HIP context on the same GPU, kept busy with pairwise
batches (many short dependent launches: what another tenant of the card looks like to a dispatcher) until the
parent removes the flag file or the time is up.  Prints BUSY_READY once the first call has finished."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    flag, seconds = sys.argv[1], float(sys.argv[2])
    import torch
    import helpers
    from instance_stixels_amd.core import Core
    case2 = helpers.build_case("drn_d_38_pairwise", 512, 1024, 64, seed=5, n_images=2)
    case = helpers.sub_case(case2, [i % 2 for i in range(8)])
    cfg = case["cfg"]
    core = Core(case["params"], case["lut"], case["odr"], max_batch=8)
    dev = torch.device("cuda", 0)
    big = torch.from_numpy(case["disparity"]).to(dev)
    seg = torch.from_numpy(case["segmentation"]).to(dev)
    p = case["params"]
    joined = torch.empty((8, p.cols, p.rows), dtype=torch.float32, device=dev)
    sec = torch.empty((8, p.cols, p.max_sections, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    t0, n = time.time(), 0
    while os.path.exists(flag) and time.time() - t0 < seconds:
        core.join_columns_ptr(big.data_ptr(), big.shape[2], False, joined.data_ptr(), 8, stream)
        core.compute_ptr(joined.data_ptr(), seg.data_ptr(), case["gf"], case["ng"], case["ig"], case["vhor"],
                         bool(cfg.pairwise), 8, sec.data_ptr(), stream=stream)
        if n % 4 == 3:
            torch.cuda.synchronize(dev)
        if n == 0:
            torch.cuda.synchronize(dev)
            print("BUSY_READY", flush=True)
        n += 1
    torch.cuda.synchronize(dev)
    core.close()
    print(f"BUSY_DONE {n} calls", flush=True)


if __name__ == "__main__":
    main()
