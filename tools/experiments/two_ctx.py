"""Throughput with TWO contexts in flight on two streams (alternate batches), against one context:
does the HBM-write-bound prepare kernel of one batch overlap the VALU-bound DP kernels of the other?
    python tools/two_ctx.py [preset] [n_ctx]"""
import sys
import time

sys.path.insert(0, ".")
import torch
import bench

preset = sys.argv[1] if len(sys.argv) > 1 else "drn_d_22_unary"
dev = torch.device("cuda", 0)
wl = bench.Workload(preset, 1024, 2048, 128, 64, 2, dev, 0)
for nctx in (1, 2, 3):
    cores = [wl.make_core() for _ in range(nctx)]
    streams = [torch.cuda.Stream(dev) for _ in range(nctx)]
    joined = [torch.empty_like(wl.d_joined) for _ in range(nctx)]
    outs = [torch.empty_like(wl.d_sections) for _ in range(nctx)]

    def step(i):
        k = i % nctx
        s = streams[k].cuda_stream
        cores[k].join_columns_ptr(wl.d_big.data_ptr(), wl.W, wl.cfg.median_join, joined[k].data_ptr(), wl.B, s)
        cores[k].compute_ptr(joined[k].data_ptr(), wl.d_seg.data_ptr(), wl.gf, wl.ng, wl.ig, wl.vh,
                             wl.cfg.pairwise, wl.B, outs[k].data_ptr(), None, None, None, s)

    for i in range(2 * nctx):
        step(i)
    torch.cuda.synchronize(dev)
    K = 24
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / K
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{preset} contexts in flight {nctx}: {dt * 1e3:.3f} ms/step, {wl.B / dt:.0f} images/s, outputs equal: {same}", flush=True)
    for c in cores:
        c.close()
