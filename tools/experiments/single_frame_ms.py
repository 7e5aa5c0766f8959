"""GPU box: ms per call of ONE 1024x2048x128 frame through the C ABI (device-resident in/out), per preset; IS_CORE_LIB / IS_* apply."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
for preset in (sys.argv[1:] or ["drn_d_38_pairwise"]):
    wl = bench.Workload(preset, 1024, 2048, 128, 1, 1, dev, 0)
    c = wl.make_core()
    for _ in range(20):
        wl.step(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        wl.step(c)
    torch.cuda.synchronize()
    print(os.environ.get("IS_CORE_LIB", "product").split("/")[-1], preset, "%.4f ms per frame" % ((time.perf_counter() - t0) / 200 * 1e3), flush=True)
    c.close()
