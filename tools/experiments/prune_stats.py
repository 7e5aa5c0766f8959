import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
    wl = bench.Workload(preset, 1024, 2048, 128, 16, 4, dev, 0)
    core = wl.make_core()
    ps = wl.prune_stats(core)
    print(preset, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in ps.items() if k != "how"})
    core.close(); wl.free()
