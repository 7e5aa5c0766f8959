"""Window misses of the DP kernels with every tile windowed (GPU box): python tools/win_miss.py [family]"""
import os, sys
os.environ["IS_P1_WIN_TILES"] = "99"
sys.path.insert(0, ".")
import torch, bench
fam = sys.argv[1] if len(sys.argv) > 1 else "scene"
dev = torch.device("cuda", 0)
for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
    wl = bench.Workload(preset, 1024, 2048, 128, 16, 4, dev, 0, family=fam)
    core = wl.make_core()
    core.set_eval_counters(True)
    wl.step(core)
    c = core.eval_counters()
    core.close()
    if preset.endswith("unary"):
        print(fam, preset, "full", c["unary_full"], "gs", c["unary_gs"], "window-miss steps", c["unary_window_miss"])
    else:
        print(fam, preset, "full", c["p1_full"], "window-miss steps", c["p1_window_miss"],
              "per tile %:", [round(100.0 * c["p1_per_tile"][t][1] / max(1, c["p1_per_tile"][t][0]), 1) for t in range(16)])
    wl.free()
