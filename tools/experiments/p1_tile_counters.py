"""Per-tile evaluation counters of the pairwise phase 1 (run on the GPU box): wave-steps per
(column, tile) and wave, split into full / window-miss / ground-sky-only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main(family="scene", batch=16):
    dev = torch.device("cuda", 0)
    wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, batch, 4, dev, 0, family=family)
    core = wl.make_core()
    core.set_eval_counters(True)
    wl.step(core)
    c = core.eval_counters()
    core.close()
    ncols = wl.B * wl.C
    print(family, "vhor", wl.vh[0], "full", c["p1_full"], "window_miss", c["p1_window_miss"], "gs", c["p1_gs"])
    for t in range(wl.H // 64):
        f, l, g = c["p1_per_tile"][t]
        print(f"tile {t:2d}: per (column, tile) wave-steps: full {f / ncols:7.1f}  window-miss {l / ncols:7.1f}  gs {g / ncols:7.1f}"
              f"   per wave: {f / ncols / 8:5.1f} {l / ncols / 8:5.1f} {g / ncols / 8:5.1f}")

if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["scene"]))
