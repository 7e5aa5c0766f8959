"""GPU box: what the branch-and-bound evaluates on one input family (both models): images/s, DP ms, the evaluated
fractions (full / ground-sky-only) and the window misses.   usage: python tools/family_prune.py <family> [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main(family="cityscapes_like", batch=32):
    dev = torch.device("cuda", 0)
    for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
        wl = bench.Workload(preset, 1024, 2048, 128, batch, 8, dev, 0, family=family)
        core = wl.make_core()
        core.set_kernel_timing(True)
        dt = wl.time_steps(core, 3)
        kt = core.kernel_times_ms()
        ps = wl.prune_stats(core)
        core.set_eval_counters(True)
        wl.step(core)
        c = core.eval_counters()
        core.close()
        print(family, preset, "images/s", round(batch / dt), "dp_ms", round(kt["dp_ms"], 2), "evaluated", round(ps["evaluated_frac"], 3),
              "full", round(ps["full_eval_frac"], 3), "gs", round(ps["ground_sky_only_frac"], 3),
              "window_miss steps", c["p1_window_miss" if wl.cfg.pairwise else "unary_window_miss"])
        wl.free()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cityscapes_like", int(sys.argv[2]) if len(sys.argv) > 2 else 32)
