"""Debug (GPU box): per-tile phase-1 step counters of ONE stixel column of a synthetic frame, by running a
frame whose 256 columns are all copies of that column (instance centres shift with the column index, the
costs do not)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

def main(col=36, family="scene"):
    dev = torch.device("cuda", 0)
    wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 1, 1, dev, 0, family=family)
    f = wl.frames[0]
    d = f.disparity.copy()
    d[:] = np.tile(d[:, 8 * col:8 * col + 8], (1, 256))
    seg = np.repeat(f.segmentation[col:col + 1], 256, axis=0)
    wl.d_big.copy_(torch.from_numpy(d[None]).to(dev))
    wl.d_seg.copy_(torch.from_numpy(seg[None]).to(dev))
    core = wl.make_core()
    core.set_eval_counters(True)
    wl.step(core)
    c = core.eval_counters()
    core.close()
    print("column", col, "full/wave per tile:", " ".join(f"{c['p1_per_tile'][t][0] / 256 / 8:.1f}" for t in range(16)))

if __name__ == "__main__":
    for c in sys.argv[1:] or ["36"]:
        main(int(c))
