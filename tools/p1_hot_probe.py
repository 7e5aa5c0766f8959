"""GPU box: what pairwise phase 1 pays for memory.  A batch of 64 copies of ONE frame whose stixel columns are all equal
(the first 8 pixel columns tiled across the image, the first column's segmentation for every column) through the product
library and through a timing-only build whose phase 1 reads the tables of column (colg mod 8) -- the same values, L2-resident:
    tools/build_variant.sh p1hot is_k_pairwise -DIS_ABL_P1HOT
    python3 tools/p1_hot_probe.py base ; python3 tools/p1_hot_probe.py p1hot        (one library per process)"""
import os, sys
which = sys.argv[1] if len(sys.argv) > 1 else "base"
if which != "base":
    os.environ["IS_CORE_LIB"] = os.path.join(os.getcwd(), f"instance_stixels_amd/lib/variants/libis_core_{which}.so")
os.environ.setdefault("IS_PW_GROUPS", "1")
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench
from instance_stixels_amd import make_config, synthetic
cfg = make_config("drn_d_38_pairwise", 1024, 2048, 128)
f = synthetic.make_frame(cfg, seed=17)
f.disparity[:] = np.tile(f.disparity[:, 904:912], (1, 256))     # (a column group through a slab of the scene)
f.segmentation[:] = f.segmentation[113][None]
dev = torch.device("cuda", 0)
wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 64, 1, dev, 0, frames=[f])
for env in ({}, {"IS_NO_PRUNE": "1"}):
    core = wl.make_core(env=env)
    core.set_kernel_timing(True)
    dt = wl.time_steps(core, 3)
    kt = core.kernel_times_ms()
    sec = wl.d_sections[0].cpu().numpy()
    print(which, env or "pruned", "images/s %.0f" % (64 / dt), {k: round(v, 3) for k, v in kt.items()},
          "checksum", int(np.abs(sec.astype(np.int64)).sum() % 1000003), flush=True)
    core.close()
