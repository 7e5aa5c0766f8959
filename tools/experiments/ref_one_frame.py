import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from instance_stixels_amd import host
dev = torch.device("cuda", 0)
for preset in ("drn_d_22_unary",):
    wr = bench.Workload(preset, 784, 1792, 128, 1, 1, dev, 0, seed0=211, invalid_disparity=0.0)
    for inst in (True, False):
        st = host.Stixels(); st.SetConfig(wr.cfg); st.SetDevice(0); st.Initialize()
        f = wr.frames[0]
        st.SetDisparityImage(f.disparity); st.SetSegmentation(f.segmentation)
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        st.Compute(wr.cfg.pairwise)
        ts = [st.time_compute(wr.cfg.pairwise, 200, inst) for _ in range(3)]
        print(os.environ.get("IS_P1_WIN_TILES", "-"), "instances", inst, ["%.3f ms" % (t * 1e3) for t in ts], flush=True)
        st.close()
