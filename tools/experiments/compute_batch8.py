import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from instance_stixels_amd import host
dev = torch.device("cuda", 0)
for preset in ("drn_d_22_unary", "drn_d_38_pairwise"):
    wh = bench.Workload(preset, 1024, 2048, 128, 8, 8, dev, 0)
    st = host.Stixels(); st.SetConfig(wh.cfg); st.SetDevice(0); st.Initialize(max_batch=8)
    road = [(wh.frames[i].vhor_image, wh.frames[i].camera_tilt, wh.frames[i].camera_height, wh.frames[i].alpha_ground) for i in wh.pick]
    t_b = st.time_compute_batch(wh.cfg.pairwise, wh.d_big.data_ptr(), wh.d_seg.data_ptr(), road, 20, False)
    t_bi = st.time_compute_batch(wh.cfg.pairwise, wh.d_big.data_ptr(), wh.d_seg.data_ptr(), road, 20, True)
    print(preset, "ComputeBatch8 images/s %.0f, with instance mappings %.0f" % (8 / t_b, 8 / t_bi), flush=True)
    st.close(); wh.free()
