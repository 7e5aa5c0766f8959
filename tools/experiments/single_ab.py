"""GPU box: single-frame Stixels::Compute() timings (ms) for a preset; env knobs apply."""
import sys, os
sys.path.insert(0, os.getcwd())
from instance_stixels_amd import make_config, synthetic, host
for preset in (sys.argv[1:] or ["drn_d_22_unary", "drn_d_38_pairwise"]):
    cfg = make_config(preset, 1024, 2048, 128)
    f = synthetic.make_frame(cfg, seed=17)
    st = host.Stixels(); st.SetConfig(cfg); st.Initialize()
    st.SetDisparityImage(f.disparity); st.SetSegmentation(f.segmentation)
    st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    a = st.time_compute(cfg.pairwise, 200, False) * 1e3
    b = st.time_compute(cfg.pairwise, 200, True) * 1e3
    print(f"{preset}: Compute {a:.4f} ms = {1e3 / a:.0f}/s; + GetInstanceStixels {b:.4f} ms = {1e3 / b:.0f}/s", flush=True)
    st.close()
