"""GPU box, debug library built with -DISF_DEBUG_HANDOVER (tools/build_variant.sh dbgho is_k_unary_fast -DISF_DEBUG_HANDOVER):
why DP workgroups distrust the hand-over of the LUT units -- timeouts, or units on another XCD (matrix reader x unit)."""
import os, sys
os.environ["IS_CORE_LIB"] = os.path.join(os.getcwd(), "instance_stixels_amd/lib/variants/libis_core_dbgho.so")
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np
import helpers
from instance_stixels_amd.core import Core
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cols = 4096 if D == 256 else 2048
case2 = helpers.build_case("drn_d_22_unary", 1024, cols, D, seed=59, n_images=2)
case = helpers.sub_case(case2, [i % 2 for i in range(8)])
for mode in (sys.argv[2:] or ["4", "1"]):
    os.environ["IS_LUT_FUSED"] = mode
    core = Core(case["params"], case["lut"], case["odr"], max_batch=8)
    core.set_eval_counters(True)
    cfg = case["cfg"]
    core.run(disparity_big=case["disparity"], segmentation=case["segmentation"], ground_function=case["gf"],
             normalization_ground=case["ng"], inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=False,
             median_join=False, want_tables=False, want_instances=False)
    c = core.eval_counters()
    print("IS_LUT_FUSED", mode, "D", D, "timeouts", c["p1_full"], "xcc mismatches", c["p1_gs"], "spins", c["lutf_spins"],
          "repaired", core.lut_fused_repaired())
    m = np.array([[c["p1_per_tile"][(8 * r + u) // 3][(8 * r + u) % 3] for u in range(8)] for r in range(8)])
    print(" reader XCC (rows) x unit XCC (columns):\n", m)
    core.close()
