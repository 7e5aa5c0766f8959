"""GPU box: phase-1 anatomy of ONE pairwise frame per call, from a variant built with
tools/build_variant.sh p1ph is_k_pairwise -DIS_ABL_P1PHASES  (s_memtime ticks of wave 0 of every phase-1 workgroup)."""
import ctypes, os, sys
os.environ["IS_CORE_LIB"] = os.path.join(os.getcwd(), "instance_stixels_amd/lib/variants/libis_core_p1ph.so")
sys.path.insert(0, os.getcwd())
import torch
import bench
from instance_stixels_amd import core
dev = torch.device("cuda", 0)
wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 1, 1, dev, 0)
c = wl.make_core()
L = core.lib()
out = (ctypes.c_ulonglong * 128)()
for _ in range(5):
    wl.step(c)
torch.cuda.synchronize()
L.isk_debug_p1phases(out, 1)
n = 20
for _ in range(n):
    wl.step(c)
torch.cuda.synchronize()
L.isk_debug_p1phases(out, 1)
names = ["prologue", "pre-pass", "walk", "wait others", "merge"]
print("per tile: shader clocks of wave 0 per workgroup (512 workgroups per launch); full steps / gs rounds of wave 0 per workgroup")
for t in range(1, 16):
    v = list(out[t * 8:(t + 1) * 8])
    parts = [v[0], v[6], v[1], v[2], v[3]]
    wgs = n * 512
    print("  tile %2d: " % t + "  ".join("%s %6.0f" % (nm, x / wgs) for nm, x in zip(names, parts)) +
          "   total %6.0f   steps %.1f  gs %.1f" % (sum(parts) / wgs, v[4] / wgs, v[5] / wgs))
c.close()
