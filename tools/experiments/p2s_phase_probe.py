"""GPU box: cycles per section of a step of the CHAIN wave of k_pw_phase2s (one pairwise frame per call), from a variant
library built with  tools/build_variant.sh p2sph is_k_pairwise -DIS_ABL_P2PHASES."""
import ctypes, os, sys
os.environ["IS_CORE_LIB"] = os.path.join(os.getcwd(), "instance_stixels_amd/lib/variants/libis_core_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "p2sph"))
sys.path.insert(0, os.getcwd())
import torch
import bench
from instance_stixels_amd import core
dev = torch.device("cuda", 0)
wl = bench.Workload("drn_d_38_pairwise", 1024, 2048, 128, 1, 1, dev, 0)
c = wl.make_core()
L = core.lib()
out = (ctypes.c_ulonglong * 8)()
for _ in range(5):
    wl.step(c)
torch.cuda.synchronize()
L.isk_debug_p2phases(out, 1)
n = 20
for _ in range(n):
    wl.step(c)
torch.cuda.synchronize()
L.isk_debug_p2phases(out, 1)
v = list(out)
names = ["prologue", "wait for slot", "slot reads", "unready polls (count)", "pairwise_step", "broadcasts", "make_step", "minima+store"]
steps = n * 256 * 1024
print("k_pw_phase2s chain wave, shader clocks per row (and per (column, tile) for the prologue):")
for k, (nm, x) in enumerate(zip(names, v)):
    if True:
        print("  %-14s %10.1f" % (nm, x / (n * 256 * 16) if k == 0 else x / steps))
print("  sum per row %.1f" % ((sum(v[1:]) - v[3]) / steps))
c.close()
