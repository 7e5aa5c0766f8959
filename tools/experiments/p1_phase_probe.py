"""Debug (GPU box): s_memtime cycles of wave 0 of every phase-1 workgroup of the pairwise DP, from a
libis_core_abl.so built with `make abl ABL=-DIS_ABL_P1PHASES`.  Extra arguments go to bench.py."""
import ctypes, os, sys, runpy
os.environ["IS_CORE_LIB"] = "instance_stixels_amd/lib/libis_core_abl.so"
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--preset", "drn_d_38_pairwise", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
            "--no-single", "--no-d2h", "--no-variants", "--min-seconds", "0"] + sys.argv[1:]
from instance_stixels_amd import core
L = core.lib()
out = (ctypes.c_ulonglong * 128)()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    L.isk_debug_p1phases(out, 1)
    names = ["prologue", "pre-pass", "walk", "wait others", "merge"]
    print("per tile, ticks of wave 0 per workgroup launch (s_memtime), share of the workgroup's life")
    for t in range(16):
        v = list(out[t * 8:(t + 1) * 8])
        parts = [v[0], v[6], v[1], v[2], v[3]]
        tot = sum(parts) or 1
        print("  tile %2d: " % t + "  ".join("%s %4.1f%%" % (n, 100.0 * x / tot) for n, x in zip(names, parts)) +
              "   total %.3g   full steps %d  gs rounds %d" % (tot, v[4], v[5]))
