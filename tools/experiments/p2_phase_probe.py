"""Debug (GPU box): s_memtime cycles per section of a phase-2 step of the pairwise DP, from a
libis_core_abl.so built with `make abl ABL=-DIS_ABL_P2PHASES`.  Extra arguments go to bench.py."""
import ctypes, os, sys, runpy
os.environ["IS_CORE_LIB"] = "instance_stixels_amd/lib/libis_core_abl.so"
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--preset", "drn_d_38_pairwise", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
            "--no-single", "--no-d2h", "--no-variants", "--min-seconds", "0"] + sys.argv[1:]
from instance_stixels_amd import core
L = core.lib()
out = (ctypes.c_ulonglong * 8)()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    L.isk_debug_p2phases(out, 1)
    v = list(out)
    tot = sum(v) or 1
    names = ["prologue", "scalar loads", "eval_segment", "LUT values", "pairwise_step", "broadcasts",
             "make_step", "minima+store"]
    print("phase-2 sections (s_memtime ticks summed over waves):")
    for n, x in zip(names, v):
        print("  %-14s %14d  %5.1f%%" % (n, x, 100.0 * x / tot))
