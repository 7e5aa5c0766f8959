"""Debug (GPU box): s_memtime cycles of wave 0 of every phase-1 workgroup of the pairwise DP, from a
libis_core_abl.so built with `make abl ABL=-DIS_ABL_P1PHASES`.  Extra arguments go to bench.py."""
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
    L.isk_debug_p1phases(out, 1)
    v = list(out)
    tot = (sum(v[:4]) + v[6]) or 1
    for n, x in zip(["prologue", "pre-pass (block bounds)", "walk", "wait for the other waves", "merge"],
                    [v[0], v[6], v[1], v[2], v[3]]):
        print("  %-26s %14d  %5.1f%%" % (n, x, 100.0 * x / tot))
    print("  wave 0: %d full steps, %d ground/sky rounds; ticks per round trip of the walk: %.0f"
          % (v[4], v[5], v[1] / max(1, v[4] + v[5])))
