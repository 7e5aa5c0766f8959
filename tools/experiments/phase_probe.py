"""Debug (GPU box): phase cycle counters of a libis_core_abl.so built with -DIS_ABL_PHASES."""
import ctypes, os, subprocess, sys, json
os.environ.setdefault("IS_CORE_LIB", "instance_stixels_amd/lib/variants/libis_core_phases.so")  # tools/build_variant.sh phases is_k_unary_fast -DIS_ABL_PHASES
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-single", "--no-d2h"] + sys.argv[1:]
import runpy
from instance_stixels_amd import core
L = core.lib()
out = (ctypes.c_ulonglong * 8)()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    L.isk_debug_phases(out, 1)
    v = list(out)
    tot = sum(v[:4]) or 1
    tot = (v[0] + v[4] + v[5] + v[6] + v[7] + v[1] + v[2] + v[3]) or 1
    names = ["ring requests", "record + 1/h table", "tile staging", "barrier", "diagonal quarters", "walk below the tile", "wait for waves", "merge"]
    vals = [v[4], v[5], v[6], v[0], v[7], v[1], v[2], v[3]]
    print("wave 0 of every workgroup, s_memtime ticks summed:")
    for n, x in zip(names, vals):
        print("  %-20s %12.4g  %5.1f%%" % (n, x, 100.0 * x / tot))
