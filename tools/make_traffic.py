"""Writes profiles/<round>_traffic.json from the PMC passes tools/profile.sh left under
gpurun_out/prof_<tag>/ (pmc3 = FETCH_SIZE, pmc4 = WRITE_SIZE + TCC hits / misses).

    python tools/make_traffic.py r02 unary=gpurun_out/prof_r02u pairwise=gpurun_out/prof_r02p

Per mode: the counters of the DP kernels (unary: k_dp_unary_fast + k_dp_unary; pairwise: every
k_pw_phase1 / k_pw_phase2 launch of a step) are summed over all dispatches of the profiled command
and divided by the number of steps it ran (warm-up + timed), i.e. "per launch of the DP of one
batch", the unit bench.py's roofline.traffic uses.  FETCH_SIZE / WRITE_SIZE are in KB.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

DP_KERNELS = {"unary": ("k_dp_unary",), "pairwise": ("k_pw_phase1", "k_pw_phase2")}  # (k_pw_phase2 matches k_pw_phase2x and k_pw_phase2_generic too)


def counters(d, which):
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(int)
    for p in ("pmc3", "pmc4"):
        for f in glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"]
                    if any(k in name for k in which):
                        tot[r["Counter_Name"]] += float(r["Counter_Value"])
                        calls[r["Counter_Name"]] += 1
    return tot, calls


def main():
    tag = sys.argv[1]
    out = {"_comment": "HBM-side traffic of the DP kernels per step (one batch), from the rocprofv3 "
                       "--pmc passes of tools/profile.sh (FETCH_SIZE and WRITE_SIZE in separate "
                       "passes, unit KB); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 being the "
                       "gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (cross-check: "
                       "TCC_MISS_sum * 128 B).  Pairwise: summed over the 2 x ntiles launches of a "
                       "step.  bench.py reports it as roofline.traffic when mode, shape and batch "
                       "match.  Written by tools/make_traffic.py."}
    for arg in sys.argv[2:]:
        mode, d = arg.split("=", 1)
        cmd = open(os.path.join(d, "command.txt")).read().strip()
        a = cmd.split()
        get = lambda flag, default: int(a[a.index(flag) + 1]) if flag in a else default
        steps = get("--steps", 5) + get("--warmup", 2)
        tot, calls = counters(d, DP_KERNELS[mode])
        if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
            raise SystemExit(f"{d}: no FETCH_SIZE / WRITE_SIZE rows for {DP_KERNELS[mode]}")
        out[mode] = {
            "kernels": list(DP_KERNELS[mode]),
            "command": cmd,
            "steps_profiled": steps,
            "dispatches_per_step": calls["FETCH_SIZE"] / steps,
            "batch": get("--batch", 64), "rows": get("--rows", 1024), "cols": get("--cols", 2048),
            "max_dis": get("--max-dis", 128),
            "fetch_size_kb": tot["FETCH_SIZE"] / steps,
            "write_size_kb": tot["WRITE_SIZE"] / steps,
            "tcc_miss": tot.get("TCC_MISS_sum", 0.0) / steps,
            "tcc_hit": tot.get("TCC_HIT_sum", 0.0) / steps,
            "bytes_per_step": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / steps * 1024.0,
        }
    path = os.path.join("profiles", f"{tag}_traffic.json")
    prev = {}
    if os.path.exists(path):
        prev = json.load(open(path))
    prev.update(out)
    json.dump(prev, open(path, "w"), indent=1)
    print(json.dumps(prev, indent=1))


if __name__ == "__main__":
    main()
