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
# the kernel whose VALU issue is reported (bench.py `valu.issue_frac`): ISA file, mangled-name pattern
VALU_KERNEL = {"unary": ("is_k_unary_fast", "k_dp_unary_fastILb0ELi2ELb0ELb1ELb0ELb1ELb0EE", "k_dp_unary_fast<false, 2, false, true, false, true, false>"),   # <false, 2, false, WIN, !GEN, LUTF, !REPAIR>: the instantiation batch 64 runs (the repair launch behind it, <.., false>, leaves at once and is not averaged in)
               "pairwise": ("is_k_pairwise", "k_pw_phase1ILb0ELi2E", "k_pw_phase1")}
N_SIMD = 256 * 4   # MI355X: 256 CUs x 4 SIMDs
# issue cycles per wave64 VALU instruction on gfx950, measured with tools/ubench:
# fp32 add / sub / mul / fma (also with one SGPR operand), VGPR-only integer add and v_mov: 2; packed
# fp32 (v_pk_*): 4.8; everything else (min / max / min3, conversions, compares, cndmask, shifts, integer
# ops with an SGPR operand, every DPP form, f64): 4
TWO_CYCLE = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32",
             "v_mov_b32")
TWO_CYCLE_INT = ("v_add_u32", "v_sub_u32", "v_subrev_u32")


def isa_cycles_per_valu(build_dir, src, pattern):
    """Static mix of the VALU instructions inside the loops of one kernel (labels the compiler marks
    `in Loop:`), priced with the measured issue costs: (cycles per instruction, instructions)."""
    path = os.path.join(build_dir, f"{src}-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(path):
        return None, 0
    lines = open(path).read().splitlines()
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pattern) + r"\w*:", l)), None)
    if start is None:
        return None, 0
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    in_loop, n, cyc = False, 0, 0.0
    for i in range(start, end):
        l = lines[i]
        if re.match(r"^\.LBB\d+_\d+:", l):
            in_loop = "in Loop:" in l or (i + 1 < end and "in Loop:" in lines[i + 1])
            continue
        x = l.strip()
        if not in_loop or not x.startswith("v_"):
            continue
        op = x.split()[0]
        n += 1
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if op.endswith("_dpp") or "row_newbcast" in x or "row_shr" in x or "quad_perm" in x:
            cyc += 4
        elif base.startswith("v_pk_"):
            cyc += 4.8
        elif base in TWO_CYCLE:
            cyc += 2
        elif base in TWO_CYCLE_INT and not re.search(r"\bs\d+\b|\bs\[", x):
            cyc += 2
        else:
            cyc += 4
    return (cyc / n if n else None), n


def valu_section(d, mode, build_dir):
    """VALU issue of the mode's dominant DP kernel from the PMC passes pmc1 (SQ_ACTIVE_INST_VALU),
    pmc2 (SQ_INSTS_VALU) and pmc5 (GRBM_GUI_ACTIVE), per dispatch average."""
    src, pattern, name = VALU_KERNEL[mode]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for p in ("pmc1", "pmc2", "pmc5"):
        for f in glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if name in r["Kernel_Name"] and "generic" not in r["Kernel_Name"]:
                        tot[r["Counter_Name"]] += float(r["Counter_Value"])
                        cnt[r["Counter_Name"]] += 1
    need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")
    if any(k not in tot for k in need):
        return None
    insts = tot["SQ_INSTS_VALU"] / cnt["SQ_INSTS_VALU"]
    active_q = tot["SQ_ACTIVE_INST_VALU"] / cnt["SQ_ACTIVE_INST_VALU"]
    cycles = tot["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"] / 8.0      # the counter sums the 8 XCDs
    cpi, n_static = isa_cycles_per_valu(build_dir, src, pattern)
    out = {"kernel": name, "dispatches_averaged": cnt["SQ_INSTS_VALU"], "simds": N_SIMD,
           "sq_insts_valu": insts, "sq_active_inst_valu_quadcycles": active_q,
           "kernel_cycles_grbm_gui_active_over_8": cycles,
           "issue_frac_pmc_upper": 4.0 * active_q / (N_SIMD * cycles),
           "isa_cycles_per_valu_inst": cpi, "isa_loop_valu_insts": n_static}
    if cpi:
        out["issue_frac"] = insts * cpi / (N_SIMD * cycles)
    out["formula"] = ("issue_frac = SQ_INSTS_VALU * c / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), c = issue cycles per "
                      "VALU wave-instruction from the static mix of the kernel's loops priced with the measured "
                      "2 / 4 / 4.8-cycle costs; issue_frac_pmc_upper = 4 * SQ_ACTIVE_INST_VALU / (same): the "
                      "counter ticks in whole quad-cycles, so it charges a 2-cycle instruction 4 cycles")
    return out


def rocprof_kernel_ms(d, which, steps):
    """ms per step of the DP kernels from the kernel-trace pass (trace/**/kernel_stats.csv): the sum over the mode's DP
    kernels of calls x average duration, over the steps the command ran -- what bench.py's HIP-event `kernel_ms`
    (roofline) has to agree with."""
    tot, per = 0.0, {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if any(k in r["Name"] for k in which):
                    ns = float(r["TotalDurationNs"])
                    tot += ns
                    per[r["Name"].split("(")[0][:80]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return (tot / steps / 1e6 if tot else None), per


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
        ms, per = rocprof_kernel_ms(d, DP_KERNELS[mode], steps)
        if ms is not None:
            out[mode]["rocprof_kernel_ms"] = ms
            out[mode]["rocprof_kernels"] = per
        v = valu_section(d, mode, os.path.join("instance_stixels_amd", "csrc", "build"))
        if v:
            out[mode]["valu"] = v
    path = os.path.join("profiles", f"{tag}_traffic.json")
    prev = {}
    if os.path.exists(path):
        prev = json.load(open(path))
    prev.update(out)
    json.dump(prev, open(path, "w"), indent=1)
    print(json.dumps(prev, indent=1))


if __name__ == "__main__":
    main()
