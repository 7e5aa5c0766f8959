#!/usr/bin/env python3
"""Checks the hidden scalar requests (srec_request*, is_kernels.h) in a kernel's ISA: between an inline-asm
s_load_dwordx16 and the inline-asm `s_waitcnt lgkmcnt(0)` that follows it on every path, no instruction
may read or write the destination SGPRs (the compiler does not know that the load is still in flight).
usage: tools/check_srec.py is_k_pairwise 'k_pw_phase1ILb0ELi2'   (after make -C instance_stixels_amd/csrc asm)"""
import re, sys
f, pat = sys.argv[1], sys.argv[2]
lines = open(f"/tmp/is_asm/{f}-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
def is_inst(l): return re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))
def sregs(l):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", l): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", l): out.add(int(m.group(1)))
    return out
bad = 0
n = 0
for i, l in enumerate(body):
    if "s_load_dwordx16" in l and i > 0 and "ASMSTART" in body[i - 1]:
        m = re.search(r"s_load_dwordx16 s\[(\d+):(\d+)\]", l)
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
        k = i + 1 # further loads of the same asm statement (srec_request_tail)
        while "ASMEND" not in body[k]:
            m2 = re.search(r"s_load_dwordx\d+ s\[(\d+):(\d+)\]", body[k])
            if m2: dst |= set(range(int(m2.group(1)), int(m2.group(2)) + 1))
            k += 1
        n += 1
        # walk every path from i+1 until an asm wait
        seen, work = set(), [k]
        while work:
            j = work.pop()
            while j < len(body):
                if j in seen: break
                seen.add(j)
                x = body[j]
                if "ASMSTART" in x and j + 1 < len(body) and "s_waitcnt lgkmcnt(0)" in body[j + 1]: break
                if is_inst(x):
                    if "s_endpgm" in x: break
                    if sregs(x) & dst:
                        print(f"line {j}: {x.strip()}   touches {sorted(dst)} requested at line {i}")
                        bad += 1
                    mb = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", x)
                    if mb:
                        work.append(labels[mb.group(2)])
                        if mb.group(1) == "s_branch": break
                j += 1
print(f"{n} hidden requests checked, {bad} violations")
sys.exit(1 if bad else 0)
