#!/usr/bin/env python3
"""Checks the hidden scalar requests (srec_request*, is_kernels.h) in the kernels' ISA: between an
inline-asm s_load_dwordx16 and the inline-asm `s_waitcnt lgkmcnt(0)` that follows it on every path, no
instruction may read or write the destination SGPRs (the compiler does not know that the load is still
in flight).  Fails closed: an indirect jump / call on such a path (s_setpc, s_swappc, s_call) or a
branch to an unknown label is a violation too.

  tools/check_srec.py is_k_pairwise 'k_pw_phase1ILb0ELi2'       one kernel, ISA in /tmp/is_asm (make asm)
  tools/check_srec.py --dir D --all is_k_pairwise is_k_unary_fast   every function of the files' ISA in D
                                                                 that holds a hidden request: what the
                                                                 default build runs on the objects it links
"""
import re
import sys


def is_inst(l):
    return re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))


def sregs(l):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", l):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", l):
        out.add(int(m.group(1)))
    return out


def check_body(name, body):
    """-> (hidden requests, violations) of one function's lines."""
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    bad = n = 0
    for i, l in enumerate(body):
        if "s_load_dwordx16" in l and i > 0 and "ASMSTART" in body[i - 1]:
            m = re.search(r"s_load_dwordx16 s\[(\d+):(\d+)\]", l)
            dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
            k = i + 1   # further loads of the same asm statement
            while "ASMEND" not in body[k]:
                m2 = re.search(r"s_load_dwordx\d+ s\[(\d+):(\d+)\]", body[k])
                if m2:
                    dst |= set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                k += 1
            n += 1
            seen, work = set(), [k]   # walk every path from the request until an asm wait
            while work:
                j = work.pop()
                while j < len(body):
                    if j in seen:
                        break
                    seen.add(j)
                    x = body[j]
                    if "ASMSTART" in x and j + 1 < len(body) and "s_waitcnt lgkmcnt(0)" in body[j + 1]:
                        break
                    if is_inst(x):
                        if "s_endpgm" in x:
                            break
                        if re.match(r"\s+(s_setpc|s_swappc|s_call|s_getpc)", x):
                            print(f"{name}: line {j}: {x.strip()}   indirect control flow while {sorted(dst)[0]}.. is in flight")
                            bad += 1
                            break
                        if sregs(x) & dst:
                            print(f"{name}: line {j}: {x.strip()}   touches {sorted(dst)} requested at line {i}")
                            bad += 1
                        mb = re.match(r"\s+(s_cbranch_\w+|s_branch)\s+(\S+)", x)
                        if mb:
                            if mb.group(2) not in labels:
                                print(f"{name}: line {j}: {x.strip()}   branch to an unknown label")
                                bad += 1
                            else:
                                work.append(labels[mb.group(2)])
                            if mb.group(1) == "s_branch":
                                break
                    j += 1
    return n, bad


def functions(lines):
    """(name, body lines) of every function of an ISA listing."""
    out, i = [], 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            end = next((k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end")), len(lines))
            out.append((m.group(1), lines[i:end]))
            i = end
        i += 1
    return out


def main(argv):
    d = "/tmp/is_asm"
    if argv and argv[0] == "--dir":
        d, argv = argv[1], argv[2:]
    if argv and argv[0] == "--all":
        total = bad = 0
        for f in argv[1:]:
            lines = open(f"{d}/{f}-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
            for name, body in functions(lines):
                n, b = check_body(name, body)
                total += n
                bad += b
        print(f"{total} hidden requests checked, {bad} violations")
        return 1 if bad or total == 0 else 0
    f, pat = argv[0], argv[1]
    lines = open(f"{d}/{f}-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    n, bad = check_body(pat, lines[start:end])
    print(f"{n} hidden requests checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
