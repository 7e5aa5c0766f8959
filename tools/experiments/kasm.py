#!/usr/bin/env python3
"""ISA of one kernel out of /tmp/is_asm/<file>-hip-amdgcn-amd-amdhsa-gfx950.s (make -C instance_stixels_amd/csrc asm).
usage: tools/kasm.py is_k_unary_fast 'k_dp_unary_fastILb0ELi2' [--hist] [--loops]"""
import re, sys, collections
f, pat = sys.argv[1], sys.argv[2]
lines = open(f"/tmp/is_asm/{f}-hip-amdgcn-amd-amdhsa-gfx950.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if "--hist" in sys.argv:
    c = collections.Counter(l.split()[0] for l in body if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";")))
    tot = sum(c.values())
    print("instructions:", tot)
    for k, v in c.most_common(70):
        print(f"{v:6d} {k}")
elif "--loops" in sys.argv:
    # basic blocks with sizes
    cur, n = None, 0
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if cur: print(cur, n)
            cur, n = m.group(1), 0
        elif re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";")):
            n += 1
    print(cur, n)
else:
    print("\n".join(body))
