"""Development driver: oracle vs HIP core on a few seeded cases, verbose diffs (run on a GPU box)."""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import helpers

CASES = [
    ("drn_d_22_unary", 64, 64, 32, {}),
    ("drn_d_22_unary", 128, 256, 32, {}),
    ("drn_d_38_pairwise", 64, 64, 32, {}),
    ("drn_d_38_pairwise", 128, 256, 32, {}),
    ("drn_d_22_unary", 128, 256, 32, dict(invalid_disparity=0.0)),
    ("drn_d_38_pairwise", 128, 256, 32, dict(invalid_disparity=0.0)),
    ("drn_d_22_unary", 136, 128, 48, dict(median_join=True)),
    ("disparity_only_unary", 128, 128, 64, {}),
    ("disparity_only_pairwise", 128, 128, 64, {}),
    ("drn_d_38_unary", 512, 512, 64, {}),
    ("drn_d_22_pairwise", 512, 512, 64, {}),
]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
fails = 0
for preset, rows, cols, D, ov in CASES:
    case = helpers.build_case(preset, rows, cols, D, seed=7, n_images=2, **ov)
    t = time.time(); got = helpers.run_core(case); tg = time.time() - t
    for img in range(2):
        t = time.time(); ref = helpers.run_oracle(case, image=img); to = time.time() - t
        errs = helpers.compare(ref, got, img, case["cfg"])
        ns = [helpers.n_sections(ref["sections"][c]) for c in range(case["cfg"].realcols)]
        print(f"{preset} {rows}x{cols}x{D} {ov} img{img}: {'OK' if not errs else 'FAIL'} "
              f"(sections/col mean {np.mean(ns):.1f}, inst {ref['inst_per_class'].tolist()}, "
              f"oracle {to:.2f}s gpu-call {tg:.2f}s)")
        for e in errs[:12]:
            print("    ", e)
        fails += bool(errs)
print("FAILS", fails)
sys.exit(1 if fails else 0)
