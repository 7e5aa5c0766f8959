"""Development fuzzer (run on a GPU box): many seeded random shapes / weights / model parameters,
both modes, HIP core against the CPU oracle, bit-exact.  tests/test_parity_gpu.py runs the first 24
of the same generator;  python tools/fuzz_parity.py [first] [count]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers, test_parity_gpu as t
fails = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 24
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for k in range(first, first + count):
    preset, rows, cols, D, ov = t._random_case(k)
    try:
        case = helpers.build_case(preset, rows, cols, D, seed=5000 + k, n_images=2, **ov)
        got = helpers.run_core(case)
        for img in range(2):
            ref = helpers.run_oracle(case, image=img)
            errs = helpers.compare(ref, got, img, case["cfg"])
            if errs:
                fails += 1
                print("FAIL", k, preset, rows, cols, D, ov, errs[:3])
    except Exception as e:
        fails += 1
        print("EXC", k, preset, rows, cols, D, repr(e)[:200])
print("fuzz done, fails =", fails)
sys.exit(1 if fails else 0)
