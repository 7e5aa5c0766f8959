"""Development fuzzer (run on a GPU box): random configurations with hostile column contents
(tests/helpers.make_hostile), HIP core against the CPU oracle, bit-exact.
python tools/fuzz_hostile.py [count]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers
import test_parity_gpu as t

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
for k in range(count):
    preset, rows, cols, D, ov = t._random_case(k)
    case = helpers.make_hostile(helpers.build_case(preset, rows, cols, D, seed=6000 + k, n_images=2, **ov),
                                seed=7000 + k)
    got = helpers.run_core(case)
    for img in range(2):
        errs = helpers.compare(helpers.run_oracle(case, image=img), got, img, case["cfg"])
        if errs:
            fails += 1
            print("FAIL", k, preset, rows, cols, D, errs[:3])
print("hostile fuzz done, fails =", fails)
sys.exit(1 if fails else 0)
