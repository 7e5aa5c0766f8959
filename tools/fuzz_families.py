"""Development fuzzer (GPU box): the random shapes / weights of tests/test_parity_gpu._random_case on frames of EVERY
input family (incl. cityscapes_like with its invalid regions when the case has an invalid-disparity value), HIP core
against the CPU oracle, bit-exact incl. the complete tables.  The IS_* knobs of the environment apply (e.g.
IS_P1_WIN_TILES=99: the windowed kernels at these small shapes).   python tools/fuzz_families.py [first] [count]"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers, test_parity_gpu as t
from instance_stixels_amd import synthetic
from oracle import oracle
fails = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for k in range(first, first + count):
    preset, rows, cols, D, ov = t._random_case(k)
    fam = synthetic.FAMILIES[k % len(synthetic.FAMILIES)]
    try:
        case = helpers.build_case(preset, rows, cols, D, seed=5000 + k, n_images=2, **ov)
        cfg = case["cfg"]
        frames = [synthetic.make_frame(cfg, seed=9000 + 2 * k + i, family=fam) for i in range(2)]
        g = [oracle.host_ground(cfg, f.vhor_image + (k % 7) - 3, f.camera_tilt, f.camera_height, f.alpha_ground) for f in frames]
        case.update(frames=frames, gf=np.stack([x[0] for x in g]), ng=np.stack([x[1] for x in g]),
                    ig=np.stack([x[2] for x in g]), vhor=np.array([x[3] for x in g], np.int32),
                    disparity=np.stack([f.disparity for f in frames]), segmentation=np.stack([f.segmentation for f in frames]))
        got = helpers.run_core(case)
        for img in range(2):
            ref = helpers.run_oracle(case, image=img)
            errs = helpers.compare(ref, got, img, cfg)
            if errs:
                fails += 1
                print("FAIL", k, fam, preset, rows, cols, D, ov, errs[:3], flush=True)
    except Exception as e:
        fails += 1
        print("EXC", k, fam, preset, rows, cols, D, repr(e)[:200], flush=True)
    if (k - first) % 40 == 39:
        print("...", k - first + 1, "cases, fails =", fails, flush=True)
print("fuzz done, fails =", fails)
sys.exit(1 if fails else 0)
