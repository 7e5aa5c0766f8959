"""Parity tests proper: the HIP path (through the C ABI / the C++ host class) against the CPU
oracle on the same seeded inputs.  Bar: bit-exact for every integer field (cuts, types, classes,
indices) AND, because the fp32 association of the reference is replicated, bit-exact for every
fp32 field (costs, mean disparity, instance centres, the full DP cost table); the contract's
tolerance (cost within 1e-4 relative) is asserted as well for documentation."""
import glob
import os

import numpy as np
import pytest

import helpers
from test_oracle_properties import load_golden, check_column_structure, GOLDEN

pytestmark = pytest.mark.gpu


def _assert_parity(case, got, images=None, cols=None):
    cfg = case["cfg"]
    for img in (range(len(case["frames"])) if images is None else images):
        ref = helpers.run_oracle(case, image=img,
                                 col_range=None if cols is None else (min(cols), max(cols) + 1))
        errs = helpers.compare(ref, got, img, cfg, cols=cols)
        assert not errs, "\n".join(errs[:10])
        # the contract's stated tolerance (north_star): costs within 1e-4 relative
        for c in (range(cfg.realcols) if cols is None else cols):
            n = helpers.n_sections(ref["sections"][c])
            a, b = ref["sections"][c][:n]["cost"], got["sections"][img][c][:n]["cost"]
            with np.errstate(invalid="ignore"):   # -inf costs (log LUT of 0) compare equal
                assert np.all((a == b) | (np.abs(a - b) <= 1e-4 * np.maximum(np.abs(a), 1e-30)))


SMALL = [
    ("drn_d_22_unary", 64, 64, 32, {}),
    ("drn_d_38_unary", 128, 256, 32, {}),
    ("drn_d_22_pairwise", 64, 64, 32, {}),
    ("drn_d_38_pairwise", 128, 256, 32, {}),
    ("drn_d_22_unary", 128, 256, 32, dict(invalid_disparity=0.0)),
    ("drn_d_38_pairwise", 128, 256, 32, dict(invalid_disparity=0.0)),
    ("drn_d_22_unary", 136, 128, 48, dict(median_join=True)),              # H % 64 != 0, D not 2^k
    ("drn_d_38_pairwise", 136, 128, 48, dict(median_join=True, invalid_disparity=0.0)),
    ("disparity_only_unary", 128, 128, 64, {}),                             # BASELINE configs[0] model
    ("disparity_only_pairwise", 128, 128, 64, {}),
    ("drn_d_22_unary", 16, 64, 16, {}),                                     # smallest legal shape
    ("drn_d_38_pairwise", 16, 64, 16, {}),
    ("drn_d_22_unary", 8, 64, 8, {}),                                       # below the reference's domain
    ("drn_d_22_unary", 64, 72, 32, dict(width_margin=8)),
]


@pytest.mark.parametrize("preset,rows,cols,D,ov", SMALL)
def test_small_cases_bit_exact(preset, rows, cols, D, ov):
    case = helpers.build_case(preset, rows, cols, D, seed=7, n_images=2, **ov)
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset,rows,cols,D,ov", [c for c in SMALL if "pairwise" in c[0]])
def test_small_pairwise_cases_one_wave_phase2(preset, rows, cols, D, ov, monkeypatch):
    """The library walks phase 2 of small calls (<= 2048 columns) with k_pw_phase2s; the one-wave
    kernel k_pw_phase2 of large batches must give the same bits on the same cases."""
    monkeypatch.setenv("IS_P2_SPLIT", "0")
    case = helpers.build_case(preset, rows, cols, D, seed=7, n_images=2, **ov)
    got = helpers.run_core(case)
    _assert_parity(case, got)


def _random_case(k):
    """Seeded random shape / weights / model parameters around the presets."""
    rng = np.random.default_rng(9000 + k)
    preset = ["drn_d_22_unary", "drn_d_38_unary", "drn_d_22_pairwise", "drn_d_38_pairwise"][k % 4]
    rows = int(rng.integers(2, 40)) * 8
    cols = int(rng.integers(1, 9)) * 8
    D = int(rng.choice([16, 24, 32, 48, 64, 96, 128]))
    ov = dict(disparity_weight=float(10.0 ** rng.uniform(-4, 0)),
              segmentation_weight=float(10.0 ** rng.uniform(-1, 1.2)),
              instance_weight=float(10.0 ** rng.uniform(-4, -1.5)),
              pord=float(rng.uniform(0.05, 0.4)), pgrav=float(rng.uniform(0.02, 0.2)),
              pblg=float(rng.uniform(0.01, 0.1)), epsilon=float(rng.uniform(1.0, 5.0)),
              sigma_disparity_object=float(rng.uniform(0.5, 2.0)),
              sigma_disparity_ground=float(rng.uniform(1.0, 3.0)))
    if k % 3 == 0:
        ov["invalid_disparity"] = 0.0
    if k % 5 == 0:
        ov["median_join"] = True
    return preset, rows, cols, D, ov


@pytest.mark.parametrize("k", range(24))
def test_random_configs_bit_exact(k):
    preset, rows, cols, D, ov = _random_case(k)
    case = helpers.build_case(preset, rows, cols, D, seed=300 + k, n_images=2, **ov)
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("k", [k for k in range(24) if k % 4 >= 2])
def test_random_pairwise_configs_two_column_phase2(k, monkeypatch):
    """The random pairwise configurations through the kernels of LARGE batches: IS_P2_SPLIT=0 makes
    a small call walk phase 2 with k_pw_phase2x (two columns per wave; k % 4 >= 2 = the pairwise
    presets of _random_case), incl. partial last tiles, invalid disparities and median joins."""
    monkeypatch.setenv("IS_P2_SPLIT", "0")
    preset, rows, cols, D, ov = _random_case(k)
    assert "pairwise" in preset
    case = helpers.build_case(preset, rows, cols, D, seed=300 + k, n_images=2, **ov)
    got = helpers.run_core(case)
    _assert_parity(case, got)
    monkeypatch.setenv("IS_P2X", "0")           # and the one-column kernel of large batches
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("k", [2, 3, 6, 7])
def test_hostile_pairwise_inputs_two_column_phase2(k, monkeypatch):
    """Generic-encoding columns next to FAST ones under the large-batch kernels: pairs with a
    generic column are left to k_pw_phase2_generic, the others to k_pw_phase2x."""
    monkeypatch.setenv("IS_P2_SPLIT", "0")
    preset, rows, cols, D, ov = _random_case(k)
    case = helpers.make_hostile(helpers.build_case(preset, rows, cols, D, seed=6000 + k, n_images=2, **ov),
                                seed=7000 + k)
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("k", range(8))
def test_hostile_inputs_bit_exact(k):
    """Random configurations whose columns are overwritten with out-of-encoding inputs (generic
    int32/int64 path, wrapping sums, subnormal disparities): still bit-exact, no faults."""
    preset, rows, cols, D, ov = _random_case(k)
    case = helpers.make_hostile(helpers.build_case(preset, rows, cols, D, seed=6000 + k, n_images=2, **ov),
                                seed=7000 + k)
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_38_unary", "drn_d_22_pairwise"])
def test_config1_shape_512x1024x64(preset):
    case = helpers.build_case(preset, 512, 1024, 64, seed=11)
    got = helpers.run_core(case)
    _assert_parity(case, got)


def test_config1_disparity_only_512x1024x64():
    for preset in ("disparity_only_unary", "disparity_only_pairwise"):
        case = helpers.build_case(preset, 512, 1024, 64, seed=12)
        got = helpers.run_core(case)
        _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_config2_full_frame_1024x2048x128(preset):
    """BASELINE configs[1]: the whole 256-column frame against the oracle (a few CPU-seconds)."""
    case = helpers.build_case(preset, 1024, 2048, 128, seed=13)
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_config3_batch_of_full_frames(preset):
    """BASELINE configs[2]/[3] shape at a reduced batch: eight distinct 1024x2048x128 frames in ONE
    call (2048 stixel columns: the batched launch geometry of bench.py, and for the pairwise model
    the two-stream split), every frame checked against the oracle."""
    case = helpers.build_case(preset, 1024, 2048, 128, seed=77, n_images=8)
    got = helpers.run_core(case, want_tables=False)
    cfg = case["cfg"]
    for img in range(8):
        ref = helpers.run_oracle(case, image=img)
        errs = helpers.compare(ref, got, img, cfg, check_tables=False)
        assert not errs, "image %d:\n" % img + "\n".join(errs[:10])


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_pruning_changes_nothing_at_full_size(preset, monkeypatch):
    """Size-independent property at BASELINE's sizes (no oracle needed): the exact branch-and-bound
    only skips candidates that cannot win, so with IS_NO_PRUNE=1 (every (vB, vT) pair evaluated,
    the reference's O(H^2) walk) the complete DP tables and all Sections of 12 full 1024x2048x128
    frames are bit-identical to the pruned run.  12 frames = 3072 columns: the one-wave phase 2 and
    the unsplit phase 1 of the pairwise mode, the geometry of large batches."""
    case4 = helpers.build_case(preset, 1024, 2048, 128, seed=91, n_images=4)
    pick = [i % 4 for i in range(12)]
    case = dict(case4)
    case["frames"] = [case4["frames"][k] for k in pick]
    for k in ("gf", "ng", "ig", "vhor", "disparity", "segmentation"):
        case[k] = case4[k][pick]
    pruned = helpers.run_core(case, want_tables=True)
    monkeypatch.setenv("IS_NO_PRUNE", "1")
    full = helpers.run_core(case, want_tables=True)
    assert np.array_equal(pruned["cost_table"].view(np.uint32), full["cost_table"].view(np.uint32))
    assert np.array_equal(pruned["index_table"], full["index_table"])
    for img in range(12):
        assert helpers.sections_equal(pruned["sections"][img], full["sections"][img])
    # and the frames that repeat inside the batch agree with each other
    for img in range(4, 12):
        assert helpers.sections_equal(pruned["sections"][img], pruned["sections"][img % 4])
    # the Sections of every column tile it exactly, top of the image first (R10), and carry the
    # clipped table cost of their last row (StixelsKernels.cu:868-944)
    H = int(case["cfg"].rows)
    for img in range(4):
        secs, ct = pruned["sections"][img], pruned["cost_table"][img]
        for c in range(secs.shape[0]):
            n = helpers.n_sections(secs[c])
            col = secs[c][:n]
            assert n >= 1 and col["vT"][0] == H - 1 and col["vB"][-1] == 0
            assert np.array_equal(col["vT"][1:], col["vB"][:-1] - 1)
            assert np.all(col["vB"] <= col["vT"])
            best = np.minimum(ct[c][col["vT"]].min(axis=1), np.float32(1e4))
            assert np.all(col["cost"] <= np.float32(1e4)) and np.all(col["cost"] >= best)


@pytest.mark.parametrize("preset", ["drn_d_38_pairwise", "drn_d_22_unary"])
@pytest.mark.parametrize("family", ["homogeneous", "many_thin_objects", "iid_noise", "low_confidence",
                                    "flat_disparity", "noisy_disparity", "cityscapes_like"])
def test_pruning_changes_nothing_on_the_input_families(family, preset, monkeypatch):
    """The same property on the other input families of bench.py (synthetic.make_frame(family=...)), both
    models, 9 full frames = 2304 columns (the two-column phase 2 and the unsplit phase 1 of large batches;
    the windowed unary ring kernel): the separable block bounds (lemmas L7 / L8) bite hardest where the
    scene is homogeneous or the CNN hesitant, i.e. exactly where the headline family exercises them least;
    the unary model has its floor on the homogeneous family.  Eight columns of two distinct frames go
    against the oracle with their complete tables (StixelsKernels.cu:600-839)."""
    from instance_stixels_amd import synthetic
    base = helpers.build_case(preset, 1024, 2048, 128, seed=7, n_images=1)
    cfg = base["cfg"]
    frames = [synthetic.make_frame(cfg, seed=300 + i, family=family) for i in range(3)]
    ground = [oracle_mod().host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
              for f in frames]
    pick = [i % 3 for i in range(9)]
    case = dict(base)
    case["frames"] = [frames[k] for k in pick]
    case["gf"] = np.stack([ground[k][0] for k in pick]); case["ng"] = np.stack([ground[k][1] for k in pick])
    case["ig"] = np.stack([ground[k][2] for k in pick])
    case["vhor"] = np.array([ground[k][3] for k in pick], np.int32)
    case["disparity"] = np.stack([frames[k].disparity for k in pick])
    case["segmentation"] = np.stack([frames[k].segmentation for k in pick])
    pruned = helpers.run_core(case, want_tables=True)
    monkeypatch.setenv("IS_NO_PRUNE", "1")
    full = helpers.run_core(case, want_tables=True)
    assert np.array_equal(pruned["cost_table"].view(np.uint32), full["cost_table"].view(np.uint32))
    assert np.array_equal(pruned["index_table"], full["index_table"])
    for img in range(9):
        assert helpers.sections_equal(pruned["sections"][img], full["sections"][img])
    # eight columns spread over two distinct frames against the oracle: Sections and complete tables
    for img, cols in ((0, (3, 100, 171, 255)), (4, (0, 64, 130, 222))):
        for c in cols:
            ref = helpers.run_oracle(case, image=img, col_range=(c, c + 1), joined=pruned["joined"][img])
            errs = helpers.compare(ref, pruned, img, cfg, cols=[c])
            assert not errs, f"{family} image {img} column {c}:\n" + "\n".join(errs[:5])


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_batch64_launch_geometry(preset):
    """BASELINE configs[2] / the per-GPU share of configs[3]: 64 full 1024x2048x128 frames in ONE
    call -- 16 384 columns, the launch geometry bench.py times (12 GB of scratch, the two-stream
    split of the pairwise DP, 131 072 unary workgroups).  Four distinct frames with per-image road
    parameters; the first, last, the images either side of the middle of the batch and four more
    are compared with the oracle."""
    from oracle import oracle
    case4 = helpers.build_case(preset, 1024, 2048, 128, seed=41, n_images=4)
    cfg = case4["cfg"]
    n = 64
    pick = [i % 4 for i in range(n)]
    case = dict(case4)
    case["frames"] = [case4["frames"][k] for k in pick]
    case["disparity"] = case4["disparity"][pick]
    case["segmentation"] = case4["segmentation"][pick]
    gf, ng, ig, vh = [], [], [], []
    for i, k in enumerate(pick):   # a different horizon for (almost) every image of the batch
        f = case4["frames"][k]
        g = oracle.host_ground(cfg, f.vhor_image + (i % 13) - 6, f.camera_tilt,
                               f.camera_height + 0.01 * (i % 5), f.alpha_ground)
        gf.append(g[0]); ng.append(g[1]); ig.append(g[2]); vh.append(g[3])
    case["gf"], case["ng"], case["ig"] = np.stack(gf), np.stack(ng), np.stack(ig)
    case["vhor"] = np.array(vh, np.int32)
    got = helpers.run_core(case, want_tables=False)
    _assert_parity(case, got, images=[0, 1, 31, 32, 33, 47, 62, 63])


def test_config5_pairwise_256_bins_column_subset():
    """BASELINE configs[4] in PAIRWISE mode: D = 256 takes the per-lane gather of the vB-side LUT
    value (LutRow<0>) in phase 1; every 16th column of a 1024x4096x256 frame against the oracle."""
    case = helpers.build_case("drn_d_38_pairwise", 1024, 4096, 256, seed=16)
    got = helpers.run_core(case, want_tables=True)
    _assert_parity(case, got, cols=list(range(0, case["cfg"].realcols, 16)))


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_tall_frame_2048_rows(preset):
    """Beyond every structural limit of the reference (one thread per row, rows < 1024): the same
    formulas at 2048 rows, 32 tiles per column."""
    case = helpers.build_case(preset, 2048, 128, 64, seed=15)
    got = helpers.run_core(case)
    _assert_parity(case, got)


def test_config5_all_512_columns_unary():
    """BASELINE configs[4], every one of the 512 stixel columns of a 1024x4096x256 frame in the
    unary mode against the oracle: Sections, instance candidates and the complete DP tables."""
    case = helpers.build_case("drn_d_22_unary", 1024, 4096, 256, seed=18)
    got = helpers.run_core(case, want_tables=True)
    _assert_parity(case, got)


def test_config5_ultrawide_1024x4096x256_column_subset():
    """BASELINE configs[4] (LDS-pressure shape): every 16th column against the oracle, structure
    checks on all 512 columns."""
    case = helpers.build_case("drn_d_22_unary", 1024, 4096, 256, seed=14)
    got = helpers.run_core(case, want_tables=True)
    cfg = case["cfg"]
    joined = got["joined"][0]
    from oracle import oracle
    assert np.array_equal(joined.view(np.uint32),
                          oracle.join_columns(cfg, case["disparity"][0]).view(np.uint32))
    cols = list(range(0, cfg.realcols, 16))
    for c in cols:
        ref = helpers.run_oracle(case, col_range=(c, c + 1), joined=joined)
        errs = helpers.compare(ref, got, 0, cfg, cols=[c])
        assert not errs, "\n".join(errs[:5])
    for c in range(cfg.realcols):
        check_column_structure(got["sections"][0][c], 1024)


def _run_with_counters(case, want_tables=True):
    """One is_compute call of the whole case with the evaluation counters on: (outputs, counters); the counters
    also say whether a unary call ran its repair launches (`lutf_repaired`, see is_debug_lut_fused_state)."""
    from instance_stixels_amd.core import Core
    cfg = case["cfg"]
    core = Core(case["params"], case["lut"], case["odr"], max_batch=len(case["frames"]))
    try:
        core.set_eval_counters(True)
        out = core.run(disparity_big=case["disparity"], segmentation=case["segmentation"],
                       ground_function=case["gf"], normalization_ground=case["ng"],
                       inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=bool(cfg.pairwise),
                       median_join=bool(cfg.median_join), want_tables=want_tables, want_instances=False)
        counters = core.eval_counters()
        counters["lutf_repaired"] = core.lut_fused_repaired()
        return out, counters
    finally:
        core.close()


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_config5_batch_of_8_windowed_kernels_as_timed(preset, monkeypatch):
    """BASELINE configs[4] at the instantiation bench.py TIMES it on: eight 1024x4096x256 frames per call =
    4096 stixel columns, which is where the library switches the D = 256 kernels to their fn windows BY
    DEFAULT (k_dp_unary_fast at any size since round 5, k_pw_phase1 from 4096 columns; a single frame = 512 columns ran
    the classic tiles, which is what the one-frame tests above exercise).  Two distinct frames, repeated;
    of one copy of each: every 16th column (Sections + complete tables) and the complete tables of eight
    more columns against the oracle (StixelsKernels.cu:600-839); the counters prove that the window path
    ran and that lanes did read outside their windows."""
    for k in ("IS_P1_WIN_TILES", "IS_NO_PRUNE"):
        monkeypatch.delenv(k, raising=False)
    case2 = helpers.build_case(preset, 1024, 4096, 256, seed=23, n_images=2)
    case = helpers.sub_case(case2, [i % 2 for i in range(8)])
    cfg = case["cfg"]
    got, counters = _run_with_counters(case)
    miss = counters["p1_window_miss" if cfg.pairwise else "unary_window_miss"]
    assert miss > 0, counters
    cols = sorted(set(range(0, cfg.realcols, 16)) | {1, 77, 130, 203, 258, 333, 410, 511})
    for img in (0, 5):
        joined = got["joined"][img]
        for c in cols:
            ref = helpers.run_oracle(case, image=img, col_range=(c, c + 1), joined=joined)
            errs = helpers.compare(ref, got, img, cfg, cols=[c])
            assert not errs, f"image {img} column {c}:\n" + "\n".join(errs[:5])
    for img in range(2, 8):     # the repeats inside the batch agree with their first copies
        assert helpers.sections_equal(got["sections"][img], got["sections"][img % 2])
        assert np.array_equal(got["cost_table"][img].view(np.uint32), got["cost_table"][img % 2].view(np.uint32))


# The only shape and mode the reference itself launches and publishes context numbers for: the 784x1792
# crop with 128 disparities and invalid_disparity = 0 (/root/reference/tests/run_test.sh:84,
# apps/run_cityscapes.cu:129-134 and :188, apps/stixels_node.cu:162-176): 224 stixel columns, P2 = 1024,
# P2S = 128, 12.25 tiles of 64 rows (H % 64 = 16: a partial last tile at full scale).
REF_SHAPE = (784, 1792, 128)


@pytest.mark.parametrize("median", [False, True])
@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_reference_operating_point_784x1792_invalid0(preset, median):
    """The reference's own operating point, full frame, all 224 columns against the oracle: Sections,
    instance candidates and the complete DP tables, both models, mean and median column joins."""
    H, W, D = REF_SHAPE
    case = helpers.build_case(preset, H, W, D, seed=31, invalid_disparity=0.0, median_join=median)
    assert case["cfg"].realcols == 224 and case["params"].rows_power2 == 1024
    assert case["params"].rows_power2_segmentation == 128
    assert (case["disparity"] == 0.0).mean() > 0.03       # the 5 % holes are there
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_reference_operating_point_invalid_regions(preset):
    """The reference's shape and mode on the cityscapes_like family: the sky and an occlusion band left of every
    object are invalid as REGIONS (21 % of the joined column entries: segments without a single valid row, valid
    counts far below the height -- the divisor of mean_valid_fast), next to 3 % pixel holes.  Two full frames in one
    call, all 224 columns against the oracle with the complete tables."""
    from instance_stixels_amd import synthetic
    H, W, D = REF_SHAPE
    case = helpers.build_case(preset, H, W, D, seed=53, n_images=2, invalid_disparity=0.0)
    cfg = case["cfg"]
    frames = [synthetic.make_frame(cfg, seed=900 + i, family="cityscapes_like") for i in range(2)]
    ground = [oracle_mod().host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
              for f in frames]
    case.update(frames=frames, gf=np.stack([g[0] for g in ground]), ng=np.stack([g[1] for g in ground]),
                ig=np.stack([g[2] for g in ground]), vhor=np.array([g[3] for g in ground], np.int32),
                disparity=np.stack([f.disparity for f in frames]),
                segmentation=np.stack([f.segmentation for f in frames]))
    got = helpers.run_core(case)
    assert (got["joined"] == 0.0).mean() > 0.1
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_reference_operating_point_batch_as_timed(preset, monkeypatch):
    """The same shape and mode at the batch bench.py times it on (variants.ref_shape_784x1792): 40 frames =
    8960 columns in one call -- the windowed HAS_INVALID instantiations of the large-batch kernels, the
    two-column phase 2 -- four distinct frames, four of the forty against the oracle, all against their
    first copies."""
    for k in ("IS_P1_WIN_TILES", "IS_NO_PRUNE"):
        monkeypatch.delenv(k, raising=False)
    H, W, D = REF_SHAPE
    case4 = helpers.build_case(preset, H, W, D, seed=37, n_images=4, invalid_disparity=0.0)
    case = helpers.sub_case(case4, [i % 4 for i in range(40)])
    got, counters = _run_with_counters(case, want_tables=False)
    assert counters["p1_window_miss" if case["cfg"].pairwise else "unary_window_miss"] > 0, counters
    for img in (0, 17, 22, 39):
        ref = helpers.run_oracle(case, image=img)
        errs = helpers.compare(ref, got, img, case["cfg"], check_tables=False)
        assert not errs, f"image {img}:\n" + "\n".join(errs[:10])
    for img in range(4, 40):
        assert helpers.sections_equal(got["sections"][img], got["sections"][img % 4])


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_reference_operating_point_through_the_host_class(preset):
    """... and one frame per call through Stixels::Compute, the reference's caller sequence
    (apps/run_cityscapes.cu:328-449) at the reference's shape and mode."""
    from instance_stixels_amd import host
    H, W, D = REF_SHAPE
    case = helpers.build_case(preset, H, W, D, seed=43, n_images=2, invalid_disparity=0.0)
    cfg = case["cfg"]
    st = host.Stixels()
    st.SetConfig(cfg)
    st.Initialize()
    try:
        for i, frame in enumerate(case["frames"]):
            st.SetDisparityImage(frame.disparity)
            st.SetSegmentation(frame.segmentation)
            st.SetRoadParameters(frame.vhor_image, frame.camera_tilt, frame.camera_height, frame.alpha_ground)
            data = st.Compute(cfg.pairwise)
            mapping = st.GetInstanceStixels()
            ref = helpers.run_oracle(case, image=i)
            got = dict(joined=ref["joined"][None], sections=data.sections[None])
            errs = helpers.compare(ref, got, 0, cfg, check_tables=False)
            assert not errs, f"frame {i}:\n" + "\n".join(errs[:10])
            assert len(mapping) == int(ref["inst_per_class"].sum())
    finally:
        st.Finish()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_vectors(path):
    from instance_stixels_amd.core import Core
    g = load_golden(path)
    core = Core(g["params"], g["lut"], g["odr"], max_batch=1)
    out = core.run(disparity_big=g["disparity"][None], segmentation=g["segmentation"][None],
                   ground_function=g["gf"], normalization_ground=g["ng"],
                   inv_sigma2_ground=g["ig"], vhor=[int(g["vhor"])], pairwise=bool(g["pairwise"]),
                   median_join=bool(g["median_join"]), want_tables=True)
    core.close()
    assert np.array_equal(out["joined"][0].view(np.uint32), g["joined"].view(np.uint32))
    C = g["params"].cols
    for c in range(C):
        n = helpers.n_sections(g["sections"][c])
        assert helpers.n_sections(out["sections"][0][c]) == n
        assert np.array_equal(out["sections"][0][c][:n].view(np.uint8), g["sections"][c][:n].view(np.uint8))
    assert np.array_equal(out["cost_table"][0].view(np.uint32), g["cost_table"].view(np.uint32))
    assert np.array_equal(out["inst_per_class"][0], g["inst_per_class"])


@pytest.mark.parametrize("vhor_image", [-5, 0, 1, 63, 126, 127, 140])
@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_horizon_edge_cases(preset, vhor_image):
    """Horizon at / beyond the image borders: ground-only, sky-only and mixed columns."""
    case = helpers.build_case(preset, 128, 64, 32, seed=31)
    from oracle import oracle
    f = case["frames"][0]
    g = oracle.host_ground(case["cfg"], vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    case["gf"][0], case["ng"][0], case["ig"][0], case["vhor"][0] = g
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_degenerate_inputs(preset):
    """All-invalid columns, constant disparity, zero segmentation, extreme offsets."""
    case = helpers.build_case(preset, 128, 128, 32, seed=33, invalid_disparity=0.0)
    d = case["disparity"][0]
    d[:, 0:8] = 0.0                       # column 0 entirely invalid
    d[:, 8:16] = 5.0                      # column 1 constant
    d[:, 16:24] = 30.98                   # column 2 at the top of the input domain (D - 1.01)
    d[64:, 24:32] = 0.0                   # column 3 invalid lower half
    s = case["segmentation"][0]
    s[4] = 0                              # column 4: zero segmentation
    s[5, 19:21, :16] = 8 * 4000           # column 5: large offsets, still a FAST column
    s[6, 19:21, :16] = -8 * 4000
    s[7, :19, :16] = 0                    # column 7: all classes tie
    s[8, 19:21, :16] = 8 * 50000          # columns 8-10: |centre| >= 2^18 -> generic int64 path
    s[9, 19:21, :16] = -8 * 50000
    s[10, 19, 3] = 2 ** 30
    s[13, 3, :16] = 200000                # column 13: class total >= 2^24 -> generic int32 sums
    s[14, 12, 5] = -7                     # column 14: negative class value -> generic path
    s[15, :19, :16] = 130000              # column 15: every class just below the fp32-exact limit
    d[:, 88:96] = 1e-30                   # column 11: tiny disparities -> generic division path
    d[40:50, 96:104] = 1e-30              # column 12: mixed
    d[10, 96:104] = 3.0e-39               # subnormal
    got = helpers.run_core(case)
    _assert_parity(case, got)


def test_pairwise_two_stream_split_matches_oracle(monkeypatch):
    """IS_PW_GROUPS=2: the pairwise DP runs two column groups on two HIP streams
    (isk_launch_dp_pairwise; not the default: measured slower beside the RCCL gather pipeline); a
    user stream is honoured around the fork / join."""
    monkeypatch.setenv("IS_PW_GROUPS", "2")
    case = helpers.build_case("drn_d_38_pairwise", 64, 2048, 32, seed=53, n_images=9)
    assert case["cfg"].realcols * 9 >= 2048
    got = helpers.run_core(case)
    _assert_parity(case, got, images=[0, 4, 8])   # first half, the image cut by the split, last
    # every image agrees with the same image computed alone (below the split threshold)
    for img in (3, 5):
        alone = helpers.run_core(helpers.sub_case(case, [img]))
        assert helpers.sections_equal(alone["sections"][0], got["sections"][img])


@pytest.mark.parametrize("split,inv", [(1, -1.0), (0, -1.0), (1, 0.0), (0, 0.0)])
def test_pairwise_phase2_window_and_fallback(split, inv, monkeypatch):
    """Phase 2 of the pairwise DP stages a <= 16-column window of the tile's lutT rows in LDS and
    reads global memory for lanes outside it.  Columns whose rows alternate between two far-apart
    disparities put every segment mean of a tile far outside any 16-column window (fallback on
    every step); smooth columns stay inside it.  IS_P2_SPLIT selects the kernel (the library picks
    by column count): 1 = k_pw_phase2s (chain + evaluator wave), 0 = k_pw_phase2 (one wave)."""
    monkeypatch.setenv("IS_P2_SPLIT", str(split))
    n_images = 2
    ov = dict(invalid_disparity=inv) if inv >= 0 else {}
    case = helpers.build_case("drn_d_38_pairwise", 256, 1536, 64, seed=77, n_images=n_images, **ov)
    d = case["disparity"]
    rows, width = d.shape[1], d.shape[2]
    pattern = np.where((np.arange(rows) // 3) % 2 == 0, 3.25, 50.5).astype(np.float32)
    d[:, :, : width // 2] = pattern[None, :, None]
    if inv >= 0:  # holes in the alternating columns too: the valid-count prefix path
        d[:, ::5, : width // 4] = inv
    got = helpers.run_core(case)
    _assert_parity(case, got)


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_one_context_many_call_sizes_back_to_back(preset):
    """One context, calls of very different sizes queued back to back on one stream WITHOUT
    host synchronisation in between: small calls run their two prepare kernels on two streams and
    split pairwise phase 1 over several workgroups, large pairwise calls use two streams for the
    two half batches; all share the context's scratch buffers and its second stream."""
    import torch
    from instance_stixels_amd.core import Core
    case = helpers.build_case(preset, 64, 2048, 32, seed=91, n_images=9)   # 256 columns / image
    p, cfg = case["params"], case["cfg"]
    dev = torch.device("cuda", 0)
    core = Core(p, case["lut"], case["odr"], max_batch=9)
    big = torch.from_numpy(case["disparity"]).to(dev)
    seg = torch.from_numpy(case["segmentation"]).to(dev)
    calls = [(0, 1), (0, 9), (3, 2), (1, 8), (8, 1), (0, 9), (4, 3)]          # (first image, count)
    outs = []
    for i0, n in calls:
        joined = torch.empty((n, p.cols, p.rows), dtype=torch.float32, device=dev)
        sec = torch.empty((n, p.cols, p.max_sections, 8), dtype=torch.int32, device=dev)
        core.join_columns_ptr(big[i0:i0 + n].data_ptr(), big.shape[2], False, joined.data_ptr(), n)
        core.compute_ptr(joined.data_ptr(), seg[i0:i0 + n].data_ptr(), case["gf"][i0:i0 + n],
                         case["ng"][i0:i0 + n], case["ig"][i0:i0 + n], case["vhor"][i0:i0 + n],
                         bool(cfg.pairwise), n, sec.data_ptr())
        outs.append((joined, sec))                                            # keep buffers alive
    torch.cuda.synchronize()
    from instance_stixels_amd.config import SECTION_DTYPE
    ref = [helpers.run_oracle(case, image=i)["sections"] for i in range(9)]
    for (i0, n), (_, sec) in zip(calls, outs):
        got = sec.cpu().numpy().view(SECTION_DTYPE).reshape(n, p.cols, p.max_sections)
        for k in range(n):
            assert helpers.sections_equal(ref[i0 + k], got[k]), (i0, n, k)
    core.close()


def test_batch_consistency_and_input_immutability():
    """Images of a batch are independent: a frame gives the same result at any batch position,
    and (unlike the reference) the segmentation input is left intact."""
    import torch
    from instance_stixels_amd.core import Core
    case = helpers.build_case("drn_d_38_pairwise", 128, 256, 32, seed=41, n_images=3)
    order = [2, 0, 1, 0, 2]
    core = Core(case["params"], case["lut"], case["odr"], max_batch=len(order))
    dev = torch.device("cuda", 0)
    seg = torch.from_numpy(case["segmentation"][order]).to(dev)
    seg_before = seg.clone()
    big = torch.from_numpy(case["disparity"][order]).to(dev)
    p = case["params"]
    joined = torch.empty((len(order), p.cols, p.rows), dtype=torch.float32, device=dev)
    sec = torch.empty((len(order), p.cols, p.max_sections, 8), dtype=torch.int32, device=dev)
    core.join_columns_ptr(big.data_ptr(), big.shape[2], False, joined.data_ptr(), len(order))
    core.compute_ptr(joined.data_ptr(), seg.data_ptr(), case["gf"][order], case["ng"][order],
                     case["ig"][order], case["vhor"][order], True, len(order), sec.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(seg, seg_before)
    out = sec.cpu().numpy()
    single = helpers.run_core(case)["sections"].view(np.int32).reshape(3, p.cols, p.max_sections, 8)
    for k, img in enumerate(order):
        for c in range(p.cols):
            n = helpers.n_sections(single[img][c].view(helpers.np.dtype(
                [("type", np.int32), ("r", np.int32, 7)])).reshape(-1))
            assert np.array_equal(out[k][c][:n + 1, 0], single[img][c][:n + 1, 0])
            assert np.array_equal(out[k][c][:n], single[img][c][:n])
    core.close()


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_host_class_call_sequence(preset, tmp_path):
    """The reference's caller sequence (apps/run_cityscapes.cu:328-449) through the C++ class."""
    from instance_stixels_amd import host
    case = helpers.build_case(preset, 128, 256, 32, seed=51)
    cfg, f = case["cfg"], case["frames"][0]
    st = host.Stixels()
    st.SetConfig(cfg)
    assert not st.IsInitialized()
    st.Initialize()
    assert st.IsInitialized() and st.GetRealCols() == 32 and st.GetMaxSections() == 200
    for _ in range(2):                     # two frames on one initialised object
        st.SetDisparityImage(f.disparity)
        st.SetSegmentation(f.segmentation)
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        data = st.Compute(cfg.pairwise)
    mapping = st.GetInstanceStixels()
    ref = helpers.run_oracle(case)
    got = dict(joined=ref["joined"][None], sections=data.sections[None])
    assert not helpers.compare(ref, got, 0, cfg, check_tables=False)
    assert (data.rows, data.cols, data.realcols, data.max_sections) == (128, 256, 32, 200)
    assert data.vhor == 128 - f.vhor_image - 1 and data.max_dis == 32
    # every instance-class stixel has a mapping entry; labels are -1 or a cluster id
    want_keys = {(int(c), int(i)) for cls in range(8)
                 for c, i in ref["inst_indices"][cls][:ref["inst_per_class"][cls]]}
    assert set(mapping) == want_keys
    # output formats (SaveStixels text, Stixels.cu:889-926)
    path = str(tmp_path / "frame.stixels")
    st.SaveStixels(data, mapping, f.alpha_ground, data.vhor, path)
    lines = open(path).read().splitlines()
    assert len(lines) == 33 and lines[-1].startswith("groundplane")
    first = lines[0].split(";")[0].split(",")
    s0 = data.sections[0][0]
    assert [int(first[0]), int(first[1]), int(first[2]), int(first[4])] == \
        [s0["type"], s0["vB"], s0["vT"], s0["semantic_class"]]
    verts = st.Get3DVertices(data)
    n_stixels = sum(helpers.n_sections(data.sections[c]) for c in range(32))
    assert len(verts) == 12 * n_stixels
    st.Finish()
    assert not st.IsInitialized()
    st.close()


@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_host_class_columns_with_many_sections(preset):
    """Stixels::Compute fetches the first 64 sections of every column with one pitched copy and the
    complete array only when a column has no terminator among them: four columns of this frame
    alternate between two object classes and two disparities every 8 rows (96 sections each)."""
    from instance_stixels_amd import host
    case = helpers.build_case(preset, 768, 64, 32, seed=5)
    cfg, f = case["cfg"], case["frames"][0]
    seg, d = case["segmentation"], case["disparity"]
    for c in range(4):
        for k in range(768 // 8):
            seg[0, c, :19, k] = 400 * 8
            seg[0, c, 11 if k % 2 == 0 else 13, k] = 0
            seg[0, c, 19:, k] = 0
            r0 = 768 - 8 * (k + 1)
            d[0, r0:r0 + 8, c * 8:(c + 1) * 8] = 20.0 if k % 2 == 0 else 10.0
    ref = helpers.run_oracle(case)
    counts = [helpers.n_sections(ref["sections"][c]) for c in range(8)]
    assert max(counts) > 64 and min(counts) < 64, counts
    st = host.Stixels()
    st.SetConfig(cfg)
    st.Initialize()
    for frame_has_many in (True, False, True):   # both copy paths, alternating on one object
        if frame_has_many:
            dd, ss, want = d[0], seg[0], ref
        else:
            plain = helpers.build_case(preset, 768, 64, 32, seed=5)
            dd, ss, want = plain["disparity"][0], plain["segmentation"][0], helpers.run_oracle(plain)
        st.SetDisparityImage(dd)
        st.SetSegmentation(ss)
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        data = st.Compute(cfg.pairwise)
        mapping = st.GetInstanceStixels()
        got = dict(joined=want["joined"][None], sections=data.sections[None])
        assert not helpers.compare(want, got, 0, cfg, check_tables=False)
        want_keys = {(int(c), int(i)) for cls in range(8)
                     for c, i in want["inst_indices"][cls][:want["inst_per_class"][cls]]}
        assert set(mapping) == want_keys
    st.Finish()
    st.close()


def test_host_class_road_parameters_change_between_frames():
    """Stixels::Compute keeps the host ground model of the last frame while the road parameters stay
    the same: frames with parameters A, B, A, A on one object must each match the oracle."""
    from instance_stixels_amd import host
    import dataclasses
    case_a = helpers.build_case("drn_d_22_unary", 128, 256, 32, seed=77)
    cfg, fa = case_a["cfg"], case_a["frames"][0]
    fb = dataclasses.replace(fa, vhor_image=fa.vhor_image + 7, camera_tilt=fa.camera_tilt * 1.5,
                             camera_height=fa.camera_height * 0.9, alpha_ground=fa.alpha_ground * 1.1)
    case_b = dict(case_a)
    case_b["frames"] = [fb]
    g = helpers.oracle.host_ground(cfg, fb.vhor_image, fb.camera_tilt, fb.camera_height, fb.alpha_ground)
    case_b["gf"], case_b["ng"], case_b["ig"] = g[0][None], g[1][None], g[2][None]
    case_b["vhor"] = np.array([g[3]], np.int32)
    refs = {"a": helpers.run_oracle(case_a), "b": helpers.run_oracle(case_b)}
    assert not helpers.sections_equal(refs["a"]["sections"], refs["b"]["sections"])  # the change matters
    st = host.Stixels()
    st.SetConfig(cfg)
    st.Initialize()
    st.SetDisparityImage(fa.disparity)
    st.SetSegmentation(fa.segmentation)
    for which in "abaa":
        f = fa if which == "a" else fb
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        data = st.Compute(cfg.pairwise)
        got = dict(joined=refs[which]["joined"][None], sections=data.sections[None])
        assert not helpers.compare(refs[which], got, 0, cfg, check_tables=False), which
    st.Finish()
    st.close()


@pytest.mark.parametrize("preset,rows,cols,D,ov", [
    ("drn_d_22_unary", 128, 64, 32, {}), ("drn_d_38_pairwise", 136, 64, 16, dict(invalid_disparity=0.0)),
    ("drn_d_22_unary", 256, 96, 128, {}), ("disparity_only_pairwise", 512, 64, 64, {})])
def test_object_lut_entries_against_the_oracle(preset, rows, cols, D, ov):
    """A4 directly (ComputeObjectLUT, StixelsKernels.cu:236-296, 959-978): every entry of the
    object data-cost prefix table the prepare kernel leaves in HBM, lutT[v][fn], against the
    oracle's d_object_lut[fn][v] of the same column -- bit for bit, all columns, two images.  (The
    DP parity tests see the table only through minima of differences of its entries.)"""
    from instance_stixels_amd.core import Core
    case = helpers.build_case(preset, rows, cols, D, seed=5, n_images=2, **ov)
    cfg, p = case["cfg"], case["params"]
    core = Core(p, case["lut"], case["odr"], max_batch=2)
    try:
        out = core.run(disparity_big=case["disparity"], segmentation=case["segmentation"],
                       ground_function=case["gf"], normalization_ground=case["ng"],
                       inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=bool(cfg.pairwise),
                       median_join=bool(cfg.median_join), want_tables=False)
        for img in range(2):
            joined = oracle_mod().join_columns(cfg, case["disparity"][img])
            assert np.array_equal(helpers.bits(joined), helpers.bits(out["joined"][img]))
            for c in range(cfg.realcols):
                want = oracle_mod().object_lut_column(p, joined[c], case["lut"])     # [D][P2 + 1]
                got = core.read_object_lut(img * cfg.realcols + c)                    # [rows + 1][D]
                assert np.array_equal(helpers.bits(want[:, : rows + 1].T), helpers.bits(got)), (img, c)
    finally:
        core.close()


@pytest.mark.parametrize("seed,ov", [(3, {}), (4, dict(invalid_disparity=0.0)), (9, dict(instance_weight=0.0))])
def test_separable_block_bounds_hold_numerically(seed, ov):
    """Lemmas L7 / L8 checked as inequalities, not through their effect: the block
    summaries the device leaves behind (is_debug_read_block_summaries), combined per lane the way
    phase 1 does it (fp32, numpy restatement of l7_lane_bounds / l7_combine / the L8 loop), against
    EVERY pairwise candidate cost of tests/independent_evaluator.py: for each 32-row block k and each
    row vT above it  LB_k <= min over the block's candidates <= UB_k  (ground, sky) and
    lb_o <= min over the block's object candidates."""
    import independent_evaluator as ie
    from instance_stixels_amd.core import Core
    rows, cols, D = 192, 64, 32
    case = helpers.build_case("drn_d_38_pairwise", rows, cols, D, seed=seed, **ov)
    cfg, p = case["cfg"], case["params"]
    F32, QB = np.float32, 32
    dw, pw, sw, iw = (F32(p.disparity_weight), F32(p.prior_weight), F32(p.segmentation_weight),
                      F32(p.instance_weight))
    REL, ABS, U22 = F32(2.0 ** -20), F32(2.0 ** -90), F32(2.0 ** -22)
    dn = lambda x: (x - np.abs(x) * U22).astype(F32)   # noqa: E731
    up = lambda x: (x + np.abs(x) * U22).astype(F32)   # noqa: E731
    # PruneRec constants (is_core.hip: sigma_od; is_k_prepare.hip: E1o, E2)
    P2 = int(p.rows_power2)
    gamma = 1.01 * (2.0 * np.log2(P2) + rows / 32.0 + 8.0) * 2.0 ** -24
    lut64 = np.asarray(case["lut"], np.float64)
    sigma_od = F32(((0.0 - min(lut64.min(), 0.0)) * rows + 2.0 * gamma * rows * np.abs(lut64).max()) * 1.001)
    E1o = F32(dw * sigma_od)
    core = Core(p, case["lut"], case["odr"], max_batch=1)
    checked = 0
    try:
        out = core.run(disparity_big=case["disparity"], segmentation=case["segmentation"],
                       ground_function=case["gf"], normalization_ground=case["ng"],
                       inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=True,
                       median_join=bool(cfg.median_join), want_tables=False)
        joined, vhor, K = out["joined"][0], int(case["vhor"][0]), int(p.segmentation_classes)
        for c in range(cfg.realcols):
            summ = core.read_block_summaries(c)                      # [n_blocks][24]
            tr = {}
            ie.evaluate_column(p, c, joined[c], case["segmentation"][0][c], case["gf"][0].astype(F32),
                               case["ng"][0].astype(F32), case["ig"][0].astype(F32), vhor,
                               np.asarray(case["lut"], F32), np.asarray(case["odr"], F32), True, trace=tr)
            H = rows
            cg = np.full((H, H), np.inf, F32); cs = cg.copy(); co = cg.copy()   # [vB][vT]
            for b, (typ, t, cgs, cob) in tr["pair"].items():
                (cg if typ == ie.GROUND else cs)[b, t] = cgs
                co[b, t] = cob
            v1 = np.arange(1, H + 1)                                 # the record of lane vT: prefixes at vT + 1

            def full(ps):                                            # full-resolution prefix at v1
                kb, m = v1 // 8, v1 % 8
                return (ps[kb] * 8 + (ps[np.minimum(kb + 1, len(ps) - 1)] - ps[kb]) * m)
            ps = tr["ps"]
            n1 = (iw * (full(ps[K]) + full(ps[K + 1])).astype(F32)).astype(F32)
            G1, K1 = tr["Gps"][v1].astype(F32), tr["Kps"][v1].astype(F32)

            def lane_b(dterm, fterm):                                # l7_b
                b = (dterm + fterm).astype(F32)
                sl = ((np.abs(dterm) + fterm) * REL + ABS).astype(F32)
                return (b - sl).astype(F32), (b + sl).astype(F32)
            with np.errstate(invalid="ignore", over="ignore"):
                fg0 = (sw * (full(ps[0]).astype(F32) + n1)).astype(F32)
                fg1 = (sw * (full(ps[1]).astype(F32) + n1)).astype(F32)
                fsk = (sw * (full(ps[10]).astype(F32) + n1)).astype(F32)
                lo_g0, hi_g0 = lane_b((dw * G1).astype(F32), fg0)
                lo_g1, hi_g1 = lane_b((dw * G1).astype(F32), fg1)
                lo_s, hi_s = lane_b((dw * K1).astype(F32), fsk)
                obj = [cl for cl in range(2, 19) if cl != 10]           # 2..9: n_c = iw N, 11..18: 0
                bo = []
                for cl in obj:
                    ft = (sw * (full(ps[cl]).astype(F32) + (n1 if cl < 10 else F32(0)))).astype(F32)
                    bo.append(((ft - E1o) - ((ft + E1o) * REL + ABS)).astype(F32))
                tot2 = float(tr["MX2"][H] + tr["MY2"][H])
                E2 = float(iw) * 2.0 ** -21 * (1 + 2.0 ** -10) * tot2
                for k in range(1, summ.shape[0]):
                    lo_row, hi_row = QB * (k - 1) + 1, min(QB * k, H - 1)
                    if lo_row > hi_row:
                        continue
                    lanes = np.arange(QB * k, H)                     # vT >= the block's top row
                    if lanes.size == 0:
                        continue
                    m = summ[k].astype(F32)
                    blk = slice(lo_row, hi_row + 1)
                    act_g, act_s, act_o = (x[blk][:, lanes].min(axis=0) for x in (cg, cs, co))
                    lbg = np.fmin(dn(m[0] + lo_g0[lanes]), dn(m[1] + lo_g1[lanes]))
                    ubg = np.fmin(up(m[4] + hi_g0[lanes]), up(m[5] + hi_g1[lanes]))
                    lbs, ubs = dn(m[2] + lo_s[lanes]), up(m[6] + hi_s[lanes])
                    gl = np.isfinite(G1[lanes])                      # (+inf ground prefix: the lane is dead for the type)
                    # (a block without rows of the type has +inf summaries: phase 1 leaves it out, has_g / has_s)
                    if np.isfinite(m[0]) or np.isfinite(m[1]):
                        assert np.all(lbg[gl] <= act_g[gl]), (c, k, "ground lower bound")
                        assert np.all(ubg[gl] >= act_g[gl]), (c, k, "ground upper bound")
                    else:
                        assert not np.isfinite(act_g).any(), (c, k, "ground candidates in a block without summary")
                    if np.isfinite(m[2]):
                        assert np.all(lbs <= act_s), (c, k, "sky lower bound")
                        assert np.all(ubs >= act_s), (c, k, "sky upper bound")
                    else:
                        assert not np.isfinite(act_s).any(), (c, k, "sky candidates in a block without summary")
                    lb_n = np.min([m[8 + j] + bo[j][lanes] for j in range(8)], axis=0).astype(F32)
                    lb_i = np.min([m[16 + j] + bo[8 + j][lanes] for j in range(8)], axis=0).astype(F32)
                    top, h = QB * k, (lanes + 1 - QB * k).astype(np.float64)   # L4's term of the block's top row
                    sx, sy = (tr[a][lanes + 1] - tr[a][top] for a in ("MX", "MY"))
                    sx2, sy2 = (tr[a][lanes + 1] - tr[a][top] for a in ("MX2", "MY2"))
                    ic = float(iw) * (sx2 - sx * sx / h + sy2 - sy * sy / h)   # the real value, binary64
                    lb_o = np.fmin(lb_n, lb_i + float(sw) * (ic - 4.0 * E2))   # (computed ic <= real + E2)
                    tol = 1e-5 * np.abs(act_o[np.isfinite(act_o)]).max(initial=1.0)
                    assert np.all(dn(lb_o.astype(F32)) <= act_o + tol), (c, k, "object lower bound")
                    checked += int(lanes.size)
    finally:
        core.close()
    assert checked > 2000


def test_core_rejects_bad_shapes():
    from instance_stixels_amd.core import Core, CoreError
    case = helpers.build_case("drn_d_22_unary", 64, 64, 32, seed=1)
    p = case["params"]
    bad = type(p).from_buffer_copy(p); bad.column_step = 4
    with pytest.raises(CoreError, match="column_step"):
        Core(bad, case["lut"], case["odr"])
    bad = type(p).from_buffer_copy(p); bad.rows_power2 = 64
    with pytest.raises(CoreError, match="rows_power2"):
        Core(bad, case["lut"], case["odr"])
    core = Core(p, case["lut"], case["odr"], max_batch=1)
    with pytest.raises(CoreError, match="n_images"):
        core.compute_ptr(1, 1, np.zeros((2, 64)), np.zeros((2, 64)), np.zeros((2, 64)), [0, 0],
                         False, 2, 1)
    core.close()


def test_missing_library_fails_loudly(monkeypatch):
    from instance_stixels_amd import core
    monkeypatch.setattr(core, "_LIB", None)
    monkeypatch.setattr(core, "LIB_PATH", "/nonexistent/libis_core.so")
    with pytest.raises(core.CoreError, match="no CPU fallback"):
        core.lib()


@pytest.mark.parametrize("shape", [(21, 98, 224, 128), (21, 128, 256, 256), (21, 12, 20, 16), (3, 70, 130, 128)])
def test_flip_and_pad_kernel(shape):
    """f4: the CNN-output -> DP-input layout transform (wrappers.py:35-61) against the oracle."""
    from instance_stixels_amd.core import flip_and_pad
    from oracle import oracle
    CH, Hs, Ws, P2S = shape
    rng = np.random.default_rng(CH * Hs)
    x = rng.normal(0, 15, (2, CH, Hs, Ws)).astype(np.float32)
    got = flip_and_pad(x, P2S)
    for i in range(2):
        assert np.array_equal(got[i], oracle.flip_and_pad(x[i], P2S))


def test_road_vdisparity_kernels_and_estimation():
    """f3: v-disparity histogram / maximum / binarisation against the oracle (bit-exact), and the
    whole RoadEstimation::Compute on a synthetic ground plane."""
    import ctypes
    import torch
    from instance_stixels_amd import core, host
    from oracle import oracle
    case = helpers.build_case("drn_d_22_unary", 256, 512, 64, seed=61)
    f = case["frames"][0]
    disp = f.disparity.copy()
    disp[::7, ::5] = 0.0                                   # zeros are skipped by the histogram
    want_v, want_b, want_m = oracle.road_vdisparity(disp, 64, 0.2)
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(disp).to(dev)
    vd = torch.empty((256, 64), dtype=torch.int32, device=dev)
    mx = torch.zeros(1, dtype=torch.int32, device=dev)
    bn = torch.empty((256, 64), dtype=torch.uint8, device=dev)
    rc = core.lib().is_road_vdisparity(d.data_ptr(), 256, 512, 64, ctypes.c_float(0.2), vd.data_ptr(),
                                       mx.data_ptr(), bn.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(vd.cpu().numpy(), want_v) and int(mx.item()) == want_m
    assert np.array_equal(bn.cpu().numpy(), want_b)

    re = host.RoadEstimation()
    re.Initialize(case["cfg"].camera_center_y * 256 / 1024, case["cfg"].baseline, case["cfg"].focal,
                  256, 512, 64)
    assert re.Compute(disp)
    assert np.array_equal(re.GetBinaryVDisparity(), want_b)
    assert abs(re.horizon_point - f.vhor_image) <= 6       # generator: ramp starts at vhor_image
    assert abs(re.slope - f.alpha_ground) < 0.05 * f.alpha_ground + 0.02
    first = (re.horizon_point, re.pitch, re.camera_height, re.slope)
    re.close()

    # the wrapper's sequence (apps/stixels_wrapper.cu:159-203): SetDisparityImage ->
    # GetInputDisparityImageOnDevice -> RoadEstimation::Compute(device pointer), both objects
    # bound to a device with SetDevice (the class runs on ITS device, on its own stream, and puts
    # the caller's current device back)
    st = host.Stixels()
    st.SetConfig(case["cfg"])
    st.SetDevice(0)
    st.Initialize()
    st.SetDisparityImage(disp)
    re = host.RoadEstimation()
    re.SetDevice(0)
    re.Initialize(case["cfg"].camera_center_y * 256 / 1024, case["cfg"].baseline, case["cfg"].focal,
                  256, 512, 64)
    assert re.GetActiveDevice() == 0
    ptr = st.GetInputDisparityImageOnDevice()
    assert ptr != 0 and re.ComputeOnDevice(ptr)
    assert np.array_equal(re.GetBinaryVDisparity(), want_b)
    assert (re.horizon_point, re.pitch, re.camera_height, re.slope) == first
    assert torch.cuda.current_device() == 0
    re.close()
    st.close()


def test_plain_cpp_caller_runs():
    """examples/run_synthetic.cpp: the reference's run_cityscapes call sequence from a g++-built
    translation unit (RoadEstimation -> SetRoadParameters -> Compute -> SaveStixels)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "run_synthetic")
    if not os.path.exists(exe):
        pytest.skip("examples/run_synthetic not built (run __graft_entry__.build())")
    for pairwise in ("0", "1"):
        out = subprocess.run([exe, "256", "512", "64", pairwise, "3"], capture_output=True, text=True,
                             timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "stixels" in out.stdout and "horizon row" in out.stdout
        hor = int(out.stdout.split("horizon row")[1].split()[0])
        assert abs(hor - int(0.45 * 256)) <= 8


def test_plain_cpp_gather_caller_runs():
    """examples/gather_batch.cpp: a g++-built caller (no HIP header, no Python, no torch in the process) creates an
    RCCL communicator through the C ABI, runs its shard through Stixels::ComputeBatchGather and compares what
    rank 0 received with ComputeBatch -- as one rank here (the box has one GPU)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "gather_batch")
    if not os.path.exists(exe):
        pytest.skip("examples/gather_batch not built (run __graft_entry__.build())")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "gathered == computed: yes" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("gather", ["compact", "fixed"])
def test_bench_force_dist_child_process_runs_the_rccl_gather(gather, tmp_path):
    """bench.py --force-dist in a FRESH child process (never a re-exec of this one): RCCL (backend
    "nccl") initialises with one rank, the pipelined gather of either payload runs inside the timed
    region, rank 0's gathered copy equals what it computed, and the timed output equals the
    oracle's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--gather", gather,
                          "--steps", "2", "--warmup", "1", "--batch", "8", "--min-seconds", "0",
                          "--no-variants", "--no-cpu-baseline", "--no-single", "--no-d2h",
                          "--no-prune-stats", "--out", str(tmp_path / "bench_full.json")],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert len(last) < 4096                       # the compact line (bench.compact_line)
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["verify_all_ok"]
    assert line["gather"]["kind"] == gather and line["gather"]["rank0_copy_equals_local"]
    full = json.load(open(os.path.join(root, line["full"])))   # the complete object beside it
    assert full["value"] == pytest.approx(line["value"], rel=1e-3)
    assert full["verify"]["ok"] and full["verify"]["rccl_gather"]["rank0_copy_equals_local"]
    assert full["gather"]["kind"] == gather and full["gather"]["bytes_per_rank_per_step"]


def test_bench_default_run_prints_one_compact_line(tmp_path):
    """The driver's contract on the GPU: `python bench.py` (here with few steps and a short CPU sample) prints ONE line
    on stdout -- strict JSON, under 4 KB, every key of the contract, `roofline` and `cpu_baseline` filled, the timed
    output verified against the oracle -- and writes the complete object beside it (round 5's 29 KB line could not be
    parsed by the driver)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    from test_bench_line import REQUIRED, ROOFLINE
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--min-seconds", "0.2", "--cpu-seconds", "1.5", "--out", str(tmp_path / "bench_full.json")],
                         capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 4096, (len(lines), len(lines[0]))
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["metric"].startswith("images/s on 1024x2048x128") and d["unit"] == "images/s" and d["n_gpus"] == 1
    assert d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 1000 and d["vs_baseline"] is None
    assert d["config"]["workload"] and d["verify_all_ok"] is True and d["lut_fused_repaired"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["kernel_ms"] > 0 and r["algorithmic_bytes_per_image"] == 15532032
    assert abs(r["achieved"] - 15532032 * 64 / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value_pruning_off"] < d["value"] and d["value_floor_families"]["images_per_s"] <= d["value"] * 1.001
    full = json.load(open(tmp_path / "bench_full.json"))
    assert full["verify"]["ok"] and full["timed_blocks"]["count"] >= 1 and full["prune"]["evaluated_frac"] < 0.5


def test_capi_rccl_gather_one_rank():
    """The C-ABI gather for C++ callers (is_comm_*, is_gather_sections, Stixels::ComputeBatchGather) in a FRESH
    child process on a one-rank RCCL communicator: ncclGather of the sizes and counts, the grouped
    point-to-point payload path, dst's go-ahead (an undersized buffer is refused with IS_ENOMEM), the unpack,
    and the host class against ComputeBatch and the oracle (tests/capi_gather_child.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "capi_gather_child.py")],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0 and "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def _mock_rccl():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(lib):
        pytest.skip("tests/mock_rccl/libmock_rccl.so not built (run __graft_entry__.build())")
    return root, lib


def test_capi_gather_two_ranks_on_one_gpu(tmp_path):
    """is_gather_sections / Stixels::ComputeBatchGather with nranks = 2: two child processes share cuda:0 and a
    communicator of tests/mock_rccl (shared memory + host staging with the semantics of the RCCL calls; real RCCL
    refuses two ranks on one device and the pool leases one-GPU boxes).  Uneven shards (3 + 2 frames: the grouped
    send / receive path of the counts), equal shards (the ncclGather path), dst = 0 and dst = 1, the refusal of an
    undersized landing buffer on EVERY rank with the collective sequence intact afterwards, the host class against
    ComputeBatch and the oracle (tests/capi_gather_ranks_child.py).  A rank that posted a collective the others do
    not match would time out in the mock with a message."""
    import subprocess
    import sys
    root, lib = _mock_rccl()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IS_RCCL_LIB=lib)
    idf = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "capi_gather_ranks_child.py"), str(r), "2", idf],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=root)
             for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    report = "\n".join(f"---- rank {r} (rc {p.returncode}):\n{o[-3000:]}" for r, (p, o) in enumerate(zip(procs, outs)))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, report


def test_plain_cpp_gather_caller_two_ranks(tmp_path):
    """examples/gather_batch (plain C++, no torch) as TWO processes on cuda:0 over the mock communicator: rank 0
    receives rank 1's frames and compares them with ComputeBatch of the same frames regenerated from rank 1's seed."""
    import subprocess
    root, lib = _mock_rccl()
    exe = os.path.join(root, "examples", "gather_batch")
    if not os.path.exists(exe):
        pytest.skip("examples/gather_batch not built (run __graft_entry__.build())")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IS_RCCL_LIB=lib)
    idf = str(tmp_path / "comm.id")
    procs = [subprocess.Popen([exe, str(r), "2", idf, "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=env, cwd=root) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 holds 6 frames of 2 ranks" in outs[0] and "gathered == computed: yes" in outs[0], outs


def test_pack_sections_kernels_match_host_logic():
    """is_pack_sections / is_unpack_sections (the compacted payload of the multi-GPU gather,
    SURVEY.md 8e) on the device against the torch restatement the gloo tests use: counts, packed
    order and the round trip, with emptied columns and a column without terminator."""
    import torch
    from instance_stixels_amd.parallel import pack_sections, unpack_sections
    case = helpers.build_case("drn_d_38_pairwise", 128, 512, 32, seed=3, n_images=3)
    got = helpers.run_core(case, want_tables=False)
    sec = torch.from_numpy(got["sections"].view(np.int32).reshape(3, -1, 200, 8).copy())
    sec[0, 5, 0, 0] = -1                               # empty column
    sec[2, 63, 0, 0] = -1
    sec[1, 7, :, 0] = 1                                # no terminator at all: 199 sections
    c_ref, p_ref = pack_sections(sec)
    dev = torch.device("cuda", 0)
    c, p = pack_sections(sec.to(dev))
    assert torch.equal(c.cpu(), c_ref) and torch.equal(p.cpu(), p_ref)
    back = unpack_sections(c, p, 200).cpu()
    assert torch.equal(back, unpack_sections(c_ref, p_ref, 200))
    c2, p2 = pack_sections(back.to(dev))
    assert torch.equal(c2.cpu(), c_ref) and torch.equal(p2.cpu(), p_ref)


def test_set_device_guard_with_two_gpus():
    """Stixels::SetDevice(d): every later call runs on d whatever the caller's current device is
    (needs two GPUs; the one-GPU box skips it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from instance_stixels_amd import host
    case = helpers.build_case("drn_d_22_unary", 128, 256, 32, seed=51)
    cfg, f = case["cfg"], case["frames"][0]
    ref = helpers.run_oracle(case)
    st = host.Stixels()
    st.SetConfig(cfg)
    st.SetDevice(1)
    torch.cuda.set_device(0)
    st.Initialize()
    for _ in range(2):
        st.SetDisparityImage(f.disparity)
        st.SetSegmentation(f.segmentation)
        st.SetRoadParameters(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        data = st.Compute(cfg.pairwise)
        assert helpers.sections_equal(ref["sections"], data.sections)
        assert len(st.GetInstanceStixels()) == int(ref["inst_per_class"].sum())
        assert torch.cuda.current_device() == 0
    # RoadEstimation bound to the same device: the wrapper's device-pointer call (stixels_wrapper.cu:187)
    re0, re1 = host.RoadEstimation(), host.RoadEstimation()
    args = (cfg.camera_center_y * 128 / 1024, cfg.baseline, cfg.focal, 128, 256, 32)
    re0.Initialize(*args)                       # current device 0, host-vector path
    assert re0.Compute(f.disparity)
    re1.SetDevice(1)
    re1.Initialize(*args)
    assert re1.GetActiveDevice() == 1 and torch.cuda.current_device() == 0
    assert re1.ComputeOnDevice(st.GetInputDisparityImageOnDevice())
    assert torch.cuda.current_device() == 0
    assert np.array_equal(re0.GetBinaryVDisparity(), re1.GetBinaryVDisparity())
    assert (re0.horizon_point, re0.slope) == (re1.horizon_point, re1.slope)
    re0.close(); re1.close()
    st.close()


# ---------------------------------------------------------------------------------------------
# The exact branch-and-bound: proof-of-exactness tests AT the bounds
# ---------------------------------------------------------------------------------------------
ADV_FAMILIES = ["negative_data_costs", "ties", "constant_centres", "weight_cutoffs",
                "horizon_in_tile", "confident_scene", "separable_bound"]


def _adversarial_case(family, k):
    """Small seeded cases built so that the slack terms of the bounds matter (lemmas L1-L6 of
    DESIGN.md section 5).  Returns a helpers case."""
    rng = np.random.default_rng(40000 + 97 * ADV_FAMILIES.index(family) + k)
    pairwise = bool(k % 2) or family == "separable_bound"
    preset = ["drn_d_22_unary", "drn_d_38_pairwise"][pairwise]
    rows = int(rng.choice([128, 136, 192, 256]))
    cols = int(rng.choice([64, 96]))
    D = int(rng.choice([16, 32]))
    ov = {}
    if k % 4 >= 2:
        ov["invalid_disparity"] = 0.0
    if family == "separable_bound":
        # lemma L7 (separable block bounds of the pairwise phase 1): several 64-row blocks of
        # candidates, homogeneous regions in which every split costs the same up to the transition
        # prior -- or exactly the same (pw = 0: whole blocks of exact ties, the smallest vB must win)
        rows = int(rng.choice([192, 256, 320, 384, 448]))
        cols, D = 64, 32
        ov.update(prior_weight=float(rng.choice([1.0, 1.0, 0.25, 0.0])),
                  disparity_weight=float(rng.choice([0.0, 1e-4, 1e-2, 1.0])),
                  segmentation_weight=float(rng.choice([1.0, 4.7095, 0.5])),
                  instance_weight=float(rng.choice([0.0, 0.003131])))
    if family == "negative_data_costs":
        # the narrowest Gaussians the reference's FastLog domain allows ((1 - pout) / (sigma *
        # sqrt(2 pi)) <= 1, Stixels.cu:79-84, 786-788): where the Gaussian's mass inside [0, D) is
        # about a half (fn near 0, ground rows near the horizon) the normalisation term is negative
        # and so are LUT minima and per-row ground costs; the sky term (sigma_sky << 1) is negative
        # for every d near 0.  sigma_od, sig_g, sig_k of PruneRec then carry the bound (L3).
        pout = float(rng.uniform(0.01, 0.1))
        smin = (1.0 - pout) / 2.5066 * 1.002
        ov.update(sigma_disparity_object=float(rng.uniform(smin, smin + 0.1)),
                  sigma_disparity_ground=float(rng.uniform(smin, smin + 0.1)),
                  sigma_sky=float(rng.uniform(0.02, 0.1)), pout=pout,
                  pout_sky=float(rng.uniform(0.01, 0.1)),
                  disparity_weight=float(10.0 ** rng.uniform(-3, 0.5)))
    elif family == "ties":
        ov.update(disparity_weight=float(rng.choice([0.0, 1e-7, 1e-3])),
                  prior_weight=float(rng.choice([0.0, 1e-6, 1.0])) if not pairwise else 1.0,
                  segmentation_weight=float(rng.choice([1.0, 2.0, 0.5])),
                  instance_weight=float(rng.choice([0.0, 1e-3])))
    elif family == "weight_cutoffs":
        # sw at / next to the 1e-5 cut-off below which the host zeroes iw, iw at / next to its own
        # 1e-8 cut-off (Stixels.cu:408-423); class values x64 (column totals still < 2^24) and small
        # dw / pw so that sw * f still decides and the bound can fire
        ov.update(segmentation_weight=float(rng.choice([1e-5, 1.1e-5, 2e-5, 1e-4])),
                  instance_weight=float(rng.choice([1e-8, 0.9e-8, 1.1e-8, 1e-7, 1e-3])),
                  disparity_weight=float(rng.choice([1e-6, 1e-4])),
                  prior_weight=1.0 if pairwise else float(rng.choice([0.0, 1e-2, 1.0])))
    case = helpers.build_case(preset, rows, cols, D, seed=41000 + k, **ov)
    seg, disp = case["segmentation"], case["disparity"]
    cfg = case["cfg"]
    Hs = rows // 8
    if family == "ties":
        # tiny integer class values: the class-group minima, hence the bounds, move in steps of
        # sw and meet the best costs exactly; whole-column ties between a pruned and a winning vB
        seg[:, :, :19, :Hs] = rng.integers(0, 3, seg[:, :, :19, :Hs].shape) * \
            (rng.random(seg[:, :, :19, :Hs].shape) < 0.3)
        seg[:, 0, :19, :Hs] = 0                       # a column where every class ties everywhere
        seg[:, 1, :19, :Hs] = 1
        seg[:, :, 19:, :Hs] = rng.integers(-2, 3, seg[:, :, 19:, :Hs].shape)
        disp[:] = np.float32(3.5)                      # constant data terms
    elif family == "weight_cutoffs":
        seg[:, :, :19, :Hs] *= 64
    elif family == "constant_centres":
        # sum(x^2) - (sum x)^2 / h cancels exactly in real arithmetic: the computed instance term
        # is pure rounding noise of either sign, amplified by large centres (E2, lemma L4)
        big = int(rng.choice([1000, 4000, 12000]))
        seg[:, :, 20, :Hs] = big                                   # mx = 8 col + 4 + big: constant
        seg[:, :, 19, :Hs] = 8 * np.arange(Hs)[None, None, :] - big  # my = big + (v mod 8)
        seg[:, ::2, 19, :Hs] += rng.integers(-1, 2, seg[:, ::2, 19, :Hs].shape)
    elif family == "separable_bound":
        # constant class values per region (road below the horizon, sky above, one object slab in
        # some columns), constant offsets, a noise-free ground ramp or a constant disparity: the
        # ground / sky costs are separable up to rounding; the horizon anywhere, also inside a block
        f = case["frames"][0]
        vhor_image = int(rng.integers(8, rows - 8))
        alpha = float(rng.choice([0.0, 0.8 * D / max(1, rows - vhor_image)]))
        g = oracle_mod().host_ground(cfg, vhor_image, f.camera_tilt, f.camera_height, alpha)
        case["gf"][0], case["ng"][0], case["ig"][0], case["vhor"][0] = g
        r = np.arange(rows, dtype=np.float32)[:, None]
        ramp = np.where(r > vhor_image, np.float32(alpha) * (r - vhor_image), 0.0).astype(np.float32)
        disp[0] = np.clip(ramp + np.float32(rng.choice([0.0, 0.25, 3.5])), 0.0, D - 1.01)
        if cfg.invalid_disparity >= 0:
            disp[0][rng.random(disp[0].shape) < 0.03] = np.float32(cfg.invalid_disparity)
        hs_hor = (rows - vhor_image) // 8            # 1/8-resolution rows (from the bottom) below the horizon
        lo_v, hi_v = int(rng.choice([0, 1, 2])), int(rng.choice([3, 20, 40]))
        seg[:, :, :19, :Hs] = hi_v
        seg[:, :, 0, :hs_hor] = lo_v                  # road
        seg[:, : cols // 16, 0, :hs_hor] = hi_v       # a sidewalk strip in the first columns
        seg[:, : cols // 16, 1, :hs_hor] = lo_v
        seg[:, :, 10, hs_hor:Hs] = lo_v               # sky
        for c in range(0, seg.shape[1], 3):           # an object slab in every third column
            a = int(rng.integers(0, Hs - 2)); b = int(rng.integers(a + 1, Hs))
            cls = int(rng.choice([2, 5, 11, 13, 18]))
            seg[:, c, :19, a:b] = hi_v
            seg[:, c, cls, a:b] = lo_v
        off = int(rng.choice([0, 0, 3, -5]))
        seg[:, :, 19:, :Hs] = off
        if k % 3 == 0:                                # a little noise: near-ties instead of exact ones
            seg[:, :, :19, :Hs] += rng.integers(0, 2, seg[:, :, :19, :Hs].shape)
    elif family == "horizon_in_tile":
        f = case["frames"][0]
        vhor_image = int(rng.choice([1, 5, 63, 64, 65, rows - 70, rows - 64, rows - 2]))
        g = oracle_mod().host_ground(cfg, vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
        case["gf"][0], case["ng"][0], case["ig"][0], case["vhor"][0] = g
    return case


def oracle_mod():
    from oracle import oracle
    return oracle


def _run_counted(case, env, monkeypatch):
    """One is_compute call with the evaluation counters on; returns (outputs, evaluated steps)."""
    from instance_stixels_amd.core import Core
    for k in ("IS_NO_PRUNE",):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = case["cfg"]
    core = Core(case["params"], case["lut"], case["odr"], max_batch=len(case["frames"]))
    try:
        core.set_eval_counters(True)
        out = core.run(disparity_big=case["disparity"], segmentation=case["segmentation"],
                       ground_function=case["gf"], normalization_ground=case["ng"],
                       inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=bool(cfg.pairwise),
                       median_join=bool(cfg.median_join), want_tables=True)
        c = core.eval_counters()
    finally:
        core.close()
    steps = c["p1_full"] + c["p1_gs"] if cfg.pairwise else c["unary_full"] + c["unary_gs"]
    return out, steps


@pytest.mark.parametrize("family", ADV_FAMILIES)
def test_branch_and_bound_is_exact_at_the_bounds(family, monkeypatch):
    """36 seeded cases per family (252 in all), each run pruned and with IS_NO_PRUNE=1 (every pair
    evaluated, the reference's walk, StixelsKernels.cu:600-839; "separable_bound": lemma L7 of
    the pairwise phase 1, all 36 cases pairwise): the complete cost_table,
    index_table and all Sections must agree bit for bit; every sixth case is also compared with
    the oracle.  The families put the inputs AT the bounds: negative per-row data costs (LUT
    minima < 0), exact ties between pruned and winning vB, constant instance centres (E2
    cancellation), sw -> 1e-5 and iw at its cut-off, +inf ground prefixes starting inside a
    tile.  The counters prove that the pruned runs really skipped work."""
    pruned_steps = full_steps = 0
    for k in range(36):
        case = _adversarial_case(family, k)
        a, sa = _run_counted(case, {}, monkeypatch)
        b, sb = _run_counted(case, {"IS_NO_PRUNE": "1"}, monkeypatch)
        tag = f"{family}[{k}] pairwise={case['cfg'].pairwise}"
        assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32)), tag
        assert np.array_equal(a["index_table"], b["index_table"]), tag
        assert helpers.sections_equal(a["sections"][0], b["sections"][0]), tag
        assert sa <= sb, tag
        pruned_steps += sa
        full_steps += sb
        if k % 6 == 0:
            _assert_parity(case, a)
    assert full_steps > 0
    # the bound must fire where it can: confident scenes and tie patterns prune a lot, slack-heavy
    # families at least something (a family that never prunes would test nothing)
    assert pruned_steps < full_steps, (family, pruned_steps, full_steps)
    if family in ("confident_scene", "horizon_in_tile"):
        assert pruned_steps < 0.8 * full_steps, (family, pruned_steps, full_steps)


@pytest.mark.parametrize("preset", ["drn_d_38_pairwise", "drn_d_22_unary"])
@pytest.mark.parametrize("rows,cols,D,family,inv", [(256, 256, 64, "scene", -1.0), (512, 512, 128, "noisy_disparity", -1.0),
                                                  (512, 256, 128, "many_thin_objects", 0.0),
                                                  (256, 512, 256, "scene", -1.0), (192, 256, 48, "scene", 0.0),
                                                  (128, 256, 36, "noisy_disparity", -1.0),
                                                  (200, 256, 128, "many_thin_objects", -1.0),
                                                  (72, 512, 64, "scene", 0.0),
                                                  (1024, 1024, 128, "scene", -1.0)])
def test_phase1_fn_windows_change_nothing(preset, rows, cols, D, family, inv, monkeypatch):
    """The fn windows of the DP kernels (is_device.h, IS_P1_WIN: a (column, tile) stages 32 lutT columns of
    its 64 rows instead of all D; lanes whose floor(mean) falls outside read global memory; k_dp_unary_fast
    and k_pw_phase1) forced for EVERY tile at any batch (IS_P1_WIN_TILES=99; by default the unary kernel
    windows every tile of every call, phase 1 the tiles below the horizon of >= 16 frames) against
    the classic tiles (IS_P1_WIN_TILES=0): complete tables and Sections bit for bit, the classic run against
    the oracle, and the counters prove that the forced run did read outside its windows."""
    from instance_stixels_amd import synthetic
    from instance_stixels_amd.core import Core
    ov = dict(invalid_disparity=inv) if inv >= 0 else {}
    base = helpers.build_case(preset, rows, cols, D, seed=11, **ov)
    cfg = base["cfg"]
    pairwise = bool(cfg.pairwise)
    f = synthetic.make_frame(cfg, seed=41, family=family)
    g = oracle_mod().host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    case = dict(base)
    case.update(frames=[f], gf=g[0][None], ng=g[1][None], ig=g[2][None], vhor=np.array([g[3]], np.int32),
                disparity=f.disparity[None], segmentation=f.segmentation[None])
    outs, misses = {}, {}
    for tiles in ("99", "0"):
        monkeypatch.setenv("IS_P1_WIN_TILES", tiles)
        core = Core(case["params"], case["lut"], case["odr"], max_batch=1)
        try:
            core.set_eval_counters(True)
            outs[tiles] = core.run(disparity_big=case["disparity"], segmentation=case["segmentation"],
                                   ground_function=case["gf"], normalization_ground=case["ng"],
                                   inv_sigma2_ground=case["ig"], vhor=case["vhor"], pairwise=pairwise,
                                   median_join=bool(cfg.median_join), want_tables=True)
            misses[tiles] = core.eval_counters()["p1_window_miss" if pairwise else "unary_window_miss"]
        finally:
            core.close()
    a, b = outs["99"], outs["0"]
    assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
    assert np.array_equal(a["index_table"], b["index_table"])
    assert helpers.sections_equal(a["sections"][0], b["sections"][0])
    assert misses["0"] == 0, misses
    if D >= 64 and rows >= 192:   # (a table hardly wider than the window, or a frame of two tiles, leaves nothing outside it)
        assert misses["99"] > 0, misses
    if rows <= 512:
        _assert_parity(case, b)


@pytest.mark.parametrize("k", range(12))
def test_classic_tiles_random_and_hostile_inputs(k, monkeypatch):
    """The 8-wave workgroups with complete lutT tiles (IS_P1_WIN_TILES=0: k_dp_unary_fast<.., WIN = false>,
    k_pw_phase1<.., false>), which a unary call no longer runs by default at any size since the windowed launch
    walks its diagonal blocks in quarters: random shapes and weights, invalid disparities, median joins, hostile
    columns, both models, two images, complete tables against the oracle."""
    preset, rows, cols, D, ov = _random_case(k)
    case = helpers.build_case(preset, rows, cols, D, seed=8500 + k, n_images=2, **ov)
    if k >= 8:
        case = helpers.make_hostile(case, seed=8600 + k)
    monkeypatch.setenv("IS_P1_WIN_TILES", "0")
    _assert_parity(case, helpers.run_core(case))


@pytest.mark.parametrize("k", range(8))
def test_carry_only_lut_hostile_and_random_inputs(k, monkeypatch):
    """Carry-only lutT (DevParams::lut_carry: the prepare kernel stores only the rows 32 k of the object data-cost
    prefix table, the windowed unary ring kernel rebuilds what it reads -- the tile, the vB-side rows, and single
    entries outside the windows through lut_entry_exact; StixelsKernels.cu:236-296, 959-978 is the association
    that must be kept) against the materialised table (IS_LUT_CARRY=0) and the oracle: random shapes and weights,
    invalid disparities, median joins, AND hostile columns -- generic-encoding columns get their complete table
    from k_object_lut_generic, columns whose pruning is off (E1o = +inf) never skip an entry.  Windows forced for
    every tile (IS_P1_WIN_TILES=99), which is what switches the carry-only form on at any batch size."""
    preset, rows, cols, D, ov = _random_case(k)
    preset = preset.replace("pairwise", "unary")
    D = max(D, 64) if k % 2 else D          # (D <= 32 has no window: the mode must then stay off by itself)
    case = helpers.build_case(preset, rows, cols, D, seed=8100 + k, n_images=2, **ov)
    if k >= 4:
        case = helpers.make_hostile(case, seed=8200 + k)
    monkeypatch.setenv("IS_P1_WIN_TILES", "99")
    outs = {}
    for carry in ("1", "0"):
        monkeypatch.setenv("IS_LUT_CARRY", carry)
        outs[carry] = helpers.run_core(case)
    a, b = outs["1"], outs["0"]
    assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
    assert np.array_equal(a["index_table"], b["index_table"])
    for img in range(2):
        assert helpers.sections_equal(a["sections"][img], b["sections"][img])
    _assert_parity(case, a)


@pytest.mark.parametrize("inv", [-1.0, 0.0])
def test_carry_only_lut_full_frames(inv, monkeypatch):
    """The carry-only form at the geometry it was measured on: eight full 1024x2048x128 frames per call (2048 columns,
    unary, every tile windowed without any knob but IS_LUT_CARRY=1), against the materialised table bit for bit
    (complete tables), one frame against the oracle; the counters show how many steps needed an entry outside
    their window after the lazy test."""
    monkeypatch.delenv("IS_P1_WIN_TILES", raising=False)
    ov = dict(invalid_disparity=inv) if inv >= 0 else {}
    case2 = helpers.build_case("drn_d_22_unary", 1024, 2048, 128, seed=57, n_images=2, **ov)
    case = helpers.sub_case(case2, [i % 2 for i in range(8)])
    outs = {}
    for carry in ("1", "0"):
        monkeypatch.setenv("IS_LUT_CARRY", carry)
        outs[carry], counters = _run_with_counters(case)
        print("IS_LUT_CARRY", carry, {k: v for k, v in counters.items() if k != "p1_per_tile"})
    a, b = outs["1"], outs["0"]
    assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
    assert np.array_equal(a["index_table"], b["index_table"])
    for img in range(8):
        assert helpers.sections_equal(a["sections"][img], b["sections"][img])
    ref = helpers.run_oracle(case, image=1)
    errs = helpers.compare(ref, a, 1, case["cfg"])
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("k", range(8))
def test_fused_lut_units_hostile_and_random_inputs(k, monkeypatch):
    """The LUT units of the prepare launch as workgroups of the unary DP launch (k_dp_unary_fast, LUTF: the default of
    every unary call whose tiles are all windowed; the DP workgroups of a column wait for that column's count) --
    against the ordinary order of launches (IS_LUT_FUSED=0) bit for bit and against the oracle: random shapes and
    weights, invalid disparities, median joins and hostile (generic-encoding) columns, whose table the same units
    build for k_dp_unary.  And the REPAIR path (IS_LUT_FUSED=2: the units publish a wrong XCC id, so every DP
    workgroup distrusts the hand-over and sets the word that makes the ordinary LUT kernel and the ordinary DP launch
    behind the fused one do the call again): the same bits."""
    preset, rows, cols, D, ov = _random_case(k)
    preset = preset.replace("pairwise", "unary")
    D = max(D, 64) if k % 2 else D
    case = helpers.build_case(preset, rows, cols, D, seed=8300 + k, n_images=2, **ov)
    if k >= 4:
        case = helpers.make_hostile(case, seed=8400 + k)
    outs, ran = {}, None
    for fused in ("1", "2", "0"):
        monkeypatch.setenv("IS_LUT_FUSED", fused)
        outs[fused], counters = _run_with_counters(case)
        if fused == "1":   # (the fused form needs windowed tiles -- D > 32, a multiple of 4 -- and 1, 2 or 4 units per column)
            ran = counters["lutf_unit_cycles"] > 0
            assert ran or D not in (64, 128, 256), (D, counters)
        assert (counters["lutf_unit_cycles"] > 0) == (fused != "0" and ran), (fused, counters)
        assert counters["lutf_repaired"] == (1 if fused == "2" and ran else 0), (fused, counters)
    b = outs["0"]
    for a in (outs["1"], outs["2"]):
        assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
        assert np.array_equal(a["index_table"], b["index_table"])
        for img in range(2):
            assert helpers.sections_equal(a["sections"][img], b["sections"][img])
    _assert_parity(case, outs["1"])


@pytest.mark.parametrize("shape", [(1024, 2048, 128, -1.0), (1024, 2048, 128, 0.0), (784, 1792, 128, 0.0),
                                   (1024, 4096, 256, -1.0)])
def test_fused_lut_units_full_frames(shape, monkeypatch):
    """The fused form at the geometries of the bench line: eight full frames per call (headline shape with and
    without an invalid value, the reference's 784x1792 crop with its partial last tile, configs[4] with four units
    per column), BY DEFAULT (no knob), complete tables against the ordinary launches (IS_LUT_FUSED=0) and against the
    repair path (IS_LUT_FUSED=2), one frame against the oracle; the counters show that the units ran (their summed
    life), how often a DP workgroup had to poll, and that no repair was needed unless forced."""
    rows, cols, D, inv = shape
    monkeypatch.delenv("IS_P1_WIN_TILES", raising=False)
    ov = dict(invalid_disparity=inv) if inv >= 0 else {}
    case2 = helpers.build_case("drn_d_22_unary", rows, cols, D, seed=59, n_images=2, **ov)
    case = helpers.sub_case(case2, [i % 2 for i in range(16 if cols < 2048 else 8)])
    outs = {}
    for fused in ("default", "2", "0"):
        if fused == "default" and D <= 128:
            monkeypatch.delenv("IS_LUT_FUSED", raising=False)
        elif fused == "default":   # (four units per column: not fused by default, measured slower)
            monkeypatch.setenv("IS_LUT_FUSED", "1")
        else:
            monkeypatch.setenv("IS_LUT_FUSED", fused)
        outs[fused], counters = _run_with_counters(case)
        print("IS_LUT_FUSED", fused, {k: v for k, v in counters.items() if k.startswith("lutf")})
        assert (counters["lutf_unit_cycles"] > 0) == (fused != "0"), counters
        assert counters["lutf_repaired"] == (1 if fused == "2" else 0), counters
    b = outs["0"]
    for a in (outs["default"], outs["2"]):
        assert np.array_equal(a["cost_table"].view(np.uint32), b["cost_table"].view(np.uint32))
        assert np.array_equal(a["index_table"], b["index_table"])
        for img in range(len(a["sections"])):
            assert helpers.sections_equal(a["sections"][img], b["sections"][img])
    ref = helpers.run_oracle(case, image=1)
    errs = helpers.compare(ref, outs["default"], 1, case["cfg"])
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("knob,value", [("IS_GRAPH", "1"), ("IS_PREPARE_OVERLAP", "0"),
                                         ("IS_PREPARE_OVERLAP", "1"), ("IS_UNARY_DIAG", "1")])
@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_launch_path_knobs_change_nothing(preset, knob, value, monkeypatch):
    """The launch-path alternatives a context can be created with -- hipGraph replay of small calls
    (IS_GRAPH=1, opt-in), the two preparation kernels in order / on two streams instead of the
    fused launch small calls use -- give the same bits; the graph is replayed: three calls on one
    context with the same buffers, the ground model changing from call to call."""
    import torch
    from instance_stixels_amd.core import Core
    from instance_stixels_amd.config import SECTION_DTYPE
    monkeypatch.setenv(knob, value)
    case = helpers.build_case(preset, 128, 256, 32, seed=71, n_images=2)
    cfg, p = case["cfg"], case["params"]
    dev = torch.device("cuda", 0)
    core = Core(p, case["lut"], case["odr"], max_batch=2)
    big = torch.from_numpy(case["disparity"]).to(dev)
    seg = torch.from_numpy(case["segmentation"]).to(dev)
    joined = torch.empty((2, p.cols, p.rows), dtype=torch.float32, device=dev)
    sec = torch.empty((2, p.cols, p.max_sections, 8), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    refs = {}
    for shift in (0, 5, 0):          # same buffers every call (one graph), another horizon in between
        g = [oracle_mod().host_ground(cfg, f.vhor_image + shift, f.camera_tilt, f.camera_height,
                                      f.alpha_ground) for f in case["frames"]]
        with torch.cuda.stream(stream):
            core.join_columns_ptr(big.data_ptr(), big.shape[2], False, joined.data_ptr(), 2, stream.cuda_stream)
            core.compute_ptr(joined.data_ptr(), seg.data_ptr(), np.stack([x[0] for x in g]),
                             np.stack([x[1] for x in g]), np.stack([x[2] for x in g]),
                             np.array([x[3] for x in g], np.int32), bool(cfg.pairwise), 2, sec.data_ptr(),
                             stream=stream.cuda_stream)
        stream.synchronize()
        got = sec.cpu().numpy().view(SECTION_DTYPE).reshape(2, p.cols, p.max_sections)
        if shift not in refs:
            c2 = dict(case)
            c2["gf"], c2["ng"], c2["ig"] = (np.stack([x[k] for x in g]) for k in range(3))
            c2["vhor"] = np.array([x[3] for x in g], np.int32)
            refs[shift] = [helpers.run_oracle(c2, image=i)["sections"] for i in range(2)]
        for i in range(2):
            assert helpers.sections_equal(refs[shift][i], got[i]), (knob, shift, i)
    core.close()
