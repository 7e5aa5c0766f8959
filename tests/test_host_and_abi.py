"""CPU tests of the host logic and of the C-ABI surface (no compute call: there is no GPU here)."""
import os
import re

import numpy as np
import pytest

from instance_stixels_amd import core, host, make_config, PRESETS
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_core_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "instance_stixels_core.h")).read()
    declared = set(re.findall(r"\b(is_[a-z0-9_]+)\s*\(", text))
    assert {"is_ctx_create", "is_ctx_destroy", "is_join_columns", "is_compute"} <= declared
    L = core.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libis_core.so does not export {name}"
    assert set(core.EXPORTS) <= declared
    assert b"gfx950" in L.is_version()
    # the counter array of is_get_eval_counters: one size in the header, the device code and the binding
    assert int(re.search(r"#define IS_EVAL_COUNTERS (\d+)", text).group(1)) == core.EVAL_COUNTERS
    dev = open(os.path.join(ROOT, "instance_stixels_amd", "csrc", "is_device.h")).read()
    assert int(re.search(r"#define IS_CNT_N (\d+)", dev).group(1)) == core.EVAL_COUNTERS


def test_gather_entry_points_reject_bad_arguments_without_a_gpu():
    """The C-ABI gather (is_gather.hip) validates its arguments before it touches RCCL or the device: null
    communicator / null sizes -> IS_EINVAL with a message (no GPU, no RCCL needed)."""
    import ctypes
    L = core.lib()
    assert L.is_gather_i32(None, 0, None, None, None, None) == -1
    assert b"null pointer" in L.is_last_error()
    tot = (ctypes.c_int64 * 1)()
    assert L.is_gather_sections(None, 0, None, None, None, None, None, None, 0, tot, None) == -1
    assert L.is_comm_destroy(None) == 0
    buf = ctypes.create_string_buffer(16)
    assert L.is_comm_unique_id(buf, 16) == -1          # the id needs 128 bytes


def test_local_rccl_declarations_match_the_installed_header():
    """is_gather.hip declares the few RCCL names it needs itself (the library dlopens RCCL and must build without the
    headers): where rccl.h is installed its values are the ones the source hard-codes."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no RCCL header on this machine")
    text = open(hdr).read()
    src = open(os.path.join(ROOT, "instance_stixels_amd", "csrc", "is_gather.hip")).read()
    assert "#include <rccl" not in src
    assert re.search(r"#define NCCL_UNIQUE_ID_BYTES 128\b", text) and "char internal[128]" in src
    assert re.search(r"ncclSuccess\s*=\s*0\b", text) and re.search(r"ncclSuccess = 0;", src)
    assert re.search(r"ncclInt32\s*=\s*2\b", text) and re.search(r"ncclInt32 = 2;", src)


def test_scene_family_of_the_generator_is_stable():
    """Throughput numbers are compared across rounds on the "scene" family: a later family added to
    synthetic.make_frame must not consume random numbers of the scene's stream (numpy 2.2, PCG64: the image's)."""
    import hashlib
    from instance_stixels_amd import synthetic
    cfg = make_config("drn_d_22_unary", 128, 256, 32)
    f = synthetic.make_frame(cfg, seed=17)
    assert hashlib.md5(f.disparity.tobytes() + f.segmentation.tobytes()).hexdigest() == "781312762ea81d7c7875b63d1cb66298"
    for fam in synthetic.FAMILIES:     # every family builds, in the reference's tensor format
        g = synthetic.make_frame(make_config("drn_d_22_unary", 64, 128, 32, invalid_disparity=0.0), seed=3, family=fam)
        assert g.disparity.shape == (64, 128) and g.segmentation.shape == (16, 21, 16) and g.segmentation.dtype == np.int32
        assert g.disparity.min() >= 0.0 and g.disparity.max() < 32


def test_host_library_exports():
    L = host.lib()
    for name in host.EXPORTS:
        assert hasattr(L, name)


def test_struct_layouts_match_reference_types():
    from instance_stixels_amd.config import StixelParams, SECTION_DTYPE
    import ctypes
    assert ctypes.sizeof(StixelParams) == 152           # 38 x 4 bytes, types.h:145-184
    assert StixelParams.vhor.offset == 0 and StixelParams.invalid_disparity.offset == 148
    assert SECTION_DTYPE.itemsize == 32                  # types.h:186-194
    assert SECTION_DTYPE.fields["cost"][1] == 20


@pytest.mark.parametrize("preset", sorted(PRESETS))
@pytest.mark.parametrize("shape", [(128, 256, 32), (512, 1024, 64), (1024, 2048, 128)])
def test_host_precompute_matches_oracle(preset, shape):
    cfg = make_config(preset, *shape)
    st = host.Stixels()
    st.SetConfig(cfg)
    st.PrecomputeHost()
    p, (lut, odr) = st.GetParameters(), st.GetLUTs()
    po, luto, odro = oracle.host_initialize(cfg)
    assert bytes(p) == bytes(po)
    assert np.array_equal(lut.view(np.uint32), luto.view(np.uint32))
    assert np.array_equal(odr.view(np.uint32), odro.view(np.uint32))
    for vhor_img in (0, shape[0] // 3, shape[0] - 1):
        st.SetRoadParameters(vhor_img, 0.05, 1.2, 0.11)
        gf, ng, ig, vh = st.GetGroundModel()
        g2 = oracle.host_ground(cfg, vhor_img, 0.05, 1.2, 0.11)
        assert vh == g2[3] == shape[0] - vhor_img - 1     # Stixels.cu:377
        for a, b in zip((gf, ng, ig), g2[:3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    st.close()


def test_instance_weight_rule():
    # Stixels.cu:408-423: divided by the segmentation weight, zeroed for tiny weights
    cfg = make_config("drn_d_22_unary", 64, 64, 32)
    p, _, _ = oracle.host_initialize(cfg)
    assert p.instance_weight == np.float32(np.float32(0.001731) / np.float32(11.241965))
    cfg.segmentation_weight = 0.0
    assert oracle.host_initialize(cfg)[0].instance_weight == 0.0
    cfg.segmentation_weight, cfg.instance_weight = 1.0, 1e-9
    assert oracle.host_initialize(cfg)[0].instance_weight == 0.0


@pytest.mark.parametrize("field,msg", [
    ("rows", "Number of rows or columns are not set."),
    ("max_dis", "Maximum disparity value is not set."),
    ("eps", "Clustering parameters are not set."),
    ("prior_weight", "Energy term weights are not set."),
    ("column_step", "Stixel width is not set."),
    ("focal", "Camera parameters are not set."),
])
def test_set_config_rejects_unset_fields(field, msg):
    # std::invalid_argument messages of Stixels.cu:292-313
    cfg = make_config("drn_d_22_unary", 64, 64, 32)
    setattr(cfg, field, -1)
    st = host.Stixels()
    with pytest.raises(ValueError, match=re.escape(msg)):
        st.SetConfig(cfg)
    with pytest.raises(ValueError):
        oracle.host_initialize(cfg)
    st.close()


def test_bench_byte_formula():
    from instance_stixels_amd import synthetic
    # C1 4 292 608 B, C2 15 532 032 B, C5 31 064 064 B
    assert synthetic.algorithmic_bytes_per_image(make_config("drn_d_22_unary", 512, 1024, 64)) == 4292608
    assert synthetic.algorithmic_bytes_per_image(make_config("drn_d_22_unary", 1024, 2048, 128)) == 15532032
    assert synthetic.algorithmic_bytes_per_image(make_config("drn_d_22_unary", 1024, 4096, 256)) == 31064064
    assert synthetic.pair_evaluations_per_image(make_config("drn_d_22_unary", 1024, 2048, 128)) == 134348800


def test_hough_lines_on_synthetic_plane():
    """f3: the host Hough transform that replaces cv::HoughLines (RoadEstimation.cu:153).  A
    v-disparity line d = alpha * (row - v0) must come back as the strongest line with
    rho / sin(theta) = v0 (the horizon row, RoadEstimation.cu:181)."""
    rows, D, v0, alpha = 256, 64, 100, 0.35
    img = np.zeros((rows, D), np.uint8)
    for r in range(v0, rows):
        c = int(round(alpha * (r - v0)))
        if c < D:
            img[r, c] = 255
    lines = host.hough_lines(img, threshold=25)
    assert len(lines) > 0
    rho, theta = abs(float(lines[0][0])), float(lines[0][1])
    assert abs(rho / np.sin(theta) - v0) <= 3.0
    # slope as computed by ComputeCameraProperties (RoadEstimation.cu:187-189)
    last = rows - 1
    down = (rho - last * np.sin(theta)) / np.cos(theta)
    slope = (0 - down) / (rho / np.sin(theta) - last)
    assert abs(slope - alpha) < 0.03
    assert len(host.hough_lines(np.zeros((64, 32), np.uint8))) == 0


def test_oracle_vdisparity_kernels():
    # RoadEstimationKernels.cu:25-60 against numpy
    rng = np.random.default_rng(2)
    d = (rng.random((40, 96)) * 31).astype(np.float32)
    d[rng.random((40, 96)) < 0.2] = 0.0
    vdisp, binary, m = oracle.road_vdisparity(d, 32, 0.2)
    want = np.zeros((40, 32), np.int32)
    for r in range(40):
        for x in d[r]:
            if x != 0:
                want[r, int(x)] += 1
    assert np.array_equal(vdisp, want) and m == want.max()
    assert np.array_equal(binary, np.where(want.astype(np.float32) > np.float32(m) * np.float32(0.2), 255, 0))
