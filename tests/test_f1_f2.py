"""SURVEY rows f1 (size-filtered DBSCAN + GetInstanceStixels) and f2 (SaveStixels text,
Get3DVertices), value level.

Pins: tests/golden/reference_python/f1_f2_reference_python.npz was produced by the reference's OWN Python
(`read_stixel_file`, `assign_instances` of tools/visualization/clustering_visualization.py, run in
the build container by tests/golden/reference_python/make_golden.py): the text files in it were
parsed by the reference reader and the instance labels in it were assigned by the reference twin
of the cuML call.  Everything else is checked against the oracle (oracle.cluster_instances, an
independent text formatter, a numpy restatement of Get3DVertices).
"""
import os

import numpy as np
import pytest

import helpers
from instance_stixels_amd import host, synthetic, make_config
from instance_stixels_amd.config import SECTION_DTYPE
from oracle import oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_python",
                            "f1_f2_reference_python.npz"))
N_CASES = int(GOLD["n_cases"])


def gold_case(k):
    rows, cols, D, seed, n_slabs, vhor, size_filter, min_pts = (int(x) for x in GOLD[f"c{k}_meta"])
    eps, alpha = (float(x) for x in GOLD[f"c{k}_fmeta"])
    preset = bytes(GOLD[f"c{k}_preset"]).decode()
    cfg = make_config(preset, rows, cols, D, size_filter=size_filter, eps=eps, min_pts=min_pts)
    secs = np.ascontiguousarray(GOLD[f"c{k}_sections"]).view(SECTION_DTYPE).reshape(
        GOLD[f"c{k}_sections"].shape[:2])
    return dict(cfg=cfg, secs=secs, vhor=vhor, alpha=alpha, seed=seed, n_slabs=n_slabs,
                text_a=bytes(GOLD[f"c{k}_text_a"]), text_b=bytes(GOLD[f"c{k}_text_b"]),
                ref_ints=GOLD[f"c{k}_ref_ints"], ref_floats=GOLD[f"c{k}_ref_floats"],
                ref_ground=GOLD[f"c{k}_ref_ground"], ref_labels=GOLD[f"c{k}_ref_labels"],
                ref_labels_b=GOLD[f"c{k}_ref_labels_b"],
                mapping={(int(u), int(v)): int(l) for u, v, l in GOLD[f"c{k}_mapping"]})


def fmt(x):
    """operator<<(ostream&, float) at the default precision 6 = printf("%g")."""
    return "%g" % float(np.float32(x))


def format_stixels(secs, mapping, alpha, vhor):
    """Independent restatement of the file format of Stixels::SaveStixels
    (/root/reference/InstanceStixels/src/Stixels.cu:889-926)."""
    lines = []
    for c in range(secs.shape[0]):
        parts = []
        for i in range(helpers.n_sections(secs[c])):
            s = secs[c][i]
            f = [str(int(s["type"])), str(int(s["vB"])), str(int(s["vT"])), fmt(s["disparity"]),
                 str(int(s["semantic_class"])), fmt(s["cost"]), fmt(s["instance_meanx"]),
                 fmt(s["instance_meany"])]
            if (c, i) in mapping:
                f.append(str(mapping[(c, i)]))
            parts.append(",".join(f) + ";")
        lines.append("".join(parts) + "\n")
    lines.append("groundplane%s,%d\n" % (fmt(alpha), vhor))
    return "".join(lines).encode()


def save_through_product(cfg, secs, mapping, alpha, vhor, tmp_path, name):
    st = host.Stixels()
    st.SetConfig(cfg)
    st.PrecomputeHost()  # host half only: SaveStixels needs no device
    C, S = secs.shape
    data = host.StixelsData(secs, int(cfg.rows), int(cfg.cols), C, S, int(cfg.max_dis), 8, 19,
                            alpha, vhor)
    f = str(tmp_path / name)
    st.SaveStixels(data, mapping, alpha, vhor, f)
    st.close()
    return open(f, "rb").read()


# ---------------------------------------------------------------------------------- f2, CPU
@pytest.mark.parametrize("k", range(N_CASES))
def test_save_stixels_bytes_equal_reference_parsed_fixture(k, tmp_path):
    g = gold_case(k)
    a = save_through_product(g["cfg"], g["secs"], {}, g["alpha"], g["vhor"], tmp_path, "a.stixels")
    b = save_through_product(g["cfg"], g["secs"], g["mapping"], g["alpha"], g["vhor"], tmp_path,
                             "b.stixels")
    assert a == g["text_a"], "SaveStixels output changed w.r.t. the text the reference reader parsed"
    assert b == g["text_b"]
    # and both equal the independent formatter
    assert a == format_stixels(g["secs"], {}, g["alpha"], g["vhor"])
    assert b == format_stixels(g["secs"], g["mapping"], g["alpha"], g["vhor"])


@pytest.mark.parametrize("k", range(N_CASES))
def test_reference_reader_recovered_the_sections(k):
    """What the reference's read_stixel_file parsed out of the product's text is the Section
    array: integers exactly, floats as the 6-significant-digit decimal of the fp32 value."""
    g = gold_case(k)
    secs = g["secs"]
    flat = np.concatenate([secs[c][:helpers.n_sections(secs[c])] for c in range(secs.shape[0])])
    assert len(flat) == len(g["ref_ints"])
    assert np.array_equal(g["ref_ints"][:, 0], flat["type"])
    assert np.array_equal(g["ref_ints"][:, 1], flat["vB"])
    assert np.array_equal(g["ref_ints"][:, 2], flat["vT"])
    assert np.array_equal(g["ref_ints"][:, 3], flat["semantic_class"])
    for j, name in enumerate(("disparity", "cost", "instance_meanx", "instance_meany")):
        want = np.array([float(fmt(x)) for x in flat[name]])
        assert np.array_equal(g["ref_floats"][:, j], want), name
        assert np.allclose(g["ref_floats"][:, j], flat[name].astype(np.float64), rtol=5e-6,
                           atol=0)
    assert g["ref_ground"][0] == float(fmt(g["alpha"])) and int(g["ref_ground"][1]) == g["vhor"]


@pytest.mark.parametrize("k", range(N_CASES))
def test_reference_reader_decodes_instance_labels(k):
    """Text B carries the labels as a ninth field; the reader turns label l of class c into
    l + 1000 c (clustering_visualization.py:104-113)."""
    g = gold_case(k)
    secs = g["secs"]
    want = []
    for c in range(secs.shape[0]):
        for i in range(helpers.n_sections(secs[c])):
            if (c, i) in g["mapping"]:
                l = g["mapping"][(c, i)]
                want.append(l + 1000 * int(secs[c][i]["semantic_class"]) if 0 <= l < 1000 else -1)
            else:
                want.append(-2)
    assert np.array_equal(g["ref_labels_b"], np.array(want, np.int32))
    # every stixel of an instance class, and only those, carries a label field
    flat_cls = g["ref_ints"][:, 3]
    assert np.array_equal(g["ref_labels_b"] != -2, flat_cls >= 11)


def vertices_numpy(cfg, secs, alpha, vhor):
    """numpy float32 restatement of Stixels::Get3DVertices
    (/root/reference/InstanceStixels/src/Stixels.cu:683-742), same operation order."""
    f32 = np.float32
    focal, base = f32(cfg.focal), f32(cfg.baseline)
    cx, cy = f32(cfg.camera_center_x), f32(cfg.camera_center_y)
    out = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(secs.shape[0]):
            for j in range(helpers.n_sections(secs[i])):
                s = secs[i][j]
                x_l = f32(i * cfg.column_step)
                x_r = f32(x_l + f32(cfg.column_step))
                y_t = f32(int(cfg.rows) - int(s["vT"]) - 1)
                y_b = f32(int(cfg.rows) - int(s["vB"]))
                top = bot = f32(0.0)
                if s["type"] == 1:
                    top = bot = f32(f32(base * focal) / f32(s["disparity"]))
                elif s["type"] == 0:
                    top = f32(f32(base * focal) / f32(f32(alpha) * f32(vhor - int(s["vT"]))))
                    bot = f32(f32(base * focal) / f32(f32(alpha) * f32(vhor - int(s["vB"]))))
                for (x, y, z) in ((x_l, y_t, top), (x_r, y_t, top), (x_r, y_b, bot), (x_l, y_b, bot)):
                    out += [f32(f32(-z / focal) * f32(cx - x)), f32(f32(-z / focal) * f32(cy - y)), z]
    return np.array(out, np.float32)


@pytest.mark.parametrize("k", range(N_CASES))
def test_get_3d_vertices_bitwise(k):
    g = gold_case(k)
    cfg, secs = g["cfg"], g["secs"]
    st = host.Stixels()
    st.SetConfig(cfg)
    st.PrecomputeHost()
    C, S = secs.shape
    data = host.StixelsData(secs, int(cfg.rows), int(cfg.cols), C, S, int(cfg.max_dis), 8, 19,
                            g["alpha"], g["vhor"])
    got = st.Get3DVertices(data)
    st.close()
    want = vertices_numpy(cfg, secs, g["alpha"], g["vhor"])
    assert got.shape == want.shape and got.size == 12 * len(g["ref_ints"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---------------------------------------------------------------------------------- f1, CPU
def fixture_candidates(g, cls):
    """Candidates of one class exactly as the reference twin forms them from the parsed text
    (get_instance_means, clustering_visualization.py:821-844): centres, size >= size_filter."""
    m = g["ref_ints"][:, 3] == cls
    X = g["ref_floats"][m][:, 2:4].astype(np.float32)
    size = g["ref_ints"][m, 2] - g["ref_ints"][m, 1] + 1
    lab = g["ref_labels"][m]
    return X, size >= int(g["cfg"].size_filter), np.where(lab >= 0, lab - 1000 * cls, -1)


@pytest.mark.parametrize("k", range(N_CASES))
def test_twin_equals_reference_assign_instances(k):
    """oracle.cluster_instances == the reference's own assign_instances (sklearn DBSCAN +
    nearest-core rule) on the fixture's candidates: identical labels, not just partitions."""
    g = gold_case(k)
    seen = 0
    for cls in range(11, 19):
        X, large, ref = fixture_candidates(g, cls)
        if len(X) == 0:
            continue
        got = oracle.cluster_instances(X, large, g["cfg"].eps, g["cfg"].min_pts)
        assert oracle.same_partition(got, ref), f"class {cls}"
        assert np.array_equal(got, ref), f"class {cls}: label ids differ"
        seen += int((ref >= 0).sum())
    assert seen > 20


def random_candidates(seed, n, k, spread=12.0, p_large=0.6):
    rng = np.random.default_rng(seed)
    cen = rng.uniform(0, 2000, (k, 2))
    X = (cen[rng.integers(0, k, n)] + rng.normal(0, spread, (n, 2))).astype(np.float32)
    return X, rng.random(n) < p_large


def sklearn_twin(X, large, eps, min_pts):
    from sklearn.cluster import DBSCAN
    n = len(X)
    lab = -np.ones(n, int)
    L, S = np.nonzero(large)[0], np.nonzero(~large)[0]
    if len(L) > min_pts:
        db = DBSCAN(eps=eps, min_samples=min_pts).fit(X[L].astype(np.float64))
        ll, ci = db.labels_, db.core_sample_indices_
        if len(ci) > 0:
            if len(S):
                d = ((X[S].astype(np.float64)[:, None, :] - X[L][ci].astype(np.float64)[None]) ** 2).sum(-1)
                cl = d.argmin(1)
                ok = d[np.arange(len(S)), cl] <= eps ** 2
                sl = -np.ones(len(S), int)
                sl[ok] = ll[ci[cl[ok]]]
                lab[S] = sl
            lab[L] = ll
    return lab


@pytest.mark.parametrize("seed", range(12))
def test_twin_equals_sklearn_dbscan(seed):
    rng = np.random.default_rng(100 + seed)
    X, large = random_candidates(seed, int(rng.integers(5, 500)), int(rng.integers(1, 7)))
    eps, mp = float(rng.uniform(10, 40)), int(rng.integers(2, 6))
    assert np.array_equal(oracle.cluster_instances(X, large, eps, mp), sklearn_twin(X, large, eps, mp))


def test_twin_edge_cases():
    X, large = random_candidates(3, 40, 2)
    # n_large <= min_pts: nothing is labelled (clustering_visualization.py:932)
    few = np.zeros(40, bool)
    few[:3] = True
    assert (oracle.cluster_instances(X, few, 25.0, 3) == -1).all()
    # no core point: points far apart
    far = (np.arange(40)[:, None] * np.array([[1000.0, 0.0]])).astype(np.float32)
    assert (oracle.cluster_instances(far, np.ones(40, bool), 25.0, 3) == -1).all()
    assert oracle.cluster_instances(np.zeros((0, 2), np.float32), np.zeros(0, bool), 25.0, 3).size == 0


# ---------------------------------------------------------------------------------- GPU
def device_cluster(cfg, per_class_sets):
    """Runs is_cluster_instances on 8 candidate sets [(X, large)]; returns labels per class."""
    from instance_stixels_amd.core import Core
    params, lut, odr = oracle.host_initialize(cfg)
    slots = params.cols * params.max_sections
    com = np.zeros((8, slots, 2), np.float32)
    cand = np.zeros((8, slots), np.uint8)
    per = np.zeros(8, np.int32)
    for c, (X, large) in enumerate(per_class_sets):
        n = len(X)
        com[c, :n] = X
        cand[c, :n] = large
        per[c] = n
    core = Core(params, lut, odr, max_batch=1)
    try:
        labels, packed = core.cluster_instances(com, cand, per)
    finally:
        core.close()
    # packed triples: (class, slot, label) in class order (cluster_instances() fills the index
    # array with (class, slot))
    want = np.concatenate([np.stack([np.full(int(per[c]), c), np.arange(int(per[c])),
                                     labels[c, :int(per[c])]], axis=1) for c in range(8)])
    assert np.array_equal(packed, want.astype(np.int32))
    return [labels[c, :int(per[c])] for c in range(8)]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_device_clustering_equals_twin_random(seed):
    cfg = make_config("drn_d_22_unary", 256, 1024, 64)
    rng = np.random.default_rng(seed)
    sets = [random_candidates(1000 * seed + c, int(rng.integers(50, 900)), int(rng.integers(1, 8)))
            for c in range(8)]
    got = device_cluster(cfg, sets)
    n_lab = 0
    for c, (X, large) in enumerate(sets):
        want = oracle.cluster_instances(X, large, cfg.eps, cfg.min_pts)
        assert oracle.same_partition(got[c], want), f"class {c}"
        assert np.array_equal(got[c], want), f"class {c}: label ids differ"
        n_lab += int((want >= 0).sum())
    assert n_lab > 400


@pytest.mark.gpu
def test_device_clustering_large_classes_global_memory_path():
    """Classes with more candidates than the LDS copies hold (CLU_LDS_N = 2048, is_k_cluster.hip)
    are clustered out of global memory: exactly 2048 (last LDS size), 2049 (first global size), a
    few thousand, next to small classes in the same launch."""
    cfg = make_config("drn_d_22_unary", 256, 1024, 64)
    sizes = [2048, 2049, 3000, 300, 2047, 4100, 0, 2500]
    sets = [random_candidates(7000 + c, n, 3 + c) if n else (np.zeros((0, 2), np.float32), np.zeros(0, bool))
            for c, n in enumerate(sizes)]
    # a long chain in the largest class: roots propagate over thousands of indices
    chain = np.stack([np.arange(4100) * 10.0, np.zeros(4100)], axis=1).astype(np.float32)[::-1].copy()
    sets[5] = (chain, np.arange(4100) % 5 > 0)
    got = device_cluster(cfg, sets)
    n_lab = 0
    for c, (X, large) in enumerate(sets):
        want = oracle.cluster_instances(X, large, cfg.eps, cfg.min_pts)
        assert np.array_equal(got[c], want), f"class {c} ({sizes[c]} candidates): label ids differ"
        n_lab += int((want >= 0).sum())
    assert n_lab > 5000


@pytest.mark.gpu
def test_device_clustering_edge_cases():
    cfg = make_config("drn_d_22_unary", 256, 1024, 64, eps=25.0, min_pts=4)
    X, large = random_candidates(7, 60, 2)
    few = np.zeros(60, bool)
    few[:4] = True                                   # n_large == min_pts: nothing labelled
    far = (np.arange(60)[:, None] * np.array([[1000.0, 0.0]])).astype(np.float32)
    chain = np.stack([np.arange(300) * 10.0, np.zeros(300)], axis=1).astype(np.float32)  # one long cluster
    rev = chain[::-1].copy()                         # roots propagate against the index order
    dup = np.repeat(np.array([[5.0, 5.0], [500.0, 5.0]], np.float32), 30, axis=0)
    nan = X.copy()
    nan[::7] = np.nan
    sets = [(X, few), (far, np.ones(60, bool)), (chain, np.ones(300, bool)), (rev, np.arange(300) % 3 > 0),
            (dup, np.ones(60, bool)), (np.zeros((0, 2), np.float32), np.zeros(0, bool)),
            (X, np.zeros(60, bool)), (nan, large)]
    got = device_cluster(cfg, sets)
    for c, (Xc, lc) in enumerate(sets):
        want = oracle.cluster_instances(Xc, lc, cfg.eps, cfg.min_pts)
        assert np.array_equal(got[c], want), f"set {c}: {got[c]} vs {want}"
    assert (got[0] == -1).all() and (got[1] == -1).all() and (got[6] == -1).all()
    assert (got[2] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(N_CASES))
def test_device_clustering_equals_reference_assign_instances(k):
    g = gold_case(k)
    sets, refs = [], []
    for cls in range(11, 19):
        X, large, ref = fixture_candidates(g, cls)
        sets.append((X, large))
        refs.append(ref)
    got = device_cluster(g["cfg"], sets)
    for c in range(8):
        assert oracle.same_partition(got[c], refs[c]), f"class {11 + c}"
        assert np.array_equal(got[c], refs[c])


@pytest.mark.gpu
@pytest.mark.parametrize("preset,seed", [("drn_d_22_unary", 4), ("drn_d_38_pairwise", 6),
                                         ("drn_d_22_unary", 11)])
def test_compute_labels_and_file_through_host_class(preset, seed, tmp_path):
    """The reference's caller sequence on the GPU (run_cityscapes.cu:346-449): Compute ->
    GetInstanceStixels -> SaveStixels; labels against the twin on the oracle's candidates, the
    file against the independent formatter of the oracle's sections."""
    ov = dict(size_filter=10) if preset.endswith("unary") else dict(size_filter=8)
    case = helpers.build_case(preset, 512, 2048, 128, seed=seed, **ov)
    cfg = case["cfg"]
    frame = synthetic.make_frame(cfg, seed=seed, n_slabs=24, offset_scale=1.0)
    case["frames"] = [frame]
    case["disparity"], case["segmentation"] = frame.disparity[None], frame.segmentation[None]
    ref = helpers.run_oracle(case)
    st = host.Stixels()
    st.SetConfig(cfg)
    st.Initialize()
    st.SetDisparityImage(frame.disparity)
    st.SetSegmentation(frame.segmentation)
    st.SetRoadParameters(frame.vhor_image, frame.camera_tilt, frame.camera_height, frame.alpha_ground)
    data = st.Compute(cfg.pairwise)
    mapping = st.GetInstanceStixels()
    f = str(tmp_path / "out.stixels")
    st.SaveStixels(data, mapping, data.alpha_ground, data.vhor, f)
    st.close()
    assert helpers.sections_equal(ref["sections"], data.sections)
    want = {}
    n_labelled = 0
    for cls in range(8):
        n = int(ref["inst_per_class"][cls])
        lab = oracle.cluster_instances(ref["inst_centerofmass"][cls][:n], ref["inst_core"][cls][:n],
                                       cfg.eps, cfg.min_pts)
        for (u, v), l in zip(ref["inst_indices"][cls][:n].tolist(), lab.tolist()):
            want[(u, v)] = l
        n_labelled += int((lab >= 0).sum())
    assert mapping == want
    assert n_labelled >= 50 and max(want.values()) >= 2
    assert open(f, "rb").read() == format_stixels(ref["sections"], want, frame.alpha_ground,
                                                  int(case["vhor"][0]))


def _twin_mapping(cfg, ref):
    """(column, section) -> label of every instance candidate of an oracle result, through the twin."""
    want = {}
    for cls in range(8):
        n = int(ref["inst_per_class"][cls])
        lab = oracle.cluster_instances(ref["inst_centerofmass"][cls][:n], ref["inst_core"][cls][:n],
                                       cfg.eps, cfg.min_pts)
        for (u, v), l in zip(ref["inst_indices"][cls][:n].tolist(), lab.tolist()):
            want[(u, v)] = l
    return want


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["drn_d_22_unary", "drn_d_38_pairwise"])
def test_batched_instance_path_eight_frames(preset):
    """The batched instance path (the reference emits candidates and clusters inside Compute for
    every frame: StixelsKernels.cu:926-942, Stixels.cu:613): eight distinct frames in ONE
    is_compute call -- candidates of the whole batch in one launch, clustering in one launch
    (grid 8 classes x 8 images) -- candidates in canonical order and labels against the oracle and
    its twin of the reference's assign_instances; then the same batch through
    Stixels::ComputeBatch, whose per-frame mappings must equal what Compute + GetInstanceStixels
    give frame by frame."""
    import torch
    ov = dict(size_filter=10) if preset.endswith("unary") else dict(size_filter=8)
    n = 8
    case = helpers.build_case(preset, 256, 1024, 64, seed=21, n_images=n, **ov)
    cfg = case["cfg"]
    frames = [synthetic.make_frame(cfg, seed=500 + i, n_slabs=10 + 2 * i, offset_scale=1.0) for i in range(n)]
    case["frames"] = frames
    case["disparity"] = np.stack([f.disparity for f in frames])
    case["segmentation"] = np.stack([f.segmentation for f in frames])
    got = helpers.run_core(case, want_tables=False)
    refs, wants = [], []
    n_labelled = 0
    for img in range(n):
        ref = helpers.run_oracle(case, image=img)
        errs = helpers.compare(ref, got, img, cfg, check_tables=False)   # incl. the candidate arrays
        assert not errs, "image %d:\n" % img + "\n".join(errs[:10])
        want = _twin_mapping(cfg, ref)
        for cls in range(8):
            m = int(ref["inst_per_class"][cls])
            lab = got["inst_labels"][img][cls][:m]
            idx = ref["inst_indices"][cls][:m]
            assert [want[(int(u), int(v))] for u, v in idx] == lab.tolist(), (img, cls)
        n_labelled += sum(1 for l in want.values() if l >= 0)
        refs.append(ref); wants.append(want)
    assert n_labelled >= 100

    st = host.Stixels()
    st.SetConfig(cfg)
    st.Initialize(max_batch=n)
    dev = torch.device("cuda", 0)
    big = torch.from_numpy(case["disparity"]).to(dev)
    seg = torch.from_numpy(case["segmentation"]).to(dev)
    road = [(f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground) for f in frames]
    data, maps = st.ComputeBatch(cfg.pairwise, big.data_ptr(), seg.data_ptr(), road)
    for img in range(n):
        assert helpers.sections_equal(refs[img]["sections"], data[img].sections), img
        assert maps[img] == wants[img], img
        assert data[img].vhor == int(case["vhor"][img])
    # a smaller call on the same object, without mappings, then frame by frame
    data3, none = st.ComputeBatch(cfg.pairwise, big[2:5].data_ptr(), seg[2:5].data_ptr(), road[2:5],
                                  with_instances=False)
    assert none is None
    for k in range(3):
        assert helpers.sections_equal(refs[2 + k]["sections"], data3[k].sections)
    f = frames[5]
    st.SetDisparityImage(f.disparity)
    st.SetSegmentation(f.segmentation)
    st.SetRoadParameters(*road[5])
    d5 = st.Compute(cfg.pairwise)
    assert helpers.sections_equal(refs[5]["sections"], d5.sections)
    assert st.GetInstanceStixels() == wants[5]
    st.close()
