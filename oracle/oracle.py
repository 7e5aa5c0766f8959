"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the product package.  Parity-pin status: see stixels_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

from instance_stixels_amd.config import (StixelConfig, StixelParams, SECTION_DTYPE,
                                          INSTANCE_CLASSES)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcConfig(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("max_dis", ctypes.c_int),
        ("invalid_disparity", ctypes.c_float), ("eps", ctypes.c_float),
        ("min_pts", ctypes.c_int), ("size_filter", ctypes.c_int),
        ("n_semantic_classes", ctypes.c_int), ("n_offset_channels", ctypes.c_int),
        ("prior_weight", ctypes.c_float), ("segmentation_weight", ctypes.c_float),
        ("instance_weight", ctypes.c_float), ("disparity_weight", ctypes.c_float),
        ("column_step", ctypes.c_int), ("focal", ctypes.c_float), ("baseline", ctypes.c_float),
        ("camera_center_x", ctypes.c_float), ("camera_center_y", ctypes.c_float),
        ("sigma_disparity_object", ctypes.c_float), ("sigma_disparity_ground", ctypes.c_float),
        ("sigma_sky", ctypes.c_float), ("pout", ctypes.c_float), ("pout_sky", ctypes.c_float),
        ("pord", ctypes.c_float), ("pgrav", ctypes.c_float), ("pblg", ctypes.c_float),
        ("pground_given_nexist", ctypes.c_float), ("pobject_given_nexist", ctypes.c_float),
        ("psky_given_nexist", ctypes.c_float), ("pnexist_dis", ctypes.c_float),
        ("pground", ctypes.c_float), ("pobject", ctypes.c_float), ("psky", ctypes.c_float),
        ("width_margin", ctypes.c_int), ("sigma_camera_tilt", ctypes.c_float),
        ("sigma_camera_height", ctypes.c_float), ("median_join", ctypes.c_int),
        ("epsilon", ctypes.c_float), ("range_objects_z", ctypes.c_float),
    ]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("stixels_oracle.c", "stixels_oracle.h")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_logf.restype = ctypes.c_float
        _LIB.orc_logf.argtypes = [ctypes.c_float]
    return _LIB


def _orc_config(cfg: StixelConfig) -> OrcConfig:
    oc = OrcConfig()
    for name, _ in OrcConfig._fields_:
        v = getattr(cfg, name)
        setattr(oc, name, int(v) if isinstance(v, (bool, np.bool_)) else v)
    return oc


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def host_initialize(cfg: StixelConfig):
    """-> (StixelParams, obj_cost_lut [D][D], obj_disparity_range [D])."""
    D = int(cfg.max_dis)
    params = StixelParams()
    lut = np.zeros((D, D), np.float32)
    rng = np.zeros(D, np.float32)
    oc = _orc_config(cfg)
    rc = lib().orc_host_initialize(ctypes.byref(oc), ctypes.byref(params),
                                   _p(lut, ctypes.c_float), _p(rng, ctypes.c_float))
    if rc != 0:
        raise ValueError("invalid StixelConfig (unset mandatory field)")
    return params, lut, rng


def host_ground(cfg: StixelConfig, vhor_image, camera_tilt, camera_height, alpha_ground):
    """-> (ground_function, normalization_ground, inv_sigma2_ground, vhor_lib)."""
    H = int(cfg.rows)
    gf, ng, ig = (np.zeros(H, np.float32) for _ in range(3))
    vhor = ctypes.c_int(0)
    oc = _orc_config(cfg)
    lib().orc_host_ground(ctypes.byref(oc), int(vhor_image), ctypes.c_float(camera_tilt),
                          ctypes.c_float(camera_height), ctypes.c_float(alpha_ground),
                          _p(gf, ctypes.c_float), _p(ng, ctypes.c_float), _p(ig, ctypes.c_float),
                          ctypes.byref(vhor))
    return gf, ng, ig, vhor.value


def join_columns(cfg: StixelConfig, disparity: np.ndarray) -> np.ndarray:
    H, W, C = int(cfg.rows), int(cfg.cols), cfg.realcols
    disparity = np.ascontiguousarray(disparity, np.float32)
    assert disparity.shape == (H, W)
    out = np.zeros((C, H), np.float32)
    lib().orc_join_columns(_p(disparity, ctypes.c_float), _p(out, ctypes.c_float),
                           int(cfg.column_step), int(bool(cfg.median_join)),
                           int(cfg.width_margin), H, W, C, ctypes.c_float(cfg.invalid_disparity))
    return out


def blelloch(arr: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(arr).copy()
    n = a.shape[0]
    assert n & (n - 1) == 0
    fn = {np.dtype(np.float32): ("orc_blelloch_f32", ctypes.c_float),
          np.dtype(np.int32): ("orc_blelloch_i32", ctypes.c_int32),
          np.dtype(np.int64): ("orc_blelloch_i64", ctypes.c_int64)}[a.dtype]
    getattr(lib(), fn[0])(_p(a, fn[1]), n)
    return a


def object_lut_column(params: StixelParams, disp_col, obj_cost_lut) -> np.ndarray:
    import math
    D, P2 = params.max_dis, params.rows_power2
    lut = np.zeros((D, P2 + 1), np.float32)
    disp_col = np.ascontiguousarray(disp_col, np.float32)
    obj_cost_lut = np.ascontiguousarray(obj_cost_lut, np.float32)
    npow2 = int(2 ** math.ceil(math.log2(params.rows)))
    lib().orc_object_lut_column(_p(disp_col, ctypes.c_float), _p(obj_cost_lut, ctypes.c_float),
                                _p(lut, ctypes.c_float), ctypes.byref(params), npow2)
    return lut


def compute(params: StixelParams, obj_cost_lut, obj_disparity_range, disp_joined, seg,
            ground_function, normalization_ground, inv_sigma2_ground, vhor, pairwise,
            col_range=None, nthreads=None, want_tables=True):
    """Runs the oracle DP on one image. Returns a dict with `sections` [C][S] (SECTION_DTYPE),
    `cost_table`/`index_table` [C][H][3] and the canonical instance-candidate arrays."""
    p = StixelParams.from_buffer_copy(params)
    p.vhor = int(vhor)
    C, H, S = p.cols, p.rows, p.max_sections
    disp_joined = np.ascontiguousarray(disp_joined, np.float32)
    seg = np.ascontiguousarray(seg, np.int32)
    assert disp_joined.shape == (C, H)
    assert seg.shape == (C, p.segmentation_channels, p.rows_power2_segmentation)
    c0, c1 = col_range if col_range is not None else (0, C)
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    sections = np.zeros((C, S), SECTION_DTYPE)
    sections["type"] = -1
    cost = np.full((C, H, 3), np.inf, np.float32) if want_tables else None
    index = np.full((C, H, 3), -1, np.int32) if want_tables else None
    com = np.zeros((INSTANCE_CLASSES, C * S, 2), np.float32)
    idx = np.zeros((INSTANCE_CLASSES, C * S, 2), np.int32)
    core = np.zeros((INSTANCE_CLASSES, C * S), np.uint8)
    per_class = np.zeros(INSTANCE_CLASSES, np.int32)
    f = ctypes.c_float
    rc = lib().orc_compute(
        ctypes.byref(p), _p(np.ascontiguousarray(obj_cost_lut, np.float32), f),
        _p(np.ascontiguousarray(obj_disparity_range, np.float32), f), _p(disp_joined, f),
        _p(seg, ctypes.c_int32), _p(np.ascontiguousarray(ground_function, np.float32), f),
        _p(np.ascontiguousarray(normalization_ground, np.float32), f),
        _p(np.ascontiguousarray(inv_sigma2_ground, np.float32), f), int(bool(pairwise)),
        int(c0), int(c1), int(nthreads), sections.ctypes.data_as(ctypes.c_void_p),
        _p(cost, f), _p(index, ctypes.c_int32), _p(com, f), _p(idx, ctypes.c_int32),
        _p(core, ctypes.c_uint8), _p(per_class, ctypes.c_int32))
    if rc != 0:
        raise ValueError("orc_compute rejected the parameters (column_step must be 8)")
    return dict(sections=sections, cost_table=cost, index_table=index, inst_centerofmass=com,
                inst_indices=idx, inst_core=core, inst_per_class=per_class)


def road_vdisparity(disparity: np.ndarray, max_dis: int, threshold: float):
    d = np.ascontiguousarray(disparity, np.float32)
    rows, cols = d.shape
    vdisp = np.zeros((rows, max_dis), np.int32)
    binary = np.zeros((rows, max_dis), np.uint8)
    m = lib().orc_road_vdisparity(_p(d, ctypes.c_float), rows, cols, int(max_dis),
                                  ctypes.c_float(threshold), _p(vdisp, ctypes.c_int32),
                                  _p(binary, ctypes.c_uint8))
    return vdisp, binary, int(m)


def flip_and_pad(cnn_out: np.ndarray, p2s: int) -> np.ndarray:
    x = np.ascontiguousarray(cnn_out, np.float32)
    CH, Hs, Ws = x.shape
    out = np.zeros((Ws, CH, p2s), np.int32)
    lib().orc_flip_and_pad(_p(x, ctypes.c_float), _p(out, ctypes.c_int32), CH, Hs, Ws, int(p2s))
    return out


def logf(x: float) -> float:
    return float(lib().orc_logf(ctypes.c_float(x)))


def cluster_instances(centerofmass, core_candidates, eps, min_pts):
    """CPU twin of the size-filtered DBSCAN of ONE instance class.

    Restates `assign_instances`
    (/root/reference/tools/visualization/clustering_visualization.py:894-960), the Python twin of
    the reference's cuML call (/root/reference/InstanceStixels/src/Stixels.cu:639-681), with the
    DBSCAN it delegates to sklearn written out: large points (core_candidates, i.e.
    height >= size_filter, :921-922) are clustered only when there are more than `min_pts` of them
    (:932); a large point is a core point when >= min_pts large points (itself included) lie
    within eps; clusters grow from the core points in index order and are numbered in that order;
    a non-core large point keeps the first cluster that reaches it; every small point takes the
    label of its nearest core point (first on ties) when that lies within eps (:941-955).
    Distances in float32: dx*dx + dy*dy <= eps*eps.  Returns int32 labels, -1 = none.
    Parity pin: tests/golden/reference_python/f1_f2_reference_python.npz holds labels produced by the reference's
    own assign_instances (tests/golden/reference_python/make_golden.py)."""
    X = np.ascontiguousarray(centerofmass, np.float32).reshape(-1, 2)
    cand = np.ascontiguousarray(core_candidates).astype(bool).reshape(-1)
    n = X.shape[0]
    labels = np.full(n, -1, np.int32)
    large = np.nonzero(cand)[0]
    if n == 0 or large.size <= min_pts:
        return labels
    eps2 = np.float32(eps) * np.float32(eps)

    def d2(a, b):  # [len(a)][len(b)] float32, no contraction
        dx = X[a, 0][:, None] - X[b, 0][None, :]
        dy = X[a, 1][:, None] - X[b, 1][None, :]
        return dx * dx + dy * dy

    with np.errstate(invalid="ignore", over="ignore"):
        near = d2(large, large) <= eps2                 # NaN coordinates: never a neighbour
    is_core = near.sum(axis=1) >= min_pts
    lab_l = np.full(large.size, -1, np.int32)
    nxt = 0
    for seed in range(large.size):                       # sklearn's dbscan_inner, in index order
        if lab_l[seed] != -1 or not is_core[seed]:
            continue
        stack = [seed]
        while stack:
            i = stack.pop()
            if lab_l[i] == -1:
                lab_l[i] = nxt
                if is_core[i]:
                    stack.extend(int(v) for v in np.nonzero(near[i] & (lab_l == -1))[0])
        nxt += 1
    labels[large] = lab_l
    cores = large[is_core]
    small = np.nonzero(~cand)[0]
    if cores.size and small.size:
        with np.errstate(invalid="ignore", over="ignore"):
            dist = d2(small, cores)
        dist = np.where(np.isnan(dist), np.float32(np.inf), dist)
        closest = dist.argmin(axis=1)
        dmin = dist[np.arange(small.size), closest]
        ok = dmin <= eps2
        labels[small[ok]] = labels[cores[closest[ok]]]
    return labels


def same_partition(a, b):
    """True when two label vectors describe the same clustering up to a renaming of the cluster
    ids (-1 = unlabelled must match exactly)."""
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    if a.shape != b.shape or not np.array_equal(a < 0, b < 0):
        return False
    fwd, bwd = {}, {}
    for x, y in zip(a.tolist(), b.tolist()):
        if x < 0:
            continue
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True
