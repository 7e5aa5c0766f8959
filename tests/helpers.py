"""Shared helpers of the parity tests: build a seeded case, run oracle / HIP core, compare."""
import numpy as np

from instance_stixels_amd import make_config, synthetic
from oracle import oracle


def build_case(preset, rows, cols, max_dis, seed=0, n_images=1, **overrides):
    zero_seg = overrides.pop("zero_segmentation", preset.startswith("disparity_only"))
    cfg = make_config(preset, rows, cols, max_dis, **overrides)
    params, lut, odr = oracle.host_initialize(cfg)
    frames = [synthetic.make_frame(cfg, seed=seed + 1000 * i, zero_segmentation=zero_seg)
              for i in range(n_images)]
    ground = [oracle.host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
              for f in frames]
    return dict(cfg=cfg, params=params, lut=lut, odr=odr, frames=frames,
                gf=np.stack([g[0] for g in ground]), ng=np.stack([g[1] for g in ground]),
                ig=np.stack([g[2] for g in ground]),
                vhor=np.array([g[3] for g in ground], np.int32),
                disparity=np.stack([f.disparity for f in frames]),
                segmentation=np.stack([f.segmentation for f in frames]))


def sub_case(case, images):
    """The same case restricted to some of its images."""
    sub = dict(case)
    sub["frames"] = [case["frames"][i] for i in images]
    for k in ("gf", "ng", "ig", "vhor", "disparity", "segmentation"):
        sub[k] = case[k][list(images)]
    return sub


def make_hostile(case, seed):
    """Overwrites whole stixel columns of a case with inputs outside the FAST encodings: negative
    class values, offsets whose squares wrap int32, class totals next to 2^24, tiny / subnormal /
    zero disparities, uniformly random disparities.  In place; returns the case."""
    rng = np.random.default_rng(seed)
    seg, d = case["segmentation"], case["disparity"]      # [n][C][21][P2S], [n][H][W]
    rows, D = int(case["cfg"].rows), int(case["cfg"].max_dis)
    for c in range(seg.shape[1]):
        mode = rng.integers(0, 6)
        px = d[:, :, 8 * c:8 * c + 8]
        if mode == 0:
            seg[:, c, :19] = rng.integers(-50, 400, seg[:, c, :19].shape)
        elif mode == 1:
            seg[:, c, 19:] = rng.integers(-16000, 16000, seg[:, c, 19:].shape)
        elif mode == 2:
            seg[:, c, :19] = rng.integers(0, 130000 // max(1, rows // 8), seg[:, c, :19].shape)
        elif mode == 3:
            px[...] = rng.choice([0.0, 1e-30, 3e-39, 0.5, D - 1.01], px.shape)
        elif mode == 4:
            px[...] = rng.uniform(0, D - 1.01, px.shape)
    return case


def run_oracle(case, image=0, col_range=None, joined=None):
    cfg = case["cfg"]
    if joined is None:
        joined = oracle.join_columns(cfg, case["disparity"][image])
    out = oracle.compute(case["params"], case["lut"], case["odr"], joined,
                         case["segmentation"][image], case["gf"][image], case["ng"][image],
                         case["ig"][image], int(case["vhor"][image]), bool(cfg.pairwise),
                         col_range=col_range)
    out["joined"] = joined
    return out


def run_core(case, max_batch=None, want_tables=True, use_join=True, joined=None):
    from instance_stixels_amd.core import Core
    cfg = case["cfg"]
    n = len(case["frames"])
    core = Core(case["params"], case["lut"], case["odr"], max_batch=max_batch or n)
    try:
        kw = dict(segmentation=case["segmentation"], ground_function=case["gf"],
                  normalization_ground=case["ng"], inv_sigma2_ground=case["ig"],
                  vhor=case["vhor"], pairwise=bool(cfg.pairwise),
                  median_join=bool(cfg.median_join), want_tables=want_tables)
        if use_join and joined is None:
            return core.run(disparity_big=case["disparity"], **kw)
        return core.run(joined=joined, **kw)
    finally:
        core.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def n_sections(sec_col):
    return int(np.argmax(sec_col["type"] == -1))


def sections_equal(a, b):
    """Bitwise equality of two [cols][max_sections] Section arrays up to (and including) each
    column's terminator; entries behind it are unspecified."""
    for ca, cb in zip(a, b):
        n = n_sections(ca)
        if n != n_sections(cb) or cb["type"][n] != -1:
            return False
        if not np.array_equal(ca[:n].view(np.int32), cb[:n].view(np.int32)):
            return False
    return True


def compare(ref, got, image, cfg, cols=None, check_tables=True):
    """Returns a list of human-readable mismatch strings (empty = parity)."""
    errs = []
    C = cfg.realcols
    cols = range(C) if cols is None else cols
    if not np.array_equal(bits(ref["joined"]), bits(got["joined"][image])):
        errs.append("joined disparity differs bitwise in %d entries" %
                    int((bits(ref["joined"]) != bits(got["joined"][image])).sum()))
    rs, gs = ref["sections"], got["sections"][image]
    for c in cols:
        nr, ng = n_sections(rs[c]), n_sections(gs[c])
        if nr != ng:
            errs.append(f"col {c}: {nr} sections in oracle vs {ng}")
            continue
        a, b = rs[c][:nr], gs[c][:nr]
        for f in ("type", "vB", "vT", "semantic_class"):
            if not np.array_equal(a[f], b[f]):
                errs.append(f"col {c}: field {f} differs: {a[f].tolist()} vs {b[f].tolist()}")
        for f in ("disparity", "instance_meanx", "instance_meany", "cost"):
            if not np.array_equal(bits(a[f]), bits(b[f])):
                rel = np.max(np.abs(a[f] - b[f]) / np.maximum(np.abs(a[f]), 1e-30))
                errs.append(f"col {c}: field {f} not bit-identical (max rel {rel:.3e})")
    if check_tables and "cost_table" in got and ref.get("cost_table") is not None:
        ci = list(cols)
        rc, gc = ref["cost_table"][ci], got["cost_table"][image][ci]
        bad = bits(rc) != bits(gc)
        if bad.any():
            w = np.argwhere(bad)[:5]
            errs.append(f"cost_table differs bitwise in {int(bad.sum())} entries, first {w.tolist()}: "
                        f"{rc[bad][:5]} vs {gc[bad][:5]}")
        ri, gi = ref["index_table"][ci], got["index_table"][image][ci]
        written = ri >= 0
        if cfg.pairwise:
            badi = written & (ri != gi)
        else:  # unary core stores the winning vB only (predecessor type resolved in back-trace)
            badi = written & ((ri // 3) != gi)
        if badi.any():
            w = np.argwhere(badi)[:5]
            errs.append(f"index_table differs in {int(badi.sum())} entries, first {w.tolist()}: "
                        f"{ri[badi][:5]} vs {gi[badi][:5]}")
    for k in ("inst_per_class",):
        if k in got and cols == range(C):
            if not np.array_equal(ref[k], got[k][image]):
                errs.append(f"{k}: {ref[k].tolist()} vs {got[k][image].tolist()}")
            else:
                S = ref["sections"].shape[1]
                for cls in range(8):
                    m = int(ref[k][cls])
                    for name in ("inst_centerofmass", "inst_indices", "inst_core"):
                        ra = ref[name][cls][:m]
                        ga = got[name][image][cls][:m]
                        if not np.array_equal(ra.view(np.uint8), ga.view(np.uint8)):
                            errs.append(f"{name}[class {cls}] differs")
    return errs
