"""Feasibility study (CPU, numpy) of the separable block bounds (DESIGN.md lemma L7) of the pairwise
phase 1: per 64-row block of candidates the minimum of A_c(vB) = T(vB) - dw G[vB] - sw (F_c[vB] + iw N[vB])
bounds every ground / sky candidate of the block for the lane owning vT from below (and the block's
best candidate from above); a block whose lower bound exceeds the smallest upper bound of any block in
every lane is never walked.  Reports how many blocks per (column, tile, type) survive."""
import sys
import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers  # noqa: E402
from oracle import oracle  # noqa: E402


def study(preset="drn_d_38_pairwise", ncols=12, seed=17, rows=1024, cols=2048, D=128, family="scene"):
    from instance_stixels_amd import synthetic, make_config
    cfg = make_config(preset, rows, cols, D)
    params, lut, odr = oracle.host_initialize(cfg)
    f = synthetic.make_frame(cfg, seed=seed, family=family)
    gf, ng, ig, vhor = oracle.host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    joined = oracle.join_columns(cfg, f.disparity)
    C, H = cfg.realcols, rows
    sw, pw, dw = float(params.segmentation_weight), float(params.prior_weight), float(params.disparity_weight)
    iw = float(params.instance_weight)
    sel = np.linspace(0, C - 1, ncols).astype(int)
    tot_blocks = tot_surv = 0
    u20 = 2.0 ** -20
    for c in sel:
        ref = oracle.compute(params, lut, odr, joined, f.segmentation, gf, ng, ig, vhor, True,
                             col_range=(int(c), int(c) + 1))
        ct = ref["cost_table"][c].astype(np.float64)
        seg = f.segmentation[c].astype(np.int64)
        x = np.repeat(seg[:, : H // 8], 8, axis=1)
        F = np.concatenate([np.zeros((21, 1), np.int64), np.cumsum(x[:21], axis=1)], axis=1)
        N = np.concatenate([[0], np.cumsum(x[19] ** 2 + x[20] ** 2)])
        # data terms: use zero (dw tiny) -- the study is about the semantic part
        pc = np.log(np.maximum(H - np.arange(H), 1).astype(np.float64))
        nlog03 = -np.log(0.3)
        T = np.full(H, np.inf)
        T[0] = pw * (np.log(2.0) + np.log(H))
        for vB in range(1, H):
            p = vB - 1
            if p < vhor:
                T[vB] = pw * (min(ct[p, 0], ct[p, 1]) + pw * (nlog03 + pc[vB]))
            else:
                p1 = ct[p, 0] + pw * (pc[vB] if gf[p] < 1 else np.inf)
                p2 = ct[p, 1] + pw * (np.log(2.0) + pc[vB])
                T[vB] = pw * min(p1, p2)
        FN = sw * (F[[0, 1, 10]] + iw * N[None, :])         # [3][H+1]
        A = T[None, :] - FN[:, :H]                            # [3][H] (class 0, 1: ground; 10: sky)
        mag = np.abs(T)[None, :] + FN[:, :H]
        Alo, Ahi = A - u20 * mag, A + u20 * mag
        ground_rows = np.arange(H) <= vhor
        Alo[:2, ~ground_rows] = np.inf; Ahi[:2, ~ground_rows] = np.inf
        Alo[2, ground_rows] = np.inf; Ahi[2, ground_rows] = np.inf
        # blocks: 0 = {0}, k = 64(k-1)+1 .. 64k
        nb = H // 64 + 1
        blk = np.zeros(H, int); blk[1:] = (np.arange(1, H) + 63) // 64
        Mlo = np.full((3, nb), np.inf); Mhi = np.full((3, nb), np.inf)
        for k in range(nb):
            m = blk == k
            Mlo[:, k] = Alo[:, m].min(axis=1); Mhi[:, k] = Ahi[:, m].min(axis=1)
        for t in range(1, H // 64):
            lo = 64 * t
            vT = np.arange(lo, lo + 64)
            B = FN[:, vT + 1]                                  # [3][64]
            Blo, Bhi = B * (1 - u20), B * (1 + u20)
            for typ, cls in (("g", [0, 1]), ("s", [2])):
                if typ == "g" and lo >= vhor:
                    continue
                nblocks = t + 1  # blocks 0..t (block t = vB in 64(t-1)+1..64t = lo)
                LB = np.min(Mlo[cls][:, :nblocks, None] + Blo[cls][:, None, :], axis=0)  # [blocks][64]
                UB = np.min(Mhi[cls][:, :nblocks, None] + Bhi[cls][:, None, :], axis=0)
                thr = UB.min(axis=0)                                # [64]
                has = np.isfinite(Mlo[cls][:, :nblocks]).any(axis=0)
                surv = ((LB <= thr[None, :]).any(axis=1)) & has
                # check against the real final costs: thr must be >= final best (sanity)
                fin = ct[vT, 0 if typ == "g" else 2]
                tot_blocks += int(has.sum()); tot_surv += int(surv.sum())
    print(f"{preset} {family}: candidate blocks {tot_blocks}, surviving {tot_surv} = {tot_surv / max(tot_blocks, 1):.3f}")


if __name__ == "__main__":
    for fam in (sys.argv[1:] or ["scene", "low_confidence", "iid_noise"]):
        study(family=fam)
