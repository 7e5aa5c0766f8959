"""Feasibility study (CPU, numpy) of separable block bounds for the OBJECT type of the pairwise phase 1
(lemma L8): per 64-row block and class the minimum of a_c(vB) = T_o(vB) - sw (F_c[vB] (+ iw N[vB])),
T_o = pw * min(p fields) with the O<-S transition left out where the segment's mean cannot exceed
epsilon; threshold = the candidate that extends the best object chain of the row below the tile.
Counts the blocks that survive per (column, tile)."""
import sys
import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import oracle  # noqa: E402


def study(preset="drn_d_38_pairwise", ncols=16, seed=17, rows=1024, cols=2048, D=128, family="scene"):
    from instance_stixels_amd import synthetic, make_config
    cfg = make_config(preset, rows, cols, D)
    params, lut, odr = oracle.host_initialize(cfg)
    f = synthetic.make_frame(cfg, seed=seed, family=family)
    gf, ng, ig, vhor = oracle.host_ground(cfg, f.vhor_image, f.camera_tilt, f.camera_height, f.alpha_ground)
    joined = oracle.join_columns(cfg, f.disparity)
    C, H = cfg.realcols, rows
    sw, pw = float(params.segmentation_weight), float(params.prior_weight)
    iw = float(params.instance_weight)
    eps, pord, pgrav, pblg = float(params.epsilon), float(params.pord), float(params.pgrav), float(params.pblg)
    sel = np.linspace(3, C - 4, ncols).astype(int)
    tot = surv = 0
    per_tile = np.zeros((H // 64, 2))
    NON, INS = [2, 3, 4, 5, 6, 7, 8, 9], [11, 12, 13, 14, 15, 16, 17, 18]
    for c in sel:
        ref = oracle.compute(params, lut, odr, joined, f.segmentation, gf, ng, ig, vhor, True,
                             col_range=(int(c), int(c) + 1))
        ct = ref["cost_table"][c].astype(np.float64)
        it = ref["index_table"][c]
        d = joined[c].astype(np.float64)
        seg = f.segmentation[c].astype(np.int64)
        x = np.repeat(seg[:, : H // 8], 8, axis=1)
        F = np.concatenate([np.zeros((21, 1), np.int64), np.cumsum(x[:21], axis=1)], axis=1).astype(np.float64)
        N = np.concatenate([[0], np.cumsum(x[19] ** 2 + x[20] ** 2)]).astype(np.float64)
        S = np.concatenate([[0], np.cumsum(d)])
        # transition fields per vB >= 1
        T12 = np.full(H, np.inf); T8 = np.full(H, np.inf)
        T12[0] = T8[0] = pw * (np.log(H) + np.log(D))      # first segment (above-horizon prior: lower)
        for vB in range(1, H):
            r = vB - 1
            cG, cO, cS = ct[r]
            pc = np.log(H - vB)
            ob = it[r, 1] // 3
            pm = max((S[r + 1] - S[ob]) / (r + 1 - ob), 0.0)
            gprev = max(gf[r], 0.0)
            base_g = -np.log(0.7) + pc
            with np.errstate(invalid="ignore", divide="ignore"):
                og = [base_g - np.log(pgrav) + np.log(max(D - gprev - eps, 1e-300)),
                      base_g - np.log(pblg) + (np.log(gprev - eps) if gprev - eps > 0 else np.nan),
                      base_g - np.log(1 - pgrav - pblg) + np.log(2 * eps)]
                p1 = np.nanmin([cG + pw * v for v in og]) if np.isfinite(cG) else np.inf
                base = (-np.log(0.7) if r < vhor else np.log(2.0)) + pc
                dif = max(odr[min(max(int(pm), 0), D - 1)], 0.0)
                hi = base - np.log(pord) + (np.log(D - pm - dif) if D - pm - dif > 0 else np.nan)
                lo = base - np.log(1 - pord) + (np.log(pm - dif) if pm - dif > 0 else np.nan)
                p2 = np.nanmin([cO + pw * hi, cO + pw * lo, np.inf])
                p3 = cS + pw * (np.log(D - eps) + pc)
            T12[vB] = pw * min(p1, p2)
            T8[vB] = pw * np.nanmin([p1, p2, p3])
        FN = np.zeros((16, H + 1))
        for i, cls in enumerate(NON):
            FN[i] = sw * (F[cls] + iw * N)
        for i, cls in enumerate(INS):
            FN[8 + i] = sw * F[cls]
        A12 = T12[None, :] - FN[:, :H]
        A8 = T8[None, :] - FN[:, :H]
        nb = H // 64 + 1
        blk = np.zeros(H, int); blk[1:] = (np.arange(1, H) + 63) // 64
        M12 = np.full((16, nb), np.inf); M8 = np.full((16, nb), np.inf)
        for k in range(nb):
            m = blk == k
            M12[:, k] = A12[:, m].min(axis=1); M8[:, k] = A8[:, m].min(axis=1)
        dmax_tile = d.reshape(H // 64, 64).max(axis=1)
        for t in range(1, H // 64):
            lo_ = 64 * t
            vT = np.arange(lo_, lo_ + 64)
            B = FN[:, vT + 1]                                   # [16][64]
            intile = np.maximum.accumulate(d[lo_:lo_ + 64])
            LB = np.zeros((t + 1, 64))
            for k in range(t + 1):
                dm = np.maximum(intile, dmax_tile[max(k - 1, 0):t].max() if k < t + 1 and t > max(k - 1, 0) else 0.0)
                flag3 = dm + 0.25 < eps
                lbA = (M12[:, k, None] + B).min(axis=0)
                lbB = (M8[:, k, None] + B).min(axis=0)
                LB[k] = np.where(flag3, lbA, lbB)
            # seed: extend the best object chain of row lo-1
            vs = it[lo_ - 1, 1] // 3
            r = lo_ - 1
            # exact-ish cost of candidate (vs, vT): T_exact(vs) + seg_o (ignoring dw*od)
            h = (vT + 1 - vs).astype(np.float64)
            segs = []
            for cls in NON:
                segs.append(F[cls][vT + 1] - F[cls][vs] + iw * (N[vT + 1] - N[vs]))
            mx = (8 * c + 4 + x[20]).astype(np.float64); my = np.trunc(np.arange(H) - x[19] + 0.5)
            MX = np.concatenate([[0], np.cumsum(mx)]); MY = np.concatenate([[0], np.cumsum(my)])
            MX2 = np.concatenate([[0], np.cumsum(mx * mx)]); MY2 = np.concatenate([[0], np.cumsum(my * my)])
            ic = iw * ((MX2[vT + 1] - MX2[vs]) - (MX[vT + 1] - MX[vs]) ** 2 / h + (MY2[vT + 1] - MY2[vs]) - (MY[vT + 1] - MY[vs]) ** 2 / h)
            for cls in INS:
                segs.append(F[cls][vT + 1] - F[cls][vs] + ic)
            seg_o = np.min(segs, axis=0)
            # the transition of vs as the final tables recorded it: cost_table[r][1] = T(vs) + seg(vs..r) => T_exact(vs) unknown per lane;
            # use the candidate's lower-bound T (T8) + a generous 12 (the priors differ by a few units)
            thr = T8[vs] + 12.0 + sw * seg_o if vs >= 1 else T8[0] + 12.0 + sw * seg_o
            fin = ct[vT, 1]
            thr = np.minimum(thr, np.where(np.isfinite(fin), fin + 1e-3 * np.abs(fin) + 10.0, np.inf))  # never below the truth
            sv = (LB <= thr[None, :] + 1.0).any(axis=1)
            tot += t + 1; surv += int(sv.sum())
            per_tile[t] += (t + 1, int(sv.sum()))
    print(f"{preset} {family}: object blocks {tot}, surviving {surv} = {surv / tot:.3f}")
    print("  per tile (blocks, surviving per column):", " ".join(f"{b / ncols:.0f}/{s / ncols:.1f}" for b, s in per_tile[1:]))


if __name__ == "__main__":
    for fam in (sys.argv[1:] or ["scene"]):
        study(family=fam)
