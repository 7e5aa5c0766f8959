"""Feasibility study (CPU, numpy) of an exact branch-and-bound on vB for the column DP.

For sample columns of a synthetic C2 frame: walk vB DOWNWARDS from each 64-row vT tile with the
8-wave striding of k_dp_unary and stop a wave once, for all its 64 lanes and every type, the
class-sum lower bound of the segment exceeds the lane's final best cost.  Reports the fraction of
full (non-diagonal) steps that would still be evaluated.
"""
import sys
import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers  # noqa: E402

NW = 8


def study(preset, ncols=24, seed=17, rows=1024, cols=2048, D=128, slack_rel=1e-5):
    case = helpers.build_case(preset, rows, cols, D, seed=seed)
    cfg = case["cfg"]
    C = cfg.realcols
    sel = np.linspace(0, C - 1, ncols).astype(int)
    H = rows
    sw = float(case["params"].segmentation_weight)
    pw = float(case["params"].prior_weight)
    vhor = int(case["vhor"][0])
    tot_full = tot_eval = 0
    n_cheap = [0]; n_full = [0]
    per_tile_full = np.zeros(H // 64)
    per_tile_eval = np.zeros(H // 64)
    for c in sel:
        ref = helpers.run_oracle(case, col_range=(int(c), int(c) + 1))
        ct = ref["cost_table"][c].astype(np.float64)          # [H][3]
        seg = case["segmentation"][0][c].astype(np.int64)     # [21][P2S]
        x = np.repeat(seg[:19, : H // 8], 8, axis=1)          # [19][H] full-res rows
        F = np.concatenate([np.zeros((19, 1), np.int64), np.cumsum(x, axis=1)], axis=1)  # [19][H+1]
        obj = [2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14, 15, 16, 17, 18]
        if cfg.pairwise:
            # accumulated part: min_t cost_table[vB-1][t] (priors >= 0 dropped), prefix-min over vB
            prev = np.concatenate([[0.0], np.min(ct[:-1], axis=1)])     # prev[vB], vB = 0..H-1
            Mpref = np.minimum.accumulate(prev)                        # min over vB' <= vB
        for T in range(H // 64):
            lo = 64 * T
            vT = np.arange(lo, lo + 64)
            best = ct[vT]                                     # [64][3]
            for w in range(NW):
                # full steps of this wave: vB = w, w+8, ... <= lo ; walked downwards
                vbs = np.arange(w, lo + 1, NW)[::-1]
                n_eval = 0
                o_done = gs_done = False
                for vB in vbs:
                    n_eval += 1
                    if o_done: n_cheap[0] += 1
                    else: n_full[0] += 1
                    d = F[:, vT + 1] - F[:, vB][:, None]      # [19][64]
                    lb_o = sw * d[obj].min(axis=0)
                    lb_g = sw * np.minimum(d[0], d[1])
                    lb_s = sw * d[10]
                    acc = pw * Mpref[vB] if cfg.pairwise else 0.0
                    lb_o = lb_o * (1 - slack_rel) + acc
                    lb_g = lb_g * (1 - slack_rel) + acc
                    lb_s = lb_s * (1 - slack_rel) + acc
                    done_o = np.all(lb_o >= best[:, 1])
                    # ground candidates exist only for vB <= vhor, sky only for vB > vhor
                    done_g = True if lo >= vhor else np.all(lb_g >= best[:, 0])
                    done_s = (vB <= vhor) or np.all(lb_s >= best[:, 2])
                    if lo < vhor and vB > vhor:
                        done_g = False
                    o_done = o_done or done_o
                    if o_done and done_g and done_s:
                        break
                tot_full += len(vbs)
                tot_eval += n_eval
                per_tile_full[T] += len(vbs)
                per_tile_eval[T] += n_eval
    print(f"  object still live in {n_full[0]} steps, gs-only (cheap) in {n_cheap[0]} steps of {tot_full} full steps")
    print(f"{preset}: vhor {vhor}; full steps evaluated {tot_eval}/{tot_full} = {tot_eval / tot_full:.3f}")
    print("  per tile:", " ".join(f"{e / max(f, 1):.2f}" for e, f in zip(per_tile_eval, per_tile_full)))
    # DP time model: diag steps are not pruned: iterations = full + diag(8 per wave per tile)
    diag = len(sel) * (H // 64) * 64
    print(f"  incl. diagonal steps: {(tot_eval + diag) / (tot_full + diag):.3f}")


if __name__ == "__main__":
    for p in sys.argv[1:] or ["drn_d_22_unary", "drn_d_38_pairwise"]:
        study(p)
