"""A second, independent evaluation of the column DP at the kernel boundary (test infrastructure).

Written from the formula sheet and the reference's kernel source
(/root/reference/InstanceStixels/src/StixelsKernels.cu, Cityscapes.h, StixelsKernels.h) -- NOT from
oracle/stixels_oracle.c: numpy float32 arrays, one lane per vT, a Python loop over vB, so that a
mis-transcribed formula in the C oracle (which shares its author with the HIP kernels) shows up as a
bitwise difference in cost_table / index_table / Sections.  Small shapes only (seconds per case).

Numerics: IEEE fp32, no contraction (numpy never fuses), logf = the correctly rounded
float32 logarithm (float64 log, rounded once).  The fp32 associations that matter are reproduced
literally: the work-efficient block scan (StixelsKernels.h:73-103) and the 32-lane Kogge-Stone scan
with carry of the object LUT (StixelsKernels.cu:236-296)."""
import numpy as np

F = np.float32
INF = F(np.inf)
GROUND, OBJECT, SKY = 0, 1, 2


def logf(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(np.asarray(x, np.float64)).astype(np.float32)


def neg_fastlog_div(v, v2):                       # NegFastLogDiv, :35-38
    return -logf(v) + logf(v2)


def block_scan(a):
    """ComputePrefixSum (StixelsKernels.h:73-103), literally: up-sweep, clear, down-sweep."""
    a = a.copy()
    n, offset, d = len(a), 1, len(a) >> 1
    while d > 0:
        t = np.arange(d)
        ai, bi = offset * (2 * t + 1) - 1, offset * (2 * t + 2) - 1
        a[bi] = a[bi] + a[ai]
        offset *= 2
        d >>= 1
    a[n - 1] = 0
    d = 1
    while d < n:
        offset >>= 1
        t = np.arange(d)
        ai, bi = offset * (2 * t + 1) - 1, offset * (2 * t + 2) - 1
        tmp = a[ai].copy()
        a[ai] = a[bi]
        a[bi] = a[bi] + tmp
        d *= 2
    return a


def object_lut(p, d, cost_lut):
    """ComputeObjectLUT (:959-978) + warp_prefix_sum (:236-273): [D][P2 + 1]."""
    H, D, P2 = p.rows, p.max_dis, p.rows_power2
    npow2 = 1 << int(np.ceil(np.log2(H)))
    dis = np.zeros(npow2, np.int64)
    dis[:H] = d.astype(np.int32)                  # (int) d, rows beyond the image use 0
    out = np.zeros((D, P2 + 1), np.float32)
    add = np.zeros(D, np.float32)
    for i in range(0, npow2, 32):
        c = cost_lut[:, dis[i:i + 32]].astype(np.float32).copy()      # [D][32]: lane = row
        c[:, 0] = c[:, 0] + add
        j = 1
        while j < 32:                              # __shfl_up: lanes >= j add the value j lanes below
            n = c[:, :-j].copy()
            c[:, j:] = c[:, j:] + n
            j *= 2
        out[:, i + 1:i + 33] = c
        add = c[:, 31].copy()
    return out


def downsampled_sum(ps, vB, vT):
    """DownsampledSum (Cityscapes.h:28-42) in wrapping int32; ps: 1/8-resolution exclusive prefix."""
    ps = ps.astype(np.int64)
    r = (ps[vT // 8] - ps[vB // 8]) * 8 + (ps[vT // 8 + 1] - ps[vT // 8]) * (vT % 8 + 1) \
        - (ps[vB // 8 + 1] - ps[vB // 8]) * (vB % 8)
    return ((r + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int64)       # wrap to int32


def i32(x):
    return ((np.asarray(x, np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31)


def evaluate_column(p, col, d, seg, gf, ng, ig, vhor, cost_lut, odr, pairwise, trace=None):
    """One stixel column -> (cost_table [H][3] f32, index_table [H][3] i32, sections list).  trace: a dict
    that receives the prefix arrays and, per vB >= 1, (type, vT, ground- or sky cost, object cost)."""
    H, D, P2, P2S, K = p.rows, p.max_dis, p.rows_power2, p.rows_power2_segmentation, p.segmentation_classes
    dw, pw, sw, iw = F(p.disparity_weight), F(p.prior_weight), F(p.segmentation_weight), F(p.instance_weight)
    inv = F(p.invalid_disparity)
    eps, maxd = F(p.epsilon), F(D)
    d = d.astype(np.float32)
    rows = np.arange(H)

    # ---- load phase (:371-446) and the block scans (:452-469)
    def scan_rows(x, dtype):
        a = np.zeros(P2, dtype)
        a[:H] = x
        return block_scan(a)
    if inv >= 0:
        valid = (d != inv).astype(np.float32)
        Vps = scan_rows(valid, np.float32)
        Sps = scan_rows(valid * d, np.float32)
    else:
        Vps = None
        Sps = scan_rows(d, np.float32)
    offy, offx = seg[K].astype(np.int64), seg[K + 1].astype(np.int64)
    mx = np.trunc((8 * col + 0.5 * 7.0) + offx[rows // 8] + 0.5).astype(np.int64)
    my = np.trunc((rows - offy[rows // 8]) + 0.5).astype(np.int64)
    MX, MY = scan_rows(mx, np.int64), scan_rows(my, np.int64)
    MX2, MY2 = scan_rows(mx * mx, np.int64), scan_rows(my * my, np.int64)
    with np.errstate(invalid="ignore", over="ignore"):
        sky_row = np.where(d != inv, np.fmin(F(p.puniform_sky), F(p.normalization_sky) + d * d * F(p.inv_sigma2_sky))
                           + F(p.nopnexists_given_sky_log), F(p.pnexists_given_sky_log)).astype(np.float32)
        md = d - gf
        gnd_row = np.where(d != inv, np.fmin(F(p.puniform), ng + md * md * ig) + F(p.nopnexists_given_ground_log),
                           F(p.pnexists_given_ground_log)).astype(np.float32)
        Kps = scan_rows(np.where(rows < vhor, F(0), sky_row), np.float32)
        Gps = scan_rows(np.where(rows >= vhor, INF, gnd_row), np.float32)
    ps = np.zeros((K + 2, P2S), np.int64)          # exclusive prefixes of the channels, offsets squared first
    for c in range(K + 2):
        x = seg[c].astype(np.int64)
        if c >= K:
            x = i32(x * x)
        ps[c] = i32(block_scan(x))                 # (int32 additions wrap; any association is exact)
    lut = object_lut(p, d, cost_lut)
    if trace is not None:
        trace.update(Gps=Gps, Kps=Kps, ps=ps, MX=MX, MY=MY, MX2=MX2, MY2=MY2, pair={})

    def mean(vB, vT):                              # ComputeMean, :47-60
        with np.errstate(invalid="ignore", divide="ignore"):
            if inv >= 0:
                vd = Vps[vT + 1] - Vps[vB]
                return np.where(vd == 0, F(0), (Sps[vT + 1] - Sps[vB]) / vd).astype(np.float32)
            return ((Sps[vT + 1] - Sps[vB]) / (vT + 1 - vB).astype(np.float32)).astype(np.float32)

    def seg_terms(vB, vT):
        h = (vT + 1.0 - vB).astype(np.float32)
        fx, fy = (MX[vT + 1] - MX[vB]).astype(np.float32), (MY[vT + 1] - MY[vB]).astype(np.float32)
        fx2, fy2 = (MX2[vT + 1] - MX2[vB]).astype(np.float32), (MY2[vT + 1] - MY2[vB]).astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            ic = iw * (fx2 - fx * fx / h + fy2 - fy * fy / h)          # :72-86
            nic = iw * i32(downsampled_sum(ps[K + 1], vB, vT) + downsampled_sum(ps[K], vB, vT)).astype(np.float32)
            g = np.fmin(downsampled_sum(ps[0], vB, vT).astype(np.float32),
                        downsampled_sum(ps[1], vB, vT).astype(np.float32)) + nic
            o = np.full(vT.shape, INF, np.float32)
            for c in range(2, 19):                 # Cityscapes.h:61-84
                if c == 10:
                    continue
                cs = (F(0) + (nic if c < 10 else ic)) + downsampled_sum(ps[c], vB, vT).astype(np.float32)
                o = np.where(o > cs, cs, o)
            s = downsampled_sum(ps[10], vB, vT).astype(np.float32) + nic
        return g, o.astype(np.float32), s, ic, nic

    ct = np.full((H, 3), INF, np.float32)
    it = np.full((H, 3), -1, np.int32)
    vT = rows
    rows_log, md_log = F(p.rows_log), F(p.max_dis_log)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        # ---- first segment, vB = 0 (:481-594)
        vB0 = np.zeros(H, np.int64)
        g, o, s, _, _ = seg_terms(vB0, vT)
        fn = mean(vB0, vT)
        fn = np.where(fn < 0, F(0), fn)
        fni = np.floor(fn).astype(np.int64)
        od = lut[fni, vT + 1] - lut[fni, 0]
        gd = Gps[vT + 1] - Gps[0]
        below = vT <= vhor
        ih = (1.0 / (vT + 1.0)).astype(np.float32)
        if pairwise:
            cg = dw * gd + pw * (logf(F(2)) + rows_log) + sw * g
            co = dw * od + pw * ((rows_log + np.where(below, logf(F(2)), F(0))) + md_log) + sw * o
        else:
            cg = dw * gd + pw * ih + sw * g
            co = dw * od + pw * ih + sw * o
        u = below & (cg < ct[:, GROUND])
        ct[u, GROUND] = cg[u]; it[u, GROUND] = GROUND
        u = co < ct[:, OBJECT]
        ct[u, OBJECT] = co[u]
        it[:, OBJECT] = OBJECT
        # ---- vB >= 1 (:600-839): row vB - 1 is final when vB starts
        for b in range(1, H):
            m = vT >= b
            t, vB = vT[m], np.full(int(m.sum()), b, np.int64)
            g, o, s, _, _ = seg_terms(vB, t)
            fn = mean(vB, t)
            fn = np.where(fn < 0, F(0), fn)
            fni = np.floor(fn).astype(np.int64)
            od = lut[fni, t + 1] - lut[fni, b]
            ih = (1.0 / (t + 1.0 - b)).astype(np.float32)
            pv = b - 1
            cG, cO, cS = ct[pv]
            if pairwise:
                pc = neg_fastlog_div(F(1), F(H - b))
                ob = int(it[pv, OBJECT]) // 3
                pm = mean(np.array([ob]), np.array([pv]))[0]
                pm = F(0) if pm < 0 else pm
            ground_range = pv < vhor
            if ground_range:                       # :687-728
                data = Gps[t + 1] - Gps[b]
                p1, p2 = cG, cO
                if pairwise:
                    prev = -logf(F(0.3)) + pc
                    p1, p2 = cG + pw * prev, cO + pw * prev
                    cost = dw * data + pw * np.fmin(p1, p2) + sw * g
                else:
                    cost = dw * data + pw * ih + sw * g
                typ = GROUND
            else:                                  # :729-775
                data = Kps[t + 1] - Kps[b]
                p1, p2 = cG, cO
                if pairwise:
                    p1 = cG + pw * (pc if gf[pv] < 1.0 else INF)
                    p2 = cO + pw * (INF if pm < eps else logf(F(2)) + pc)
                    cost = dw * data + pw * np.fmin(p1, p2) + sw * s
                else:
                    cost = dw * data + pw * ih + sw * s
                typ = SKY
            u = cost < ct[t, typ]
            cgs = cost
            ct[t[u], typ] = cost[u]
            it[t[u], typ] = b * 3 + (GROUND if p1 < p2 else OBJECT)
            # object (:777-837)
            q1, q2, q3 = np.full(t.shape, cG, np.float32), np.full(t.shape, cO, np.float32), np.full(t.shape, cS, np.float32)
            if pairwise:
                fp = max(gf[pv], F(0))
                c1 = -logf(F(0.7)) + pc
                c1 = np.where(fn > fp + eps, c1 + neg_fastlog_div(F(p.pgrav), maxd - fp - eps),
                              np.where(fn < fp - eps, c1 + neg_fastlog_div(F(p.pblg), fp - eps),
                                       c1 + neg_fastlog_div(F(1) - F(p.pgrav) - F(p.pblg), F(2) * eps))).astype(np.float32)
                c2 = (-logf(F(0.7)) if pv < vhor else logf(F(2))) + pc
                dif = max(odr[int(pm)], F(0))
                c2 = np.where(fn > pm + dif, c2 + neg_fastlog_div(F(p.pord), maxd - pm - dif),
                              np.where(fn < pm - dif, c2 + neg_fastlog_div(F(1) - F(p.pord), pm - dif), INF)).astype(np.float32)
                c3 = np.where(fn > eps, neg_fastlog_div(F(1), maxd - eps) + pc, INF).astype(np.float32)
                q1, q2, q3 = q1 + pw * c1, q2 + pw * c2, q3 + pw * c3
                cost = dw * od + pw * np.fmin(np.fmin(q1, q2), q3) + sw * o
            else:
                cost = dw * od + pw * ih + sw * o
            if trace is not None:
                trace["pair"][b] = (typ, t, cgs, cost)
            u = cost < ct[t, OBJECT]
            prev = np.where(q1 < q2, GROUND, OBJECT)
            prev = np.where(q3 < np.fmin(q1, q2), SKY, prev)
            ct[t[u], OBJECT] = cost[u]
            it[t[u], OBJECT] = (b * 3 + prev)[u]

    # ---- back-trace (:843-955)
    secs = []
    v = H - 1
    lg, lo, ls = ct[v]
    typ = OBJECT
    if lg < lo:
        typ = GROUND
    if ls < np.fmin(lg, lo):
        typ = SKY
    idx = v * 3 + typ
    while True:
        e = int(it[idx // 3, idx % 3])
        pvT = e // 3 - 1 if e >= 0 else int(np.trunc(e / 3)) - 1
        vB = pvT + 1
        a_vB, a_vT = np.array([vB]), np.array([v])
        disp = mean(a_vB, a_vT)[0]
        cost = np.fmin(ct[v, typ], F(1e4))
        hh = F(v + 1 - vB)
        mxm = F(MX[v + 1] - MX[vB]) / hh
        mym = F(MY[v + 1] - MY[vB]) / hh
        stype = typ
        if typ == GROUND:
            cls = 0 if F(downsampled_sum(ps[0], a_vB, a_vT)[0]) < F(downsampled_sum(ps[1], a_vB, a_vT)[0]) else 1
        elif typ == SKY or disp < 1.0:
            stype, cls = SKY, 10
        else:
            _, _, _, ic, nic = seg_terms(a_vB, a_vT)
            best, cls = INF, 2
            for c in range(2, 19):
                if c == 10:
                    continue
                cs = (F(0) + (nic[0] if c < 10 else ic[0])) + F(downsampled_sum(ps[c], a_vB, a_vT)[0])
                if best > cs:
                    best, cls = cs, c
        secs.append((stype, vB, v, F(disp), cls, F(cost), F(mxm), F(mym)))
        typ = e % 3
        v = pvT
        idx = pvT * 3 + typ
        if pvT == -1:
            break
    return ct, it, secs


def evaluate(p, joined, seg, gf, ng, ig, vhor, cost_lut, odr, pairwise):
    C = p.cols
    gf, ng, ig = (np.asarray(x, np.float32) for x in (gf, ng, ig))
    out = [evaluate_column(p, c, joined[c], seg[c], gf, ng, ig, int(vhor), np.asarray(cost_lut, np.float32),
                           np.asarray(odr, np.float32), bool(pairwise)) for c in range(C)]
    return (np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), [o[2] for o in out])
