/* is_k_pairwise.hip -- the pairwise column DP (two launches per 64-row tile).  See is_kernels.h. */
#include "is_kernels.h"

/* ====================================================================================== */
/* A7-A9  pairwise DP                                                                      */
/* ====================================================================================== */
/* Everything of a transition INTO a segment starting at vB that does not depend on the lane:
 * the final costs of row vB-1 combined with the transition priors (StixelsKernels.cu:88-199,
 * 687-837).  Built once, when row vB-1 becomes final, from the row's costs, the winning
 * object chain (previous_mean) and the frame's PriorRec; read by every later segment through
 * scalar loads.  64 bytes. */
struct __attribute__((aligned(64))) StepRec {
    float pwmp;      /* pw * fminf(p1, p2) of the ground (vB-1 < vhor) or sky transition       */
    int idx_gs;      /* vB*3 + (p1 < p2 ? GROUND : OBJECT)                    :723-727, 769-773 */
    float g_hi_thr, g_lo_thr;   /* g_prev + epsilon, g_prev - epsilon                 :132, 136 */
    float p1_hi, p1_lo, p1_mid; /* cG + pw * GetPriorCostObjectFromGround, three cases :120-144 */
    float o_hi_thr, o_lo_thr;   /* pm + dif, pm - dif                                 :159, 163 */
    float p2_hi, p2_lo, p2_mid; /* cO + pw * GetPriorCostObjectFromObject, three cases :146-171 */
    float p3_yes, p3_no;        /* cS + pw * GetPriorCostObjectFromSky, fn > eps or not :173-183 */
    /* branch-and-bound (DESIGN.md "Pruning"): running minima over the rows vB' <= vB of the SAME
     * 64-row block (= the StepRecs one phase-2 launch builds) of what a segment starting at vB'
     * adds before its own data / semantic terms: q_o = pw * (smallest of the eight p fields),
     * q_gs = pwmp */
    float q_o, q_gs;
};
static_assert(sizeof(StepRec) == 64, "StepRec must be 64 bytes");
typedef const __attribute__((address_space(4))) StepRec* cstep_t;

struct StepVals { /* register copy of a StepRec, always passed by value */
    float pwmp;
    int idx_gs;
    float g_hi_thr, g_lo_thr, p1_hi, p1_lo, p1_mid, o_hi_thr, o_lo_thr, p2_hi, p2_lo, p2_mid, p3_yes,
        p3_no, q_o, q_gs;
};

__device__ __forceinline__ void store_step(StepRec* dst, const StepVals v) {
    float4* d = reinterpret_cast<float4*>(dst);
    d[0] = make_float4(v.pwmp, __builtin_bit_cast(float, v.idx_gs), v.g_hi_thr, v.g_lo_thr);
    d[1] = make_float4(v.p1_hi, v.p1_lo, v.p1_mid, v.o_hi_thr);
    d[2] = make_float4(v.o_lo_thr, v.p2_hi, v.p2_lo, v.p2_mid);
    d[3] = make_float4(v.p3_yes, v.p3_no, v.q_o, v.q_gs);
}

/* A scalar-loaded value made opaque to the optimiser: without this, LLVM rewrites the selects
 * between neighbouring record fields (p1_hi / p1_lo / p1_mid ...) into a per-lane indexed load
 * from a scratch copy of the record. */
__device__ __forceinline__ float opaque_s(float x) {
    asm volatile("" : "+s"(x));
    return x;
}

__device__ __forceinline__ StepVals sload_step(const StepRec* p) {
    cstep_t q = (cstep_t)p;
    StepVals r;
    r.pwmp = q->pwmp; r.idx_gs = q->idx_gs;
    r.g_hi_thr = q->g_hi_thr; r.g_lo_thr = q->g_lo_thr;
    r.p1_hi = opaque_s(q->p1_hi); r.p1_lo = opaque_s(q->p1_lo); r.p1_mid = opaque_s(q->p1_mid);
    r.o_hi_thr = q->o_hi_thr; r.o_lo_thr = q->o_lo_thr;
    r.p2_hi = opaque_s(q->p2_hi); r.p2_lo = opaque_s(q->p2_lo); r.p2_mid = opaque_s(q->p2_mid);
    r.p3_yes = opaque_s(q->p3_yes); r.p3_no = opaque_s(q->p3_no);
    r.q_o = q->q_o; r.q_gs = q->q_gs;
    return r;
}

/* the two halves of sload_step apart: the loads where they are requested, the pinning where the
 * values are first needed (pinning makes the compiler wait for the loads) */
__device__ __forceinline__ StepVals sload_step_raw(const StepRec* p) {
    cstep_t q = (cstep_t)p;
    StepVals r;
    r.pwmp = q->pwmp; r.idx_gs = q->idx_gs;
    r.g_hi_thr = q->g_hi_thr; r.g_lo_thr = q->g_lo_thr;
    r.p1_hi = q->p1_hi; r.p1_lo = q->p1_lo; r.p1_mid = q->p1_mid;
    r.o_hi_thr = q->o_hi_thr; r.o_lo_thr = q->o_lo_thr;
    r.p2_hi = q->p2_hi; r.p2_lo = q->p2_lo; r.p2_mid = q->p2_mid;
    r.p3_yes = q->p3_yes; r.p3_no = q->p3_no;
    r.q_o = q->q_o; r.q_gs = q->q_gs;
    return r;
}
/* (opaque_u: readfirstlane in front of the pin -- a no-op for a value that lives in an SGPR; when the
 * register allocator has carried a field around the loop in a VGPR (it feeds v_cndmask, which takes
 * one scalar operand only) the plain "+s" pin asks for a VGPR -> SGPR copy, which does not exist) */
__device__ __forceinline__ float opaque_u(float x) {
    return opaque_s(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))));
}
__device__ __forceinline__ void pin_step(StepVals& r) {
    r.p1_hi = opaque_u(r.p1_hi); r.p1_lo = opaque_u(r.p1_lo); r.p1_mid = opaque_u(r.p1_mid);
    r.p2_hi = opaque_u(r.p2_hi); r.p2_lo = opaque_u(r.p2_lo); r.p2_mid = opaque_u(r.p2_mid);
    r.p3_yes = opaque_u(r.p3_yes); r.p3_no = opaque_u(r.p3_no);
}

/* sload_rec with every field pinned into its SGPR right here: the compiler otherwise sinks the
 * loads of the fields towards their first uses and the serial chain of phase 2 waits for scalar
 * memory twice per step (mean first, the class prefixes later) instead of once */
__device__ __forceinline__ RowRec sload_rec_pinned(const RowRec* p) {
    RowRec r = sload_rec(p);
    asm volatile("" : "+s"(r.Fg0), "+s"(r.Fg1), "+s"(r.Fon[0]), "+s"(r.Fon[1]), "+s"(r.Fon[2]),
                      "+s"(r.Fon[3]), "+s"(r.Fon[4]), "+s"(r.Fon[5]), "+s"(r.Fon[6]), "+s"(r.Fon[7]));
    asm volatile("" : "+s"(r.Foi[0]), "+s"(r.Foi[1]), "+s"(r.Foi[2]), "+s"(r.Foi[3]), "+s"(r.Foi[4]),
                      "+s"(r.Foi[5]), "+s"(r.Foi[6]), "+s"(r.Foi[7]), "+s"(r.Fsky), "+s"(r.Fnic));
    asm volatile("" : "+s"(r.G), "+s"(r.K), "+s"(r.S), "+s"(r.V), "+s"(r.MX), "+s"(r.MY),
                      "+s"(r.MX2h), "+s"(r.MX2l), "+s"(r.MY2h), "+s"(r.MY2l));
    return r;
}

__device__ __forceinline__ float readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

/* -is_logf(v) + is_logf(v2) (NegFastLogDiv, :35-38) for TWO argument pairs at once: the lower
 * half of the wave evaluates pair A, the upper half pair B, so the serial chain pays for one
 * logarithm instead of two.  v is a compile-time-like constant whose log is passed in. */
__device__ __forceinline__ void neg_fastlog_div2(float neg_log_va, float v2a, float neg_log_vb,
                                                 float v2b, const double* s_invc,
                                                 const double* s_logc, float* outa, float* outb) {
    const bool upper = threadIdx.x >= 32;
    const float arg = upper ? v2b : v2a;
    const float l = is_logf_t(arg, s_invc, s_logc);
    const float la = readlane_f(l, 0), lb = readlane_f(l, 32);
    *outa = neg_log_va + la;
    *outb = neg_log_vb + lb;
}

/* the same for TWO columns in the two halves of the wave (k_pw_phase2x): the four logarithms run in
 * the four 16-lane quarters, each half then reads its own two through ds_bpermute */
__device__ __forceinline__ void neg_fastlog_div2x(float neg_log_va, float v2a, float neg_log_vb,
                                                  float v2b, const double* s_invc,
                                                  const double* s_logc, float* outa, float* outb) {
    const int lane = threadIdx.x;
    const float arg = (lane & 16) ? v2b : v2a;
    const float l = is_logf_t(arg, s_invc, s_logc);
    const int base = (lane & 32) << 2; /* byte address of lane 0 / 32 for ds_bpermute */
    const float la = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base, __builtin_bit_cast(int, l)));
    const float lb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base + 64, __builtin_bit_cast(int, l)));
    *outa = neg_log_va + la;
    *outb = neg_log_vb + lb;
}

/* StepRec of vB = r + 1 from the final row r.  All inputs are wave-uniform; every lane computes
 * the same values.  (TWOCOL: uniform per half of the wave, see k_pw_phase2x.) */
/* register (SGPR) copy of a PriorRec: loaded at the top of a step together with the step's RowRec,
 * so that the serial chain waits for scalar memory once per step, not twice */
struct PriorVals {
    float pc, g_from, s_from_g, o_from_s, og_hi, og_lo, og_mid, g_prev;
};
__device__ __forceinline__ PriorVals sload_prior(const PriorRec* p) {
    cprior_t q = (cprior_t)p;
    PriorVals v;
    v.pc = q->pc; v.g_from = q->g_from; v.s_from_g = q->s_from_g; v.o_from_s = q->o_from_s;
    v.og_hi = q->og_hi; v.og_lo = q->og_lo; v.og_mid = q->og_mid; v.g_prev = q->g_prev;
    return v;
}

/* rcp_h: RN(1 / (r + 1 - obj_vB)) for the exact-division shortcut of FAST columns (fast_div,
 * is_kernels.h: the prefix difference is 0 or within [2^-84, 2^75] there), 0 = IEEE division */
template <bool HAS_INVALID, bool TWOCOL = false>
__device__ __forceinline__ StepVals make_step(const DevParams& P, float S_r1, float V_r1, float S_ob,
                                             float V_ob, const float* s_odr, const double* s_invc,
                                             const double* s_logc, const PriorVals* pr, int vhor, int r,
                                             float cG, float cO, float cS, int obj_vB, float rcp_h = 0.0f) {
    const int vB = r + 1;
    const float pw = P.pw;
    StepVals st;
    /* previous_mean = ComputeMean(previous_object_vB, previous_vT), :47-60, :675-685; S_r1 / V_r1:
     * the disparity / valid-count prefixes at r + 1, S_ob / V_ob: at obj_vB */
    float pm;
    if (HAS_INVALID) {
        const float valid_dif = V_r1 - V_ob;
        pm = (valid_dif == 0) ? 0 : (S_r1 - S_ob) / valid_dif;
    } else if (rcp_h != 0.0f) {
        pm = fast_div(S_r1 - S_ob, (float)(r + 1 - obj_vB), rcp_h);
    } else {
        pm = (S_r1 - S_ob) / (float)(r + 1 - obj_vB);
    }
    if (pm < 0) pm = 0;
    const float pc = pr->pc;

    if (r < vhor) { /* ground, :687-728 */
        const float prev_cost = pr->g_from;
        const float p1 = cG + pw * prev_cost;
        const float p2 = cO + pw * prev_cost;
        st.pwmp = pw * min_raw(p1, p2); /* (computed values: quiet NaNs only, see min_raw) */
        st.idx_gs = vB * 3 + ((p1 < p2) ? IS_GROUND : IS_OBJECT);
    } else { /* sky, :729-775 */
        const float p1 = cG + pw * pr->s_from_g;
        const float so = (pm < P.epsilon) ? IS_INF : (P.log2c + pc); /* :88-96 */
        const float p2 = cO + pw * so;
        st.pwmp = pw * min_raw(p1, p2);
        st.idx_gs = vB * 3 + ((p1 < p2) ? IS_GROUND : IS_OBJECT);
    }
    /* object from ground, :120-144 */
    st.g_hi_thr = pr->g_prev + P.epsilon;
    st.g_lo_thr = pr->g_prev - P.epsilon;
    st.p1_hi = cG + pw * pr->og_hi;
    st.p1_lo = cG + pw * pr->og_lo;
    st.p1_mid = cG + pw * pr->og_mid;
    /* object from object, :146-171 */
    float base = (r < vhor) ? P.nlog07 : P.log2c;
    base += pc;
    int k = (int)pm;
    k = min(max(k, 0), P.D - 1);
    float dif = s_odr[k];
    if (dif < 0.0f) dif = 0.0f;
    st.o_hi_thr = pm + dif;
    st.o_lo_thr = pm - dif;
    float nl_hi, nl_lo;
    if (TWOCOL)
        neg_fastlog_div2x(P.nlog_pord, P.max_disf - pm - dif, P.nlog_1mpord, st.o_lo_thr, s_invc, s_logc,
                          &nl_hi, &nl_lo);
    else
        neg_fastlog_div2(P.nlog_pord, P.max_disf - pm - dif, P.nlog_1mpord, st.o_lo_thr, s_invc, s_logc,
                         &nl_hi, &nl_lo);
    st.p2_hi = cO + pw * (base + nl_hi);
    st.p2_lo = cO + pw * (base + nl_lo);
    st.p2_mid = cO + pw * IS_INF;
    /* object from sky, :173-183 */
    st.p3_yes = cS + pw * pr->o_from_s;
    st.p3_no = cS + pw * IS_INF;
    return st;
}

struct PairBest {
    float g, o, s;
    int ig, io, is; /* vB*3 + prev type */
};

/* ---- separable block bounds of the ground / sky candidates (lemma L7) --------
 * A ground candidate costs  C(vB, vT) = fl(fl(dw fl(G[vT+1] - G[vB])) + T[vB]) + fl(sw fl(f + n)),
 * f = min_c (F_c[vT+1] - F_c[vB]) over the classes c of the type, n = fl(iw float(N[vT+1] - N[vB])),
 * T = StepRec.pwmp (sky: K instead of G, one class).  As real numbers
 *     dw (G1 - G0) + T + sw (f + iw dN) = min_c [ a_c(vB) + b_c(vT) ],
 *     a_c(vB) = T - dw G[vB] - sw (F_c[vB] + iw N[vB]),   b_c(vT) = dw G[vT+1] + sw (F_c[vT+1] + iw N[vT+1]),
 * and every term of C passes through at most five roundings, so with u = 2^-24
 *     | C - min_c (a_c + b_c) | <= 5.01 u (mu_c(vB) + nu_c(vT)),  mu_c = |T| + dw |G0| + sw (F_c0 + iw N0),
 *                                                                nu_c = dw |G1| + sw (F_c1 + iw N1)
 * for every class c (f <= F_c[vT+1], dN <= N[vT+1]: class values and squared offsets are >= 0 in a
 * column whose pruning is on).  The computed fp32 values a^_c, b^_c lie within 5.01 u mu_c / nu_c of
 * a_c, b_c.  Hence, with slack(x) = 2^-20 x + 2^-90 (16 u: 10.02 u for the two errors, 1 u for the
 * rounding of the subtraction below, the rest covers mu^ >= (1 - u)^4 mu),
 *     alo_c = fl(a^_c - slack(mu^_c)) <= a_c - 10.02 u mu_c,   ahi_c = fl(a^_c + slack(mu^_c)) >= a_c + 10.02 u mu_c
 * and likewise blo_c / bhi_c:   min_c (alo_c(vB) + blo_c(vT)) <= C(vB, vT) <= ahi_c(vB) + bhi_c(vT) for EVERY c.
 * Phase 2 leaves, per 64-row block of candidates (= the rows whose StepRecs one launch builds) and
 * class, the minima of alo_c and ahi_c over the block's rows of the type (l7_row_bounds + a wave
 * minimum; rows of the other type, rows beyond the image and NaN / +inf rows -- whose candidates
 * cost NaN / +inf and never win -- count as +inf).  Phase 1 turns them into a lower bound of every
 * candidate of the block and an upper bound of the block's best one per lane (pw_phase1_body). */
#define IS_L7_REL 0x1p-20f
#define IS_L7_ABS 0x1p-90f
struct L7Row { float lo_g0, lo_g1, lo_s, hi_g0, hi_g1, hi_s; };
__device__ __forceinline__ void l7_ab(float t, float dterm, float fterm, float* lo, float* hi) {
    const float a = (t - dterm) - fterm;
    const float mu = (__builtin_fabsf(t) + __builtin_fabsf(dterm)) + fterm; /* fterm >= 0 */
    const float sl = mu * IS_L7_REL + IS_L7_ABS;
    *lo = a - sl;
    *hi = a + sl;
}
/* bounds of the candidate row vB whose record fields are given (FAST encoding), T = its pwmp;
 * ground_row: vB - 1 < vhor.  `ok`: vB is a candidate row at all (1 <= vB <= H - 1). */
__device__ __forceinline__ L7Row l7_row_bounds(const DevParams& P, float T, float G, float K, float Fg0,
                                               float Fg1, float Fsky, int Fnic, bool ground_row, bool ok) {
    L7Row r;
    const float n = P.iw * (float)Fnic;
    const float dg = P.dw * G, dk = P.dw * K;
    l7_ab(T, dg, P.sw * (Fg0 + n), &r.lo_g0, &r.hi_g0);
    l7_ab(T, dg, P.sw * (Fg1 + n), &r.lo_g1, &r.hi_g1);
    l7_ab(T, dk, P.sw * (Fsky + n), &r.lo_s, &r.hi_s);
    const bool g = ok && ground_row, s = ok && !ground_row;
    /* (NaN bounds -- a NaN or infinite T -- are skipped by the minima: fminf ignores a quiet NaN) */
    r.lo_g0 = g ? r.lo_g0 : IS_INF; r.hi_g0 = g ? r.hi_g0 : IS_INF;
    r.lo_g1 = g ? r.lo_g1 : IS_INF; r.hi_g1 = g ? r.hi_g1 : IS_INF;
    r.lo_s = s ? r.lo_s : IS_INF;   r.hi_s = s ? r.hi_s : IS_INF;
    return r;
}
/* minimum over a group of 2^LOG lanes (NaN-skipping), every lane gets it */
template <int LOG>
__device__ __forceinline__ float l7_group_min(float x) {
    x = __builtin_fminf(x, IS_INF); /* a NaN becomes +inf */
#pragma unroll
    for (int m = (1 << LOG) >> 1; m >= 1; m >>= 1) x = __builtin_fminf(x, __shfl_xor(x, m, 64));
    return x;
}
template <int LOG>
__device__ __forceinline__ L7Row l7_group_min_row(L7Row r) {
    r.lo_g0 = l7_group_min<LOG>(r.lo_g0); r.lo_g1 = l7_group_min<LOG>(r.lo_g1); r.lo_s = l7_group_min<LOG>(r.lo_s);
    r.hi_g0 = l7_group_min<LOG>(r.hi_g0); r.hi_g1 = l7_group_min<LOG>(r.hi_g1); r.hi_s = l7_group_min<LOG>(r.hi_s);
    return r;
}
__device__ __forceinline__ void l7_store(float* dst /* 8 floats, 32-byte aligned */, const L7Row& r) {
    float4* d = reinterpret_cast<float4*>(dst);
    d[0] = make_float4(r.lo_g0, r.lo_g1, r.lo_s, 0.0f);
    d[1] = make_float4(r.hi_g0, r.hi_g1, r.hi_s, 0.0f);
}
/* summary of a block that must never be skipped (generic columns; phase 1 ignores the summaries of
 * a column whose pruning is off anyway) */
__device__ __forceinline__ L7Row l7_never() {
    L7Row r;
    r.lo_g0 = r.lo_g1 = r.lo_s = -IS_INF;
    r.hi_g0 = r.hi_g1 = r.hi_s = IS_INF;
    return r;
}
#define IS_L7_F 24 /* floats per block summary: 8 of lemma L7 (ground / sky) + 16 of lemma L8 (object classes) */

/* ---- lemma L8: separable block bounds of the OBJECT candidates ------------------------------------
 * cost_o(vB, vT) = fl(fl(fl(dw od) + fl(pw mp)) + fl(sw seg_o)),  seg_o = min(on, oi),  on = fl(nic + f_on),
 * oi = fl(ic + f_oi)  (pairwise_step, eval_segment).  By L1 (monotone rounding) with fl(dw od) >= -E1o (L3),
 * mp >= m8(vB) = the smallest of the eight p fields of StepRec(vB), the computed ic >= -E2 (L4) and
 * nic, f >= 0:
 *     cost_o >= min_c [ T8(vB) - sw (F_c[vB] + n_c[vB]) ]  +  [ sw (F_c[vT+1] + n_c[vT+1]) - e_c ]  -  roundings,
 * c over the sixteen object classes, T8 = fl(pw m8), n_c = iw N for the non-instance classes 2..9 and 0
 * for the instance classes 11..18, e_c = E1o (+ sw E2 for an instance class): per class the SAME
 * separable form as lemma L7, with the same slack rule (2^-20 of the magnitudes + 2^-90 on either side,
 * 2^-22 on the sum).  Phase 2 leaves min over the block's rows of alo_c = fl(a^_c - slack) for the
 * sixteen classes; the pre-pass of phase 1 turns them into the lower bound of every object candidate
 * of a lower block, per lane -- what the sticky closure of the object type compares with the lane's
 * best cost (s_lb).  Where round 3 evaluated the block's TOP row (the smallest transition term of the
 * block + the smallest semantic term of the block: a chain of short objects, every split near-optimal,
 * lost a whole block of accumulated path cost), the per-class form loses nothing but the instance term. */
/* Per lane the sixteen alo_c of its candidate row (record `rec`, T8 = pw * m8 of its StepRec; `ok`: the row
 * is a candidate, 1 <= vB <= H - 1), then the minima over a group of 2^LOG lanes (LOG = 4 or 5) as a
 * reduce-scatter, eight classes at a time (the kernels that call this have no registers to spare):
 * every exchange halves the classes a lane is responsible for, 4 + 2 + 1 shuffles + the full-minimum
 * steps of the remaining strides instead of eight full butterflies; the lanes that end up with a class
 * store it.  dst: the group's 16 floats. */
template <int LOG>
__device__ __forceinline__ void l8_group_min_store(const DevParams& P, float T8, const RowRec* rec /* global */,
                                                   bool ok, float* dst) {
    static_assert(LOG == 4 || LOG == 5, "groups of 16 or 32 lanes");
    const int lane = threadIdx.x;
    /* the class prefixes are RE-READ from the record in memory (an L2 hit), eight at a time: the lane's
     * register copy would have to stay alive across the whole walk of the calling kernel, which has no
     * registers to spare (k_pw_phase2x: 124 of 128) */
    asm volatile("" ::: "memory");
    const float4* r4 = reinterpret_cast<const float4*>(rec);
    const float n = P.iw * (float)__float_as_int(r4[4].w); /* Fnic, dword 19 */
#pragma unroll
    for (int bq = 0; bq < 2; bq++) { /* the non-instance classes 2..9, then the instance classes 11..18 */
        float f[8];
        if (bq == 0) { /* Fon[0..7] = dwords 2..9 */
            const float4 a = r4[0], b = r4[1], c = r4[2];
            f[0] = a.z; f[1] = a.w; f[2] = b.x; f[3] = b.y; f[4] = b.z; f[5] = b.w; f[6] = c.x; f[7] = c.y;
        } else { /* Foi[0..7] = dwords 10..17 */
            const float4 a = r4[2], b = r4[3], c = r4[4];
            f[0] = a.z; f[1] = a.w; f[2] = b.x; f[3] = b.y; f[4] = b.z; f[5] = b.w; f[6] = c.x; f[7] = c.y;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float ft = P.sw * (bq == 0 ? f[j] + n : f[j]);
            const float a = T8 - ft;
            const float mu = __builtin_fabsf(T8) + ft;
            const float lo = a - (mu * IS_L7_REL + IS_L7_ABS);
            /* (a NaN bound -- NaN / infinite T8: the candidates cost NaN / +inf -- counts as +inf) */
            v[j] = ok ? __builtin_fminf(lo, IS_INF) : IS_INF;
        }
        int base = 0;
#pragma unroll
        for (int step = 0; step < 3; step++) { /* 8 -> 4 -> 2 -> 1 classes per lane */
            const int m = 1 << (LOG - 1 - step);
            const int half = 4 >> step;
            const bool up = (lane & m) != 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j < half) {
                    const float send = up ? v[j] : v[j + half];
                    const float keep = up ? v[j + half] : v[j];
                    v[j] = __builtin_fminf(keep, __shfl_xor(send, m, 64));
                }
            }
            base += up ? half : 0;
        }
#pragma unroll
        for (int m = 1 << (LOG - 4); m >= 1; m >>= 1) v[0] = __builtin_fminf(v[0], __shfl_xor(v[0], m, 64));
        if ((lane & ((1 << (LOG - 3)) - 1)) == 0) dst[bq * 8 + base] = v[0];
    }
}
/* ---- the same minima on the DPP / permlane path (IS_L78_DPP, groups of 32 lanes) --------------------
 * __shfl_xor is a ds_bpermute (LDS crossbar, an address VGPR, an lgkmcnt wait) and __builtin_fminf
 * re-quiets a value that went through a bitcast (v_max x, x): 3 instructions and a round trip per
 * exchange.  Here an exchange is v_permlane16_swap (distance 16: the rows of TWO values swapped by one
 * instruction, so that one minimum reduces both -- each lands in its own row parity) or a v_mov_b32_dpp
 * (row_ror / quad_perm, bank-masked where the two halves of a pair go different ways) + a bare v_min_f32.
 * The results are the minima of the same numbers: bit-identical to the shuffle version. */
#ifndef IS_L78_DPP
#define IS_L78_DPP 1
#endif
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(src), CTRL, 0xf, BANK, false));
}
#define IS_DPP_ROR4 0x124
#define IS_DPP_ROR8 0x128
#define IS_DPP_ROR12 0x12C
#define IS_DPP_XOR1 0xB1 /* quad_perm:[1,0,3,2] */
#define IS_DPP_XOR2 0x4E /* quad_perm:[2,3,0,1] */
#define IS_DPP_ID 0xE4   /* quad_perm:[0,1,2,3] */
/* rows 0 / 2 of the result: min of a over lanes 0..31 / 32..63, rows 1 / 3: min of b (quiet-NaN-free inputs) */
__device__ __forceinline__ float pair_min32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float x = min_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
    x = min_raw(x, dpp_mov<IS_DPP_ROR8>(x, x));
    x = min_raw(x, dpp_mov<IS_DPP_ROR4>(x, x));
    x = min_raw(x, dpp_mov<IS_DPP_XOR2>(x, x));
    x = min_raw(x, dpp_mov<IS_DPP_XOR1>(x, x));
    return x;
}
/* sixteen values per lane -> lane l of a 32-lane group holds the group's minimum of value (l & 31) >> 1 */
__device__ __forceinline__ float scatter_min32x16(const float (&v)[16], int lane) {
    float w[8], x[4], y[2];
#pragma unroll
    for (int j = 0; j < 8; j++) { /* distance 16: odd rows take value j + 8 */
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 8]), false, false);
        w[j] = min_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { /* distance 8: lanes 8..15 of a row (banks 2, 3) take value j + 4 */
        const float keep = dpp_mov<IS_DPP_ID, 0xC>(w[j], w[j + 4]);
        float recv = dpp_mov<IS_DPP_ROR8, 0x3>(w[j], w[j]);
        recv = dpp_mov<IS_DPP_ROR8, 0xC>(recv, w[j + 4]);
        x[j] = min_raw(keep, recv);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) { /* distance 4: banks 1, 3 take value j + 2 (row_ror:n: lane l reads lane l - n) */
        const float keep = dpp_mov<IS_DPP_ID, 0xA>(x[j], x[j + 2]);
        float recv = dpp_mov<IS_DPP_ROR12, 0x5>(x[j], x[j]);
        recv = dpp_mov<IS_DPP_ROR4, 0xA>(recv, x[j + 2]);
        y[j] = min_raw(keep, recv);
    }
    const bool up = (lane & 2) != 0; /* distance 2 */
    const float keep = up ? y[1] : y[0], send = up ? y[0] : y[1];
    float z = min_raw(keep, dpp_mov<IS_DPP_XOR2>(send, send));
    return min_raw(z, dpp_mov<IS_DPP_XOR1>(z, z));
}
/* l7_row_bounds + l7_group_min_row + l7_store and l8_group_min_store for groups of 32 lanes */
__device__ __forceinline__ void l78_store32(const DevParams& P, float T, float T8, const RowRec* rec /* global */,
                                            bool ground_row, bool ok, float* slot, int lane) {
    const float4* r4 = reinterpret_cast<const float4*>(rec);
    const float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3], c4 = r4[4]; /* Fg0 Fg1 Fon0..7 Foi0..7 Fsky Fnic */
    const float2 gk = *reinterpret_cast<const float2*>((const float*)rec + 20); /* G K */
    L7Row r = l7_row_bounds(P, T, gk.x, gk.y, c0.x, c0.y, c4.z, __float_as_int(c4.w), ground_row, ok);
    r.lo_g0 = min_raw(r.lo_g0, IS_INF); r.lo_g1 = min_raw(r.lo_g1, IS_INF); r.lo_s = min_raw(r.lo_s, IS_INF); /* NaN -> +inf */
    r.hi_g0 = min_raw(r.hi_g0, IS_INF); r.hi_g1 = min_raw(r.hi_g1, IS_INF); r.hi_s = min_raw(r.hi_s, IS_INF);
    const float m0 = pair_min32(r.lo_g0, r.hi_g0), m1 = pair_min32(r.lo_g1, r.hi_g1), m2 = pair_min32(r.lo_s, r.hi_s);
    if ((lane & 15) == 0) /* lane 0 of the group: the lo minima (floats 0..3), lane 16: the hi minima (floats 4..7) */
        reinterpret_cast<float4*>(slot)[(lane >> 4) & 1] = make_float4(m0, m1, m2, 0.0f);
    const float n = P.iw * (float)__float_as_int(c4.w);
    const float f[16] = {c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w, c4.x, c4.y};
    const float t8 = ok ? T8 : IS_INF; /* not a candidate row: inf - inf = NaN -> +inf below */
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float ft = P.sw * (j < 8 ? f[j] + n : f[j]);
        const float a = t8 - ft;
        const float mu = __builtin_fabsf(t8) + ft;
        v[j] = min_raw(a - (mu * IS_L7_REL + IS_L7_ABS), IS_INF);
    }
    const float z = scatter_min32x16(v, lane);
    if ((lane & 1) == 0) slot[8 + ((lane & 31) >> 1)] = z;
}

/* The summaries (lemmas L7, L8) of the bound blocks of the candidate rows vB = tile_lo + base + g + 1, g = this
 * lane's index in its group of 2^LOG lanes (one lane per row, one group per block).  Called at the END of
 * a phase-2 walk, when nothing else is alive: the per-row transition terms -- pwmp = StepRec field 0 and
 * T8 = pw * m8 from the side array t8col -- are read back from memory (this wave stored them during
 * the walk; the caller has waited for its stores), the record fields from the record. */
template <int LOG>
__device__ __forceinline__ void l78_block_summaries(const DevParams& P, const RowRec* rcol, const StepRec* scol,
                                                    const float* t8col, float* bsum_col /* the column's summaries */,
                                                    int tile_lo, int row /* of the tile: base + g */, int vhor) {
    const int H = P.H;
    const int vB = tile_lo + row + 1;
    const bool ok = vB < H;
    const int vBc = min(vB, H - 1);
    const float T = ((const float*)(scol + vBc))[0];
    const float T8 = t8col[vBc];
    const RowRec* rec = rcol + vBc;
    float* const slot = bsum_col + (size_t)(((tile_lo >> 6) * IS_QPT) + (row >> IS_QB_LOG) + 1) * IS_L7_F;
    if (IS_L78_DPP && LOG == 5) {
        l78_store32(P, T, T8, rec, vB - 1 < vhor, ok, slot, (int)(threadIdx.x & 63));
        return;
    }
    const float4* r4 = reinterpret_cast<const float4*>(rec);
    const float4 c0 = r4[0], c4 = r4[4];      /* Fg0 Fg1 . . | Foi6 Foi7 Fsky Fnic */
    const float2 gk = *reinterpret_cast<const float2*>((const float*)rec + 20); /* G K */
    L7Row sum = l7_row_bounds(P, T, gk.x, gk.y, c0.x, c0.y, c4.z, __float_as_int(c4.w), vB - 1 < vhor, ok);
    sum = l7_group_min_row<LOG>(sum);
    if ((threadIdx.x & ((1 << LOG) - 1)) == 0) l7_store(slot, sum);
    l8_group_min_store<LOG>(P, T8, rec, ok, slot + 8);
}
/* (generic columns: never a bound) */
__device__ __forceinline__ void l8_store_never(float* dst, bool writer) {
    if (writer) {
        float4* d = reinterpret_cast<float4*>(dst);
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = make_float4(-IS_INF, -IS_INF, -IS_INF, -IS_INF);
    }
}

/* ---- phase-1 side of lemma L7 ----------------------------------------------------------------
 * Per lane (vT) and class: blo_c / bhi_c from the lane's own record; per block k and type: a lower
 * bound LB_k of every candidate of the block and an upper bound UB_k of its best one,
 *     LB_k = min_c dn(Mlo_c[k] + blo_c),   UB_k = min_c up(Mhi_c[k] + bhi_c),   dn / up(s) = s -+ 2^-22 |s|
 * (4 u |s|: covers the rounding of the sum and of the correction itself).  thr = min_k UB_k is an
 * upper bound of the lane's FINAL cost of the type (some candidate of some block costs no more), so
 * a block with LB_k > thr in every lane holds no candidate that wins or ties anywhere in the tile:
 * its 64 rows are never visited.  In a homogeneous road or sky stretch -- where the sticky bounds
 * never close, every split being a near-optimal candidate -- exactly the block with the stretch's
 * first row survives (tools/l7_study.py). */
#ifndef IS_P1_L7
#define IS_P1_L7 1
#endif
#define IS_P1_L7_WORDS 136 /* LDS words of the exchange: [2][64] threshold keys + 2 x 64-bit masks, padded */
/* LDS words per bound block of a phase-1 launch: 64 object bounds (one per lane) + its 24-float summary + the
 * 8 instance-prefix dwords of the record at its top row */
#define IS_P1_SUM_F (IS_L7_F + 8)
#define IS_P1_BLK_WORDS (64 + IS_P1_SUM_F)
static_assert(IS_QB_LOG >= 3 && IS_QB_LOG <= 5, "bound blocks: 8 .. 32 rows (a phase of k_pw_phase2x holds 32 rows)");
struct L7B { float lo_g0, lo_g1, lo_s, hi_g0, hi_g1, hi_s; };
__device__ __forceinline__ void l7_b(float dterm, float fterm, float* lo, float* hi) {
    const float b = dterm + fterm;
    const float nu = __builtin_fabsf(dterm) + fterm;
    const float sl = nu * IS_L7_REL + IS_L7_ABS;
    *lo = b - sl;
    *hi = b + sl;
}
__device__ __forceinline__ L7B l7_lane_bounds(const DevParams& P, const RowRec& my) {
    L7B b;
    const float n = P.iw * (float)my.Fnic;
    const float dg = P.dw * my.G, dk = P.dw * my.K;
    l7_b(dg, P.sw * (my.Fg0 + n), &b.lo_g0, &b.hi_g0);
    l7_b(dg, P.sw * (my.Fg1 + n), &b.lo_g1, &b.hi_g1);
    l7_b(dk, P.sw * (my.Fsky + n), &b.lo_s, &b.hi_s);
    return b;
}
/* summary of block k (wave-uniform); block 0 = the first segment: a ground candidate
 * with T = pw * first_g and every prefix 0 (:196-199, :481-594) */
__device__ __forceinline__ L7Row l7_block_summary(const DevParams& P, const float* entry /* of block k */, int k) {
    L7Row m;
    if (k == 0) {
        l7_ab(P.pw * P.first_g, 0.0f, 0.0f, &m.lo_g0, &m.hi_g0);
        m.lo_g1 = m.lo_g0; m.hi_g1 = m.hi_g0;
        m.lo_s = m.hi_s = IS_INF;
    } else { /* (the LDS copy of the block's summary: broadcast reads) */
        const float4 a = *reinterpret_cast<const float4*>(entry);
        const float4 c = *reinterpret_cast<const float4*>(entry + 4);
        m.lo_g0 = a.x; m.lo_g1 = a.y; m.lo_s = a.z;
        m.hi_g0 = c.x; m.hi_g1 = c.y; m.hi_s = c.z;
    }
    return m;
}
__device__ __forceinline__ float l7_dn(float s) { return s - __builtin_fabsf(s) * 0x1p-22f; }
__device__ __forceinline__ float l7_up(float s) { return s + __builtin_fabsf(s) * 0x1p-22f; }
__device__ __forceinline__ void l7_combine(const L7Row& m, const L7B& b, float* lbg, float* ubg, float* lbs,
                                           float* ubs) {
    *lbg = __builtin_fminf(l7_dn(m.lo_g0 + b.lo_g0), l7_dn(m.lo_g1 + b.lo_g1));
    *ubg = __builtin_fminf(l7_up(m.hi_g0 + b.hi_g0), l7_up(m.hi_g1 + b.hi_g1));
    *lbs = l7_dn(m.lo_s + b.lo_s);
    *ubs = l7_up(m.hi_s + b.hi_s);
}
/* order-preserving map float -> unsigned (for ds_min_u32); NaNs are never mapped */
__device__ __forceinline__ unsigned l7_key(float x) {
    const unsigned u = __float_as_uint(x);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float l7_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

/* One (vB >= 1, vT) evaluation of the pairwise model for the lane owning vT; `st` is the
 * wave-uniform StepRec of vB.  SKY: vB-1 >= vhor (:729), else ground (:687). */
/* DESC: the caller walks vB downwards, so a candidate of EQUAL cost replaces the best one (the
 * reference's ascending strict < keeps the smallest vB among equal costs) */
template <bool SKY, bool ALL_LANES = false, bool NOGROUND = false, bool DESC = false>
__device__ __forceinline__ void pairwise_step(const DevParams& P, const StepVals st, int vB,
                                              bool live, float od, const SegTerms& t, PairBest& b) {
    /* ALL_LANES (phase 1): every lane with vT < H is live and rows vT >= H are never stored */
    constexpr bool CMPX = IS_CMPX_UPDATE && ALL_LANES;
    static_assert(!DESC || CMPX, "descending walks are whole-wave steps");
    if (SKY) { /* :729-775 */
        const float cost = P.dw * t.sd + st.pwmp + P.sw * t.seg_s;
        if (CMPX && DESC) {
            take_if_le(b.s, b.is, cost, st.idx_gs);
        } else if (CMPX) {
            take_if_less(b.s, b.is, cost, st.idx_gs);
        } else {
            const bool u = live && (cost < b.s);
            b.s = u ? cost : b.s;
            b.is = u ? st.idx_gs : b.is;
        }
    } else if (!NOGROUND) { /* :687-728; NOGROUND: tile at / above the horizon, see unary_step */
        const float cost = P.dw * t.gd + st.pwmp + P.sw * t.seg_g;
        if (CMPX && DESC) {
            take_if_le(b.g, b.ig, cost, st.idx_gs);
        } else if (CMPX) {
            take_if_less(b.g, b.ig, cost, st.idx_gs);
        } else {
            const bool u = live && (cost < b.g);
            b.g = u ? cost : b.g;
            b.ig = u ? st.idx_gs : b.ig;
        }
    }
    /* object, :777-837 */
    const float fn = t.mean;
    const float p1 = (fn > st.g_hi_thr) ? st.p1_hi : ((fn < st.g_lo_thr) ? st.p1_lo : st.p1_mid);
    const float p2 = (fn > st.o_hi_thr) ? st.p2_hi : ((fn < st.o_lo_thr) ? st.p2_lo : st.p2_mid);
    const float p3 = (fn > P.epsilon) ? st.p3_yes : st.p3_no;
    /* (min_raw: the p values are computed fp32 values -- quiet NaNs only --, see is_kernels.h) */
    const float m12 = min_raw(p1, p2);
    const float mp = min_raw(m12, p3);
    const float cost = P.dw * od + P.pw * mp + P.sw * t.seg_o;
    /* min_prev: OBJECT (1), GROUND (0) if p1 < p2, SKY (2) if p3 < fminf(p1, p2), :828-835;
     * vB * 3 on the scalar unit (left to itself the compiler folds it into a 64-bit vector mad) */
    int base3 = vB * 3;
    asm volatile("" : "+s"(base3));
    int sel = (p1 < p2) ? IS_GROUND : IS_OBJECT;
    sel = (p3 < m12) ? IS_SKY : sel;
    const int idx = base3 + sel;
    if (CMPX && DESC) {
        take_if_le_v(b.o, b.io, cost, idx);
    } else if (CMPX) {
        take_if_less_v(b.o, b.io, cost, idx);
    } else {
        const bool u = live && (cost < b.o);
        b.o = u ? cost : b.o;
        b.io = u ? idx : b.io;
    }
}

/* ---- phase 1: warming the L2 for the scalar loads ---------------------------------------------
 * A step takes the record and the StepRec of its vB through scalar loads (SGPR operands), and a
 * wave cannot keep two such record sets in SGPRs, so those loads cannot be issued a step ahead:
 * they were HBM misses on the critical path of every step (~1.5 us under load).  Instead a step
 * now TOUCHES the two lines of the vB that comes IS_P1_TOUCH_AHEAD steps later with one LDS-DMA
 * instruction (two lanes, no VGPR, destination a scratch word nobody reads), issued just before
 * the step's own row prefetch so that it never lengthens a vmcnt wait; when the scalar loads come
 * they hit the L2. */
#ifndef IS_P1_WIN_MIN_COLS
#define IS_P1_WIN_MIN_COLS 4096
#endif
#ifndef IS_P1_WIN_WAVES
#define IS_P1_WIN_WAVES 4
#endif
#ifndef IS_P2_GATHER_MASKED
#define IS_P2_GATHER_MASKED 1
#endif
#ifndef IS_P1_TOUCH_AHEAD
#define IS_P1_TOUCH_AHEAD 2
#endif
/* (measured on MI355X, batch 64, ms per step of the pairwise DP: no touch 25.35, 1 step ahead
 * 24.2, 2 steps 22.96, 3 steps 24.3, 4 steps 23.2, 8 steps 24.2; touching the lutT row of the step
 * after next as well +0.35; the same trick on the record / priors of phase 2 +0.6: both removed) */
__device__ __forceinline__ void touch_step(const RowRec* rcol, const StepRec* scol, int vB, int lane,
                                           unsigned lds_scratch) {
    if (lane < 2)
        dma_dword(lane == 0 ? (const float*)(rcol + vB) : (const float*)(scol + vB), lds_scratch);
}
/* the four vB of the next ground / sky round: records in lanes 0..3, StepRecs in lanes 4..7 */
__device__ __forceinline__ void touch_round(const RowRec* rcol, const StepRec* scol, int vB0, int nw,
                                            int lo, int lane, unsigned lds_scratch) {
    if (lane < 8) {
        const int v = max(vB0 - (lane & 3) * nw, lo);
        dma_dword(lane < 4 ? (const float*)(rcol + v) : (const float*)(scol + v), lds_scratch);
    }
}

/* The pairwise DP of one 64-row tile is split over two launches (per tile, bottom-up):
 *
 *  phase 1  k_pw_phase1: segments that START in earlier tiles (vB <= tile_lo).  Their
 *           predecessor rows are final (StepRec written by earlier launches, read with scalar
 *           loads), so all (vB, vT) pairs are independent: same structure, occupancy and issue
 *           bound as the unary kernel.  Writes the merged partial minima of the tile.
 *  phase 2  k_pw_phase2: the 64x64 diagonal block, where step vB needs the final row vB-1 of
 *           the same tile: one wavefront per column walks the 63 steps; the finished row is
 *           broadcast with v_readlane, its StepRec is computed uniformly and published.
 *
 * The serial chain of the reference (rows x __syncthreads, StixelsKernels.cu:600-603) is thus
 * confined to phase 2, 1/16 of the pair evaluations at 1024 rows. */
/* ---- phase 1, FAST columns: the vB record as DPP operands ---------------------------------------
 * The record of vB is wave-uniform.  Through scalar loads its 32 dwords are free SGPR operands,
 * but the loads cannot be issued ahead (a wave cannot hold two records in SGPRs) and every step
 * waits for scalar memory.  Here the record is fetched by two ordinary vector loads instead --
 * lane l of every 16-lane row holds dwords (l & 15) and 16 + (l & 15) -- TWO steps ahead
 * (vmcnt-tracked, six VGPRs in flight), and a subtraction takes dword k as a DPP operand:
 *     v_subrev_f32_dpp d, R, mine row_newbcast:k        d = mine - R[lane k of my row]
 * A DPP instruction issues in 4 cycles instead of 2 (tools/ubench/dpp_rate.hip), ~60 cycles more
 * per step, and frees 32 SGPRs (the kernel had 72 SGPR spills).  The StepRec stays in SGPRs and is
 * double buffered (16 + 16), loaded one step ahead.  `asm volatile`: a DPP read of a lane that
 * EXEC has switched off returns 0, so these instructions must never sink into divergent code. */
#ifndef IS_P1_DPP
#define IS_P1_DPP 1
#endif
#ifndef IS_P1_DPP_INV
#define IS_P1_DPP_INV 1 /* the DPP / scalar-operand step also with an invalid-disparity value (mean_valid_fast) */
#endif
#ifndef IS_P1_TILE0_DIRECT
#define IS_P1_TILE0_DIRECT 1
#endif
#ifndef IS_P1_SREC_LATE
#define IS_P1_SREC_LATE 1 /* StepRec of the next step: loaded at the end of the step, pinned (= waited for) at its use */
#endif
#ifndef IS_P1_SREC
#define IS_P1_SREC 1 /* class-prefix half of the vB record as scalar operands (see eval_segment_mix) */
#endif
#ifdef IS_ABL_P1PHASES
/* debug build only: s_memtime cycles of wave 0 of every phase-1 workgroup in prologue / walk /
 * waiting for the other waves / merge, plus the number of full and ground-sky rounds of wave 0 */
__device__ unsigned long long g_p1phase[16 * 8]; /* [min(tile, 15)][counter] */
#define ISP1_MARK(k)                                                              \
    do {                                                                          \
        const unsigned long long now__ = __builtin_readcyclecounter();            \
        if (threadIdx.x == 0) atomicAdd(&g_p1phase[min(tile, 15) * 8 + (k)], now__ - t_p1); \
        t_p1 = now__;                                                             \
    } while (0)
#define ISP1_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_p1phase[min(tile, 15) * 8 + (k)], 1ull); } while (0)
extern "C" void isk_debug_p1phases(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p1phase), sizeof(g_p1phase));
    if (reset) {
        unsigned long long z[16 * 8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p1phase), z, sizeof(z));
    }
}
#else
#define ISP1_MARK(k)
#define ISP1_COUNT(k)
#endif

template <bool FAST, bool HAS_INVALID, int NR, bool WIN>
__device__ __forceinline__ void pw_phase1_body(const DevParams& P, char* smem, int colg, int tile,
                                               const RowRec* __restrict__ recs,
                                               const float* __restrict__ lutT,
                                               const StepRec* __restrict__ steps,
                                               const float* __restrict__ rcp, int vhor,
                                               int split, int nsplit,
                                               const PruneRec* __restrict__ prec,
                                               float* __restrict__ part_cost,
                                               int* __restrict__ part_idx,
                                               unsigned long long* __restrict__ counters,
                                               const float* __restrict__ joined,
                                               const float* __restrict__ cost_T,
                                               const float* __restrict__ blksum) {
    const int H = P.H, D = P.D;
    /* the vT-side tile: all D lutT columns of the 64 rows, or the fn window [win_lo, win_lo + IS_P1_WIN) of
     * wide tables (is_device.h) */
    constexpr bool windowed = WIN; /* (the launch picks the instantiation: tile < P.win_tiles) */
    const int win_w = windowed ? IS_P1_WIN : D;
    const int DP = win_w + 1;
    float* s_tile = (float*)smem;             /* [64][win_w + 1] */
    float* s_rcp = s_tile + IS_TILE * DP;     /* [H+1]     */
    float* s_scr = s_rcp + ((H + 1 + 3) & ~3); /* [8 per wave] landing area of the L2-warming DMAs */
    const int tid = threadIdx.x, lane = tid & 63;
    /* `nsplit` workgroups share the vB range of one (column, tile): together they behave like
     * one workgroup of nsplit * nwl waves (few columns = small batches: more of the chip works
     * on the latency chain); their partial minima are merged by phase 2 */
    const int nwl = blockDim.x >> 6;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = nwl * nsplit;
    const int w = split * nwl + wl;
    const int tile_lo = tile * IS_TILE;
#ifdef IS_ABL_P1HOT /* timing-only ablation for batches whose columns are all EQUAL (tools/p1_hot_probe.py): phase 1 reads the
                     * tables of column (colg mod 8) -- the same values, from lines that stay in the XCD's L2 -- so that the
                     * difference to the product build is what phase 1 pays for memory latency and bandwidth */
    const int colr = colg & 7;
#else
    const int colr = colg;
#endif
    const RowRec* rcol = recs + (size_t)colr * (H + 1);
    const float* lcol = lutT + (size_t)colr * (H + 1) * D;
    const StepRec* scol = steps + (size_t)colr * H;

#if IS_P1_TILE0_DIRECT
    /* Tile 0 has ONE candidate per row, the first segment (vB = 0, :481-594), and it belongs to wave 0
     * of split 0: that wave computes it with its operands straight from global memory (1/h, the two
     * lutT values of the lane) and writes the partial minima in the form the merge below would; no
     * tile staging, no barrier, the other waves leave at once.  (The full prologue + merge made the
     * tile-0 launch as long as a launch with work: 0.24 ms per 64 frames.) */
    if (tile == 0) {
        if (wl != 0) return;
        const int vT = lane;
        const int vTc = min(vT, H - 1);
        const bool live = vT < H;
        float cg = IS_INF, co = IS_INF;
        int ig = -1, io = IS_OBJECT;
        if (split == 0) {
            const RowRec my = load_rec(rcol + vTc + 1);
            const RowRec rb = sload_rec(rcol);
            const int h = vTc + 1;
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, rcp[min(h, H)], D, P.iw, rcp);
            const float od = lcol[(size_t)min(vT + 1, H) * D + (unsigned)t.fni] - lcol[(unsigned)t.fni];
            const bool below = vT <= vhor;
            const float cost_g = P.dw * t.gd + P.pw * P.first_g + P.sw * t.seg_g;
            const bool ug = live && below && (cost_g < IS_INF); /* (a +inf candidate keeps the initial index) */
            cg = ug ? cost_g : cg;
            ig = ug ? IS_GROUND : ig;
            const float prior = below ? P.first_o_below : P.first_o_above;
            const float cost = P.dw * od + P.pw * prior + P.sw * t.seg_o;
            co = (live && cost < IS_INF) ? cost : co;
        }
        const size_t o = (((size_t)colg * nsplit + split) * 3) * 64 + lane;
        part_cost[o] = cg;            part_idx[o] = ig;
        part_cost[o + 64] = co;       part_idx[o + 64] = io;
        part_cost[o + 128] = IS_INF;  part_idx[o + 128] = -1;
        if (counters != nullptr && lane == 0 && split == 0) atomicAdd(counters + IS_CNT_P1_FULL, 1ull);
        return;
    }
#endif

#ifdef IS_ABL_P1PHASES
    unsigned long long t_p1 = __builtin_readcyclecounter();
#endif
    const int vB_last = min(tile_lo, H - 1);
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    /* What the pre-pass of the branch-and-bound reads -- the separable summaries of the lower bound
     * blocks (lemmas L7, L8: 24 floats per block) -- is requested HERE, in front of the tile staging, and
     * lands in LDS with it: one memory round trip for the whole prologue (round 3 chased a record and a
     * StepRec per block top through scalar loads, one after the other: the 0.25 ms floor of a launch is
     * a chain of such round trips). */
    const int NLB = tile * IS_QPT + 1; /* bound blocks 0 .. tile * IS_QPT of this tile's candidates */
    float* s_lb = s_scr + 8 * nwl + IS_P1_L7_WORDS;   /* [NLB][64 lanes] object block bounds      */
    float* s_sum = s_lb + NLB * 64;                   /* [NLB][IS_P1_SUM_F]: 24 summary floats + the instance
                                                       * prefixes (dwords 24..31) of the record at the block's TOP row */
    const float* bcol = blksum + (size_t)colr * (P.ntiles * IS_QPT + 1) * IS_L7_F;
    constexpr int PRE_N = 2; /* floats per thread: NLB * 32 <= PRE_N * nthreads at 1024 rows */
    float pre_v[PRE_N];
    const int nthr = (int)blockDim.x;
    const bool pre_regs = FAST && IS_PRUNE && NLB * IS_P1_SUM_F <= PRE_N * nthr;
    auto pre_load = [&](int i) -> float { /* entry i of s_sum */
        const int k = i / IS_P1_SUM_F, j = i - k * IS_P1_SUM_F;
        if (j >= IS_L7_F) return ((const float*)(rcol + (size_t)k * IS_QB))[24 + (j - IS_L7_F)];
        return k >= 1 ? bcol[k * IS_L7_F + j] : 0.0f; /* (the summary of block 0 is analytic) */
    };
    if (pre_regs) {
#pragma unroll
        for (int j = 0; j < PRE_N; j++) {
            const int i = tid + j * nthr;
            pre_v[j] = i < NLB * IS_P1_SUM_F ? pre_load(i) : 0.0f;
        }
    }
    const int win_lo = windowed ? __builtin_amdgcn_readfirstlane(P.win_lo[(size_t)colr * P.ntiles + tile]) : 0;
    if (windowed) stage_window_and_rcp(s_tile, s_rcp, lcol, rcp, tile_lo, H, D, win_lo, tid, (int)blockDim.x);
    else stage_tile_and_rcp(s_tile, s_rcp, lcol, rcp, tile_lo, H, D, tid, (int)blockDim.x);
    const RowRec my = load_rec(rcol + vTc + 1);
    if (pre_regs) {
#pragma unroll
        for (int j = 0; j < PRE_N; j++) {
            const int i = tid + j * nthr;
            if (i < NLB * IS_P1_SUM_F) s_sum[i] = pre_v[j];
        }
    } else if (FAST && IS_PRUNE) { /* tall frames: a plain loop */
        for (int i = tid; i < NLB * IS_P1_SUM_F; i += nthr) s_sum[i] = pre_load(i);
    }
    const float* my_tile = s_tile + lane * DP;
    const bool live = vT < H;
    /* (round 6, one frame per call: the record / StepRec lines of the wave's first 3 or 5 steps touched here, in front
     * of the barrier and the pre-pass -- 1.563-1.588 against 1.574-1.576 ms per frame: nothing, as in round 4) */
    /* lemma L7: per-lane thresholds (order-preserving keys, +inf) and the two survive masks */
    unsigned* s_thr = (unsigned*)(s_scr + 8 * nwl); /* [2][64] + [4] */
    if (tid < 2 * 64) s_thr[tid] = 0xFF800000u;
    if (tid < 4) s_thr[2 * 64 + tid] = 0u;
    __syncthreads();
    ISP1_MARK(0);

    PairBest b;
    b.g = b.o = b.s = IS_INF;
    b.ig = b.is = -1;
    b.io = IS_OBJECT; /* :592 */
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)lcol, 0, (H + 1) * D * (int)sizeof(float), 0x00020000);
    const int lane4 = lane * 4;
    /* WIN: ONE load fetches the window columns of a vB row (lane l: staged column l & 31 -- both ends of a segment
     * read the SAME lutT column, so the row needs what the tile holds); classic: NR loads, the whole row */
    constexpr int NRW = windowed ? 1 : NR;
    const int lane4r = windowed ? IS_WIN_COL(win_lo, lane & (IS_P1_WIN - 1)) * 4 : lane4;
    const unsigned scr = lds_addr(s_scr + 8 * wl);
    /* lutT[vT + 1][fni] of this lane: from the staged tile / window; outside the window from global memory */
    int n_winmiss = 0; /* steps in which some lane read outside the window (evaluation counters) */
    auto vt_value = [&](int fni) -> float {
        if (!windowed) return my_tile[fni];
        const int fo = IS_WIN_FIND(win_lo, fni);
        const bool inw = fo >= 0;
        float v = my_tile[inw ? fo : 0];
        if (__builtin_amdgcn_ballot_w64(!inw) != 0ull) { /* (never without a window: fni < D) */
            n_winmiss++;
            if (!inw) v = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(lrsrc, ((vTc + 1) * D + fni) * 4, 0, 0));
        }
        return v;
    };
    /* lutT[vT + 1][fni] - lutT[vB][fni]: `row` = the registers of row vB */
    auto od_value = [&](const LutRow<NRW>& row, int fni, int vB_row) -> float {
        if (!windowed) return my_tile[fni] - pick_lut<NRW>(row, fni);
        const int fo = IS_WIN_FIND(win_lo, fni);
        const bool in_t = fo >= 0; /* the tile's window = the row registers' */
        const bool in_r = in_t;
        float vt = my_tile[in_t ? fo : 0];
        float vb = pick_lut<NRW>(row, in_t ? fo : 0);
        if (__builtin_amdgcn_ballot_w64(!in_t) != 0ull) {
            n_winmiss++;
            if (!in_t) vt = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(lrsrc, ((vTc + 1) * D + fni) * 4, 0, 0));
            if (!in_r) vb = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(lrsrc, (vB_row * D + fni) * 4, 0, 0));
        }
        return vt - vb;
    };
    if (FAST && IS_PRUNE) {
        /* FAST columns: vB downwards with the exact branch-and-bound of DESIGN.md "Pruning".  The
         * candidates of a tile fall into BLOCKS: block k >= 1 = the vB values 64 (k-1) + 1 .. 64 k
         * whose StepRecs phase 2 of tile k - 1 built, block 0 = the first segment vB = 0.  A
         * candidate vB' of block k with vB' <= vB costs lane vT at least
         *     fl(fl(q[vB] - E1) + fl(sw * m(vB, vT)))      (same block as vB)
         *     fl(fl(q[64 k] - E1) + fl(sw * m(64 k, vT)))  (lower block, from its TOP row)
         * m = the monotone semantic term of the type (seg_g / seg_s / seg_o_lower_bound), q =
         * StepRec.q_o / q_gs = the smallest transition term pw * min_prev of the block's rows up to
         * vB (phase 2 restarts the running minimum with every tile, so q[64 k] is the minimum of
         * the whole block), E1 / E2 the slacks of PruneRec: the monotonicity argument of the unary
         * kernel, block by block.  The accumulated path cost inside q grows with vB, so a bound
         * that may use the block's own q instead of the minimum over ALL lower rows closes a type a
         * few rows below the region of the lane instead of vT / 16 rows further down.
         *
         * Pre-pass: the waves evaluate the semantic terms at the tops of all lower blocks (at most
         * tile values of vB, no LUT access, no update) and leave, per type and block k, the
         * smallest bound of the blocks BELOW k in LDS (s_lb); the walk closes a type for good when
         * the bound of its own block and that entry both exceed the lane's best cost. */
        /* (round 4: the BOUND blocks are IS_QB = 32 rows, half a tile -- see is_device.h; "64 k" above
         * reads IS_QB k.  Only the OBJECT type keeps the per-block array s_lb; the ground / sky types
         * close when their own block's bound holds and the separable summaries (lemma L7) leave no
         * surviving block below.) */
        const unsigned long long dead = ~__builtin_amdgcn_ballot_w64(live);
        const unsigned long long gdead = dead | __builtin_amdgcn_ballot_w64(my.G == IS_INF);
        /* Separable block bounds of the ground / sky candidates (lemma L7, see l7_row_bounds): bit k
         * of mask_g / mask_s = block k may hold the winning ground / sky candidate of some lane of
         * this tile; the walk below never enters the other blocks.  Off (all ones) when the column's
         * pruning is off or the column has more blocks than the masks have bits (then the ground /
         * sky types only close in block 0: slower, never wrong). */
        unsigned long long mask_g = ~0ull, mask_s = ~0ull;
        L7B bl7;
        bool l7 = false;
        {
            cprune_t pq0 = (cprune_t)prec;
            const float pE1o = pq0->E1o;
            const float pE1gs = __builtin_fmaxf(pq0->E1g, pq0->E1s);
            const bool pnog = IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor;
            l7 = IS_P1_L7 && (pE1gs < IS_INF) && NLB <= 63;
            if (l7) {
                bl7 = l7_lane_bounds(P, my);
                float thr_g = IS_INF, thr_s = IS_INF;
                for (int k = wl; k < NLB; k += nwl) {
                    const L7Row m = l7_block_summary(P, s_sum + (size_t)k * IS_P1_SUM_F, k);
                    float lbg, ubg, lbs, ubs;
                    l7_combine(m, bl7, &lbg, &ubg, &lbs, &ubs);
                    thr_g = __builtin_fminf(thr_g, ubg);
                    thr_s = __builtin_fminf(thr_s, ubs);
                }
                if (!pnog && thr_g == thr_g) atomicMin(&s_thr[lane], l7_key(thr_g));
                if (thr_s == thr_s) atomicMin(&s_thr[64 + lane], l7_key(thr_s));
            }
            /* lemma L8: the lower bound of every OBJECT candidate of the lower block k, per lane */
            float bo[16]; /* sw (F_c[vT+1] + n_c[vT+1]) - e_c - slack, per object class */
            {
                const float n1 = P.iw * (float)my.Fnic;
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const float ft = P.sw * (c < IS_N_ON ? my.Fon[c] + n1 : my.Foi[c - IS_N_ON]);
                    bo[c] = (ft - pE1o) - ((ft + pE1o) * IS_L7_REL + IS_L7_ABS);
                }
            }
            /* the instance classes also keep lemma L4's term of the block's TOP row: the computed
             * ic(vB', vT) >= ic(top, vT) - 2 E2 for every vB' <= top (a sum of squared deviations only grows
             * with the segment), so oi' >= fl(fl(ic_top - 3 E2) + f_oi') like in seg_o_lower_bound; it is
             * what makes a candidate that starts inside an instance object and ends in the sky above it
             * expensive.  ic_top: the expression of eval_segment on the staged prefixes of the top row. */
            const float E2x3 = 3.0f * pq0->E2;
            const float t0 = P.pw * __builtin_fminf(P.first_o_below, P.first_o_above); /* block 0: the first segment, :189-199 */
            for (int k = wl; k < NLB - 1; k += nwl) {
                float lb_n, lb_i; /* non-instance / instance classes */
                if (k == 0) {
                    const float a0 = t0 - (__builtin_fabsf(t0) * IS_L7_REL + IS_L7_ABS); /* every prefix is 0 at vB = 0 */
                    lb_n = a0 + bo[0];
                    lb_i = a0 + bo[IS_N_ON];
#pragma unroll
                    for (int c = 1; c < IS_N_ON; c++) {
                        lb_n = __builtin_fminf(lb_n, a0 + bo[c]);
                        lb_i = __builtin_fminf(lb_i, a0 + bo[IS_N_ON + c]);
                    }
                } else {
                    const float4* m4 = reinterpret_cast<const float4*>(s_sum + k * IS_P1_SUM_F + 8);
                    const float4 m0 = m4[0], m1 = m4[1], m2 = m4[2], m3 = m4[3];
                    const float mm[16] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w,
                                          m2.x, m2.y, m2.z, m2.w, m3.x, m3.y, m3.z, m3.w};
                    lb_n = mm[0] + bo[0];
                    lb_i = mm[IS_N_ON] + bo[IS_N_ON];
#pragma unroll
                    for (int c = 1; c < IS_N_ON; c++) {
                        lb_n = __builtin_fminf(lb_n, mm[c] + bo[c]);
                        lb_i = __builtin_fminf(lb_i, mm[IS_N_ON + c] + bo[IS_N_ON + c]);
                    }
                }
                { /* ic(top, vT), top = IS_QB k (FAST encoding, eval_segment's expression and order) */
                    const float4 t0i = *reinterpret_cast<const float4*>(s_sum + k * IS_P1_SUM_F + IS_L7_F);     /* MX MY MX2h MX2l */
                    const float2 t1i = *reinterpret_cast<const float2*>(s_sum + k * IS_P1_SUM_F + IS_L7_F + 4); /* MY2h MY2l */
                    const int h = vTc + 1 - k * IS_QB;
                    const float rh = s_rcp[h];
                    const float meanx = my.MX - t0i.x, meany = my.MY - t0i.y;
                    const float meanx2 = (my.MX2h - t0i.z) + (my.MX2l - t0i.w);
                    const float meany2 = (my.MY2h - t1i.x) + (my.MY2l - t1i.y);
                    const float ic = P.iw * (meanx2 - fast_div(meanx * meanx, (float)h, rh) + meany2 -
                                             fast_div(meany * meany, (float)h, rh));
                    lb_i = lb_i + P.sw * (ic - E2x3);
                }
                float lb_o = l7_dn(__builtin_fminf(lb_n, lb_i));
                /* a NaN bound (inf - inf with the pruning switched off, NaN inputs) must never
                 * close anything: -inf */
                lb_o = (lb_o == lb_o) ? lb_o : -IS_INF;
                s_lb[k * 64 + lane] = lb_o;
            }
            __syncthreads();
            if (tid < 64) { /* entry k <- smallest bound of the blocks below k */
                float* q = s_lb + lane;
                float run = IS_INF;
                for (int k = 0; k < NLB - 1; k++) {
                    const float v = q[k * 64];
                    q[k * 64] = run;
                    run = (v < run) ? v : run;
                }
                q[(NLB - 1) * 64] = run;
            }
            if (l7) { /* a block survives when its lower bound reaches the threshold in some lane */
                const float thr_g = l7_unkey(s_thr[lane]), thr_s = l7_unkey(s_thr[64 + lane]);
                unsigned long long mg = 0ull, ms = 0ull;
                for (int k = wl; k < NLB; k += nwl) {
                    const L7Row m = l7_block_summary(P, s_sum + (size_t)k * IS_P1_SUM_F, k);
                    float lbg, ubg, lbs, ubs;
                    l7_combine(m, bl7, &lbg, &ubg, &lbs, &ubs);
                    const bool has_g = !pnog && (m.lo_g0 < IS_INF || m.lo_g1 < IS_INF);
                    const bool has_s = m.lo_s < IS_INF;
                    if (has_g && (__builtin_amdgcn_ballot_w64(lbg > thr_g) | gdead) != ~0ull) mg |= 1ull << k;
                    if (has_s && (__builtin_amdgcn_ballot_w64(lbs > thr_s) | dead) != ~0ull) ms |= 1ull << k;
                }
                if (lane == 0) {
                    atomicOr(&s_thr[2 * 64 + 0], (unsigned)mg);
                    atomicOr(&s_thr[2 * 64 + 1], (unsigned)(mg >> 32));
                    atomicOr(&s_thr[2 * 64 + 2], (unsigned)ms);
                    atomicOr(&s_thr[2 * 64 + 3], (unsigned)(ms >> 32));
                }
            }
            __syncthreads();
            if (l7) {
                mask_g = (unsigned long long)__builtin_amdgcn_readfirstlane(s_thr[2 * 64 + 0]) |
                         ((unsigned long long)__builtin_amdgcn_readfirstlane(s_thr[2 * 64 + 1]) << 32);
                mask_s = (unsigned long long)__builtin_amdgcn_readfirstlane(s_thr[2 * 64 + 2]) |
                         ((unsigned long long)__builtin_amdgcn_readfirstlane(s_thr[2 * 64 + 3]) << 32);
            }
            ISP1_MARK(6); /* (debug build: the pre-pass) */
        }
        if (w <= vB_last) {
            cprune_t pq = (cprune_t)prec;
            const float E1o = pq->E1o, E2 = 3.0f * pq->E2; /* see seg_o_lower_bound */
            const float E1gs = __builtin_fmaxf(pq->E1g, pq->E1s);
            const bool nog = IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor;
            int vB = vB_last - (vB_last - w) % nw; /* the wave's largest vB */
            LutRow<NRW> next_row;
            load_lut_row<NRW>(next_row, lrsrc, lcol, vB, D, lane4r);
            /* The bounds are sticky per type (a bound that holds at vB holds at every smaller vB:
             * the class minima only grow, the running minima q only grow, the best cost cannot
             * change any more).  The OBJECT bound -- minimum over the object classes only --
             * explodes as soon as the segment leaves an object, long before the ground / sky bound
             * of a homogeneous road or sky region does (there every split is a near-optimal
             * candidate, nothing can be pruned).  Once the object type is closed the remaining
             * candidates are ground / sky ones: two class differences, no instance term, no mean,
             * no LUT access -- a fifth of the instructions of a full step; they are evaluated FOUR
             * vB at a time so that one scalar-load latency covers four steps. */
            bool done = false, o_closed = false;
            int n_full = 0, n_gs = 0; /* evaluation counters (wave-uniform) */
            /* lower bound of seg_o(vB', vT) for every vB' below the last fully evaluated step; before
             * the first one: on >= 0, and X = fl(fl(ic - 3 E2) + f_oi) >= -4 E2 (1 + u) since the computed
             * ic >= -E2 and f_oi >= 0 (`E2` here is 3 E2 of PruneRec: -2 * that = -6 E2) */
            float lbseg = -2.0f * E2;
#define IS_P1_NEXT_ROW() load_lut_row<NRW>(next_row, lrsrc, lcol, max(vB - nw, 0), D, lane4r)
            /* (with an invalid-disparity value the step used to keep the scalar-load form: valid-count operands and
             * an IEEE division per step took the DPP form to 90+ VGPRs.  Round 5: the mean of a FAST column through
             * mean_valid_fast -- a table read and the exact-division shortcut -- so both run the same step) */
            constexpr bool USE_DPP = IS_P1_DPP && (!HAS_INVALID || IS_P1_DPP_INV);
            /* record of vB (c_*), of vB - nw (n_*): vector loads, two steps ahead; StepRec one step
             * ahead in a second set of SGPRs */
            const int l15 = lane & 15;
            float c_r0 = 0.0f, c_r1 = 0.0f, n_r0 = 0.0f, n_r1 = 0.0f;
            StepVals st_next;
            isk_f16v S; /* IS_P1_SREC: the class-prefix half of the record of vB as scalars (eval_segment_mix) */
            if (USE_DPP) { /* (requesting these before the tile staging of the prologue measured 2.5 % slower) */
                const float* q0 = (const float*)(rcol + vB);
                const float* q1 = (const float*)(rcol + max(vB - nw, 0));
                if (!IS_P1_SREC) { c_r0 = q0[l15]; n_r0 = q1[l15]; }
                c_r1 = q0[16 + l15];
                n_r1 = q1[16 + l15];
                st_next = sload_step(scol + vB);
                if (IS_P1_SREC) srec_request(S, rcol + vB);
            }
#define IS_P1_DRAIN() if (USE_DPP && IS_P1_SREC) srec_arrived(S) /* a request is in flight after every step */
            /* IS_P1_SREC: the StepRec and the scalar half of the record of the next step are requested
             * when the step has read its own for the last time (the bounds): one set of registers
             * each instead of two, and the loads have the first half of the next step to arrive */
#define IS_P1_REQUEST_NEXT(last_use)                                                               \
            if (USE_DPP && IS_P1_SREC) {                                                           \
                const int vn = max(vB - nw, 0);                                                    \
                const StepRec* sn = scol + vn;                                                     \
                asm volatile("" : "+s"(sn) : "s"(__builtin_amdgcn_readfirstlane((int)(last_use))));    \
                st_next = IS_P1_SREC_LATE ? sload_step_raw(sn) : sload_step(sn);                   \
                srec_request_next(S, rcol + vn);                                                   \
            }
#define IS_P1_STEP(SKY, NOG)                                                                       \
            ISP1_COUNT(4);                                                                         \
            const LutRow<NRW> row = next_row;                                                       \
            if (IS_P1_TOUCH_AHEAD > 0)                                                             \
                touch_step(rcol, scol, max(vB - IS_P1_TOUCH_AHEAD * nw, 0), lane, scr);            \
            IS_P1_NEXT_ROW();                                                                      \
            const int h = vTc + 1 - vB;                                                            \
            const int kb = (vB + IS_QB - 1) >> IS_QB_LOG; /* the bound block of vB (>= 1) */        \
            const float* lbp = s_lb + kb * 64 + lane;                                              \
            bool ok_o = false, ok_x = (NOG);                                                       \
            {                                                                                      \
                n_full++;                                                                          \
                SegTerms t;                                                                        \
                StepVals st;                                                                       \
                float od;                                                                          \
                if (USE_DPP) {                                                                     \
                    const float r0 = c_r0, r1 = c_r1;                                              \
                    st = st_next;                                                                  \
                    c_r0 = n_r0; c_r1 = n_r1;                                                      \
                    {                                                                              \
                        const float* q2 = (const float*)(rcol + max(vB - 2 * nw, 0));              \
                        if (!IS_P1_SREC) n_r0 = q2[l15];                                           \
                        n_r1 = q2[16 + l15];                                                       \
                    }                                                                              \
                    constexpr int WANT = (SKY) ? IS_WANT_SKY : ((NOG) ? 0 : IS_WANT_GROUND);       \
                    if (IS_P1_SREC) {                                                              \
                        if (IS_P1_SREC_LATE) pin_step(st);                                         \
                        srec_arrived(S);                                                           \
                        t = eval_segment_mix<HAS_INVALID, WANT>(my, S, r1, (float)h, s_rcp[h], D, P.iw, s_rcp); \
                    } else {                                                                       \
                        t = eval_segment_dpp<HAS_INVALID, WANT>(my, r0, r1, (float)h, s_rcp[h], D, P.iw, s_rcp); \
                    }                                                                              \
                    od = od_value(row, t.fni, vB);                                \
                    if (!IS_P1_SREC) {                                                             \
                        /* the next StepRec: requested only now -- every LDS wait is an lgkmcnt(0) \
                         * wait and would wait for this scalar load too (SMEM returns out of order) */ \
                        const StepRec* sn = scol + max(vB - nw, 0);                                \
                        asm volatile("" : "+s"(sn) : "v"(od));                                     \
                        st_next = sload_step(sn);                                                  \
                    }                                                                              \
                } else {                                                                           \
                    const RowRec rb = sload_rec(rcol + vB);                                        \
                    st = sload_step(scol + vB);                                                    \
                    t = eval_segment<true, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);      \
                    od = od_value(row, t.fni, vB);                                \
                }                                                                                  \
                pairwise_step<SKY, true, NOG, true>(P, st, vB, live, od, t, b);                    \
                const float lb_o = min_raw((st.q_o - E1o) + P.sw * seg_o_lower_bound(t, E2), lbp[0]); \
                ok_o = (__builtin_amdgcn_ballot_w64(lb_o > b.o) | dead) == ~0ull;                  \
                if (SKY) {                                                                         \
                    const float lb_s = (st.q_gs - E1gs) + P.sw * t.seg_s;                          \
                    ok_x = (__builtin_amdgcn_ballot_w64(lb_s > b.s) | dead) == ~0ull &&            \
                           IS_L7_NONE_BELOW(mask_s, kb);                                           \
                } else if (!(NOG)) {                                                               \
                    const float lb_g = (st.q_gs - E1gs) + P.sw * t.seg_g;                          \
                    ok_x = (__builtin_amdgcn_ballot_w64(lb_g > b.g) | gdead) == ~0ull &&           \
                           IS_L7_NONE_BELOW(mask_g, kb);                                           \
                }                                                                                  \
            }                                                                                      \
            if (!(NOG)) ok_x = ok_x || IS_L7_NONE_LE((SKY) ? mask_s : mask_g, kb)
            /* ground / sky candidates of up to four vB (vB, vB - nw, ...) >= lo; closes `x_closed`
             * when the bound of the last one holds; leaves vB at the next unvisited value */
#define IS_P1_GS4(SKY, lo, x_dead, x_closed)                                                       \
            {                                                                                      \
                const int n_here = min(4, (vB - (lo)) / nw + 1);                                   \
                ISP1_COUNT(5);                                                                     \
                n_gs += n_here;                                                                    \
                if (IS_P1_TOUCH_AHEAD > 0) touch_round(rcol, scol, vB - 4 * nw, nw, (lo), lane, scr); \
                float c_f[4], c_cost[4];                                                           \
                int c_idx[4];                                                                      \
                float c_q[4];                                                                      \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                    \
                    const int vj = max(vB - j * nw, (lo)); /* clamped: the loads are unconditional */ \
                    crec_t rq = (crec_t)(rcol + vj);                                               \
                    cstep_t sq = (cstep_t)(scol + vj);                                             \
                    const float nic = P.iw * (float)(my.Fnic - rq->Fnic);                          \
                    c_f[j] = (SKY) ? (my.Fsky - rq->Fsky)                                          \
                                   : __builtin_fminf(my.Fg0 - rq->Fg0, my.Fg1 - rq->Fg1);         \
                    const float data = (SKY) ? (my.K - rq->K) : (my.G - rq->G);                    \
                    c_f[j] += nic; /* f + nic: also monotone (lemma L2), and part of the cost */   \
                    c_cost[j] = P.dw * data + sq->pwmp + P.sw * c_f[j];                            \
                    c_idx[j] = sq->idx_gs;                                                         \
                    c_q[j] = sq->q_gs;                                                             \
                }                                                                                  \
                float f_last = c_f[0], q_last = c_q[0];                                            \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                    \
                    if (j < n_here) {                                                              \
                        if (SKY) take_if_le(b.s, b.is, c_cost[j], c_idx[j]);                       \
                        else take_if_le(b.g, b.ig, c_cost[j], c_idx[j]);                           \
                        f_last = c_f[j]; q_last = c_q[j];                                          \
                    }                                                                              \
                }                                                                                  \
                const int vb_l = vB - (n_here - 1) * nw; /* the lowest vB just evaluated */      \
                const float lb_x = (q_last - E1gs) + P.sw * f_last;                                \
                if ((__builtin_amdgcn_ballot_w64(lb_x > ((SKY) ? b.s : b.g)) | (x_dead)) == ~0ull && \
                    IS_L7_NONE_BELOW((SKY) ? mask_s : mask_g, (vb_l + IS_QB - 1) >> IS_QB_LOG))    \
                    x_closed = true;                                                               \
                vB -= n_here * nw;                                                                 \
            }
            /* lemma L7: no block at or below block k holds a possible winner */
#define IS_L7_NONE_LE(mask, k) (((mask) & ((2ull << min((k), 62)) - 1ull)) == 0ull)
            /* ... strictly below block k */
#define IS_L7_NONE_BELOW(mask, k) (((mask) & ((1ull << min((k), 63)) - 1ull)) == 0ull)
            /* in front of a ground / sky round: close the type when no surviving block is left at or
             * below the block of vB; jump over the blocks that cannot hold the winner to this wave's
             * largest vB in the next surviving block (block 0 = the first segment, vB = 0) */
#define IS_P1_L7_SKIP(mask, x_closed)                                                              \
            {                                                                                      \
                const int kb = min((vB + IS_QB - 1) >> IS_QB_LOG, 62);                             \
                const unsigned long long at_or_below = (mask) & ((2ull << kb) - 1ull);             \
                if (at_or_below == 0ull) { x_closed = true; break; }                               \
                if (!(((mask) >> kb) & 1ull)) {                                                    \
                    const int top = (63 - __builtin_clzll(at_or_below)) << IS_QB_LOG;              \
                    int dd = (top - w) % nw;                                                       \
                    dd = dd < 0 ? dd + nw : dd;                                                    \
                    vB = top - dd;                                                                 \
                    continue;                                                                      \
                }                                                                                  \
            }
            const int sky_lo = max(vhor + 1, 1);
            for (; vB >= sky_lo; vB -= nw) { /* sky range: vB - 1 >= vhor */
                IS_P1_STEP(true, false);
                IS_P1_REQUEST_NEXT(ok_x);
                if (ok_o && ok_x) { done = true; break; }
                if (ok_o) { o_closed = true; vB -= nw; break; }
            }
            IS_P1_DRAIN();
            if (!done && o_closed) {
                /* a tile with a sky range lies above the horizon: no ground candidates, and the
                 * first segment's object candidate is closed too -- only sky candidates are left */
                bool s_closed = false;
                while (vB >= sky_lo && !s_closed) {
                    IS_P1_L7_SKIP(mask_s, s_closed)
                    IS_P1_GS4(true, sky_lo, dead, s_closed)
                }
                done = true;
            }
            if (!done && nog) {
                for (; vB >= 1; vB -= nw) { /* ground range, ground candidates are +inf */
                    IS_P1_STEP(false, true);
                    IS_P1_REQUEST_NEXT(ok_o);
                    if (ok_o) { done = true; break; }
                }
                IS_P1_DRAIN();
            } else if (!done) {
                for (; vB >= 1; vB -= nw) { /* ground range: vB - 1 < vhor */
                    IS_P1_STEP(false, false);
                    IS_P1_REQUEST_NEXT(ok_x);
                    if (ok_o && ok_x) { done = true; break; }
                    if (ok_o) { o_closed = true; vB -= nw; break; }
                }
                IS_P1_DRAIN();
                if (!done && o_closed) {
                    bool g_closed = false;
                    while (vB >= 1 && !g_closed) {
                        IS_P1_L7_SKIP(mask_g, g_closed)
                        IS_P1_GS4(false, 1, gdead, g_closed)
                    }
                    if (g_closed) done = true; /* else vB <= 0: the first segment is still to come */
                }
            }
#undef IS_P1_STEP
#undef IS_P1_NEXT_ROW
#undef IS_P1_GS4
#undef IS_P1_L7_SKIP
#undef IS_L7_NONE_LE
#undef IS_L7_NONE_BELOW
#undef IS_P1_DRAIN
#undef IS_P1_REQUEST_NEXT
            if (!done && vB == 0) { /* first segment, :481-594 */
                n_full++;
                const RowRec rb = sload_rec(rcol);
                const int h = vTc + 1;
                const SegTerms t = eval_segment<true, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);
                const float od = vt_value(t.fni) - lcol[(unsigned)t.fni];
                const bool below = vT <= vhor;
                const float cost_g = P.dw * t.gd + P.pw * P.first_g + P.sw * t.seg_g;
                const bool ug = live && below && (cost_g <= b.g);
                b.g = ug ? cost_g : b.g;
                b.ig = ug ? IS_GROUND : b.ig;
                const float prior = below ? P.first_o_below : P.first_o_above;
                const float cost = P.dw * od + P.pw * prior + P.sw * t.seg_o;
                const bool uo = live && (cost <= b.o);
                b.o = uo ? cost : b.o;
                b.io = uo ? IS_OBJECT : b.io;
            }
            if (counters != nullptr && lane == 0) {
                atomicAdd(counters + IS_CNT_P1_FULL, (unsigned long long)n_full);
                atomicAdd(counters + IS_CNT_P1_GS, (unsigned long long)n_gs);
                atomicAdd(counters + IS_CNT_P1_WINMISS, (unsigned long long)n_winmiss);
                unsigned long long* ct = counters + IS_CNT_TILE0 + 3 * min(tile, 63);
                atomicAdd(ct + 0, (unsigned long long)n_full);
                atomicAdd(ct + 1, (unsigned long long)n_winmiss);
                atomicAdd(ct + 2, (unsigned long long)n_gs);
            }
        }
    } else {
        int vB = w;
        LutRow<NRW> next_row;
        if (vB <= vB_last) load_lut_row<NRW>(next_row, lrsrc, lcol, vB == 0 ? min(nw, H) : vB, D, lane4r);
        if (vB == 0) { /* first segment, :481-594 */
            const RowRec rb = sload_rec(rcol);
            const int h = vTc + 1;
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);
            const float od = vt_value(t.fni) - lcol[(unsigned)t.fni];
            const bool below = vT <= vhor;
            const float cost_g = P.dw * t.gd + P.pw * P.first_g + P.sw * t.seg_g;
            const bool ug = live && below && (cost_g < b.g);
            b.g = ug ? cost_g : b.g;
            b.ig = ug ? IS_GROUND : b.ig;
            const float prior = below ? P.first_o_below : P.first_o_above;
            const float cost = P.dw * od + P.pw * prior + P.sw * t.seg_o;
            b.o = (live && cost < b.o) ? cost : b.o;
            vB += nw;
        }
        if (IS_SKIP_GROUND_ABOVE_HORIZON && tile_lo >= vhor) {
            for (; vB <= min(vhor, vB_last); vB += nw) { /* ground range, ground candidate = +inf */
                const RowRec rb = sload_rec(rcol + vB);
                const StepVals st = sload_step(scol + vB);
                const LutRow<NRW> row = next_row;
                load_lut_row<NRW>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4r);
                const int h = vTc + 1 - vB;
                const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);
                const float od = od_value(row, t.fni, vB);
                pairwise_step<false, true, true>(P, st, vB, live, od, t, b);
            }
        }
        for (; vB <= min(vhor, vB_last); vB += nw) { /* ground range: vB-1 < vhor */
            const RowRec rb = sload_rec(rcol + vB);
            const StepVals st = sload_step(scol + vB);
            const LutRow<NRW> row = next_row;
            load_lut_row<NRW>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4r);
            const int h = vTc + 1 - vB;
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);
            const float od = od_value(row, t.fni, vB);
            pairwise_step<false, true>(P, st, vB, live, od, t, b);
        }
        for (; vB <= vB_last; vB += nw) { /* sky range */
            const RowRec rb = sload_rec(rcol + vB);
            const StepVals st = sload_step(scol + vB);
            const LutRow<NRW> row = next_row;
            load_lut_row<NRW>(next_row, lrsrc, lcol, min(vB + nw, H), D, lane4r);
            const int h = vTc + 1 - vB;
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)h, s_rcp[h], D, P.iw, s_rcp);
            const float od = od_value(row, t.fni, vB);
            pairwise_step<true, true>(P, st, vB, live, od, t, b);
        }
    }
    /* merge the waves: min cost, ties -> smallest vB (a finite cost always has a real index).
     * Every L2-warming DMA must have landed before this wave can end: its LDS target would
     * otherwise be written after the workgroup's LDS has been handed to another one. */
    wait_vmcnt<0>();
    ISP1_MARK(1);
    __syncthreads();
    ISP1_MARK(2);
    float* m_cost = (float*)smem;               /* [nwl][3][64] (aliases the tile) */
    int* m_idx = (int*)(m_cost + nwl * 3 * 64); /* [nwl][3][64] */
    m_cost[(wl * 3 + 0) * 64 + lane] = b.g; m_idx[(wl * 3 + 0) * 64 + lane] = b.ig;
    m_cost[(wl * 3 + 1) * 64 + lane] = b.o; m_idx[(wl * 3 + 1) * 64 + lane] = b.io;
    m_cost[(wl * 3 + 2) * 64 + lane] = b.s; m_idx[(wl * 3 + 2) * 64 + lane] = b.is;
    __syncthreads();
    if (tid < 3 * 64) {
        const int type = tid >> 6;
        float c = m_cost[(0 * 3 + type) * 64 + lane];
        int ix = m_idx[(0 * 3 + type) * 64 + lane];
        for (int ww = 1; ww < nwl; ww++) {
            const float c2 = m_cost[(ww * 3 + type) * 64 + lane];
            const int ix2 = m_idx[(ww * 3 + type) * 64 + lane];
            const bool take = (c2 < c) || (c2 == c && c2 < IS_INF && (ix2 / 3) < (ix / 3));
            if (take) { c = c2; ix = ix2; }
        }
        /* a row without a finite candidate keeps the initial index (a descending walk records
         * +inf candidates, the reference's strict < never does; :592 for the object type) */
        if (!(c < IS_INF)) ix = (type == IS_OBJECT) ? IS_OBJECT : -1;
        const size_t o = (((size_t)colg * nsplit + split) * 3 + type) * 64 + lane;
        part_cost[o] = c;
        part_idx[o] = ix;
    }
    ISP1_MARK(3);
}

#ifndef ISP1_OCC_INV
#define ISP1_OCC_INV (IS_P1_DPP_INV ? 6 : 8) /* with an invalid-disparity value (8: the scalar-load form of the step, 64 VGPRs) */
#endif
#ifndef ISP1_OCC
#define ISP1_OCC 6 /* waves per SIMD phase 1 is compiled for: 80 VGPRs, no spills (8: 64 VGPRs + spills) */
#endif
#ifndef ISP1_OCC_WIN
#define ISP1_OCC_WIN ISP1_OCC /* the windowed instantiation (4-wave workgroups, 8-26 KB of LDS) */
#endif
template <bool HAS_INVALID, int NR, bool WIN = false>
__global__ __launch_bounds__(IS_UNARY_WAVES * 64, HAS_INVALID ? ISP1_OCC_INV : (WIN ? ISP1_OCC_WIN : ISP1_OCC)) void k_pw_phase1(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const StepRec* __restrict__ steps, const float* __restrict__ rcp,
    const int* __restrict__ vhor_arr, const int* __restrict__ col_flags,
    const PruneRec* __restrict__ prune, float* __restrict__ part_cost, int* __restrict__ part_idx,
    unsigned long long* __restrict__ counters, const float* __restrict__ joined,
    const float* __restrict__ cost_T, const float* __restrict__ blksum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = col_base + (int)(blockIdx.x / (unsigned)nsplit);
    const int split = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (unsigned)nsplit));
    if (colg >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
        pw_phase1_body<true, HAS_INVALID, NR, WIN>(P, smem, colg, tile, recs, lutT, steps, rcp, vhor, split,
                                              nsplit, prune + colg, part_cost, part_idx, counters, joined, cost_T,
                                              blksum);
    else
        pw_phase1_body<false, HAS_INVALID, NR, WIN>(P, smem, colg, tile, recs, lutT, steps, rcp, vhor, split,
                                               nsplit, prune + colg, part_cost, part_idx, counters, joined, cost_T,
                                               blksum);
}

/* ---- phase 2: the fn window --------------------------------------------------------------
 * Every segment of the diagonal block lies inside the tile, so its mean disparity lies between
 * the smallest and the largest (valid) disparity of the tile's rows, and floor(mean) -- the lutT
 * column it reads -- inside [floor(dmin) - 1, floor(dmax) + 1] (the +-1 covers the rounding of the
 * tree-summed prefixes the mean is computed from).  On real and synthetic scenes that window is a
 * few columns wide (a road ramp moves by ~12 disparities over 64 rows, an object by ~1), so the
 * wave stages only lutT[tile_lo .. tile_lo+64][lo .. lo+W) in LDS -- 65 x W floats, W <= 32 --
 * instead of gathering two values per lane and step from the 33 KB tile in global memory (64 + 4
 * cache lines per step: that gather throughput, at 5.3 TB/s of L2 misses, bounded the kernel).
 * Exactness does not rest on the window: a lane whose floor(mean) falls outside reads global
 * memory as before. */
#ifndef ISP2_WMAX
#define ISP2_WMAX 16
#endif
#define ISP2_WS (ISP2_WMAX + 1) /* row stride: lanes reading one column of 64 rows hit 32 banks */
#define ISP2_ROWS (IS_TILE + 1)

#ifdef IS_ABL_P2PHASES
/* debug build only: s_memtime cycles of the sections of a phase-2 step, summed over all waves */
__device__ unsigned long long g_p2phase[8];
#define ISP2_MARK_INIT() unsigned long long t_p2 = __builtin_readcyclecounter()
#define ISP2_MARK(k)                                                              \
    do {                                                                          \
        const unsigned long long now__ = __builtin_readcyclecounter();            \
        acc_p2[k] += now__ - t_p2;                                                \
        t_p2 = now__;                                                             \
    } while (0)
extern "C" void isk_debug_p2phases(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2phase), sizeof(g_p2phase));
    if (reset) {
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p2phase), z, sizeof(z));
    }
}
#else
#define ISP2_MARK_INIT()
#define ISP2_MARK(k)
#endif

__device__ __forceinline__ float wave_min_f(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = __builtin_fminf(x, __shfl_xor(x, m, 64));
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x = __builtin_fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}

template <bool FAST, bool HAS_INVALID>
__device__ __forceinline__ void pw_phase2_body(const DevParams& P, char* smem, int colg, int tile,
                                               const RowRec* __restrict__ recs,
                                               const float* __restrict__ lutT,
                                               const float* __restrict__ joined,
                                               const PriorRec* __restrict__ priors,
                                               const float* __restrict__ odr,
                                               const float* __restrict__ rcp,
                                               const float* __restrict__ sv_arr, int vhor,
                                               int nsplit, const float* __restrict__ part_cost,
                                               const int* __restrict__ part_idx,
                                               StepRec* __restrict__ steps,
                                               float* __restrict__ cost_table,
                                               int32_t* __restrict__ index_table,
                                               float* __restrict__ blksum, float* __restrict__ t8row) {
    const int H = P.H, D = P.D;
    const int lane = threadIdx.x;
    double* s_invc = (double*)smem;                    /* [32] */
    double* s_logc = s_invc + IS_LOG_TABLE_SIZE;       /* [32] */
    float* s_odr = (float*)(s_logc + IS_LOG_TABLE_SIZE); /* [D] */
    float* s_rcp = s_odr + D;                          /* [IS_TILE+1] RN(1/h), h <= 64 */
    float* s_win = s_rcp + (IS_TILE + 1);              /* [65][ISP2_WS] lutT window */
    const int tile_lo = tile * IS_TILE;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const PriorRec* pcol = priors + (size_t)(colg / P.C) * H;
    StepRec* scol = steps + (size_t)colg * H;
    const float* sv = sv_arr + (size_t)colg * 2 * (H + 1);
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);

    /* the window [lo, lo + W) of lutT columns the tile's segments can select */
    int lo, W;
    {
        const float d = joined[(size_t)colg * H + vTc];
        const bool ok = (vT < H) && !(HAS_INVALID && d == P.invalid);
        const float dmin = wave_min_f(ok ? d : IS_INF);
        const float dmax = wave_max_f(ok ? d : -IS_INF);
        /* any junk (NaN, no valid row, values outside [0, D)) still yields a window inside the
         * table; lanes that leave it take the global path */
        int l = (int)__builtin_fminf(__builtin_fmaxf(dmin, 1.0f), (float)D) - 1;
        l = min(max(l, 0), D - 1);
        int h = (int)__builtin_fminf(__builtin_fmaxf(dmax, 0.0f), (float)(D - 1)) + 1;
        h = min(max(h, l), min(D - 1, l + ISP2_WMAX - 1));
        lo = __builtin_amdgcn_readfirstlane(l);
        W = __builtin_amdgcn_readfirstlane(h - l + 1);
    }
    /* rows tile_lo .. tile_lo + 64 (vB side: row r, vT side: row vT + 1) x window columns.  A
     * wave instruction covers 64 / Wp rows of Wp = 2^k >= W columns; every load is issued before
     * the first LDS store (fully unrolled: one memory round trip for the whole window). */
    {
        int lg = 0;
        while ((1 << lg) < W) lg++;
        lg = __builtin_amdgcn_readfirstlane(lg);
        const int f = lane & ((1 << lg) - 1);
        const int j0 = lane >> lg, dj = 64 >> lg;
        constexpr int NL = (ISP2_ROWS * ISP2_WMAX + 63) / 64; /* loads per lane at the widest window */
        float tmp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + k * dj;
            tmp[k] = (j < ISP2_ROWS && f < W) ? lcol[(size_t)min(tile_lo + j, H) * D + lo + f] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + k * dj;
            if (j < ISP2_ROWS && f < W) s_win[j * ISP2_WS + f] = tmp[k];
        }
    }
    if (lane == 0) is_log_tables(s_invc, s_logc);
    for (int i = lane; i < D; i += 64) s_odr[i] = odr[i];
    for (int i = lane; i <= IS_TILE; i += 64) s_rcp[i] = rcp[min(i, H)];
    const RowRec my = load_rec(rcol + vTc + 1);
    const float* my_row = lcol + (size_t)(vTc + 1) * D;
    const float* my_win = s_win + (vTc + 1 - tile_lo) * ISP2_WS;
    PairBest b; /* partial minima of phase 1 (its nsplit workgroups merged: min cost, then smallest vB) */
    {
        const size_t o = (size_t)colg * nsplit * 3 * 64 + lane;
        b.g = part_cost[o]; b.ig = part_idx[o];
        b.o = part_cost[o + 64]; b.io = part_idx[o + 64];
        b.s = part_cost[o + 128]; b.is = part_idx[o + 128];
        for (int sp = 1; sp < nsplit; sp++) {
            const size_t q = o + (size_t)sp * 3 * 64;
            float c2 = part_cost[q]; int i2 = part_idx[q];
            if ((c2 < b.g) || (c2 == b.g && c2 < IS_INF && (i2 / 3) < (b.ig / 3))) { b.g = c2; b.ig = i2; }
            c2 = part_cost[q + 64]; i2 = part_idx[q + 64];
            if ((c2 < b.o) || (c2 == b.o && c2 < IS_INF && (i2 / 3) < (b.io / 3))) { b.o = c2; b.io = i2; }
            c2 = part_cost[q + 128]; i2 = part_idx[q + 128];
            if ((c2 < b.s) || (c2 == b.s && c2 < IS_INF && (i2 / 3) < (b.is / 3))) { b.s = c2; b.is = i2; }
        }
    }
    __syncthreads();

    const int n_rows = min(IS_TILE, H - tile_lo);
    StepVals st;
    st.pwmp = IS_INF; st.idx_gs = -1;
    st.g_hi_thr = st.g_lo_thr = st.p1_hi = st.p1_lo = st.p1_mid = st.o_hi_thr = st.o_lo_thr = 0.0f;
    st.p2_hi = st.p2_lo = st.p2_mid = st.p3_yes = st.p3_no = 0.0f;
    /* running minima of the transition terms (StepRec.q_o / q_gs) over the rows of THIS tile:
     * phase 1 bounds block by block (see pw_phase1_body) */
    float q_o = IS_INF, q_gs = IS_INF;
    st.q_o = q_o; st.q_gs = q_gs;
    int ob_cached = -1;
    float S_obc = 0.0f, V_obc = 0.0f;
#ifdef IS_ABL_P2PHASES
    unsigned long long acc_p2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ISP2_MARK_INIT();
    ISP2_MARK(0); /* prologue */
    for (int s = 0; s < n_rows; s++) {
        const int r = tile_lo + s; /* row that becomes final in this step */
        PriorVals pv = sload_prior(pcol + min(r + 1, H - 1));
        if (s > 0) { /* segments starting at vB = r: lanes vT >= r */
            const RowRec rb = sload_rec_pinned(rcol + r);
            pv.pc = opaque_s(pv.pc); /* the record and the priors arrive behind one wait */
            const int hc = max(vTc + 1 - r, 1);
            const bool live = (vT < H) && (vT >= r);
            ISP2_MARK(1); /* scalar loads */
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)hc, s_rcp[hc], D, P.iw, s_rcp);
            ISP2_MARK(2); /* eval_segment */
            const int fo = t.fni - lo;
            const bool inwin = (unsigned)fo < (unsigned)W;
            const int foc = inwin ? fo : 0;
            float od = my_win[foc] - s_win[s * ISP2_WS + foc];
            if (__builtin_amdgcn_ballot_w64(live && !inwin) != 0ull) { /* outside the window: rare */
                /* (only the lanes outside fetch: with every lane gathering, a step of this kind pulled up to
                 * 128 lines -- 16 KB -- for the few values it needed) */
                if (IS_P2_GATHER_MASKED ? (live && !inwin) : true) {
                    const float og = my_row[(unsigned)t.fni] - (lcol + (size_t)r * D)[(unsigned)t.fni];
                    od = inwin ? od : og;
                }
            }
            ISP2_MARK(3); /* LUT values */
            if (r - 1 < vhor)
                pairwise_step<false>(P, st, r, live, od, t, b);
            else
                pairwise_step<true>(P, st, r, live, od, t, b);
            ISP2_MARK(4); /* pairwise_step */
        }
        /* lane s holds the final values of row r: broadcast, derive the StepRec of vB = r+1.
         * (Also for the last row of the image, whose StepRec nobody reads: an unconditional
         * assignment keeps `st` in place -- a guarded one costs 16 register moves per step.) */
        {
            const float cG = readlane_f(b.g, s), cO = readlane_f(b.o, s), cS = readlane_f(b.s, s);
            const int ob = __builtin_amdgcn_readlane(b.io, s) / 3; /* start of the best object chain */
            /* the disparity / valid-count prefixes at r + 1 and at ob: lane k holds the record of
             * prefix index tile_lo + k + 1; an index at or below the tile comes from memory through a
             * scalar load, repeated only when the chain's start changes (inside an object: never) */
            const float S_r1 = readlane_f(my.S, s), V_r1 = HAS_INVALID ? readlane_f(my.V, s) : 0.0f;
            float S_ob, V_ob = 0.0f;
            if (ob > tile_lo) {
                S_ob = readlane_f(my.S, ob - 1 - tile_lo);
                if (HAS_INVALID) V_ob = readlane_f(my.V, ob - 1 - tile_lo);
            } else {
                if (ob != ob_cached) {
                    typedef const __attribute__((address_space(4))) float* cflt_t;
                    S_obc = *(cflt_t)(sv + ob);
                    if (HAS_INVALID) V_obc = *(cflt_t)(sv + (H + 1) + ob);
                    ob_cached = ob;
                }
                S_ob = S_obc; V_ob = V_obc;
            }
            float rcp_h = 0.0f; /* RN(1 / h) of the chain's height for the exact-division shortcut */
            if (FAST && !HAS_INVALID) {
                typedef const __attribute__((address_space(4))) float* cflt_t;
                rcp_h = *(cflt_t)(rcp + min(max(r + 1 - ob, 1), H));
            }
            ISP2_MARK(5); /* broadcasts */
            st = make_step<HAS_INVALID>(P, S_r1, V_r1, S_ob, V_ob, s_odr, s_invc, s_logc, &pv, vhor, r,
                                        cG, cO, cS, ob, rcp_h);
            ISP2_MARK(6); /* make_step */
            /* fminf skips NaN fields: a candidate that selects one costs NaN and never wins */
            const float m8 = min_raw(min3_raw(st.p1_hi, st.p1_lo, st.p1_mid),
                                     min3_raw(min3_raw(st.p2_hi, st.p2_lo, st.p2_mid), st.p3_yes, st.p3_no));
            if ((s & (IS_QB - 1)) == 0) q_o = q_gs = IS_INF; /* vB = r + 1 starts a bound block */
            const float pwm8 = P.pw * m8;
            q_o = min_raw(q_o, pwm8);
            q_gs = min_raw(q_gs, st.pwmp);
            st.q_o = q_o; st.q_gs = q_gs;
            if (lane == 0 && r + 1 < H) {
                store_step(scol + r + 1, st);
                t8row[(size_t)colg * H + r + 1] = pwm8; /* (lemma L8: the row's own transition bound) */
            }
            ISP2_MARK(7); /* running minima + store */
        }
    }
#ifdef IS_ABL_P2PHASES
    if (lane == 0)
        for (int k = 0; k < 8; k++) atomicAdd(&g_p2phase[k], acc_p2[k]);
#endif
    if (vT < H) {
        const size_t o = ((size_t)colg * H + vT) * 3;
        cost_table[o + 0] = b.g; cost_table[o + 1] = b.o; cost_table[o + 2] = b.s;
        index_table[o + 0] = b.ig; index_table[o + 1] = b.io; index_table[o + 2] = b.is;
    }
    { /* summaries of the bound blocks of the candidate rows vB = tile_lo + 1 .. tile_lo + 64 (lemmas L7, L8) */
        float* const bcolw = blksum + (size_t)colg * (P.ntiles * IS_QPT + 1) * IS_L7_F;
        if (FAST) {
            __builtin_amdgcn_s_waitcnt(0); /* this wave's StepRec / T8 stores have reached the L2 */
            l78_block_summaries<IS_QB_LOG>(P, rcol, scol, t8row + (size_t)colg * H, bcolw, tile_lo, lane, vhor);
        } else {
            float* const slot = bcolw + (size_t)(tile * IS_QPT + (lane >> IS_QB_LOG) + 1) * IS_L7_F;
            const bool writer = (lane & (IS_QB - 1)) == 0;
            if (writer) l7_store(slot, l7_never());
            l8_store_never(slot + 8, writer);
        }
    }
}

#ifndef ISP2_OCC
#define ISP2_OCC 6 /* waves per SIMD the kernel is compiled for */
#endif
template <bool HAS_INVALID>
__global__ __launch_bounds__(64, ISP2_OCC) void k_pw_phase2(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ joined, const PriorRec* __restrict__ priors,
    const float* __restrict__ odr, const float* __restrict__ rcp,
    const float* __restrict__ sv_arr, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const float* __restrict__ part_cost,
    const int* __restrict__ part_idx, StepRec* __restrict__ steps, float* __restrict__ cost_table,
    int32_t* __restrict__ index_table, float* __restrict__ blksum, float* __restrict__ t8row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = col_base + blockIdx.x;
    if (colg >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
        pw_phase2_body<true, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,
                                          nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
    else
        pw_phase2_body<false, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,
                                           nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
}

/* ====================================================================================== */
/* phase 2, TWO columns per wavefront (k_pw_phase2x, large batches)                         */
/* ====================================================================================== */
/* In k_pw_phase2 a step issues for 64 lanes although only 64 - s of them hold a live row, and
 * ~45 % of its instructions (make_step, the broadcasts, the running minima) are wave-uniform work
 * that every lane repeats.  Here a wave walks the diagonal blocks of two neighbouring columns X, Y
 * of one image at once, X in lanes 0..31, Y in lanes 32..63, a lane owning vT = base + (lane & 31):
 *
 *   phase L  base = tile_lo:      steps s = 0..31, candidates vB = tile_lo + s for lanes >= s,
 *                                 row tile_lo + s finalised (lower-left triangle)
 *   phase S  base = tile_lo + 32: the 32 x 32 square: candidates vB = tile_lo + 1 .. + 31 (their
 *                                 StepRecs are final: read back from memory) for all 32 rows
 *   phase U  base = tile_lo + 32: steps s = 32..63 like phase L (upper-right triangle)
 *
 * 95 steps for two columns instead of 2 x 64, and every "uniform" value is uniform per half: the
 * instructions that computed one column's StepRec 64 times now compute two columns' 32 times.  The
 * record of vB comes as DPP operands (each 16-lane row reads its own column's record), the finished
 * row is broadcast inside its half with ds_bpermute, the lutT windows, S / V prefixes of the tile
 * rows and the StepRec handed from phase L to phase U live in LDS per column.  Arithmetic and
 * operand order are those of pw_phase2_body: eval_segment_dpp == eval_segment<true> (phase 1 uses
 * both), pairwise_step and make_step are the same functions.  Pairs with a generic column (or a
 * last odd column) run pw_phase2_body column by column. */
#ifndef ISP2X_OCC
#define ISP2X_OCC 4 /* waves per SIMD k_pw_phase2x is compiled for (each wave: two columns) */
#endif
/* the two-column kernel: windows of at most 15 columns at a row stride of 15 (odd as it is: no padding column) --
 * with 16 + 1 the wave's LDS was 11.1 KB = 14 waves per CU, now 9.6 KB = 16 (what its 128 VGPRs allow) */
#ifndef ISP2X_WMAX
#define ISP2X_WMAX 15
#endif
#define ISP2X_WS (ISP2X_WMAX | 1)
#define ISP2X_WF (ISP2_ROWS * ISP2X_WS)        /* floats of one column's lutT window */

__device__ __forceinline__ float half_bcast_f(float x, int src_lane) { /* src_lane: per lane */
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ int half_bcast_i(int x, int src_lane) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, x);
}
__device__ __forceinline__ float half_min_f(float x) { /* minimum over the lane's 32-lane half */
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x = __builtin_fminf(x, __shfl_xor(x, m, 64));
    return x;
}
__device__ __forceinline__ float half_max_f(float x) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) x = __builtin_fmaxf(x, __shfl_xor(x, m, 64));
    return x;
}

template <bool HAS_INVALID>
__device__ __forceinline__ void pw_phase2x_body(const DevParams& P, char* smem, int col0, int tile,
                                                const RowRec* __restrict__ recs,
                                                const float* __restrict__ lutT,
                                                const float* __restrict__ joined,
                                                const PriorRec* __restrict__ priors,
                                                const float* __restrict__ odr,
                                                const float* __restrict__ rcp,
                                                const float* __restrict__ sv_arr, int vhor, int nsplit,
                                                const float* __restrict__ part_cost,
                                                const int* __restrict__ part_idx,
                                                StepRec* __restrict__ steps,
                                                float* __restrict__ cost_table,
                                                int32_t* __restrict__ index_table,
                                                float* __restrict__ blksum, float* __restrict__ t8row) {
    const int H = P.H, D = P.D;
    const int lane = threadIdx.x, li = lane & 31, l15 = lane & 15, hbase = lane & 32;
    const int half = lane >> 5;
    double* s_invc = (double*)smem;                      /* [32] */
    double* s_logc = s_invc + IS_LOG_TABLE_SIZE;         /* [32] */
    float* s_odr = (float*)(s_logc + IS_LOG_TABLE_SIZE); /* [D -> x4] */
    float* s_rcp = s_odr + ((D + 3) & ~3);               /* [IS_TILE+1 -> x4] */
    float* s_win = s_rcp + ((IS_TILE + 1 + 3) & ~3);     /* [2][65][ISP2X_WS] lutT windows */
    /* [2][66] S prefixes of the tile's rows; with an invalid value also [2][68] BYTES: the valid-count prefixes as
     * offsets from the count at the tile's first row (0 .. 64, exact) -- as floats they cost the kernel its 16th
     * wave per CU (10.3 instead of 9.9 KB of LDS per wave) */
    float* s_SV = s_win + 2 * ISP2X_WF;
    unsigned char* s_Vb = (unsigned char*)(s_SV + 2 * 66);
    float* s_st = s_SV + 2 * 66 + (HAS_INVALID ? 36 : 0); /* [2][16] StepRec handed from phase L to phase U */
    const int tile_lo = tile * IS_TILE;
    const int colg = col0 + half;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const PriorRec* pcol = priors + (size_t)(col0 / P.C) * H; /* (both columns: the same image) */
    StepRec* scol = steps + (size_t)colg * H;
    const float* sv = sv_arr + (size_t)colg * 2 * (H + 1);
    const int n_rows = min(IS_TILE, H - tile_lo);
    float* my_winbase = s_win + half * ISP2X_WF;
    float* my_S = s_SV + half * 66;
    unsigned char* my_Vb = s_Vb + half * 68; /* (never touched without an invalid value) */
    float* my_Vbase = s_SV + 2 * 66 + 34 + half; /* the count at the tile's first row (an exact integer), kept in LDS: a register held through the walk spilled */

    /* ---- prologue: per column the fn window [lo, lo + W) of the tile's rows (pw_phase2_body) */
    int lo, W;
    {
        float dmin = IS_INF, dmax = -IS_INF;
        for (int k = 0; k < 2; k++) {
            const int v = tile_lo + li + 32 * k;
            const float d = joined[(size_t)colg * H + min(v, H - 1)];
            const bool ok = (v < H) && !(HAS_INVALID && d == P.invalid);
            dmin = __builtin_fminf(dmin, ok ? d : IS_INF);
            dmax = __builtin_fmaxf(dmax, ok ? d : -IS_INF);
        }
        dmin = half_min_f(dmin);
        dmax = half_max_f(dmax);
        int l = (int)__builtin_fminf(__builtin_fmaxf(dmin, 1.0f), (float)D) - 1;
        l = min(max(l, 0), D - 1);
        int hh = (int)__builtin_fminf(__builtin_fmaxf(dmax, 0.0f), (float)(D - 1)) + 1;
        hh = min(max(hh, l), min(D - 1, l + ISP2X_WMAX - 1));
        lo = l;
        W = hh - l + 1;
    }
    { /* the window rows tile_lo .. tile_lo + 64 of this lane's column: 32 lanes, 16 columns at most */
        constexpr int NL = (ISP2_ROWS * 16 + 31) / 32; /* sixteen lanes per row, those beyond the window idle */
        const int f = li & 15, j0 = li >> 4; /* two rows per sweep of the half */
        float tmp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + 2 * k;
            tmp[k] = (j < ISP2_ROWS && f < W) ? lcol[(size_t)min(tile_lo + j, H) * D + lo + f] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + 2 * k;
            if (j < ISP2_ROWS && f < W) my_winbase[j * ISP2X_WS + f] = tmp[k];
        }
    }
    for (int j = li; j <= IS_TILE; j += 32) { /* S / V prefixes at tile_lo + j */
        const RowRec* q = rcol + min(tile_lo + j, H);
        my_S[j] = q->S;
        if (HAS_INVALID) {
            const float vb0 = rcol[min(tile_lo, H)].V;
            my_Vb[j] = (unsigned char)(int)(q->V - vb0);
            if (j == 0) *my_Vbase = vb0;
        }
    }
    if (lane == 0) is_log_tables(s_invc, s_logc);
    for (int i = lane; i < D; i += 64) s_odr[i] = odr[i];
    for (int i = lane; i <= IS_TILE; i += 64) s_rcp[i] = rcp[min(i, H)];
    __syncthreads();

    StepVals st;
    st.pwmp = IS_INF; st.idx_gs = -1;
    st.g_hi_thr = st.g_lo_thr = st.p1_hi = st.p1_lo = st.p1_mid = st.o_hi_thr = st.o_lo_thr = 0.0f;
    st.p2_hi = st.p2_lo = st.p2_mid = st.p3_yes = st.p3_no = 0.0f;
    float q_o = IS_INF, q_gs = IS_INF; /* (per tile, see pw_phase2_body) */
    st.q_o = q_o; st.q_gs = q_gs;
    int ob_cached = -1;
    float S_obc = 0.0f, V_obc = 0.0f;
    /* partial minima of phase 1 for the rows base + li (its nsplit workgroups merged) */
    auto load_best = [&](int row_off, PairBest& b) {
        const size_t o = (size_t)colg * nsplit * 3 * 64 + row_off + li;
        b.g = part_cost[o]; b.ig = part_idx[o];
        b.o = part_cost[o + 64]; b.io = part_idx[o + 64];
        b.s = part_cost[o + 128]; b.is = part_idx[o + 128];
        for (int sp = 1; sp < nsplit; sp++) {
            const size_t q = o + (size_t)sp * 3 * 64;
            float c2 = part_cost[q]; int i2 = part_idx[q];
            if ((c2 < b.g) || (c2 == b.g && c2 < IS_INF && (i2 / 3) < (b.ig / 3))) { b.g = c2; b.ig = i2; }
            c2 = part_cost[q + 64]; i2 = part_idx[q + 64];
            if ((c2 < b.o) || (c2 == b.o && c2 < IS_INF && (i2 / 3) < (b.io / 3))) { b.o = c2; b.io = i2; }
            c2 = part_cost[q + 128]; i2 = part_idx[q + 128];
            if ((c2 < b.s) || (c2 == b.s && c2 < IS_INF && (i2 / 3) < (b.is / 3))) { b.s = c2; b.is = i2; }
        }
    };
    auto store_rows = [&](int vT, const PairBest& b) {
        if (vT < H) {
            const size_t o = ((size_t)colg * H + vT) * 3;
            cost_table[o + 0] = b.g; cost_table[o + 1] = b.o; cost_table[o + 2] = b.s;
            index_table[o + 0] = b.ig; index_table[o + 1] = b.io; index_table[o + 2] = b.is;
        }
    };
    /* the candidates that start at vB = r for the lanes `live`: eval + lutT values + update */
    auto candidates = [&](const RowRec& my, int vTc, const float* my_win, int r, bool live, float R0, float R1,
                          const StepVals& stv, PairBest& b) {
        const int hc = max(vTc + 1 - r, 1);
        const float rh = s_rcp[min(hc, IS_TILE)];
        const bool sky = !(r - 1 < vhor);
        SegTerms t;
        if (sky) t = eval_segment_dpp<HAS_INVALID, IS_WANT_SKY>(my, R0, R1, (float)hc, rh, D, P.iw, s_rcp);
        else t = eval_segment_dpp<HAS_INVALID, IS_WANT_GROUND>(my, R0, R1, (float)hc, rh, D, P.iw, s_rcp);
        const int fo = t.fni - lo;
        const bool inwin = (unsigned)fo < (unsigned)W;
        const int foc = inwin ? fo : 0;
        float od = my_win[foc] - my_winbase[(r - tile_lo) * ISP2X_WS + foc];
        if (__builtin_amdgcn_ballot_w64(live && !inwin) != 0ull) { /* outside the window: rare */
            if (IS_P2_GATHER_MASKED ? (live && !inwin) : true) { /* (only the lanes outside fetch, see pw_phase2_body) */
                const float og = (lcol + (size_t)(vTc + 1) * D)[(unsigned)t.fni] - (lcol + (size_t)r * D)[(unsigned)t.fni];
                od = inwin ? od : og;
            }
        }
        if (sky) pairwise_step<true>(P, stv, r, live, od, t, b);
        else pairwise_step<false>(P, stv, r, live, od, t, b);
    };
    /* row r is final in lane `src` of each half: broadcast it, build and publish StepRec(r + 1) */
    auto finalize = [&](int r, int src, const PairBest& b, float myS_r1, float myV_r1) {
        const PriorVals pv = sload_prior(pcol + min(r + 1, H - 1));
        const int sl = hbase + src;
        const float cG = half_bcast_f(b.g, sl), cO = half_bcast_f(b.o, sl), cS = half_bcast_f(b.s, sl);
        const int ob = half_bcast_i(b.io, sl) / 3; /* start of the best object chain */
        const float S_r1 = half_bcast_f(myS_r1, sl);
        const float V_r1 = HAS_INVALID ? half_bcast_f(myV_r1, sl) : 0.0f;
        float S_ob, V_ob = 0.0f;
        const bool below = ob <= tile_lo;
        if (__builtin_amdgcn_ballot_w64(below && ob != ob_cached) != 0ull) {
            if (below && ob != ob_cached) { /* repeated only when the chain's start changes */
                S_obc = sv[ob];
                if (HAS_INVALID) V_obc = sv[(H + 1) + ob];
                ob_cached = ob;
            }
        }
        const int oi = min(max(ob - tile_lo, 0), IS_TILE);
        S_ob = below ? S_obc : my_S[oi];
        if (HAS_INVALID) V_ob = below ? V_obc : *my_Vbase + (float)my_Vb[oi];
        st = make_step<HAS_INVALID, true>(P, S_r1, V_r1, S_ob, V_ob, s_odr, s_invc, s_logc, &pv, vhor, r, cG,
                                          cO, cS, ob);
        const float m8 = min_raw(min3_raw(st.p1_hi, st.p1_lo, st.p1_mid),
                                 min3_raw(min3_raw(st.p2_hi, st.p2_lo, st.p2_mid), st.p3_yes, st.p3_no));
        if (((r - tile_lo) & (IS_QB - 1)) == 0) q_o = q_gs = IS_INF; /* vB = r + 1 starts a bound block */
        const float pwm8 = P.pw * m8;
        q_o = min_raw(q_o, pwm8);
        q_gs = min_raw(q_gs, st.pwmp);
        st.q_o = q_o; st.q_gs = q_gs;
        if (li == 0 && r + 1 < H) {
            store_step(scol + r + 1, st);
            t8row[(size_t)colg * H + r + 1] = pwm8; /* (lemma L8: the row's own transition bound) */
        }
    };
    auto rec_dpp = [&](int v, float& R0, float& R1) { /* the record of v of this lane's column, DPP layout */
        const float* q = (const float*)(rcol + min(v, H));
        R0 = q[l15];
        R1 = q[16 + l15];
    };

    /* ================= phase L: rows tile_lo .. tile_lo + 31 ================= */
    {
        const int vT = tile_lo + li, vTc = min(vT, H - 1);
        const RowRec my = load_rec(rcol + vTc + 1);
        const float* my_win = my_winbase + (vTc + 1 - tile_lo) * ISP2X_WS;
        PairBest b;
        load_best(0, b);
        float n0, n1;
        rec_dpp(tile_lo + 1, n0, n1);
        const int nL = min(32, n_rows);
        for (int s = 0; s < nL; s++) {
            const int r = tile_lo + s;
            if (s > 0) {
                const float R0 = n0, R1 = n1;
                rec_dpp(r + 1, n0, n1);
                candidates(my, vTc, my_win, r, (vT < H) && (vT >= r), R0, R1, st, b);
            }
            finalize(r, s, b, my.S, my.V);
        }
        store_rows(vT, b);
        if (li == 0) { /* StepRec(tile_lo + 32) for phase U */
            float* d = s_st + half * 16;
            d[0] = st.pwmp; d[1] = __builtin_bit_cast(float, st.idx_gs); d[2] = st.g_hi_thr; d[3] = st.g_lo_thr;
            d[4] = st.p1_hi; d[5] = st.p1_lo; d[6] = st.p1_mid; d[7] = st.o_hi_thr; d[8] = st.o_lo_thr;
            d[9] = st.p2_hi; d[10] = st.p2_lo; d[11] = st.p2_mid; d[12] = st.p3_yes; d[13] = st.p3_no;
            d[14] = st.q_o; d[15] = st.q_gs;
        }
    }
    /* the summaries of the bound blocks (lemmas L7, L8) of both phases' rows, at the very end: nothing of
     * the walk is alive any more (computed inside the phases they cost the walk registers it does not have) */
    auto summaries = [&](int rows_done) {
        __builtin_amdgcn_s_waitcnt(0); /* this wave's StepRec / T8 stores have reached the L2 */
        float* const bcolw = blksum + (size_t)colg * (P.ntiles * IS_QPT + 1) * IS_L7_F;
        for (int base = 0; base < rows_done; base += 32)
            l78_block_summaries<IS_QB_LOG>(P, rcol, scol, t8row + (size_t)colg * H, bcolw, tile_lo, base + li, vhor);
    };
    if (n_rows <= 32) {
        summaries(32);
        return;
    }
    /* this wave's StepRec stores of phase L must have reached the L2 before phase S reads them */
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    /* ================= phases S and U: rows tile_lo + 32 .. tile_lo + 63 ================= */
    {
        const int vT = tile_lo + 32 + li, vTc = min(vT, H - 1);
        const RowRec my = load_rec(rcol + vTc + 1);
        const float* my_win = my_winbase + (vTc + 1 - tile_lo) * ISP2X_WS;
        PairBest b;
        load_best(32, b);
        const bool live_all = vT < H;
        float n0, n1;
        rec_dpp(tile_lo + 1, n0, n1);
        for (int sp = 1; sp < 32; sp++) { /* the square: vB = tile_lo + sp, every row of the half */
            const int r = tile_lo + sp;
            const float R0 = n0, R1 = n1;
            rec_dpp(r + 1, n0, n1);
            /* the StepRec of r (built in phase L by this wave): requested now, needed after the eval */
            const float4* sq = reinterpret_cast<const float4*>(scol + r);
            const float4 a0 = sq[0], a1 = sq[1], a2 = sq[2], a3 = sq[3];
            StepVals sv_;
            sv_.pwmp = a0.x; sv_.idx_gs = __builtin_bit_cast(int, a0.y); sv_.g_hi_thr = a0.z; sv_.g_lo_thr = a0.w;
            sv_.p1_hi = a1.x; sv_.p1_lo = a1.y; sv_.p1_mid = a1.z; sv_.o_hi_thr = a1.w;
            sv_.o_lo_thr = a2.x; sv_.p2_hi = a2.y; sv_.p2_lo = a2.z; sv_.p2_mid = a2.w;
            sv_.p3_yes = a3.x; sv_.p3_no = a3.y; sv_.q_o = a3.z; sv_.q_gs = a3.w;
            candidates(my, vTc, my_win, r, live_all, R0, R1, sv_, b);
        }
        { /* StepRec(tile_lo + 32), the running minima with it */
            const float* d = s_st + half * 16;
            st.pwmp = d[0]; st.idx_gs = __builtin_bit_cast(int, d[1]); st.g_hi_thr = d[2]; st.g_lo_thr = d[3];
            st.p1_hi = d[4]; st.p1_lo = d[5]; st.p1_mid = d[6]; st.o_hi_thr = d[7]; st.o_lo_thr = d[8];
            st.p2_hi = d[9]; st.p2_lo = d[10]; st.p2_mid = d[11]; st.p3_yes = d[12]; st.p3_no = d[13];
            st.q_o = d[14]; st.q_gs = d[15];
            q_o = st.q_o; q_gs = st.q_gs;
        }
        rec_dpp(tile_lo + 32, n0, n1);
        for (int s = 32; s < n_rows; s++) {
            const int r = tile_lo + s;
            const float R0 = n0, R1 = n1;
            rec_dpp(r + 1, n0, n1);
            candidates(my, vTc, my_win, r, (vT < H) && (vT >= r), R0, R1, st, b);
            finalize(r, s - 32, b, my.S, my.V);
        }
        store_rows(vT, b);
    }
    summaries(64);
}

template <bool HAS_INVALID>
__global__ __launch_bounds__(64, ISP2X_OCC) void k_pw_phase2x(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ joined, const PriorRec* __restrict__ priors,
    const float* __restrict__ odr, const float* __restrict__ rcp,
    const float* __restrict__ sv_arr, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const float* __restrict__ part_cost,
    const int* __restrict__ part_idx, StepRec* __restrict__ steps, float* __restrict__ cost_table,
    int32_t* __restrict__ index_table, float* __restrict__ blksum, float* __restrict__ t8row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int col0 = col_base + 2 * (int)blockIdx.x; /* col_base even, P.C even: one image per pair */
    if (col0 >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[col0 / P.C]);
    const int f0 = __builtin_amdgcn_readfirstlane(col_flags[col0]);
    const int f1 = col0 + 1 < ncols ? __builtin_amdgcn_readfirstlane(col_flags[col0 + 1]) : 1;
    if (f0 == 0 && f1 == 0)
        pw_phase2x_body<HAS_INVALID>(P, smem, col0, tile, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor, nsplit,
                                     part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
    /* (a pair with a generic column: k_pw_phase2_generic walks it, column by column) */
}

/* The columns k_pw_phase2x leaves out: both columns of every pair that contains a generic-encoding
 * column.  A small grid that leaves at once when k_prepare_columns counted no generic column. */
template <bool HAS_INVALID>
__global__ __launch_bounds__(64, 2) void k_pw_phase2_generic(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ joined, const PriorRec* __restrict__ priors,
    const float* __restrict__ odr, const float* __restrict__ rcp,
    const float* __restrict__ sv_arr, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const float* __restrict__ part_cost,
    const int* __restrict__ part_idx, StepRec* __restrict__ steps, float* __restrict__ cost_table,
    int32_t* __restrict__ index_table, const int* __restrict__ n_generic, float* __restrict__ blksum, float* __restrict__ t8row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (__builtin_amdgcn_readfirstlane(*n_generic) == 0) return;
    for (int colg = col_base + (int)blockIdx.x; colg < ncols; colg += (int)gridDim.x) {
        const int f0 = __builtin_amdgcn_readfirstlane(col_flags[colg & ~1]);
        const int f1 = (colg | 1) < ncols ? __builtin_amdgcn_readfirstlane(col_flags[colg | 1]) : 1;
        if (f0 == 0 && f1 == 0) continue; /* k_pw_phase2x has the pair */
        const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
        if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
            pw_phase2_body<true, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,
                                              nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
        else
            pw_phase2_body<false, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,
                                               nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
        __syncthreads();
    }
}


/* ====================================================================================== */
/* phase 2, split over the waves of a workgroup (k_pw_phase2s)                              */
/* ====================================================================================== */
/* A lone wavefront retires a dependent instruction every ~8 cycles, and a step of k_pw_phase2 is
 * ~420 instructions of which only the chain
 *     final row r-1 -> StepRec(r) -> candidate (r, r) -> final row r
 * is inherently serial (~1.3 us per step, 82 us per tile at one frame).  Here the work of a
 * (column, tile) is split over ISP2S_WAVES wavefronts:
 *
 *   wave 0 (chain)      per step: reads the state-independent terms of the step from an LDS slot,
 *                       applies the StepRec (two adds and the three threshold selects per
 *                       candidate), updates the running minima, broadcasts the finished row,
 *                       builds and publishes the next StepRec;
 *   waves 1.. (evaluators) everything of a step that does not depend on the DP state -- the
 *                       scalar loads of the vB record and the priors, eval_segment, the two lutT
 *                       values, the weights -- for steps s = e, e + NE, ..., several steps ahead,
 *                       into a ring of ISP2S_SLOTS LDS slots.
 *
 * Hand-over through LDS sequence numbers (seq[slot] = step it holds, cons = last step consumed);
 * both sides poll with s_sleep.  Dead lanes (vT < r, vT >= H) get +inf data terms: their costs are
 * +inf or NaN and never pass a `<`, so the chain needs no lane mask.  Arithmetic and operand order
 * are those of pairwise_step / make_step: cost = (dw * data + pw-term) + sw * seg. */
#ifndef ISP2S_WAVES
#define ISP2S_WAVES 2
#endif
#define ISP2S_NE (ISP2S_WAVES - 1)
#ifndef ISP2S_SLOTS
#define ISP2S_SLOTS 4
#endif
#define ISP2S_SLOT_F (8 * 64 + 16) /* [lane][A_gs, B_gs, A_o, B_o, fn, 3 unused] + the PriorVals of vB = r + 1 */
#define ISP2S_SPIN_LIMIT (1 << 26)

/* wave-uniform wait until *p >= want; a bound that a correct run never reaches turns a would-be
 * hang into an abort */
/* The flags live in LDS and are accessed through LDS-typed pointers with LDS-only fences: as plain `volatile int*`
 * (a generic pointer) every poll was a FLAT load with sc0 sc1 behind `s_waitcnt vmcnt(0)`, and every release fence a
 * wait for ALL vector memory of the wave -- i.e. each step of the serial chain waited for the global stores of the
 * StepRec it had just written (round 6: one 1024x2048 pairwise frame 1.56 -> see DESIGN.md section 9). */
typedef __attribute__((address_space(3))) int isp2s_flag_t;
typedef __attribute__((address_space(3))) float isp2s_lds_f_t;
typedef float isp2s_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) isp2s_f4 isp2s_lds_f4_t;
#define ISP2S_FENCE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define ISP2S_FENCE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
__device__ __forceinline__ int isp2s_wait_ge(volatile isp2s_flag_t* p, int want) { /* -> polls that found it unready */
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(*p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > ISP2S_SPIN_LIMIT) __builtin_trap();
    }
    ISP2S_FENCE_ACQUIRE();
    return spins;
}

template <bool SKY>
__device__ __forceinline__ void pairwise_step_pre(const DevParams& P, const StepVals st, int vB, float a_gs,
                                                  float b_gs, float a_o, float b_o, float fn,
                                                  PairBest& b) {
    const float cost_gs = a_gs + st.pwmp + b_gs; /* :716-719 / :762-765 */
    if (SKY) take_if_less(b.s, b.is, cost_gs, st.idx_gs);
    else take_if_less(b.g, b.ig, cost_gs, st.idx_gs);
    /* object, :777-837 */
    const float p1 = (fn > st.g_hi_thr) ? st.p1_hi : ((fn < st.g_lo_thr) ? st.p1_lo : st.p1_mid);
    const float p2 = (fn > st.o_hi_thr) ? st.p2_hi : ((fn < st.o_lo_thr) ? st.p2_lo : st.p2_mid);
    const float p3 = (fn > P.epsilon) ? st.p3_yes : st.p3_no;
    const float m12 = __builtin_fminf(p1, p2);
    const float mp = __builtin_fminf(m12, p3);
    const float cost = a_o + P.pw * mp + b_o;
    const int base_o = vB * 3 + IS_OBJECT;
    int idx = (p1 < p2) ? (base_o - 1) : base_o;
    idx = (p3 < m12) ? (base_o + 1) : idx;
    take_if_less_v(b.o, b.io, cost, idx);
}

template <bool FAST, bool HAS_INVALID>
__device__ __forceinline__ void pw_phase2s_body(const DevParams& P, char* smem, int colg, int tile,
                                                const RowRec* __restrict__ recs,
                                                const float* __restrict__ lutT,
                                                const float* __restrict__ joined,
                                                const PriorRec* __restrict__ priors,
                                                const float* __restrict__ odr,
                                                const float* __restrict__ rcp,
                                                const float* __restrict__ sv_arr, int vhor,
                                                int nsplit, const float* __restrict__ part_cost,
                                                const int* __restrict__ part_idx,
                                                StepRec* __restrict__ steps,
                                                float* __restrict__ cost_table,
                                                int32_t* __restrict__ index_table,
                                                float* __restrict__ blksum, float* __restrict__ t8row) {
    const int H = P.H, D = P.D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* s_invc = (double*)smem;                      /* [32] */
    double* s_logc = s_invc + IS_LOG_TABLE_SIZE;         /* [32] */
    float* s_odr = (float*)(s_logc + IS_LOG_TABLE_SIZE); /* [D -> x4] */
    float* s_rcp = s_odr + ((D + 3) & ~3);               /* [IS_TILE+1 -> x4] */
    float* s_win = s_rcp + ((IS_TILE + 1 + 3) & ~3);     /* [65][ISP2_WS] lutT window */
    float* s_ring = s_win + ((ISP2_ROWS * ISP2_WS + 3) & ~3); /* [ISP2S_SLOTS][ISP2S_SLOT_F] */
    volatile isp2s_flag_t* s_seq = (volatile isp2s_flag_t*)(isp2s_flag_t*)(int*)(s_ring + ISP2S_SLOTS * ISP2S_SLOT_F); /* [SLOTS] + cons */
    volatile isp2s_flag_t* s_cons = s_seq + ISP2S_SLOTS;
    const int tile_lo = tile * IS_TILE;
    const RowRec* rcol = recs + (size_t)colg * (H + 1);
    const float* lcol = lutT + (size_t)colg * (H + 1) * D;
    const PriorRec* pcol = priors + (size_t)(colg / P.C) * H;
    StepRec* scol = steps + (size_t)colg * H;
    const float* sv = sv_arr + (size_t)colg * 2 * (H + 1);
    const int vT = tile_lo + lane;
    const int vTc = min(vT, H - 1);
    const int n_rows = min(IS_TILE, H - tile_lo);

    /* ---- prologue, all waves: the fn window (every wave computes it), tables, flags */
    int lo, W;
    {
        const float d = joined[(size_t)colg * H + vTc];
        const bool ok = (vT < H) && !(HAS_INVALID && d == P.invalid);
        const float dmin = wave_min_f(ok ? d : IS_INF);
        const float dmax = wave_max_f(ok ? d : -IS_INF);
        int l = (int)__builtin_fminf(__builtin_fmaxf(dmin, 1.0f), (float)D) - 1;
        l = min(max(l, 0), D - 1);
        int h = (int)__builtin_fminf(__builtin_fmaxf(dmax, 0.0f), (float)(D - 1)) + 1;
        h = min(max(h, l), min(D - 1, l + ISP2_WMAX - 1));
        lo = __builtin_amdgcn_readfirstlane(l);
        W = __builtin_amdgcn_readfirstlane(h - l + 1);
    }
    { /* window rows: every load issued before the first LDS store (see pw_phase2_body) */
        int lg = 0;
        while ((1 << lg) < W) lg++;
        lg = __builtin_amdgcn_readfirstlane(lg);
        const int f = tid & ((1 << lg) - 1);
        const int j0 = tid >> lg, dj = (ISP2S_WAVES * 64) >> lg;
        constexpr int NL = (ISP2_ROWS * ISP2_WMAX + ISP2S_WAVES * 64 - 1) / (ISP2S_WAVES * 64);
        float tmp[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + k * dj;
            tmp[k] = (j < ISP2_ROWS && f < W) ? lcol[(size_t)min(tile_lo + j, H) * D + lo + f] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int j = j0 + k * dj;
            if (j < ISP2_ROWS && f < W) s_win[j * ISP2_WS + f] = tmp[k];
        }
    }
    if (tid == 0) is_log_tables(s_invc, s_logc);
    for (int i = tid; i < D; i += ISP2S_WAVES * 64) s_odr[i] = odr[i];
    for (int i = tid; i <= IS_TILE; i += ISP2S_WAVES * 64) s_rcp[i] = rcp[min(i, H)];
    if (tid < ISP2S_SLOTS) s_seq[tid] = -1;
    if (tid == ISP2S_SLOTS) *s_cons = 0;
    __syncthreads();

    if (w == 0) {
        /* ================================ chain wave ================================ */
        float myS, myV = 0.0f; /* prefixes at vT + 1 (lane k: prefix index tile_lo + k + 1) */
        {
            const RowRec* mr = rcol + vTc + 1;
            myS = mr->S;
            if (HAS_INVALID) myV = mr->V;
        }
        PairBest b;
        {
            const size_t o = (size_t)colg * nsplit * 3 * 64 + lane;
            b.g = part_cost[o]; b.ig = part_idx[o];
            b.o = part_cost[o + 64]; b.io = part_idx[o + 64];
            b.s = part_cost[o + 128]; b.is = part_idx[o + 128];
            for (int sp = 1; sp < nsplit; sp++) {
                const size_t q = o + (size_t)sp * 3 * 64;
                float c2 = part_cost[q]; int i2 = part_idx[q];
                if ((c2 < b.g) || (c2 == b.g && c2 < IS_INF && (i2 / 3) < (b.ig / 3))) { b.g = c2; b.ig = i2; }
                c2 = part_cost[q + 64]; i2 = part_idx[q + 64];
                if ((c2 < b.o) || (c2 == b.o && c2 < IS_INF && (i2 / 3) < (b.io / 3))) { b.o = c2; b.io = i2; }
                c2 = part_cost[q + 128]; i2 = part_idx[q + 128];
                if ((c2 < b.s) || (c2 == b.s && c2 < IS_INF && (i2 / 3) < (b.is / 3))) { b.s = c2; b.is = i2; }
            }
        }
        /* S (and V) prefix at the START row of this lane's best object candidate, carried beside b.io: the row that
         * becomes final hands it to make_step through one v_readlane.  (Until round 6 the chain fetched it with a scalar
         * load from global memory whenever the start lay below the tile -- a round trip of its own on every such row:
         * "broadcasts" 330 of the 2030 clocks a row costs, tools/experiments/p2s_phase_probe.py.) */
        float So, Vo = 0.0f;
        {
            const int ob0 = max(b.io, 0) / 3;
            So = sv[ob0];
            if (HAS_INVALID) Vo = sv[(H + 1) + ob0];
        }
        float S_vb = 0.0f, V_vb = 0.0f; /* the prefixes at index r = the start row of this step's candidates */
        StepVals st;
        st.pwmp = IS_INF; st.idx_gs = -1;
        st.g_hi_thr = st.g_lo_thr = st.p1_hi = st.p1_lo = st.p1_mid = st.o_hi_thr = st.o_lo_thr = 0.0f;
        st.p2_hi = st.p2_lo = st.p2_mid = st.p3_yes = st.p3_no = 0.0f;
        float q_o = IS_INF, q_gs = IS_INF; /* (per tile, see pw_phase2_body) */
        st.q_o = q_o; st.q_gs = q_gs;
#ifdef IS_ABL_P2PHASES /* (debug build: the chain wave's sections, tools/experiments/p2s_phase_probe.py) */
        unsigned long long acc_p2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        ISP2_MARK_INIT();
        ISP2_MARK(0); /* prologue */
        for (int s = 0; s < n_rows; s++) {
            const int r = tile_lo + s;
            PriorVals pv;
            if (s > 0) {
                const int q = s % ISP2S_SLOTS;
                /* flag and data are requested TOGETHER: LDS operations of a wave execute in order and the evaluator
                 * writes the data before the flag, so data read behind a flag that says "ready" is the step's data --
                 * one LDS round trip instead of two on the chain.  (volatile: the compiler keeps the program order of
                 * the accesses.)  The evaluator is ahead in practice; otherwise poll, then read again. */
                volatile isp2s_lds_f_t* vs = (volatile isp2s_lds_f_t*)(isp2s_lds_f_t*)(s_ring + q * ISP2S_SLOT_F);
                /* slot layout: [64 lanes][8 floats: a_gs, b_gs, a_o, b_o, fn, -, -, -] then the eight PriorVals: five LDS
                 * instructions per step (flag, b128 + b32 per lane, two broadcast b128) instead of fourteen */
                volatile isp2s_lds_f4_t* v4 = (volatile isp2s_lds_f4_t*)(isp2s_lds_f4_t*)(s_ring + q * ISP2S_SLOT_F);
                int seen = s_seq[q];
                isp2s_f4 lv = v4[2 * lane];
                float fn = vs[8 * lane + 4];
                isp2s_f4 p0 = v4[128], p1 = v4[129];
                if (__builtin_amdgcn_readfirstlane(seen) < s) {
#ifdef IS_ABL_P2PHASES
                    acc_p2[3] += 1ull + (unsigned long long)isp2s_wait_ge(s_seq + q, s); /* (section 3: a count) */
#else
                    isp2s_wait_ge(s_seq + q, s);
#endif
                    lv = v4[2 * lane];
                    fn = vs[8 * lane + 4];
                    p0 = v4[128]; p1 = v4[129];
                }
                ISP2_MARK(1); /* slot flag + data */
                const float a_gs = lv.x, b_gs = lv.y, a_o = lv.z, b_o = lv.w;
                pv.pc = p0.x; pv.g_from = p0.y; pv.s_from_g = p0.z; pv.o_from_s = p0.w;
                pv.og_hi = p1.x; pv.og_lo = p1.y; pv.og_mid = p1.z; pv.g_prev = p1.w;
                /* the slot is free again once these reads have executed (LDS operations of a wave
                 * execute in order) */
                ISP2S_FENCE_RELEASE();
                if (lane == 0) *s_cons = s;
                ISP2_MARK(2); /* slot reads */
                if (r - 1 < vhor)
                    pairwise_step_pre<false>(P, st, r, a_gs, b_gs, a_o, b_o, fn, b);
                else
                    pairwise_step_pre<true>(P, st, r, a_gs, b_gs, a_o, b_o, fn, b);
                { /* lanes whose object candidate of THIS step (index 3 r + type) has just become their best */
                    const bool up = (unsigned)(b.io - 3 * r) < 3u;
                    So = up ? S_vb : So;
                    if (HAS_INVALID) Vo = up ? V_vb : Vo;
                }
                ISP2_MARK(4); /* pairwise_step */
            } else {
                pv = sload_prior(pcol + min(r + 1, H - 1));
            }
            { /* (unconditional: see pw_phase2_body) */
                const float cG = readlane_f(b.g, s), cO = readlane_f(b.o, s), cS = readlane_f(b.s, s);
                const int ob = __builtin_amdgcn_readlane(b.io, s) / 3;
                const float S_r1 = readlane_f(myS, s), V_r1 = HAS_INVALID ? readlane_f(myV, s) : 0.0f;
                const float S_ob = readlane_f(So, s);
                const float V_ob = HAS_INVALID ? readlane_f(Vo, s) : 0.0f;
                S_vb = S_r1; V_vb = V_r1; /* prefix index r + 1 = the start row of the next step's candidates */
                ISP2_MARK(5); /* broadcasts */
                /* (measured here in round 6, neither kept: the exact-division shortcut for chains of up to 64 rows with
                 * its reciprocal out of a register through v_readlane -- make_step 844 -> 994 clocks, both divisions stay
                 * in the code --, and object_disparity_range[k] out of registers instead of LDS -- no measurable change) */
                st = make_step<HAS_INVALID>(P, S_r1, V_r1, S_ob, V_ob, s_odr, s_invc, s_logc, &pv, vhor, r,
                                            cG, cO, cS, ob);
                ISP2_MARK(6); /* make_step */
                const float m8 = min_raw(min3_raw(st.p1_hi, st.p1_lo, st.p1_mid),
                                         min3_raw(min3_raw(st.p2_hi, st.p2_lo, st.p2_mid), st.p3_yes, st.p3_no));
                if ((s & (IS_QB - 1)) == 0) q_o = q_gs = IS_INF; /* vB = r + 1 starts a bound block */
                const float pwm8 = P.pw * m8;
                q_o = min_raw(q_o, pwm8);
                q_gs = min_raw(q_gs, st.pwmp);
                st.q_o = q_o; st.q_gs = q_gs;
                if (lane == 0 && r + 1 < H) {
                    store_step(scol + r + 1, st);
                    t8row[(size_t)colg * H + r + 1] = pwm8;
                }
                ISP2_MARK(7); /* running minima + store */
            }
        }
#ifdef IS_ABL_P2PHASES
        if (lane == 0)
            for (int k = 0; k < 8; k++) atomicAdd(&g_p2phase[k], acc_p2[k]);
#endif
        if (vT < H) {
            const size_t o = ((size_t)colg * H + vT) * 3;
            cost_table[o + 0] = b.g; cost_table[o + 1] = b.o; cost_table[o + 2] = b.s;
            index_table[o + 0] = b.ig; index_table[o + 1] = b.io; index_table[o + 2] = b.is;
        }
        { /* block summaries (lemmas L7, L8, see pw_phase2_body) */
            float* const bcolw = blksum + (size_t)colg * (P.ntiles * IS_QPT + 1) * IS_L7_F;
            if (FAST) {
                __builtin_amdgcn_s_waitcnt(0);
                l78_block_summaries<IS_QB_LOG>(P, rcol, scol, t8row + (size_t)colg * H, bcolw, tile_lo, lane, vhor);
            } else {
                float* const slot = bcolw + (size_t)(tile * IS_QPT + (lane >> IS_QB_LOG) + 1) * IS_L7_F;
                const bool writer = (lane & (IS_QB - 1)) == 0;
                if (writer) l7_store(slot, l7_never());
                l8_store_never(slot + 8, writer);
            }
        }
    } else {
        /* ================================ evaluator waves ================================ */
        const RowRec my = load_rec(rcol + vTc + 1);
        const float* my_row = lcol + (size_t)(vTc + 1) * D;
        const float* my_win = s_win + (vTc + 1 - tile_lo) * ISP2_WS;
        for (int s = w; s < n_rows; s += ISP2S_NE) { /* s >= 1 */
            const int r = tile_lo + s;
            const PriorVals pv = sload_prior(pcol + min(r + 1, H - 1));
            const RowRec rb = sload_rec_pinned(rcol + r);
            const int hc = max(vTc + 1 - r, 1);
            const bool live = (vT < H) && (vT >= r);
            const SegTerms t = eval_segment<FAST, HAS_INVALID>(my, rb, (float)hc, s_rcp[hc], D, P.iw, s_rcp);
            const int fo = t.fni - lo;
            const bool inwin = (unsigned)fo < (unsigned)W;
            const int foc = inwin ? fo : 0;
            float od = my_win[foc] - s_win[s * ISP2_WS + foc];
            if (__builtin_amdgcn_ballot_w64(live && !inwin) != 0ull) { /* outside the window: rare */
                /* (only the lanes outside fetch: with every lane gathering, a step of this kind pulled up to
                 * 128 lines -- 16 KB -- for the few values it needed) */
                if (IS_P2_GATHER_MASKED ? (live && !inwin) : true) {
                    const float og = my_row[(unsigned)t.fni] - (lcol + (size_t)r * D)[(unsigned)t.fni];
                    od = inwin ? od : og;
                }
            }
            const bool ground = r - 1 < vhor; /* :687 / :729 */
            const float a_gs = live ? P.dw * (ground ? t.gd : t.sd) : IS_INF;
            const float b_gs = P.sw * (ground ? t.seg_g : t.seg_s);
            const float a_o = live ? P.dw * od : IS_INF;
            const float b_o = P.sw * t.seg_o;
            const int q = s % ISP2S_SLOTS;
            float* slot = s_ring + q * ISP2S_SLOT_F;
            if (s >= ISP2S_SLOTS) isp2s_wait_ge(s_cons, s - ISP2S_SLOTS); /* the slot's last tenant is consumed */
            *reinterpret_cast<float4*>(slot + 8 * lane) = make_float4(a_gs, b_gs, a_o, b_o);
            slot[8 * lane + 4] = t.mean;
            if (lane == 0) {
                *reinterpret_cast<float4*>(slot + 512) = make_float4(pv.pc, pv.g_from, pv.s_from_g, pv.o_from_s);
                *reinterpret_cast<float4*>(slot + 516) = make_float4(pv.og_hi, pv.og_lo, pv.og_mid, pv.g_prev);
            }
            ISP2S_FENCE_RELEASE();
            if (lane == 0) s_seq[q] = s;
        }
    }
}

template <bool HAS_INVALID>
__global__ __launch_bounds__(ISP2S_WAVES * 64, 5) void k_pw_phase2s(
    const DevParams P, int col_base, int ncols, int tile, int nsplit,
    const RowRec* __restrict__ recs, const float* __restrict__ lutT,
    const float* __restrict__ joined, const PriorRec* __restrict__ priors,
    const float* __restrict__ odr, const float* __restrict__ rcp,
    const float* __restrict__ sv_arr, const int* __restrict__ vhor_arr,
    const int* __restrict__ col_flags, const float* __restrict__ part_cost,
    const int* __restrict__ part_idx, StepRec* __restrict__ steps, float* __restrict__ cost_table,
    int32_t* __restrict__ index_table, float* __restrict__ blksum, float* __restrict__ t8row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int colg = col_base + blockIdx.x;
    if (colg >= ncols) return;
    const int vhor = __builtin_amdgcn_readfirstlane(vhor_arr[colg / P.C]);
    if (__builtin_amdgcn_readfirstlane(col_flags[colg]) == 0)
        pw_phase2s_body<true, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr,
                                           vhor, nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
    else
        pw_phase2s_body<false, HAS_INVALID>(P, smem, colg, tile, recs, lutT, joined, priors, odr, rcp, sv_arr,
                                            vhor, nsplit, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row);
}

extern "C" {

static size_t p1_lds_bytes(const DevParams* P, int nwaves, bool windowed) {
    const size_t rcp = sizeof(float) * (((size_t)P->H + 1 + 3) & ~(size_t)3);
    const size_t tile = sizeof(float) * (size_t)IS_TILE * ((windowed ? IS_P1_WIN : P->D) + 1);
    const size_t merge = (size_t)nwaves * 3 * 64 * 8; /* aliases the tile after the loop */
    /* + the object block bounds of the pre-pass: [ntiles * IS_QPT + 1][64] (a launch of tile t uses
     * t * IS_QPT + 1 entries) */
    return (tile > merge ? tile : merge) + rcp + sizeof(float) * 8 * (size_t)nwaves +
           sizeof(float) * IS_P1_L7_WORDS + sizeof(float) * IS_P1_BLK_WORDS * ((size_t)P->ntiles * IS_QPT + 1) + 32;
}
size_t isk_pairwise_lds_bytes(const DevParams* P, int nwaves) { return p1_lds_bytes(P, nwaves, false); }
size_t isk_phase2_lds_bytes(const DevParams* P) {
    size_t need = sizeof(double) * 2 * IS_LOG_TABLE_SIZE +
                  sizeof(float) * (P->D + (IS_TILE + 1) + (size_t)ISP2_ROWS * ISP2_WS) + 16;
    /* IS_P2_LDS: a floor on the allocation = an occupancy throttle for experiments */
    const size_t floor_bytes = P->knob_p2_lds_floor > 0 ? (size_t)P->knob_p2_lds_floor : 0;
    return need > floor_bytes ? need : floor_bytes;
}

size_t isk_phase2x_lds_bytes(const DevParams* P) {
    const size_t x = sizeof(double) * 2 * IS_LOG_TABLE_SIZE +
                     sizeof(float) * (((P->D + 3) & ~3) + ((IS_TILE + 1 + 3) & ~3) + 2 * (size_t)ISP2X_WF +
                                      2 * 66 + (P->invalid >= 0 ? 36 : 0) + 2 * 16) + 32;
    const size_t one = isk_phase2_lds_bytes(P); /* the one-column fallback inside the kernel */
    return x > one ? x : one;
}

size_t isk_phase2s_lds_bytes(const DevParams* P) {
    return sizeof(double) * 2 * IS_LOG_TABLE_SIZE +
           sizeof(float) * (((P->D + 3) & ~3) + ((IS_TILE + 1 + 3) & ~3) +
                            (((size_t)ISP2_ROWS * ISP2_WS + 3) & ~(size_t)3) +
                            (size_t)ISP2S_SLOTS * ISP2S_SLOT_F) +
           sizeof(int) * (ISP2S_SLOTS + 1 + 2 * ISP2S_WAVES) + 32;
}

hipError_t isk_launch_dp_pairwise(const DevParams* P, int ncols, int nwaves, const RowRec* recs,
                                  const float* lutT, const float* joined, const PriorRec* priors, const float* odr,
                                  const float* rcp, const float* sv_arr, const int* vhor,
                                  const int* col_flags, const PruneRec* prune, StepRec* steps,
                                  float* part_cost, int* part_idx, float* cost_table,
                                  int32_t* index_table, unsigned long long* counters,
                                  const float* cost_T, const int* n_generic, float* blksum, float* t8row,
                                  hipStream_t stream, hipStream_t* aux, int n_aux,
                                  hipEvent_t ev_fork, hipEvent_t* ev_join) {
    /* windowed tiles (P->win_tiles): IS_P1_WIN_WAVES waves per workgroup -- the smaller LDS footprint lets a CU
     * hold more, smaller workgroups (measured at D = 64: 4 waves x 4 workgroups beat 8 x 3 on every tile) */
    int nwaves_win = IS_P1_WIN_WAVES < nwaves ? IS_P1_WIN_WAVES : nwaves;
    if (P->knob_pw_waves > 0) nwaves_win = nwaves; /* (experiments: one wave count for all tiles) */
    const size_t lds2 = isk_phase2_lds_bytes(P);
    /* Columns are independent: with enough of them the batch is cut into groups whose
     * phase-1 / phase-2 chains (2 x ntiles dependent launches each) run on their own streams, so
     * that the tails and the latency-bound serial phase 2 of one group share the CUs with the
     * other groups' launches. */
    /* few columns: two workgroups per (column, tile) in phase 1 */
    /* (measured on MI355X, frames/s of one / two 256-column frames per call: 1 workgroup per
     * (column, tile) 540 / 903, 2: 587 / 931, 3: 584 / -, 4: 561 / -) */
    static_assert(2 * IS_PW_SPLIT_MAX_COLS <= IS_PW_SPLIT_TARGET_WGS,
                  "the context reserves IS_PW_SPLIT_TARGET_WGS partial-minima slots for the split phase 1");
    int nsplit = ncols <= IS_PW_SPLIT_MAX_COLS ? 2 : 1;
    if (nsplit > IS_PW_MAX_SPLIT) nsplit = IS_PW_MAX_SPLIT;
    int groups = ncols / IS_PAIRWISE_SPLIT_MIN_COLS;
    groups = groups < 1 ? 1 : groups;
    /* measured on MI355X at batch 64 after the pruning of phase 1: 1 group 29.8 ms, 2 groups 30.2,
     * 4 groups 30.1, 8 groups 30.6 per step -- launches of different streams barely overlap, so the
     * default is one group; IS_PW_GROUPS overrides */
    if (groups > IS_PAIRWISE_MAX_GROUPS) groups = IS_PAIRWISE_MAX_GROUPS;
    if (groups > n_aux + 1) groups = n_aux + 1;
    if (P->knob_pw_groups >= 1 && P->knob_pw_groups <= n_aux + 1) groups = P->knob_pw_groups;
    hipError_t e;
/* the vB-side lutT row in registers (LutRow<2>, D <= 128): slower than the per-lane gather while
 * phase 1 was issue-bound (41.4 vs 38.0 ms per 64 frames, round 1), faster now that the pruned
 * phase 1 is latency-bound (30.4 vs 31.2 ms): no memory access on the chain mean -> fn -> value */
#ifndef IS_PW_PHASE1_ROW_REGS
#define IS_PW_PHASE1_ROW_REGS 1
#endif
#define IS_LAUNCH_P1(INV, c0, c1, st)                                                              \
    do {                                                                                           \
        if (win_t)                                                                                 \
            hipLaunchKernelGGL((k_pw_phase1<INV, 2, true>), dim3(((c1) - (c0)) * nsplit),          \
                               dim3(nw_t * 64), lds1_t, st, *P, c0, c1, tile, nsplit, recs, lutT,  \
                               steps, rcp, vhor, col_flags, prune, part_cost, part_idx, counters, \
                               joined, cost_T, blksum);                                            \
        else if (IS_PW_PHASE1_ROW_REGS && P->D <= 128)                                             \
            hipLaunchKernelGGL((k_pw_phase1<INV, 2>), dim3(((c1) - (c0)) * nsplit),                \
                               dim3(nw_t * 64), lds1_t, st, *P, c0, c1, tile, nsplit, recs, lutT,  \
                               steps, rcp, vhor, col_flags, prune, part_cost, part_idx, counters, \
                               joined, cost_T, blksum);                                            \
        else                                                                                       \
            hipLaunchKernelGGL((k_pw_phase1<INV, 0>), dim3(((c1) - (c0)) * nsplit),                \
                               dim3(nw_t * 64), lds1_t, st, *P, c0, c1, tile, nsplit, recs, lutT,  \
                               steps, rcp, vhor, col_flags, prune, part_cost, part_idx, counters,  \
                               joined, cost_T, blksum);                                            \
    } while (0)
#define IS_LAUNCH_P2(INV, c0, c1, st)                                                              \
    hipLaunchKernelGGL(k_pw_phase2<INV>, dim3((c1) - (c0)), dim3(64), lds2, st, *P, c0, c1, tile,  \
                       nsplit, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor, col_flags,      \
                       part_cost,                                                                  \
                       part_idx, steps, cost_table, index_table, blksum, t8row)
#define IS_LAUNCH_P2X(INV, c0, c1, st)                                                             \
    do {                                                                                           \
        hipLaunchKernelGGL(k_pw_phase2x<INV>, dim3(((c1) - (c0) + 1) / 2), dim3(64), lds2x, st, *P, c0, \
                           c1, tile, nsplit, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,   \
                           col_flags, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row); \
        hipLaunchKernelGGL(k_pw_phase2_generic<INV>, dim3(min((c1) - (c0), 512)), dim3(64), lds2, st, \
                           *P, c0, c1, tile, nsplit, recs, lutT, joined, priors, odr, rcp, sv_arr, \
                           vhor, col_flags, part_cost, part_idx, steps, cost_table, index_table,   \
                           n_generic, blksum, t8row);                                                     \
    } while (0)
#define IS_LAUNCH_P2S(INV, c0, c1, st)                                                             \
    hipLaunchKernelGGL(k_pw_phase2s<INV>, dim3((c1) - (c0)), dim3(ISP2S_WAVES * 64), lds2s, st, *P, \
                       c0, c1, tile, nsplit, recs, lutT, joined, priors, odr, rcp, sv_arr, vhor,   \
                       col_flags, part_cost, part_idx, steps, cost_table, index_table, blksum, t8row)
    const bool inv = P->invalid >= 0;
    /* phase 2 split over four waves per column (k_pw_phase2s) while the columns are too few to fill
     * the chip with one wave each: it shortens the serial chain of a column (one frame: 82 -> 74 us
     * per tile) but spends four wave slots per column, which costs throughput at large batches
     * (batch 64: 32.7 vs 25.4 ms per step).  IS_P2_SPLIT=0/1 overrides. */
    bool split2 = ncols <= IS_P2_SPLIT_MAX_COLS;
    if (P->knob_p2_split >= 0) split2 = P->knob_p2_split != 0;
    const size_t lds2s = isk_phase2s_lds_bytes(P);
    /* large batches: two columns per wave in phase 2 (k_pw_phase2x); needs an even number of
     * columns per image (a pair never straddles two images); IS_P2X=0 selects k_pw_phase2 */
    const size_t lds2x = isk_phase2x_lds_bytes(P);
    const bool two_col = !split2 && (P->C % 2) == 0 && P->knob_p2x != 0 && lds2x <= 64 * 1024;
    if (groups > 1) {
        if ((e = hipEventRecord(ev_fork, stream)) != hipSuccess) return e;
        for (int g = 1; g < groups; g++)
            if ((e = hipStreamWaitEvent(aux[g - 1], ev_fork, 0)) != hipSuccess) return e;
    }
    for (int tile = 0; tile < P->ntiles; tile++) {
        /* (the block bounds of tile t need t + 1 of the ntiles + 1 entries lds1 has room for) */
        /* (large batches only: a call of a few frames does not fill the chip, there the eight waves per
         * column are the parallelism: one frame 1.57 ms classic, 1.63 ms windowed; frames/s at batch 4 / 8 /
         * 16 / 32: 1461 / 2068 / 2795 / 3178 classic, 1424 / 2010 / 2798 / 3269 windowed.  IS_P1_WIN_TILES
         * forces the window for that many tiles at any batch: tests) */
        /* (any D: the windowed instantiation keeps 64 columns of a vB row in one register per lane) */
        const bool win_t = IS_P1_WINDOWED(P->D) &&
                           P->win_lo != nullptr && tile < P->win_tiles &&
                           (P->knob_win_tiles >= 0 || ncols >= IS_P1_WIN_MIN_COLS);
        const int nw_t = win_t ? nwaves_win : nwaves;
        const size_t lds1_t = p1_lds_bytes(P, nw_t, win_t) -
                              sizeof(float) * IS_P1_BLK_WORDS * (size_t)IS_QPT * (size_t)(P->ntiles - tile);
        for (int g = 0; g < groups; g++) {
            const int c0 = (int)((long long)ncols * g / groups) & ~1; /* (even: column pairs) */
            const int c1 = g + 1 == groups ? ncols : ((int)((long long)ncols * (g + 1) / groups) & ~1);
            hipStream_t st = g == 0 ? stream : aux[g - 1];
            if (inv) IS_LAUNCH_P1(true, c0, c1, st); else IS_LAUNCH_P1(false, c0, c1, st);
            if (split2) {
                if (inv) IS_LAUNCH_P2S(true, c0, c1, st); else IS_LAUNCH_P2S(false, c0, c1, st);
            } else if (two_col) {
                if (inv) IS_LAUNCH_P2X(true, c0, c1, st); else IS_LAUNCH_P2X(false, c0, c1, st);
            } else {
                if (inv) IS_LAUNCH_P2(true, c0, c1, st); else IS_LAUNCH_P2(false, c0, c1, st);
            }
        }
    }
#undef IS_LAUNCH_P1
#undef IS_LAUNCH_P2
#undef IS_LAUNCH_P2S
#undef IS_LAUNCH_P2X
    for (int g = 1; g < groups; g++) {
        if ((e = hipEventRecord(ev_join[g - 1], aux[g - 1])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ev_join[g - 1], 0)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t isk_set_lds_pairwise(const DevParams* P, int nwaves_pair) {
    hipError_t e;
    const int c = (int)isk_pairwise_lds_bytes(P, nwaves_pair);
    e = hipFuncSetAttribute((const void*)k_pw_phase1<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_pw_phase1<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, c);
    return e;
}

} /* extern "C" */
