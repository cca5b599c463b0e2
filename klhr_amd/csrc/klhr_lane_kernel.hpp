// klhr_lane_kernel.hpp — lane-per-chain kernel for small model dimension (D <= 26).
//
// Why a second mapping.  Profiling the octet kernel (profiles/r01_first) showed that at D = 10 the log-density work
// is small and the wavefront spends its instructions on per-chain *scalar* work (SciPy's optimiser logic, RNG,
// sinh-arcsinh scalars) that the octet mapping replicates in all 8 lanes of a chain.  Here every lane owns one
// chain (64 chains per wavefront) so nothing is replicated, and the 8 quadrature nodes of a KL evaluation become
// 8 independent accumulator chains in registers (model_eval_nv<MODEL, 8>): one pass over the chain's LDS-resident
// (theta_i, rho_i) pairs serves all 8 nodes, and the wavefront — alone on its SIMD at 65 536 chains per GPU, with up to
// 512 VGPRs — gets its latency hiding from instruction-level parallelism instead of occupancy.
//
// Same chain-state layout in HBM, same phases / probes / diagnostics as klhr_iter_kernel: the host chooses the
// kernel per model dimension (klhr_api.hip: use_lane_kernel), and the parity tests run both.
//
// LDS: per chain D double2 (theta, rho) + D doubles scratch, chain stride S ≡ 2 (mod 32) doubles so that the 16-lane
// groups of a ds_read_b128 hit 64 distinct banks.
#pragma once
#include "klhr_kernels.hpp"

namespace klhr {

#pragma clang fp contract(off)
// Lane-private rows of the chain state live in HBM ([chain][i] layout) and a wavefront alone on its SIMD has nothing
// to hide a load's latency behind, so rows are fetched in chunks of LANE_CH elements — all loads of a chunk in flight
// at once — into registers, worked on there, and stored back.  Arithmetic and its order are unchanged.
constexpr int LANE_CH = 13;        // the lane kernel serves D <= 26 (klhr_api.hip): two chunks

template <class T>
__device__ __forceinline__ void lane_load_chunk(const T *src, int stride, int i0, int D, T (&r)[LANE_CH]) {
#pragma unroll
    for (int k = 0; k < LANE_CH; k++) { const int i = i0 + k < D ? i0 + k : D - 1; r[k] = src[(long long)i * stride]; }
}

// klhr_sinh.py:203-213 for the chain owned by this lane (sequential, unfused: rounds like NumPy)
// Leaves r UNNORMALISED in TR[i].y and returns ||r + tol||: the caller takes the line sums from (theta, r) first — the split pipeline's
// arithmetic (klhr_split.hpp: sums on r, scaled by 1 / ||r|| afterwards), so that both mappings hand the fit the same bits — and then
// normalises (lane_normalise).
__device__ inline double lane_direction(const KParams &p, long long c, double2 *TR, const double *Z, double u_choice) {
    const int D = p.D, nc = p.ncol;
    const double *ev = p.eigvals + c * nc;
    const double *evec = p.eigvecs + c * (long long)D * nc;
    double pj[4] = {0, 0, 0, 0};
    double sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < nc) sum += ev[j];
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < nc) pj[j] = ev[j] / sum;
    int pick = -1;
    if (p.flags & FLAG_EIGEN_ONE) {
        double cdf[4], accu = 0, last = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { accu += pj[j]; cdf[j] = accu; if (j == nc - 1) last = accu; }
        pick = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < nc && cdf[j] / last <= u_choice) pick = j + 1;
        if (pick >= nc) pick = nc - 1;
    }
    double nn = 0;
    for (int i0 = 0; i0 < D; i0 += LANE_CH) {
        double e0[LANE_CH], e1[LANE_CH], cv[LANE_CH];
        int rk[LANE_CH];
        lane_load_chunk(evec, nc, i0, D, e0);
        if (nc > 1) lane_load_chunk(evec + 1, nc, i0, D, e1);
        lane_load_chunk(p.cov + c * (long long)D, 1, i0, D, cv);
        lane_load_chunk(p.rank + c * (long long)D, 1, i0, D, rk);
#pragma unroll
        for (int k = 0; k < LANE_CH; k++) {
            const int i = i0 + k;
            if (i < D) {
                double m = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j < nc) {
                        double e = j == 0 ? e0[k] : (j == 1 ? e1[k] : evec[i * nc + j]);
                        if (pick >= 0) { if (j == pick) m = e; }
                        else m += ((p.variant == VAR_GAUSS || p.variant == VAR_SLICE) ? ev[j] : pj[j]) * e;      // klhr.py:150
                    }
                }
                double r = m + sqrt(cv[k]) * Z[rk[k]];
                nn += (r + p.tol) * (r + p.tol);
                TR[i].y = r;
            }
        }
    }
    return sqrt(nn);
}
__device__ inline void lane_normalise(double2 *TR, int D, double nn) {
    for (int i = 0; i < D; i++) TR[i].y = TR[i].y / nn;
}

// onlinemoments.py:10-18
__device__ inline void lane_om_update(double *m, double *v, long long N, int D, const double2 *TR, const double *Z, bool fromZ, double clipb) {
    const double w = 1.0 / (double)N;
    for (int i0 = 0; i0 < D; i0 += LANE_CH) {
        double mr[LANE_CH], vr[LANE_CH];
        lane_load_chunk(m, 1, i0, D, mr);
        lane_load_chunk(v, 1, i0, D, vr);
#pragma unroll
        for (int k = 0; k < LANE_CH; k++) {
            const int i = i0 + k;
            if (i < D) {
                double x = fromZ ? clipc(Z[i], clipb) : TR[i].x;
                double mi = mr[k], vi = vr[k];
                double d = x - mi;
                mi += d * w;
                vi += -vi * w + d * d * w * (1 - w);
                m[i] = mi;
                v[i] = vi;
            }
        }
    }
}

// onlinepca.py:13-26; u = theta - mean in Z (LDS, lane-private), V rows in HBM (lane-private).  The column is held in
// registers (two chunks) across the three passes over it.
__device__ inline void lane_pca_update(const KParams &p, long long c, long long n_new, double *Z) {
    const int D = p.D, K = p.J;
    double *V = p.pca_v + c * (long long)D * K;
    const long long lim = n_new < K ? n_new : K;
    for (int col = 0; col < lim; col++) {
        if (col == n_new - 1) {
            for (int i = 0; i < D; i++) V[i * K + col] = Z[i];
        } else {
            const double w = ((double)(n_new - 1) - p.l) / (double)n_new;
            double va[LANE_CH], vb[LANE_CH];
            lane_load_chunk(V + col, K, 0, D, va);
            if (D > LANE_CH) lane_load_chunk(V + col, K, LANE_CH, D, vb);
            double s = 0, udv = 0;
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) { double vi = i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH]; s += vi * vi; }
            double nv = sqrt(s);
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) udv += Z[i] * (i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH]);
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) {
                double &vr = i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH];
                vr = w * vr + (1 - w) * Z[i] * udv / (nv + PCA_TOL);
                V[i * K + col] = vr;
            }
            s = 0; udv = 0;
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) { double vi = i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH]; s += vi * vi; }
            nv = sqrt(s);
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) udv += Z[i] * (i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH]);
#pragma unroll
            for (int i = 0; i < 2 * LANE_CH; i++) if (i < D) Z[i] = Z[i] - udv * (i < LANE_CH ? va[i % LANE_CH] : vb[i % LANE_CH]) / (nv * nv + PCA_TOL);
        }
    }
}

// klhr_sinh.py:268-280 (+ rank permutation for the following windows)
__device__ inline void lane_close_window(const KParams &p, long long c, long long omN, double *Z) {
    const int D = p.D, K = p.J, nc = p.ncol;
    double *mean = p.mean + c * (long long)D, *cov = p.cov + c * (long long)D;
    double *om_m = p.om_m + c * (long long)D, *om_v = p.om_v + c * (long long)D;
    double *V = p.pca_v + c * (long long)D * K;
    double *evec = p.eigvecs + c * (long long)D * nc, *ev = p.eigvals + c * nc;
    int *rank = p.rank + c * (long long)D;
    for (int i = 0; i < D; i++) {
        mean[i] = om_m[i];
        double cv = omN > 2 ? om_v[i] * (double)omN / (double)(omN - 1) : 1.0;
        if (p.flags & FLAG_SCALE_DIR_COV) {
            double *omd_m = p.omd_m + c * (long long)D, *omd_v = p.omd_v + c * (long long)D;
            double vd = omN > 2 ? omd_v[i] * (double)omN / (double)(omN - 1) : 1.0;
            cv = cv / (p.tol + vd);
            omd_m[i] = 0.0; omd_v[i] = 0.0;
        }
        cov[i] = cv;
        Z[i] = cv;
        om_m[i] = 0.0; om_v[i] = 0.0;
    }
    for (int i = 0; i < D; i++) {
        double ci = Z[i];
        int r = 0;
        for (int j = 0; j < D; j++) { double cj = Z[j]; r += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
        rank[i] = r;
    }
    double vals[4] = {0, 0, 0, 0};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < K) {
            double s = 0;
            for (int i = 0; i < D; i++) { double vi = V[i * K + j]; s += vi * vi; }
            vals[j] = sqrt(s);
            if (!isfinite(vals[j])) bad = true;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { if (bad) vals[j] = 0.0; vals[j] += PCA_TOL; }
    for (int i = 0; i < D; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) { evec[i * nc + j] = V[i * K + j] / vals[j]; V[i * K + j] = 0.0; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < K) ev[j] = vals[j];
}
#pragma clang fp contract(fast)

// KL(eta) value + gradient (klhr_sinh.py:163-176) as a sum over *node groups*.
//
// The 8 quadrature nodes are processed in groups of G in {4, 2, 1} register-resident accumulator chains
// (model_eval_nv<MODEL, G>; G = 2 is not instantiated at present); a group returns the partial sums (KL, dKL/deta0..3) of its nodes, and partials are combined
// in one fixed binary tree over the node index, ((0+1)+(2+3))+((4+5)+(6+7)).  Because of that fixed tree — and because
// every per-node quantity is computed by the same (unfused or explicitly fused) instruction sequence whatever G is —
// the result does not depend on *who* evaluates a group: the chain's own lane (two groups of 4, the normal case), or
// eight lanes of the wavefront, one node each, on its behalf (node-shared evaluation in the fit loop below).
#ifndef KLHR_NVG
#define KLHR_NVG 4
#endif

// extra LDS behind the 64 chain regions (doubles): published trial points [64][4], group partials [64][5],
// owner table [64] ints, quadrature table asinh(x)[8], w[8]
// (+ the requesters' line sums [64][7] for the closed-form evaluation of the separable targets)
constexpr int LANE_XT = 0, LANE_PART = 256, LANE_OWN = 576, LANE_Q = 608, LANE_ST = 624, LANE_LDS_EXTRA = 1072;

struct KLScal { double s_, invd, e21, esh; };

// per-evaluation scalars of the sinh-arcsinh family; SUB: d == 1, e = x2 (sub_klhr_sinh.py:92-97)
template <bool SUB>
__device__ __forceinline__ KLScal kl_scalars(const KParams &p, double x1, double x2, double x3) {
    KLScal k;
    k.s_ = exp(clipc(x1, p.scale_clip)) + p.tol;
    k.invd = SUB ? 1.0 : 1.0 / (exp(clipc(x2, p.scale_clip)) + p.tol);
    k.e21 = SUB ? -x1 : x2 - x1;
    k.esh = SUB ? x2 : x3;
    return k;
}

#pragma clang fp contract(off)
// contribution of one node to (KL, dKL/deta): explicit operations only, so every instantiation rounds identically
template <bool SUB>
__device__ __forceinline__ void kl_node_terms(double w, double a, double sh, double ch, double lch, double logp, double gdot,
                                              const KLScal &k, double (&c)[5]) {
    const double th = sh * fast_rcp(ch);
    const double laj = k.e21 - lch;
    c[0] = w * (laj - logp);
    const double wg = w * gdot;
    c[1] = -wg;
    c[2] = -fma(wg, k.s_ * sh, w);
    const double sc = k.s_ * ch;
    if (SUB) { c[3] = -fma(wg, sc, w * th); c[4] = 0.0; }
    else {
        c[3] = fma(wg, sc * a, w * fma(th, a, 1.0));
        c[4] = -fma(wg, sc * k.invd, w * (th * k.invd));
    }
}

// partial sums of nodes n0 .. n0+G-1 for the chain whose (theta, rho) pairs sit at TR; asx / wq point at the group's
// first node (kernel-argument tables on the normal path, the LDS copy when n0 is a per-lane value)
template <int MODEL, bool SUB, int G>
__device__ __forceinline__ void lane_kl_group(const KParams &p, const double2 *TR, double x0, const KLScal &k, const double *asx,
                                              const double *wq, double (&part)[5]) {
    double sh[G], ch[G], lch[G], t[G], av[G], logp[G], gdot[G];
#pragma unroll
    for (int n = 0; n < G; n++) {
        const double a = (asx[n] + k.esh) * k.invd;
        const double ac = clipc(a, p.scale_clip);
        // one exp serves sinh and cosh: |abs error| <= 2^-53 * cosh, the same size as the rounding of the sums
        // these values enter (KL value/gradient stay within the 1e-12 parity bound, tests/test_gpu_parity.py)
        const double E = exp(ac), Ei = fast_rcp(E);
        sh[n] = 0.5 * (E - Ei);
        ch[n] = 0.5 * (E + Ei);
        lch[n] = log_cosh_from_exp(fabs(ac), fmin(E, Ei));
        t[n] = fma(k.s_, sh[n], x0);
        av[n] = a;
    }
    model_eval_nv<MODEL, G>(p.md, TR, t, SUB ? p.grad_clip : p.scale_clip, logp, gdot);
    double c[G][5];
#pragma unroll
    for (int n = 0; n < G; n++) kl_node_terms<SUB>(wq[n], av[n], sh[n], ch[n], lch[n], logp[n], gdot[n], k, c[n]);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        if constexpr (G == 4) part[j] = (c[0][j] + c[1][j]) + (c[2][j] + c[3][j]);
        else if constexpr (G == 2) part[j] = c[0][j] + c[1][j];
        else part[j] = c[0][j];
    }
}
#pragma clang fp contract(fast)

// whole evaluation by the chain's own lane (probes; the fit loop inlines the same two groups)
template <int MODEL, bool SUB = false>
__device__ __forceinline__ void lane_eval_kl(const KParams &p, const double2 *TR, const double *xt, double &f, double *g) {
    const KLScal k = kl_scalars<SUB>(p, xt[1], xt[2], xt[3]);
    double P0[5], P1[5];
    lane_kl_group<MODEL, SUB, 4>(p, TR, xt[0], k, p.asx, p.wq, P0);
    lane_kl_group<MODEL, SUB, 4>(p, TR, xt[0], k, p.asx + 4, p.wq + 4, P1);
    f = add_rn(P0[0], P1[0]);
#pragma unroll
    for (int j = 0; j < 4; j++) g[j] = add_rn(P0[j + 1], P1[j + 1]);
}

// KLHR objective (klhr.py:109-124): 8 nodes y_n = s x_n + m, raw gradient — as node groups with the same fixed summation
// tree as lane_kl_group, so that the tail of a KLHR fit can be node-shared too.  part = (sum w logp, sum w g.rho,
// sum w g.rho x_n s) over the group's nodes; xq / wq point at the group's first node.
#pragma clang fp contract(off)
template <int MODEL, int G>
__device__ __forceinline__ void lane_gauss_group(const KParams &p, const double2 *TR, double m, double s_, const double *xq, const double *wq,
                                                 double (&part)[5]) {
    double t[G], logp[G], gdot[G];
#pragma unroll
    for (int n = 0; n < G; n++) t[n] = fma(s_, xq[n], m);
    model_eval_nv<MODEL, G>(p.md, TR, t, INFINITY, logp, gdot);
    double c[G][3];
#pragma unroll
    for (int n = 0; n < G; n++) {
        const double wgr = wq[n] * gdot[n];
        c[n][0] = wq[n] * logp[n];
        c[n][1] = wgr;
        c[n][2] = (wgr * xq[n]) * s_;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if constexpr (G == 4) part[j] = (c[0][j] + c[1][j]) + (c[2][j] + c[3][j]);
        else if constexpr (G == 2) part[j] = c[0][j] + c[1][j];
        else part[j] = c[0][j];
    }
    part[3] = part[4] = 0.0;
}
#pragma clang fp contract(fast)

template <int MODEL>
__device__ __forceinline__ void lane_eval_kl_gauss(const KParams &p, const double2 *TR, const double *xt, double &f, double *g) {
    const double s_ = exp(clipc(xt[1], p.scale_clip));
    double P0[5], P1[5];
    lane_gauss_group<MODEL, 4>(p, TR, xt[0], s_, p.xq, p.wq, P0);
    lane_gauss_group<MODEL, 4>(p, TR, xt[0], s_, p.xq + 4, p.wq + 4, P1);
    f = -(add_rn(P0[0], P1[0]) + xt[1]);
    g[0] = -add_rn(P0[1], P1[1]);
    g[1] = -(add_rn(P0[2], P1[2]) + 1.0);
    g[2] = g[3] = 0.0;
}

template <int MODEL>
__device__ __forceinline__ void lane_eval_line(const KParams &p, const double2 *TR, double tv, double &f, double *g) {
    double t[1] = {tv}, logp[1], gdot[1];
    model_eval_nv<MODEL, 1>(p.md, TR, t, INFINITY, logp, gdot);
    f = -logp[0]; g[0] = -gdot[0]; g[1] = g[2] = g[3] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// Closed-form line evaluation for the separable targets (funnel, ill-normal, normal).  Restricted to a line theta + t*rho
// their log density is a closed form in t through a handful of per-line sums
//     sum_i (a_i + t r_i)^2 = A + 2tB + t^2 C,      sum_i (a_i + t r_i) r_i = B + tC         (a = theta or theta/s, r = rho or rho/s),
// so after ONE O(D) pass per iteration every evaluation of a fit costs O(1) instead of O(D).  The element-wise gradient clip
// of _logp_grad (klhr_sinh.py:158-161) is inactive whenever max_i |g_i(t)| <= clip, which max|a|, max|r| bound; the rare point
// where the bound fails (measured on the reference: 1 in 2 000-5 000 KL evaluations) is recomputed by the element-wise loop.
// Used by this kernel for D <= 26 and by the split-phase pipeline (klhr_split.hpp) above that.
// (contract(on): FMAs are formed per expression in the front end — line_point is inlined into kernels that must agree bit for bit)
#pragma clang fp contract(on)
struct LineStat { double v0, r0, A, B, C, m0, m1; };

#define KLHR_FB(p, f, c) (p).fitbuf[(long long)(f) * (p).fstride + (c)]

__device__ __forceinline__ LineStat load_stat(const KParams &p, long long c) {
    LineStat s;
    s.v0 = KLHR_FB(p, F_S0, c); s.r0 = KLHR_FB(p, F_S1, c); s.A = KLHR_FB(p, F_S2, c); s.B = KLHR_FB(p, F_S3, c);
    s.C = KLHR_FB(p, F_S4, c); s.m0 = KLHR_FB(p, F_S5, c); s.m1 = KLHR_FB(p, F_S6, c);
    return s;
}

// log density and clip(grad).rho at theta + t*rho from the line sums.  Returns true when the element-wise clip at +-cl may
// be active at this point (the caller then recomputes the point with the element-wise loop); CLIP = false: no clipping
// (logp_grad_rho, klhr_sinh.py:178-180; klhr.py's raw gradient; log-density values of the MH step).
template <int MODEL, bool CLIP>
__device__ __forceinline__ bool line_point(const KParams &p, const LineStat &s, double t, double cl, double &logp, double &gdot) {
    double S = fma(t, fma(t, s.C, 2.0 * s.B), s.A);
    S = S < 0.0 ? 0.0 : S;                       // a sum of squares: rounding of the closed form may leave -1e-17
    const double Sr = fma(t, s.C, s.B);
    if constexpr (MODEL == MODEL_FUNNEL) {
        const double Da = (double)p.md.n;
        const double v = fma(t, s.r0, s.v0);
        const double ev = exp(-v);
        const double g0 = -v * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev * S;
        logp = -v * v * (1.0 / 18.0) - 0.5 * Da * v - 0.5 * ev * S;
        gdot = (CLIP ? clipc(g0, cl) : g0) * s.r0 - ev * Sr;
        if constexpr (CLIP) return !(ev * (s.m0 + fabs(t) * s.m1) <= cl);
        return false;
    } else {
        logp = -0.5 * S;
        gdot = -Sr;
        if constexpr (CLIP) return !(s.m0 + fabs(t) * s.m1 <= cl);
        return false;
    }
}


template <int MODEL>
constexpr bool is_separable() { return MODEL == MODEL_FUNNEL || MODEL == MODEL_ILL_NORMAL || MODEL == MODEL_NORMAL; }

// the line sums of the chain whose (theta_i, rho_i) pairs sit at TR (rho already normalised)
template <int MODEL>
__device__ __forceinline__ LineStat lane_line_stats(const KParams &p, const double2 *TR) {
    LineStat st = {0, 0, 0, 0, 0, 0, 0};
    const int D = p.D;
    for (int i = 0; i < D; i++) {
        const double2 e = TR[i];
        double a = e.x, r = e.y;
        if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = p.md.aux[i]; a = a * is; r = r * is; }
        if (MODEL == MODEL_FUNNEL && i == 0) { st.v0 = a; st.r0 = r; }
        else {
            st.A = fma(a, a, st.A); st.B = fma(a, r, st.B); st.C = fma(r, r, st.C);
            if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = p.md.aux[i]; a = a * is; r = r * is; }     // |g_i| = |x_i| / s_i^2
            st.m0 = fmax(st.m0, fabs(a)); st.m1 = fmax(st.m1, fabs(r));
            if (a != a || r != r) st.m0 = NAN;
        }
    }
    return st;
}

#pragma clang fp contract(off)
// the line sums of a freshly drawn direction, from (theta_i, r_i) with r still unnormalised and nn = ||r + tol||: the split pipeline's
// operations in the split pipeline's order (sums on r, then B / nn, C / nn^2, max|r| / nn, r_0 / nn)
template <int MODEL>
__device__ __forceinline__ LineStat lane_line_stats_unnormalised(const KParams &p, const double2 *TR, double nn) {
    LineStat st = lane_line_stats<MODEL>(p, TR);
    const double inv = 1.0 / nn;
    st.B = st.B * inv; st.C = st.C * inv * inv; st.m1 = st.m1 * inv;
    if (MODEL == MODEL_FUNNEL) st.r0 = st.r0 / nn;
    return st;
}
// contribution of ONE quadrature node to (KL, dKL/deta) from the line sums: per-node arithmetic identical whoever evaluates
// it (the chain's own lane, or another lane in the node-shared tail), exact element-wise loop where the clip bound fails
template <int MODEL, bool SUB>
__device__ __forceinline__ void cf_node(const KParams &p, const LineStat &st, const double2 *TR, double x0, const KLScal &k, double asx, double w,
                                        double (&c)[5]) {
    const double cl = SUB ? p.grad_clip : p.scale_clip;
    const double a = (asx + k.esh) * k.invd;
    const double ac = clipc(a, p.scale_clip);
    const double E = exp(ac), Ei = fast_rcp(E);
    const double sh = 0.5 * (E - Ei), ch = 0.5 * (E + Ei);
    const double lch = log_cosh_from_exp(fabs(ac), fmin(E, Ei));
    const double t = fma(k.s_, sh, x0);
    double logp, gdot;
    if (line_point<MODEL, true>(p, st, t, cl, logp, gdot)) {      // the rare node behind the clip bound: the element-wise loop, for this node only
        const double t1[1] = {t};
        double lp1[1], gd1[1];
        model_eval_nv<MODEL, 1>(p.md, TR, t1, cl, lp1, gd1);
        logp = lp1[0]; gdot = gd1[0];
    }
    kl_node_terms<SUB>(w, a, sh, ch, lch, logp, gdot, k, c);
}
// KLHR's objective (klhr.py:109-124): raw gradient, the closed form always applies
template <int MODEL>
__device__ __forceinline__ void cf_gauss_node(const KParams &p, const LineStat &st, double m, double s_, double xq, double w, double (&c)[5]) {
    const double t = fma(s_, xq, m);
    double logp, gdot;
    (void)line_point<MODEL, false>(p, st, t, INFINITY, logp, gdot);
    const double wgr = w * gdot;
    c[0] = w * logp;
    c[1] = wgr;
    c[2] = (wgr * xq) * s_;
    c[3] = c[4] = 0.0;
}
// the chain's own lane: all 8 nodes, summed in the fixed tree ((0+1)+(2+3))+((4+5)+(6+7)) of lane_kl_group.  The four nodes of
// a half are computed side by side without a branch between them (four independent dependency chains through exp / log cosh:
// the wavefront is alone on its SIMD and hides latency only through instruction-level parallelism); the rare node whose
// clip bound fails is patched afterwards with the element-wise loop.  Per node the operations are those of cf_node.
template <int MODEL, int VARIANT>
__device__ __forceinline__ void cf_eval_kl(const KParams &p, const LineStat &st, const double2 *TR, const double *xt, const double *qa, const double *qw,
                                           double &f, double *g) {
    constexpr bool SUB = VARIANT == VAR_SUB, GAUSS = VARIANT == VAR_GAUSS;
    const KLScal k = kl_scalars<SUB>(p, xt[1], xt[2], xt[3]);
    const double sg = GAUSS ? exp(clipc(xt[1], p.scale_clip)) : 0.0;
    const double cl = SUB ? p.grad_clip : p.scale_clip;
    double acc5[5] = {0, 0, 0, 0, 0};
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        double c[4][5];
        if constexpr (GAUSS) {
#pragma unroll
            for (int n = 0; n < 4; n++) cf_gauss_node<MODEL>(p, st, xt[0], sg, qa[4 * h + n], qw[4 * h + n], c[n]);
        } else {
            double av[4], sh[4], ch[4], lch[4], t[4], logp[4], gdot[4];
            bool need[4], any_need = false;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const double a = (qa[4 * h + n] + k.esh) * k.invd;
                const double ac = clipc(a, p.scale_clip);
                const double E = exp(ac), Ei = fast_rcp(E);
                sh[n] = 0.5 * (E - Ei); ch[n] = 0.5 * (E + Ei);
                lch[n] = log_cosh_from_exp(fabs(ac), fmin(E, Ei));
                t[n] = fma(k.s_, sh[n], xt[0]);
                av[n] = a;
                need[n] = line_point<MODEL, true>(p, st, t[n], cl, logp[n], gdot[n]);
                any_need = any_need || need[n];
            }
            if (__any(any_need)) {
                // (only the nodes whose own bound failed: the rule of every form of the KL evaluation, klhr_split.hpp included)
#pragma unroll
                for (int n = 0; n < 4; n++) {
                    if (need[n]) {
                        const double t1[1] = {t[n]};
                        double lp1[1], gd1[1];
                        model_eval_nv<MODEL, 1>(p.md, TR, t1, cl, lp1, gd1);
                        logp[n] = lp1[0]; gdot[n] = gd1[0];
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < 4; n++) kl_node_terms<SUB>(qw[4 * h + n], av[n], sh[n], ch[n], lch[n], logp[n], gdot[n], k, c[n]);
        }
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double part = (c[0][j] + c[1][j]) + (c[2][j] + c[3][j]);
            acc5[j] = h == 0 ? part : acc5[j] + part;
        }
    }
    if constexpr (GAUSS) { f = -(acc5[0] + xt[1]); g[0] = -acc5[1]; g[1] = -(acc5[2] + 1.0); g[2] = g[3] = 0.0; }
    else { f = acc5[0]; g[0] = acc5[1]; g[1] = acc5[2]; g[2] = acc5[3]; g[3] = acc5[4]; }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// OVERRELAX is a compile-time switch: the over-relaxed proposal's code (binomial/beta loops, Phi, Phi^-1) would otherwise
// sit in the register allocation of every launch (measured: +1.5 KB scratch per lane, -8 % throughput).
template <int MODEL, bool OVERRELAX, int VARIANT>
__global__ __launch_bounds__(64, 1) void klhr_lane_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const long long c0 = (long long)blockIdx.x * 64;
    const long long c = c0 + lane;
    const bool active = c < p.C;
    const int nch = (p.C - c0) < 64 ? (int)(p.C - c0) : 64;
    const int D = p.D;
    double2 *TR = reinterpret_cast<double2 *>(lds + (size_t)lane * p.S);
    double *Z = lds + (size_t)lane * p.S + 2 * (size_t)D;
    double *shx = lds + (size_t)64 * p.S;            // node-shared evaluation: see the fit loop
    if (lane < 8) { shx[LANE_Q + lane] = VARIANT == VAR_GAUSS ? p.xq[lane] : p.asx[lane]; shx[LANE_Q + 8 + lane] = p.wq[lane]; }

    // ---- cooperative, coalesced load of the block's theta rows into LDS
    {
        const double *src = p.theta + c0 * (long long)D;
        int cc = 0, i = lane;
        while (i >= D) { i -= D; cc++; }
        for (int idx = lane; idx < nch * D; idx += 64) {
            double2 v; v.x = src[idx]; v.y = 0.0;
            reinterpret_cast<double2 *>(lds + (size_t)cc * p.S)[i] = v;
            i += 64;
            while (i >= D) { i -= D; cc++; }
        }
    }
    Pcg64 rng = {0, 0, 0, 0};
    long long draw = 0, omN = 0, pcan = 0, gevals = 0;
    int cidx = 0;
    double acc = 0;
    double sm_x = 0, sm_cnt = 0, moved_f = 0;
    int Kc = p.K;
    if (VARIANT == VAR_GAUSS && active) { sm_x = p.kstate[c * 4 + 0]; sm_cnt = p.kstate[c * 4 + 1]; Kc = (int)p.kstate[c * 4 + 2]; moved_f = p.kstate[c * 4 + 3]; }
    if (active) {
        rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
        draw = p.counters[c * 4 + 0]; omN = p.counters[c * 4 + 1]; pcan = p.counters[c * 4 + 2]; cidx = (int)p.counters[c * 4 + 3];
        acc = p.acc[c];
        gevals = p.grad_evals[c];
    }
    KLHR_WAVE_SYNC();

    double eta[4] = {0, 0, 0, 0};
    bool fit_capped = false;            // KLHR_MAX_PASSES reached (klhr_kernels.hpp)
    // diagnostics go straight to memory when requested (wave-uniform flag): they are not carried in registers
    const bool diag = (p.flags & FLAG_DIAG) && !(p.phases & (PH_KL | PH_LINE));
    double *dd = p.d_diag + (active ? c : 0) * KLHR_NDIAG;

#ifdef KLHR_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter(), st_dir = 0, st_line = 0, st_kl = 0, st_klsh = 0, st_adv = 0, st_mh = 0,
                       st_adapt = 0, st_out = 0, st_trips = 0, st_shtrips = 0, st_kltrips = 0;
#endif
    for (int iter = 0; iter < p.n_iters; ++iter) {
        KLHR_STAMP(st_out);
        double nn_dir = 1.0;
        // =============================================================== direction
        if (active) {
            if (p.phases & PH_DIR) {
                const double u_choice = (p.flags & FLAG_EIGEN_ONE) ? pcg_next_double(rng) : 0.0;
                for (int k = 0; k < D; k++) Z[k] = rng_std_normal_call(rng);
                nn_dir = lane_direction(p, c, TR, Z, u_choice);
                // (separable target with the fit in this launch: the fit's head takes the line sums from the unnormalised direction first)
                if (!(is_separable<MODEL>() && (p.phases & PH_FIT))) lane_normalise(TR, D, nn_dir);
                if (p.flags & FLAG_DIAG) {
                    double *rg = p.d_rho + c * (long long)D;
                    const bool unn = is_separable<MODEL>() && (p.phases & PH_FIT);
                    for (int i = 0; i < D; i++) rg[i] = unn ? TR[i].y / nn_dir : TR[i].y;
                }
            } else {
                const double *rg = p.d_rho + c * (long long)D;
                for (int i = 0; i < D; i++) TR[i].y = rg[i];
            }
        }

        KLHR_STAMP(st_dir);
        // =============================================================== fit
        if (p.phases & PH_FIT) {
          if constexpr (is_separable<MODEL>()) {
            // ---- separable targets: closed-form evaluations from the line sums (one O(D) pass), the two stages of fit() as
            // two loops — the mode search runs opt_advance<N = 1> and never pays for a KL evaluation
            constexpr bool SUB = VARIANT == VAR_SUB, GAUSS = VARIANT == VAR_GAUSS;
            constexpr int NFIT = GAUSS ? 2 : (SUB ? 3 : 4);
            // (a direction drawn in this launch is still unnormalised: its line sums are the split pipeline's — sums on r, scaled by 1 / ||r|| —,
            //  then it is normalised; a given direction — klhr_fit — is normalised already: nn = 1)
            LineStat st = {0, 0, 0, 0, 0, 0, 0};
            if (active) {
                if (p.phases & PH_DIR) { st = lane_line_stats_unnormalised<MODEL>(p, TR, nn_dir); lane_normalise(TR, D, nn_dir); }
                else st = lane_line_stats<MODEL>(p, TR);
            }
            {
                double *sl = shx + LANE_ST + lane * 7;
                sl[0] = st.v0; sl[1] = st.r0; sl[2] = st.A; sl[3] = st.B; sl[4] = st.C; sl[5] = st.m0; sl[6] = st.m1;
            }
            Opt o;
            double fit_n2 = 0, fit_n3 = 0;
            {
                double x0v[4] = {0, 0, 0, 0};
                if (active) {
                    x0v[0] = rng_std_normal_call(rng) * p.initscale;
                    if (VARIANT != VAR_GAUSS) {              // klhr.py's fit draws no further normals (klhr.py:134-139)
                        (void)rng_std_normal_call(rng);
                        (void)rng_std_normal_call(rng);
                        fit_n2 = rng_std_normal_call(rng) * p.initscale;
                        if (VARIANT == VAR_SINH) fit_n3 = rng_std_normal_call(rng) * p.initscale;     // size = 3 for SUBKLHRSINH
                    }
                }
                if (diag && active) dd[KLHR_D_X0] = x0v[0];
                opt_begin(o, 1, x0v, 1e-5);
            }
            int npass = 0, trips = 0;
            bool done = !active;
            // ---- stage 1: minimize(logp_grad_rho, x0, BFGS, gtol = 1e-5) (klhr_sinh.py:183-190)
            for (;;) {
                if (!done) {
                    double f = 0, g[4] = {0, 0, 0, 0};
                    if (opt_cache_hit(o)) { f = o.lf; g[0] = o.lg[0]; }
                    else {
                        double lp, gd;
                        (void)line_point<MODEL, false>(p, st, o.xt[0], INFINITY, lp, gd);
                        f = -lp; g[0] = -gd;
                        opt_cache_store(o, f, g);
                        npass++;
                    }
                    opt_advance<false, 1>(o, f, g);
                    if (o.pc == PC_DONE) done = true;
                }
                if (__all(done) || ++trips > KLHR_MAX_PASSES) break;
            }
            KLHR_STAMP(st_line);          // (stamps build: the whole mode search, evaluations + optimiser steps)
            if (active) {
                if (!done) fit_capped = true;
                gevals += o.nfev;
                if (diag) {
                    dd[KLHR_D_MODE] = o.xk[0]; dd[KLHR_D_HINV1] = o.H[0]; dd[KLHR_D_NFEV1] = o.nfev; dd[KLHR_D_NIT1] = o.k;
                    dd[KLHR_D_STATUS1] = o.status; dd[KLHR_D_FALLBACKS] = o.fallbacks;
                }
                double s1 = o.H[0];
                s1 = (s1 > 0 ? 1.0 : 0.0) * 0.5 * log(s1);
                double init[4] = {o.xk[0], s1, fit_n2, fit_n3};
                if (diag) { dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = init[2]; dd[KLHR_D_INIT3] = init[3]; }
                opt_begin(o, NFIT, init, GAUSS ? 1e-5 : 1e-3);
            }
            // ---- stage 2: minimize(KL, init, BFGS) (klhr_sinh.py:191-201), node-shared in the tail of the fit (see below)
            done = !active;
            trips = 0;
            for (;;) {
                double f = 0, g[4] = {0, 0, 0, 0};
                const bool hit = !done && opt_cache_hit(o);
                const bool ev4 = !done && !hit;
                const unsigned long long m4 = __ballot(ev4);
                const int n4 = __popcll(m4);
                KLHR_STAMP(st_adv);
#ifdef KLHR_STAMPS
                st_trips++;
                if (n4 > 0) st_kltrips++;
                if (n4 > 0 && n4 <= 8) st_shtrips++;
#endif
                if (n4 > 8) {
                    if (ev4) cf_eval_kl<MODEL, VARIANT>(p, st, TR, o.xt, shx + LANE_Q, shx + LANE_Q + 8, f, g);
                    KLHR_STAMP(st_kl);
                } else if (n4 > 0) {
                    // at most 8 chains still fitting: each request is served by 8 lanes, one quadrature node each (line sums and
                    // trial point published in LDS), partial terms back through LDS, summed by the owner in the same fixed tree
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m4 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m4, 0u));
                    if (ev4) {
                        reinterpret_cast<int *>(shx + LANE_OWN)[rank] = lane;
#pragma unroll
                        for (int j = 0; j < 4; j++) shx[LANE_XT + lane * 4 + j] = o.xt[j];
                    }
                    KLHR_WAVE_SYNC();
                    const int r = lane >> 3, node = lane & 7;
                    if (r < n4) {
                        const int ow = reinterpret_cast<const int *>(shx + LANE_OWN)[r];
                        const double x0 = shx[LANE_XT + ow * 4], x1 = shx[LANE_XT + ow * 4 + 1], x2 = shx[LANE_XT + ow * 4 + 2],
                                     x3 = shx[LANE_XT + ow * 4 + 3];
                        const double *sl = shx + LANE_ST + ow * 7;
                        const LineStat so = {sl[0], sl[1], sl[2], sl[3], sl[4], sl[5], sl[6]};
                        const double2 *TRo = reinterpret_cast<const double2 *>(lds + (size_t)ow * p.S);
                        double part[5];
                        if constexpr (GAUSS) cf_gauss_node<MODEL>(p, so, x0, exp(clipc(x1, p.scale_clip)), shx[LANE_Q + node], shx[LANE_Q + 8 + node], part);
                        else {
                            const KLScal k = kl_scalars<SUB>(p, x1, x2, x3);
                            cf_node<MODEL, SUB>(p, so, TRo, x0, k, shx[LANE_Q + node], shx[LANE_Q + 8 + node], part);
                        }
#pragma unroll
                        for (int j = 0; j < 5; j++) shx[LANE_PART + lane * 5 + j] = part[j];
                    }
                    KLHR_WAVE_SYNC();
                    if (ev4) {
                        const double *pp = shx + LANE_PART + rank * 40;
                        double r5[5];
#pragma unroll
                        for (int j = 0; j < 5; j++)
                            r5[j] = add_rn(add_rn(add_rn(pp[j], pp[5 + j]), add_rn(pp[10 + j], pp[15 + j])),
                                              add_rn(add_rn(pp[20 + j], pp[25 + j]), add_rn(pp[30 + j], pp[35 + j])));
                        if constexpr (GAUSS) { f = -(r5[0] + o.xt[1]); g[0] = -r5[1]; g[1] = -(r5[2] + 1.0); g[2] = g[3] = 0.0; }
                        else { f = r5[0]; g[0] = r5[1]; g[1] = r5[2]; g[2] = r5[3]; g[3] = r5[4]; }
                    }
                    KLHR_STAMP(st_klsh);
                }
                if (!done) {
                    if (hit) { f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3]; }
                    else { opt_cache_store(o, f, g); npass++; }
                    opt_advance<false, NFIT>(o, f, g);
                    if (o.pc == PC_DONE) {
                        gevals += (long long)o.nfev * 8;
                        for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                        if (diag) {
                            dd[KLHR_D_NFEV2] = o.nfev; dd[KLHR_D_NIT2] = o.k; dd[KLHR_D_STATUS2] = o.status; dd[KLHR_D_FUN2] = o.old_fval;
                            dd[KLHR_D_FALLBACKS] += o.fallbacks; dd[KLHR_D_PASSES] = npass;
                        }
                        done = true;
                    }
                }
                if (__all(done) || ++trips > KLHR_MAX_PASSES) break;
            }
            if (!done) { fit_capped = true; for (int k = 0; k < 4; k++) eta[k] = o.xk[k]; }
            if (active && (p.flags & FLAG_DIAG)) {
                for (int k = 0; k < 4; k++) p.d_eta[c * 4 + k] = eta[k];
            }
          } else {
            Opt o;
            // fit() consumes exactly five normals — x0, then size-4 init of which the first two are discarded
            // (klhr_sinh.py:184,191-193) — and nothing in between touches the generator, so all five are drawn here,
            // with the whole wavefront converged, instead of inside the divergent stage-1 -> stage-2 transition.
            double fit_n2 = 0, fit_n3 = 0;
            {
                double x0v[4] = {0, 0, 0, 0};
                if (active) {
                    x0v[0] = rng_std_normal_call(rng) * p.initscale;
                    if (VARIANT != VAR_GAUSS) {              // klhr.py's fit draws no further normals (klhr.py:134-139)
                        (void)rng_std_normal_call(rng);
                        (void)rng_std_normal_call(rng);
                        fit_n2 = rng_std_normal_call(rng) * p.initscale;
                        if (VARIANT == VAR_SINH) fit_n3 = rng_std_normal_call(rng) * p.initscale;     // size = 3 for SUBKLHRSINH
                    }
                }
                if (diag && active) dd[KLHR_D_X0] = x0v[0];
                opt_begin(o, 1, x0v, 1e-5);
            }
            bool done = !active;
            int npass = 0, trips = 0;
            for (;;) {
                double f = 0, g[4] = {0, 0, 0, 0};
                const bool hit = !done && opt_cache_hit(o);
                const bool ev1 = !done && !hit && o.n == 1;
                const bool ev4 = !done && !hit && o.n != 1;
                KLHR_STAMP(st_adv);
                if (ev1) lane_eval_line<MODEL>(p, TR, o.xt[0], f, g);
                KLHR_STAMP(st_line);
#ifdef KLHR_STAMPS
                st_trips++;
#endif
                {
                    // ---- KL evaluation, node-shared in the tail of the fit.
                    // Chains finish their fits after different numbers of evaluations (mean ~20 passes, slowest of 64
                    // ~30; in 40 % of the passes at most 8 chains of the wavefront are still fitting).  With n4 <= 8
                    // requesting chains each request is served by 8 lanes: lane j evaluates quadrature node j % 8 of
                    // the chain ranked j / 8 among the requesters, reading that chain's (theta, rho) from its LDS
                    // region, and the owner adds the 8 partials in the fixed tree of lane_kl_group.  (Serving 9..32
                    // requesters with 4 or 2 lanes each was measured too: such passes are rare, no gain.)
                    constexpr bool SUB = VARIANT == VAR_SUB, GAUSS = VARIANT == VAR_GAUSS;
                    const unsigned long long m4 = __ballot(ev4);
                    const int n4 = __popcll(m4);
#ifdef KLHR_STAMPS
                    if (n4 > 0) st_kltrips++;
                    if (n4 > 0 && n4 <= 8) st_shtrips++;
#endif
                    if (n4 > 8) {
                        if (ev4) {
                            const KLScal k = kl_scalars<SUB>(p, o.xt[1], o.xt[2], o.xt[3]);      // (s_ = exp(clip(eta1)) serves KLHR)
                            const double sg = GAUSS ? exp(clipc(o.xt[1], p.scale_clip)) : 0.0;
                            double acc5[5] = {0, 0, 0, 0, 0};
#pragma unroll 1
                            for (int h = 0; h < 2; h++) {
                                double part[5];
                                if constexpr (GAUSS) lane_gauss_group<MODEL, 4>(p, TR, o.xt[0], sg, shx + LANE_Q + 4 * h, shx + LANE_Q + 8 + 4 * h, part);
                                else lane_kl_group<MODEL, SUB, 4>(p, TR, o.xt[0], k, shx + LANE_Q + 4 * h, shx + LANE_Q + 8 + 4 * h, part);
#pragma unroll
                                for (int j = 0; j < 5; j++) acc5[j] = h == 0 ? part[j] : add_rn(acc5[j], part[j]);
                            }
                            if constexpr (GAUSS) { f = -(acc5[0] + o.xt[1]); g[0] = -acc5[1]; g[1] = -(acc5[2] + 1.0); g[2] = g[3] = 0.0; }
                            else { f = acc5[0]; g[0] = acc5[1]; g[1] = acc5[2]; g[2] = acc5[3]; g[3] = acc5[4]; }
                        }
                        KLHR_STAMP(st_kl);
                    } else if (n4 > 0) {
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m4 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m4, 0u));
                        if (ev4) {
                            reinterpret_cast<int *>(shx + LANE_OWN)[rank] = lane;
#pragma unroll
                            for (int j = 0; j < 4; j++) shx[LANE_XT + lane * 4 + j] = o.xt[j];
                        }
                        KLHR_WAVE_SYNC();
                        const int r = lane >> 3, node = lane & 7;
                        if (r < n4) {
                            const int ow = reinterpret_cast<const int *>(shx + LANE_OWN)[r];
                            const double x0 = shx[LANE_XT + ow * 4], x1 = shx[LANE_XT + ow * 4 + 1], x2 = shx[LANE_XT + ow * 4 + 2],
                                         x3 = shx[LANE_XT + ow * 4 + 3];
                            const double2 *TRo = reinterpret_cast<const double2 *>(lds + (size_t)ow * p.S);
                            double part[5];
                            if constexpr (GAUSS) lane_gauss_group<MODEL, 1>(p, TRo, x0, exp(clipc(x1, p.scale_clip)), shx + LANE_Q + node, shx + LANE_Q + 8 + node, part);
                            else {
                                const KLScal k = kl_scalars<SUB>(p, x1, x2, x3);
                                lane_kl_group<MODEL, SUB, 1>(p, TRo, x0, k, shx + LANE_Q + node, shx + LANE_Q + 8 + node, part);
                            }
#pragma unroll
                            for (int j = 0; j < 5; j++) shx[LANE_PART + lane * 5 + j] = part[j];
                        }
                        KLHR_WAVE_SYNC();
                        if (ev4) {
                            const double *pp = shx + LANE_PART + rank * 40;
                            double r5[5];
#pragma unroll
                            for (int j = 0; j < 5; j++)
                                r5[j] = add_rn(add_rn(add_rn(pp[j], pp[5 + j]), add_rn(pp[10 + j], pp[15 + j])),
                                                  add_rn(add_rn(pp[20 + j], pp[25 + j]), add_rn(pp[30 + j], pp[35 + j])));
                            if constexpr (GAUSS) { f = -(r5[0] + o.xt[1]); g[0] = -r5[1]; g[1] = -(r5[2] + 1.0); g[2] = g[3] = 0.0; }
                            else { f = r5[0]; g[0] = r5[1]; g[1] = r5[2]; g[2] = r5[3]; g[3] = r5[4]; }
                        }
                        KLHR_STAMP(st_klsh);
                    }
                }
                if (!done) {
                    if (hit) { f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3]; }
                    else { opt_cache_store(o, f, g); npass++; }
                    opt_advance(o, f, g);
                    if (o.pc == PC_DONE) {
                        if (o.n == 1) {
                            gevals += o.nfev;
                            if (diag) {
                                dd[KLHR_D_MODE] = o.xk[0]; dd[KLHR_D_HINV1] = o.H[0]; dd[KLHR_D_NFEV1] = o.nfev; dd[KLHR_D_NIT1] = o.k;
                                dd[KLHR_D_STATUS1] = o.status; dd[KLHR_D_FALLBACKS] = o.fallbacks;
                            }
                            double s = o.H[0];
                            s = (s > 0 ? 1.0 : 0.0) * 0.5 * log(s);
                            double init[4] = {o.xk[0], s, fit_n2, fit_n3};
                            if (diag) { dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = init[2]; dd[KLHR_D_INIT3] = init[3]; }
                            if (VARIANT == VAR_GAUSS) opt_begin(o, 2, init, 1e-5);      // klhr.py:136-139: default gtol
                            else opt_begin(o, VARIANT == VAR_SUB ? 3 : 4, init, 1e-3);
                        } else {
                            gevals += (long long)o.nfev * 8;
                            for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                            if (diag) {
                                dd[KLHR_D_NFEV2] = o.nfev; dd[KLHR_D_NIT2] = o.k; dd[KLHR_D_STATUS2] = o.status; dd[KLHR_D_FUN2] = o.old_fval;
                                dd[KLHR_D_FALLBACKS] += o.fallbacks; dd[KLHR_D_PASSES] = npass;
                            }
                            done = true;
                        }
                    }
                }
                if (__all(done) || ++trips > KLHR_MAX_PASSES) break;
            }
            if (!done) { fit_capped = true; for (int k = 0; k < 4; k++) eta[k] = o.xk[k]; }
            if (active && (p.flags & FLAG_DIAG)) {
                for (int k = 0; k < 4; k++) p.d_eta[c * 4 + k] = eta[k];
            }
          }
        } else if (active) {
            for (int k = 0; k < 4; k++) eta[k] = p.d_eta[c * 4 + k];
        }

        // =============================================================== KL / line probes
        if ((p.phases & (PH_KL | PH_LINE)) && active) {
            double f, g[4];
            if constexpr (is_separable<MODEL>()) {
                const LineStat stp = lane_line_stats<MODEL>(p, TR);
                if (p.phases & PH_KL) cf_eval_kl<MODEL, VARIANT>(p, stp, TR, eta, shx + LANE_Q, shx + LANE_Q + 8, f, g);
                else {
                    double lp, gd;
                    (void)line_point<MODEL, false>(p, stp, eta[0], INFINITY, lp, gd);
                    f = -lp; g[0] = -gd; g[1] = g[2] = g[3] = 0.0;
                }
            } else if (p.phases & PH_KL) {
                if constexpr (VARIANT == VAR_GAUSS) lane_eval_kl_gauss<MODEL>(p, TR, eta, f, g);
                else lane_eval_kl<MODEL, VARIANT == VAR_SUB>(p, TR, eta, f, g);
            } else lane_eval_line<MODEL>(p, TR, eta[0], f, g);
            double *dp = p.d_diag + c * KLHR_NDIAG;
            dp[0] = f; dp[1] = g[0]; dp[2] = g[1]; dp[3] = g[2]; dp[4] = g[3];
        }

        KLHR_STAMP(st_adv);
        // =============================================================== MH step
        if ((p.phases & PH_MH) && active) {
            draw += 1;
            double zp;
            if constexpr (VARIANT == VAR_GAUSS && OVERRELAX) { const ZpOut zo = gauss_overrelaxed_zp_call(rng, eta[0], eta[1], Kc, p.scale_clip); zp = zo.zp; rng = zo.rng; }
            else if constexpr (VARIANT == VAR_GAUSS) zp = gauss_zp<false>(rng, eta, Kc, p.scale_clip);
            else if constexpr (OVERRELAX) { const ZpOut zo = overrelaxed_zp_call(rng, eta[0], eta[1], eta[2], eta[3], p.K, p.tol, p.scale_clip, VARIANT == VAR_SUB); zp = zo.zp; rng = zo.rng; }
            else zp = sa_T(rng_std_normal_call(rng), eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
            double lp2[2], gd2[2];
            if constexpr (is_separable<MODEL>()) {
                // (the line sums of this iteration's direction: the fit's own, from the lane's LDS slot, when this launch drew the direction
                //  and ran the fit; a stand-alone MH launch — klhr_metropolis — has a given, normalised direction)
                LineStat stm;
                if ((p.phases & PH_DIR) && (p.phases & PH_FIT)) {
                    const double *sl = shx + LANE_ST + lane * 7;
                    stm.v0 = sl[0]; stm.r0 = sl[1]; stm.A = sl[2]; stm.B = sl[3]; stm.C = sl[4]; stm.m0 = sl[5]; stm.m1 = sl[6];
                } else stm = lane_line_stats<MODEL>(p, TR);
                (void)line_point<MODEL, false>(p, stm, zp, INFINITY, lp2[0], gd2[0]);
                (void)line_point<MODEL, false>(p, stm, 0.0, INFINITY, lp2[1], gd2[1]);
            } else {
                double t2[2] = {zp, 0.0};
                model_eval_nv<MODEL, 2>(p.md, TR, t2, INFINITY, lp2, gd2);
            }
            double r = lp2[0];
            r -= lp2[1];
            if constexpr (VARIANT == VAR_GAUSS) {
                const double gm = eta[0], gs = exp(clipc(eta[1], p.scale_clip));
                r += gauss_logq(0.0, gm, gs);
                r -= gauss_logq(zp, gm, gs);
            } else {
                r += sa_log_q(0.0, eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
                r -= sa_log_q(zp, eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
            }
            const double logu = log(rng_uniform(rng));
            const double mn = (r != r) ? r : (r < 0 ? r : 0.0);
            const int a = logu < mn ? 1 : 0;
            const double ad = (double)a, na = (double)(1 - a);
            double d2 = 0.0;
            for (int i = 0; i < D; i++) {
                double2 v = TR[i];
                double thp = add_rn(mul_rn(zp, v.y), v.x);
                double nx = add_rn(mul_rn(ad, thp), mul_rn(na, v.x));
                if (VARIANT == VAR_GAUSS) d2 += (nx - v.x) * (nx - v.x);
                v.x = nx;
                TR[i] = v;
            }
            if (VARIANT == VAR_GAUSS) moved_f = (sqrt(d2) > 0.0) ? 1.0 : 0.0;
            acc += (ad - acc) / (double)draw;
            if (diag) { dd[KLHR_D_ZP] = zp; dd[KLHR_D_R] = r; dd[KLHR_D_LOGU] = logu; dd[KLHR_D_ACCEPT] = a; }
        }

        KLHR_STAMP(st_mh);
        // =============================================================== closure-or-update
        if ((p.phases & PH_ADAPT) && active) {
            bool closed = false;
            if (!(p.warmup < p.windowsize) && p.nclos > 0) {
                closed = draw == p.closures[cidx];
                if (closed && cidx < p.nclos - 1) cidx++;
            }
            if (closed) {
                lane_close_window(p, c, omN, Z);
                omN = 0;
                pcan = 0;
                if (VARIANT == VAR_GAUSS) {      // klhr.py:212-214
                    Kc = (int)(sm_x < 1.0 ? 1.0 : (sm_x > 50.0 ? 50.0 : sm_x));
                    sm_x = (double)p.K; sm_cnt = 0.0;
                }
            } else {
                omN += 1;
                if (VARIANT == VAR_GAUSS) smoother_update(sm_x, sm_cnt, moved_f != 0.0);
                if (p.flags & FLAG_SCALE_DIR_COV) {
                    double lp_, gd_;
                    model_eval<MODEL>(p.md, [&](int i, double &xi, double &ri) { xi = TR[i].x; ri = 0.0; }, INFINITY,
                                      [&](int i, double gi) { Z[i] = gi; }, lp_, gd_);
                    lane_om_update(p.omd_m + c * (long long)D, p.omd_v + c * (long long)D, omN, D, TR, Z, true,
                                   VARIANT == VAR_GAUSS ? INFINITY : (VARIANT == VAR_SUB ? p.grad_clip : p.scale_clip));
                }
                lane_om_update(p.om_m + c * (long long)D, p.om_v + c * (long long)D, omN, D, TR, Z, false, 0.0);
                for (int i0 = 0; i0 < D; i0 += LANE_CH) {
                    double mr[LANE_CH];
                    lane_load_chunk(p.mean + c * (long long)D, 1, i0, D, mr);
#pragma unroll
                    for (int k = 0; k < LANE_CH; k++) if (i0 + k < D) Z[i0 + k] = sub_rn(TR[i0 + k].x, mr[k]);
                }
                pcan += 1;
                lane_pca_update(p, c, pcan, Z);
            }
        }

        KLHR_STAMP(st_adapt);
        // =============================================================== draws: cooperative, coalesced store
        if (p.draws) {
            KLHR_WAVE_SYNC();
            double *dst = p.draws + ((long long)iter * p.C + c0) * D;
            int cc = 0, i = lane;
            while (i >= D) { i -= D; cc++; }
            for (int idx = lane; idx < nch * D; idx += 64) {
                dst[idx] = reinterpret_cast<const double2 *>(lds + (size_t)cc * p.S)[i].x;
                i += 64;
                while (i >= D) { i -= D; cc++; }
            }
            KLHR_WAVE_SYNC();
        }
    }

#ifdef KLHR_STAMPS
    if (lane == 0 && (p.flags & FLAG_DIAG)) {      // diagnostic build: cycle totals of the wavefront's phases in its first chain's diag row
        double *ds = p.d_diag + c0 * KLHR_NDIAG;
        ds[0] = (double)st_dir; ds[1] = (double)st_line; ds[2] = (double)st_kl; ds[3] = (double)st_klsh; ds[4] = (double)st_adv;
        ds[5] = (double)st_mh; ds[6] = (double)st_adapt; ds[7] = (double)st_out; ds[8] = (double)st_trips; ds[9] = (double)st_kltrips;
        ds[10] = (double)st_shtrips;
    }
#endif
    // ---- store chain state
    KLHR_WAVE_SYNC();
    {
        double *dst = p.theta + c0 * (long long)D;
        int cc = 0, i = lane;
        while (i >= D) { i -= D; cc++; }
        for (int idx = lane; idx < nch * D; idx += 64) {
            dst[idx] = reinterpret_cast<const double2 *>(lds + (size_t)cc * p.S)[i].x;
            i += 64;
            while (i >= D) { i -= D; cc++; }
        }
    }
    if (active) {
        bool fin = true;
        for (int i = 0; i < D; i++) if (!isfinite(TR[i].x)) fin = false;
        if (!fin) atomicOr(&p.status[c], 2);
        if (fit_capped) atomicOr(&p.status[c], 8);
        p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il;
        p.counters[c * 4 + 0] = draw; p.counters[c * 4 + 1] = omN; p.counters[c * 4 + 2] = pcan; p.counters[c * 4 + 3] = cidx;
        p.acc[c] = acc;
        p.grad_evals[c] = gevals;
        if (VARIANT == VAR_GAUSS) { p.kstate[c * 4 + 0] = sm_x; p.kstate[c * 4 + 1] = sm_cnt; p.kstate[c * 4 + 2] = (double)Kc; p.kstate[c * 4 + 3] = moved_f; }
    }
}

}  // namespace klhr
