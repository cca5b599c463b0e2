// klhr_split.hpp — the split-phase pipeline: KLHRSINH.draw() (klhr_sinh.py:262-289) for line-separable targets.
//
// For funnel, ill-normal and normal (stan/funnel.stan:4-10, stan/ill-normal.stan:4-11, stan/normal.stan:4-8) the log
// density restricted to a line theta + t*rho is a closed form in t through a handful of per-line sums:
//     sum_i (a_i + t r_i)^2 = A + 2tB + t^2 C,      sum_i (a_i + t r_i) r_i = B + tC
// (a = theta or theta/s, r = rho or rho/s).  One O(D) pass per iteration produces (A, B, C, max|a|, max|r|, ...), after
// which every one of the ~120 log-density(+gradient) evaluations of a fit (klhr_sinh.py:163-201) costs O(1) instead of
// O(D).  The element-wise gradient clip of _logp_grad (klhr_sinh.py:158-161) is inactive whenever
// max_i |g_i(t)| <= clip, which the two maxima bound; the rare evaluation where that bound fails (measured on the
// reference: 1 in 2 000-5 000 KL evaluations) is recomputed by the exact element-wise loop, so clipping semantics are kept.
//
// With the fit reduced to O(1) state per chain the iteration splits into two kernels with the mapping each part wants:
//   klhr_stream_kernel  lane = chain, wavefront = tile of 64 chains (klhr_stream16_kernel: tiles of 16 chains, FOUR lanes per chain
//                       — quad mode, below).  Everything that is O(D) per chain and sequential
//                       in the coordinate index (NumPy's summation orders, the PCG64 stream): MH accept/reject + theta
//                       update + OnlineMoments + CCIPCA (+ window closure) of iteration k, then the direction draw, the
//                       line sums and the fit's five normals of iteration k+1.  The chain state lives in HBM in a TILED
//                       layout [tile][coordinate][W lanes]: lane-per-chain access to coordinate i of a tile is one
//                       contiguous segment.  This kernel moves all the algorithmic bytes of an iteration.
//   klhr_fit12_kernel   (many chains) lane = chain: the 1-D mode search (klhr_sinh.py:183-187, opt_advance<N = 1>), then the KL
//                       fit (klhr_sinh.py:194-199, opt_advance<N = 4 | 3 | 2>, 8 Gauss-Hermite nodes per lane)
//   klhr_fitn_kernel    (few chains) the same two stages with 8 lanes per chain, one quadrature node each
//   klhr_tile_kernel    (up to 16 384 chains; Gaussian targets 32 768) whole iterations of a 16-chain tile — quad-mode streaming body
//                       on two wavefronts (tail | head) + node-parallel fit body — in one launch
// Results: RNG stream, direction, OnlineMoments, CCIPCA, closures bit-exact as in the monolithic kernels; objective values
// agree with the element-wise loops to rounding (closed form instead of a D-term sum), inside the stated tolerances.
#pragma once
#include "klhr_lane_kernel.hpp"

namespace klhr {

// (per-chain fit record F_*: klhr_kernels.hpp)
// (LineStat, line_point: klhr_lane_kernel.hpp — the lane-per-chain kernel evaluates the same closed forms)
// The exact element-wise evaluation of four points of one chain's line (model_eval's arithmetic, in index order), from
// the tiled theta / direction rows in HBM: the rare path behind line_point's clip bound.  Out of line, by value.
struct Pts4 { double lp[4], gd[4]; };
template <int MODEL>
__device__ __attribute__((noinline)) Pts4 exact_points4(const double *theta_t, const double *r_t, const double *aux, int D, int W, int nmodel,
                                                        double nn, double t0, double t1, double t2, double t3, double cl) {
    const double t[4] = {t0, t1, t2, t3};
    Pts4 o;
    if constexpr (MODEL == MODEL_FUNNEL) {
        const double th0 = theta_t[0], r0 = r_t[0] / nn;
        const double Da = (double)nmodel;
        double v[4], ev[4], ss[4], gd[4];
#pragma unroll
        for (int n = 0; n < 4; n++) { v[n] = fma(t[n], r0, th0); ev[n] = exp(-v[n]); ss[n] = 0.0; gd[n] = 0.0; }
        for (int i = 1; i < D; i++) {
            const double a = theta_t[(long long)i * W], r = r_t[(long long)i * W] / nn;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const double xi = fma(t[n], r, a);
                ss[n] += xi * xi;
                gd[n] += clipc(-ev[n] * xi, cl) * r;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const double g0 = -v[n] * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev[n] * ss[n];
            o.gd[n] = clipc(g0, cl) * r0 + gd[n];
            o.lp[n] = -v[n] * v[n] * (1.0 / 18.0) - 0.5 * Da * v[n] - 0.5 * ev[n] * ss[n];
        }
    } else if constexpr (MODEL == MODEL_AR1) {
        // model_eval's chained pass (device_models.hpp): g_{i-1} is complete once z_i is known
        const double a = 0.9, b = sqrt(1 - a * a);
        double rp = r_t[0] / nn;
        double xp[4], lp[4], gprev[4], gd[4];
#pragma unroll
        for (int n = 0; n < 4; n++) { xp[n] = fma(t[n], rp, theta_t[0]); lp[n] = -0.5 * xp[n] * xp[n]; gprev[n] = -xp[n]; gd[n] = 0.0; }
        for (int i = 1; i < D; i++) {
            const double th = theta_t[(long long)i * W], r = r_t[(long long)i * W] / nn;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const double xi = fma(t[n], r, th);
                const double z = (xi - a * xp[n]) / b;
                lp[n] -= 0.5 * z * z;
                const double gfull = gprev[n] + a * z / b;
                gd[n] += clipc(gfull, cl) * rp;
                gprev[n] = -z / b;
                xp[n] = xi;
            }
            rp = r;
        }
#pragma unroll
        for (int n = 0; n < 4; n++) { o.gd[n] = gd[n] + clipc(gprev[n], cl) * rp; o.lp[n] = lp[n]; }
    } else {
        double acc[4] = {0, 0, 0, 0}, gd[4] = {0, 0, 0, 0};
        for (int i = 0; i < D; i++) {
            const double a = theta_t[(long long)i * W], r = r_t[(long long)i * W] / nn;
            const double is = MODEL == MODEL_ILL_NORMAL ? aux[i] : 1.0;
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const double xi = fma(t[n], r, a);
                double gi;
                if constexpr (MODEL == MODEL_ILL_NORMAL) { const double z = xi * is; acc[n] += z * z; gi = -z * is; }
                else { acc[n] += xi * xi; gi = -xi; }
                gd[n] += clipc(gi, cl) * r;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; n++) { o.gd[n] = gd[n]; o.lp[n] = -0.5 * acc[n]; }
    }
    return o;
}

// tiled row of chain c: element i of a [C][D] array sits at base + i*W
__device__ __forceinline__ long long tile_base(long long c, int D, int W) { return (c / W) * (long long)D * W + (c % W); }

// ---- KL(eta) value + gradient (klhr_sinh.py:163-176) from the line sums: the lane kernel's node groups (same per-node
// arithmetic, same fixed summation tree ((0+1)+(2+3))+((4+5)+(6+7))) with line_point in place of the O(D) loop
#pragma clang fp contract(off)
template <int MODEL, bool SUB>
__device__ __forceinline__ void split_kl_group(const KParams &p, const LineStat &st, long long c, double nn, double x0, const KLScal &k,
                                               const double *asx, const double *wq, double (&part)[5]) {
    double sh[4], ch[4], lch[4], t[4], av[4], logp[4], gdot[4];
    const double cl = SUB ? p.grad_clip : p.scale_clip;
    bool need = false, need_n[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double a = (asx[n] + k.esh) * k.invd;
        const double ac = clipc(a, p.scale_clip);
        const double E = exp(ac), Ei = fast_rcp(E);
        sh[n] = 0.5 * (E - Ei);
        ch[n] = 0.5 * (E + Ei);
        lch[n] = log_cosh_from_exp(fabs(ac), fmin(E, Ei));
        t[n] = fma(k.s_, sh[n], x0);
        av[n] = a;
        need_n[n] = line_point<MODEL, true>(p, st, t[n], cl, logp[n], gdot[n]);
        need = need || need_n[n];
    }
    if (__any(need)) {
        // the rare node behind the clip bound takes the element-wise value — that node only, in every form of the KL evaluation (the lane
        // kernel's cf_eval_kl / cf_node, node_eval_kl below), so that the forms and the mappings return the same bits
        if (need) {
            const long long tb = tile_base(c, p.D, p.tw);
            const Pts4 e = exact_points4<MODEL>(p.theta + tb, p.d_rho + tb, p.md.aux, p.D, p.tw, p.md.n, nn, t[0], t[1], t[2], t[3], cl);
#pragma unroll
            for (int n = 0; n < 4; n++) if (need_n[n]) { logp[n] = e.lp[n]; gdot[n] = e.gd[n]; }
        }
    }
    double cc[4][5];
#pragma unroll
    for (int n = 0; n < 4; n++) kl_node_terms<SUB>(wq[n], av[n], sh[n], ch[n], lch[n], logp[n], gdot[n], k, cc[n]);
#pragma unroll
    for (int j = 0; j < 5; j++) part[j] = (cc[0][j] + cc[1][j]) + (cc[2][j] + cc[3][j]);
}

// KLHR objective (klhr.py:109-124): raw gradient, so the closed form needs no clip bound
template <int MODEL>
__device__ __forceinline__ void split_gauss_group(const KParams &p, const LineStat &st, double m, double s_, const double *xq, const double *wq,
                                                  double (&part)[5]) {
    double c3[4][3];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const double t = fma(s_, xq[n], m);
        double logp, gdot;
        (void)line_point<MODEL, false>(p, st, t, INFINITY, logp, gdot);
        const double wgr = wq[n] * gdot;
        c3[n][0] = wq[n] * logp;
        c3[n][1] = wgr;
        c3[n][2] = (wgr * xq[n]) * s_;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) part[j] = (c3[0][j] + c3[1][j]) + (c3[2][j] + c3[3][j]);
    part[3] = part[4] = 0.0;
}
#pragma clang fp contract(fast)

template <int MODEL, int VARIANT>
__device__ __forceinline__ void split_eval_kl(const KParams &p, const LineStat &st, long long c, double nn, const double *xt, double &f, double *g) {
    double acc5[5] = {0, 0, 0, 0, 0};
    if constexpr (VARIANT == VAR_GAUSS) {
        const double sg = exp(clipc(xt[1], p.scale_clip));
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            double part[5];
            split_gauss_group<MODEL>(p, st, xt[0], sg, p.xq + 4 * h, p.wq + 4 * h, part);
#pragma unroll
            for (int j = 0; j < 5; j++) acc5[j] = h == 0 ? part[j] : add_rn(acc5[j], part[j]);
        }
        f = -(acc5[0] + xt[1]); g[0] = -acc5[1]; g[1] = -(acc5[2] + 1.0); g[2] = g[3] = 0.0;
    } else {
        constexpr bool SUB = VARIANT == VAR_SUB;
        const KLScal k = kl_scalars<SUB>(p, xt[1], xt[2], xt[3]);
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            double part[5];
            split_kl_group<MODEL, SUB>(p, st, c, nn, xt[0], k, p.asx + 4 * h, p.wq + 4 * h, part);
#pragma unroll
            for (int j = 0; j < 5; j++) acc5[j] = h == 0 ? part[j] : add_rn(acc5[j], part[j]);
        }
        f = acc5[0]; g[0] = acc5[1]; g[1] = acc5[2]; g[2] = acc5[3]; g[3] = acc5[4];
    }
}

// ------------------------------------------------------------------------------------------------
// Node-parallel fit for FEW chains (klhr_split.hip: at most 16 384 per launch): both stages of fit() in one kernel, 8 lanes per
// chain.  With lane = chain a launch lasts as long as its slowest chain's sequence of passes, however few chains there are,
// and a KL pass is 8 quadrature nodes one after the other.  Here lane l8 of a chain's octet evaluates node l8 and a butterfly
// (grp_sum8: the tree ((0+1)+(2+3))+((4+5)+(6+7)) of split_kl_group / split_eval_kl, additions commuted) hands every lane the
// same sums; the optimiser state is replicated in the 8 lanes (identical inputs, identical steps).  A wavefront holds 8 chains,
// not 64: fewer optimiser states to serialise per pass.  Per node and per step the arithmetic is that of klhr_fit1_kernel /
// klhr_fit2_kernel: the two forms return the same bits (tests/test_gpu_parity.py compares them).
#pragma clang fp contract(off)
template <int MODEL, int VARIANT>
__device__ __forceinline__ void node_eval_kl(const KParams &p, const LineStat &st, long long c, double nn, const double *xt, double q_a, double q_x,
                                             double q_w, double &f, double *g) {
    double cc[5];
    if constexpr (VARIANT == VAR_GAUSS) {
        const double sg = exp(clipc(xt[1], p.scale_clip));
        const double t = fma(sg, q_x, xt[0]);
        double logp, gdot;
        (void)line_point<MODEL, false>(p, st, t, INFINITY, logp, gdot);
        const double wgr = q_w * gdot;
        cc[0] = q_w * logp; cc[1] = wgr; cc[2] = (wgr * q_x) * sg; cc[3] = cc[4] = 0.0;
        const double a0 = grp_sum8(cc[0]), a1 = grp_sum8(cc[1]), a2 = grp_sum8(cc[2]);
        f = -(a0 + xt[1]); g[0] = -a1; g[1] = -(a2 + 1.0); g[2] = g[3] = 0.0;
    } else {
        constexpr bool SUB = VARIANT == VAR_SUB;
        const KLScal k = kl_scalars<SUB>(p, xt[1], xt[2], xt[3]);
        const double cl = SUB ? p.grad_clip : p.scale_clip;
        const double a = (q_a + k.esh) * k.invd;
        const double ac = clipc(a, p.scale_clip);
        const double E = exp(ac), Ei = fast_rcp(E);
        const double sh = 0.5 * (E - Ei), ch = 0.5 * (E + Ei);
        const double lch = log_cosh_from_exp(fabs(ac), fmin(E, Ei));
        const double t = fma(k.s_, sh, xt[0]);
        double logp, gdot;
        bool need = line_point<MODEL, true>(p, st, t, cl, logp, gdot);
        if (__any(need)) {
            // the rare point behind the clip bound: the element-wise loop, for this node only (exact_points4 takes four points; the lane
            // passes its own four times)
            if (need) {
                const long long tb = tile_base(c, p.D, p.tw);
                const Pts4 e = exact_points4<MODEL>(p.theta + tb, p.d_rho + tb, p.md.aux, p.D, p.tw, p.md.n, nn, t, t, t, t, cl);
                logp = e.lp[0]; gdot = e.gd[0];
            }
        }
        kl_node_terms<SUB>(q_w, a, sh, ch, lch, logp, gdot, k, cc);
        f = grp_sum8(cc[0]); g[0] = grp_sum8(cc[1]); g[1] = grp_sum8(cc[2]); g[2] = grp_sum8(cc[3]);
        g[3] = SUB ? 0.0 : grp_sum8(cc[4]);
    }
}
#pragma clang fp contract(fast)

template <int MODEL, int VARIANT>
__device__ __forceinline__ void fitn_body(const KParams &p, const long long task, const int l8) {
    constexpr int N = VARIANT == VAR_GAUSS ? 2 : (VARIANT == VAR_SUB ? 3 : 4);
    if (task >= p.C) return;                    // (whole octets leave: the butterflies stay inside an octet)
    const bool diag = (p.flags & FLAG_DIAG) != 0, lead = l8 == 0;
    // this lane's quadrature node (a per-lane index into the kernel-argument tables would go through scratch)
    double q_a = p.asx[0], q_x = p.xq[0], q_w = p.wq[0];
#pragma unroll
    for (int n = 1; n < 8; n++) if (l8 == n) { q_a = p.asx[n]; q_x = p.xq[n]; q_w = p.wq[n]; }
    const LineStat st = load_stat(p, task);
    const double nn = KLHR_FB(p, F_NN, task);
    Opt o;
    int npass = 0;
    // ---- stage 1: minimize(logp_grad_rho, x0, BFGS, gtol = 1e-5) (klhr_sinh.py:183-190), every lane of the octet alike
    {
        double x0v[4] = {KLHR_FB(p, F_X0, task), 0, 0, 0};
        if (diag && lead) p.d_diag[task * KLHR_NDIAG + KLHR_D_X0] = x0v[0];
        opt_begin(o, 1, x0v, 1e-5);
    }
    bool live = true;
    while (__any(live)) {
        if (live) {
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
            if (o.pc == PC_DONE || npass > KLHR_MAX_PASSES) live = false;
        }
    }
    const bool capped1 = o.pc != PC_DONE;
    const double mode = o.xk[0];
    double s1 = o.H[0];
    s1 = (s1 > 0 ? 1.0 : 0.0) * 0.5 * log(s1);
    const int npass1 = npass, nfev1 = o.nfev;
    if (lead) {
        KLHR_FB(p, F_MODE, task) = mode; KLHR_FB(p, F_INIT1, task) = s1;
        KLHR_FB(p, F_PASS, task) = (double)npass1; KLHR_FB(p, F_NFEV1, task) = (double)nfev1;
        if (capped1) atomicOr(&p.status[task], 8);
        if (diag) {
            double *dd = p.d_diag + task * KLHR_NDIAG;
            dd[KLHR_D_MODE] = o.xk[0]; dd[KLHR_D_HINV1] = o.H[0]; dd[KLHR_D_NFEV1] = o.nfev; dd[KLHR_D_NIT1] = o.k;
            dd[KLHR_D_STATUS1] = o.status; dd[KLHR_D_FALLBACKS] = o.fallbacks;
        }
    }
    const int fb1 = o.fallbacks;
    // ---- stage 2: minimize(KL, init, BFGS, gtol = 1e-3) (klhr_sinh.py:191-201; klhr.py:134-139; sub_klhr_sinh.py:184-192)
    {
        const double init[4] = {mode, s1, VARIANT == VAR_GAUSS ? 0.0 : KLHR_FB(p, F_N2, task), VARIANT == VAR_SINH ? KLHR_FB(p, F_N3, task) : 0.0};
        if (diag && lead) {
            double *dd = p.d_diag + task * KLHR_NDIAG;
            dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = init[2]; dd[KLHR_D_INIT3] = init[3];
        }
        opt_begin(o, N, init, VARIANT == VAR_GAUSS ? 1e-5 : 1e-3);
    }
    npass = 0;
    live = true;
    while (__any(live)) {
        if (live) {
            double f = 0, g[4] = {0, 0, 0, 0};
            if (opt_cache_hit(o)) { f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3]; }
            else {
                node_eval_kl<MODEL, VARIANT>(p, st, task, nn, o.xt, q_a, q_x, q_w, f, g);
#ifdef KLHR_DBG_CHAIN
                if (task == KLHR_DBG_CHAIN && l8 == 0) printf("N pass %d x %a %a %a %a f %a g %a %a %a %a\n", npass, o.xt[0], o.xt[1], o.xt[2], o.xt[3], f, g[0], g[1], g[2], g[3]);
#endif
                opt_cache_store(o, f, g);
                npass++;
            }
            opt_advance<false, N>(o, f, g);
            if (o.pc == PC_DONE || npass > KLHR_MAX_PASSES) live = false;
        }
    }
    if (l8 < 4) p.d_eta[task * 4 + l8] = l8 == 0 ? o.xk[0] : (l8 == 1 ? o.xk[1] : (l8 == 2 ? o.xk[2] : o.xk[3]));
    if (lead) {
        KLHR_FB(p, F_NFEV2, task) = (double)o.nfev;
        if (o.pc != PC_DONE) atomicOr(&p.status[task], 8);
        if (diag) {
            double *dd = p.d_diag + task * KLHR_NDIAG;
            dd[KLHR_D_NFEV2] = o.nfev; dd[KLHR_D_NIT2] = o.k; dd[KLHR_D_STATUS2] = o.status; dd[KLHR_D_FUN2] = o.old_fval;
            dd[KLHR_D_FALLBACKS] = fb1 + o.fallbacks; dd[KLHR_D_PASSES] = npass1 + npass;
        }
    }
}

template <int MODEL, int VARIANT>
__global__ __launch_bounds__(64) void klhr_fitn_kernel(const KParams p) {
    const int lane = threadIdx.x & 63;
    fitn_body<MODEL, VARIANT>(p, (long long)blockIdx.x * 8 + (lane >> 3), lane & 7);
}

// ------------------------------------------------------------------------------------------------
// The fit kernel for MANY chains (klhr_fit12_kernel; above 32 768 chains): one wavefront = 64 chains, both stages of fit() in one
// launch, in three steps:
//   1. the 1-D mode search, lane = chain, opt_advance<N = 1> (klhr_sinh.py:183-190);
//   2. the KL fit, lane = chain (8 quadrature nodes per lane, opt_advance<N = 4 | 3 | 2>), until at most FIT_THR = 8 of the wavefront's 64
//      fits are still live.  The passes of a KL fit are narrowly distributed with a thin long tail (funnel-100: median 13, 90 % <= 14, 1 % >= 28,
//      rare 60-75): run to completion in this form a wavefront lasts as long as its slowest chain (42 % of the lanes active on the round-2
//      kernel; an experiment that capped every fit at 14 passes took the kernel from 0.62 to 0.35 ms);
//   3. the stragglers: the <= 8 survivors leave their optimiser state, line sums and counters in LDS (80 doubles each) and are finished
//      NODE-PARALLEL by the wavefront's 8 octets — lane l8 of an octet evaluates quadrature node l8, the butterfly hands every lane the same
//      sums, the optimiser state is replicated in the 8 lanes (node_eval_kl, as in klhr_fitn_kernel): a pass of 8 lanes per chain has about
//      a third of the latency of a lane = chain pass, and below ~16 live lanes it is also the cheaper form per chain.
// Which form runs a pass has no influence on its arithmetic (the two forms return the same bits: tests/test_gpu_parity.py compares this kernel
// with klhr_fitn_kernel and klhr_tile_kernel over populations, with the exact patch made a common path), nor has the lane a chain sits in.
// Measured, funnel-100 x 131 072 (same box, back to back): kernel alone 0.62 -> 0.52 ms; the iteration with four overlapped chain groups
// 1.46e8 -> 1.47e8 it/s — it is bound by the streaming kernel's LDS-limited concurrency, not by this kernel.
constexpr int FIT_THR = 8, FITSLOT_DOUBLES = 80;
// (element by element with constant indices: a struct copy would pin the optimiser state in scratch memory)
__device__ __forceinline__ void fitslot_store(double *q, const Opt &o, const LineStat &st, double nn, long long task, int npass, int npass1) {
#pragma unroll
    for (int k = 0; k < 4; k++) { q[k] = o.xk[k]; q[4 + k] = o.gfk[k]; q[8 + k] = o.pk[k]; q[12 + k] = o.xt[k]; q[16 + k] = o.lx[k]; q[20 + k] = o.lg[k]; }
#pragma unroll
    for (int k = 0; k < 16; k++) q[24 + k] = o.H[k];
#pragma unroll
    for (int k = 0; k < L_N; k++) q[40 + k] = o.ls[k];
    q[53] = o.old_fval; q[54] = o.old_old_fval; q[55] = o.derphi0; q[56] = o.gtol; q[57] = o.areq; q[58] = o.lf;
    q[59] = st.v0; q[60] = st.r0; q[61] = st.A; q[62] = st.B; q[63] = st.C; q[64] = st.m0; q[65] = st.m1; q[66] = nn;
    q[67] = (double)task; q[68] = (double)npass; q[69] = (double)npass1;
    q[70] = (double)o.n; q[71] = (double)o.k; q[72] = (double)o.nfev; q[73] = (double)o.status; q[74] = (double)o.fallbacks; q[75] = (double)o.pc;
    q[76] = (double)o.it; q[77] = (double)o.i2; q[78] = (double)((o.have ? 1 : 0) | (o.brackt ? 2 : 0) | (o.stage2 ? 4 : 0));
}
__device__ __forceinline__ void fitslot_load(const double *q, Opt &o, LineStat &st, double &nn, long long &task, int &npass, int &npass1) {
#pragma unroll
    for (int k = 0; k < 4; k++) { o.xk[k] = q[k]; o.gfk[k] = q[4 + k]; o.pk[k] = q[8 + k]; o.xt[k] = q[12 + k]; o.lx[k] = q[16 + k]; o.lg[k] = q[20 + k]; }
#pragma unroll
    for (int k = 0; k < 16; k++) o.H[k] = q[24 + k];
#pragma unroll
    for (int k = 0; k < L_N; k++) o.ls[k] = q[40 + k];
    o.old_fval = q[53]; o.old_old_fval = q[54]; o.derphi0 = q[55]; o.gtol = q[56]; o.areq = q[57]; o.lf = q[58];
    st.v0 = q[59]; st.r0 = q[60]; st.A = q[61]; st.B = q[62]; st.C = q[63]; st.m0 = q[64]; st.m1 = q[65]; nn = q[66];
    task = (long long)q[67]; npass = (int)q[68]; npass1 = (int)q[69];
    o.n = (int)q[70]; o.k = (int)q[71]; o.nfev = (int)q[72]; o.status = (int)q[73]; o.fallbacks = (int)q[74]; o.pc = (int)q[75];
    o.it = (int)q[76]; o.i2 = (int)q[77];
    const int fl = (int)q[78];
    o.have = (fl & 1) != 0; o.brackt = (fl & 2) != 0; o.stage2 = (fl & 4) != 0;
}

// Stage 1 of fit(): minimize(logp_grad_rho, x0, BFGS, gtol = 1e-5) (klhr_sinh.py:183-190), lane = chain
template <int MODEL>
__device__ __forceinline__ void fit1_body(const KParams &p, const long long task, const bool active) {
    const bool diag = (p.flags & FLAG_DIAG) != 0;
    Opt o;
    LineStat st = {0, 0, 0, 0, 0, 0, 0};
    int npass = 0;
    {
        double x0v[4] = {0, 0, 0, 0};
        if (active) {
            st = load_stat(p, task);
            x0v[0] = KLHR_FB(p, F_X0, task);
            if (diag) p.d_diag[task * KLHR_NDIAG + KLHR_D_X0] = x0v[0];
        }
        opt_begin(o, 1, x0v, 1e-5);
    }
    bool live = active;
    while (__any(live)) {
        if (live) {
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
            const bool capped = npass > KLHR_MAX_PASSES;            // safety valve (klhr_kernels.hpp)
            if (o.pc == PC_DONE || capped) {
                double s = o.H[0];
                s = (s > 0 ? 1.0 : 0.0) * 0.5 * log(s);
                KLHR_FB(p, F_MODE, task) = o.xk[0];
                KLHR_FB(p, F_INIT1, task) = s;
                KLHR_FB(p, F_PASS, task) = (double)npass;
                KLHR_FB(p, F_NFEV1, task) = (double)o.nfev;       // (stores only: a load here would stall the whole wavefront's pass)
                if (capped && o.pc != PC_DONE) atomicOr(&p.status[task], 8);
                if (diag) {
                    double *dd = p.d_diag + task * KLHR_NDIAG;
                    dd[KLHR_D_MODE] = o.xk[0]; dd[KLHR_D_HINV1] = o.H[0]; dd[KLHR_D_NFEV1] = o.nfev; dd[KLHR_D_NIT1] = o.k;
                    dd[KLHR_D_STATUS1] = o.status; dd[KLHR_D_FALLBACKS] = o.fallbacks;
                }
                live = false;
            }
        }
    }
}

// the results of a finished KL fit (klhr_sinh.py:199-201): eta, evaluation count, status, diagnostics
__device__ __forceinline__ void fit2_finish(const KParams &p, const Opt &o, const long long task, const int npass, const int npass1, const bool diag) {
    KLHR_FB(p, F_NFEV2, task) = (double)o.nfev;
    if (o.pc != PC_DONE) atomicOr(&p.status[task], 8);
    if (diag) {
        double *dd = p.d_diag + task * KLHR_NDIAG;
        dd[KLHR_D_NFEV2] = o.nfev; dd[KLHR_D_NIT2] = o.k; dd[KLHR_D_STATUS2] = o.status; dd[KLHR_D_FUN2] = o.old_fval;
        dd[KLHR_D_FALLBACKS] += o.fallbacks; dd[KLHR_D_PASSES] = npass1 + npass;       // (diagnostic launches only)
    }
}

template <int MODEL, int VARIANT>
__global__ __launch_bounds__(64) void klhr_fit12_kernel(const KParams p) {
    constexpr int N = VARIANT == VAR_GAUSS ? 2 : (VARIANT == VAR_SUB ? 3 : 4);
    __shared__ double slots[FIT_THR * FITSLOT_DOUBLES];
    const int lane = threadIdx.x & 63, l8 = lane & 7;
    long long task = (long long)blockIdx.x * 64 + lane;
    const bool active = task < p.C;
    const bool diag = (p.flags & FLAG_DIAG) != 0;
    // ---- 1. mode search
    fit1_body<MODEL>(p, task, active);
    // ---- 2. KL fit, lane = chain, until at most FIT_THR of this wavefront's fits are live
    Opt o;
    LineStat st = {0, 0, 0, 0, 0, 0, 0};
    double nn = 1.0;
    int npass = 0, npass1 = 0;
    {
        double init[4] = {0, 0, 0, 0};
        if (active) {
            st = load_stat(p, task);
            nn = KLHR_FB(p, F_NN, task);
            npass1 = (int)KLHR_FB(p, F_PASS, task);
            init[0] = KLHR_FB(p, F_MODE, task); init[1] = KLHR_FB(p, F_INIT1, task);
            init[2] = VARIANT == VAR_GAUSS ? 0.0 : KLHR_FB(p, F_N2, task);
            init[3] = VARIANT == VAR_SINH ? KLHR_FB(p, F_N3, task) : 0.0;
            if (diag) {
                double *dd = p.d_diag + task * KLHR_NDIAG;
                dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = init[2]; dd[KLHR_D_INIT3] = init[3];
            }
        }
        opt_begin(o, N, init, VARIANT == VAR_GAUSS ? 1e-5 : 1e-3);
    }
    bool live = active;
    while (__popcll(__ballot(live)) > FIT_THR) {
        if (live) {
            double f = 0, g[4] = {0, 0, 0, 0};
            if (opt_cache_hit(o)) { f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3]; }
            else {
                split_eval_kl<MODEL, VARIANT>(p, st, task, nn, o.xt, f, g);
#ifdef KLHR_DBG_CHAIN
                if (task == KLHR_DBG_CHAIN) printf("L pass %d x %a %a %a %a f %a g %a %a %a %a\n", npass, o.xt[0], o.xt[1], o.xt[2], o.xt[3], f, g[0], g[1], g[2], g[3]);
#endif
                opt_cache_store(o, f, g);
                npass++;
            }
            opt_advance<false, N>(o, f, g);
            if (o.pc == PC_DONE || npass > KLHR_MAX_PASSES) {
#pragma unroll
                for (int k = 0; k < 4; k++) p.d_eta[task * 4 + k] = o.xk[k];
                fit2_finish(p, o, task, npass, npass1, diag);
                live = false;
            }
        }
    }
    // ---- 3. the stragglers: records through LDS (wave-level synchronisation: the workgroup is this wavefront), then node-parallel
    const unsigned long long m = __ballot(live);
    const int n = __popcll(m);
    if (n == 0) return;
    if (live) fitslot_store(slots + (size_t)__popcll(m & ((1ull << lane) - 1ull)) * FITSLOT_DOUBLES, o, st, nn, task, npass, npass1);
    KLHR_WAVE_SYNC();
    // this lane's quadrature node (a per-lane index into the kernel-argument tables would go through scratch)
    double q_a = p.asx[0], q_x = p.xq[0], q_w = p.wq[0];
#pragma unroll
    for (int k = 1; k < 8; k++) if (l8 == k) { q_a = p.asx[k]; q_x = p.xq[k]; q_w = p.wq[k]; }
    live = (lane >> 3) < n;                        // (whole octets: the butterflies stay inside an octet)
    if (live) fitslot_load(slots + (size_t)(lane >> 3) * FITSLOT_DOUBLES, o, st, nn, task, npass, npass1);
    while (__any(live)) {
        if (live) {
            double f = 0, g[4] = {0, 0, 0, 0};
            if (opt_cache_hit(o)) { f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3]; }
            else {
                node_eval_kl<MODEL, VARIANT>(p, st, task, nn, o.xt, q_a, q_x, q_w, f, g);
                opt_cache_store(o, f, g);
                npass++;
            }
            opt_advance<false, N>(o, f, g);
            if (o.pc == PC_DONE || npass > KLHR_MAX_PASSES) {
                if (l8 < 4) p.d_eta[task * 4 + l8] = l8 == 0 ? o.xk[0] : (l8 == 1 ? o.xk[1] : (l8 == 2 ? o.xk[2] : o.xk[3]));
                if (l8 == 0) fit2_finish(p, o, task, npass, npass1, diag);
                live = false;
            }
        }
    }
}

// KL / line probes of the C ABI (klhr_kl_eval, klhr_line_eval): one evaluation per chain at eta = d_eta
template <int MODEL, int VARIANT>
__global__ __launch_bounds__(64) void klhr_probe_kernel(const KParams p) {
    const long long c = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= p.C) return;
    const LineStat st = load_stat(p, c);
    const double nn = KLHR_FB(p, F_NN, c);
    double eta[4], f = 0, g[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) eta[k] = p.d_eta[c * 4 + k];
    for (int rep = 0; rep < p.n_iters; rep++) {
        if (p.phases & PH_KL) split_eval_kl<MODEL, VARIANT>(p, st, c, nn, eta, f, g);
        else {
            double lp, gd;
            (void)line_point<MODEL, false>(p, st, eta[0], INFINITY, lp, gd);
            f = -lp; g[0] = -gd; g[1] = g[2] = g[3] = 0.0;
        }
    }
    double *dd = p.d_diag + c * KLHR_NDIAG;
    dd[0] = f; dd[1] = g[0]; dd[2] = g[1]; dd[3] = g[2]; dd[4] = g[3];
}

// ------------------------------------------------------------------------------------------------
// The streaming kernel.  lane = chain; a wavefront owns tile blockIdx.x (W chains, W = 16 or 64; lanes >= W idle) of every
// state array: element i of chain (tile, lane) of a [C][D] array sits at (tile*D + i)*W + lane.  Narrow tiles put more
// wavefronts — more loads in flight — on the machine when there are few chains.
// A row is walked CH elements at a time: all loads of a chunk are issued before the first use, whole chunks carry no bounds
// checks (one base address per array and chunk, immediate offsets k*W*8), the D % CH remaining elements go one by one.
// (quad mode, QN = 4: lane q of a chain's quad takes the elements q, q + 4, ... one at a time — element-wise work only)
template <int CH, int QN = 1, class F>
__device__ __forceinline__ void for_chunks(int D, int q, F f) {
    if constexpr (QN == 1) {
        int i0 = 0;
        for (; i0 + CH <= D; i0 += CH) f(std::integral_constant<int, CH>(), i0);
        for (; i0 < D; ++i0) f(std::integral_constant<int, 1>(), i0);
    } else {
        for (int i = q; i < D; i += QN) f(std::integral_constant<int, 1>(), i);
    }
}
#ifndef KLHR_ST_CH
#define KLHR_ST_CH 8
#endif
#ifndef KLHR_ST_CH1
#define KLHR_ST_CH1 16
#endif
#ifndef KLHR_RING_STEPS
#define KLHR_RING_STEPS 2
#endif
// (inside functions templated on the tile width W and on CHDIV.  16-chain tiles, quad mode: a step is four coordinates and the chunk
//  KLHR_ST_CH / CHDIV steps — CHDIV = 4: 250 registers, two wavefronts per SIMD for the streaming kernel's pair mode; CHDIV = 8:
//  170 registers, which lets klhr_tile_kernel — this body plus the fit's — run two wavefronts per SIMD)
#define ST_CH (W == 16 ? KLHR_ST_CH / CHDIV : KLHR_ST_CH)          // passes over several arrays
#define ST_CH1 (W == 16 ? KLHR_ST_CH1 / CHDIV : KLHR_ST_CH1)       // passes over one or two arrays

// Software-pipelined form: `load(buf, s0, full)` fills a register buffer with the chunk of CH steps at s0, `compute(buf, s0,
// full)` consumes it.  The loads of chunk j+1 are issued before chunk j is computed (two buffers): a wavefront of this kernel
// has at most one or two neighbours on its SIMD (the LDS column scratch bounds the occupancy), so it hides its own memory latency.
// A step is QN coordinates (lane = chain: 1; quad mode: 4, one per lane of the chain's quad — element (s0 + k) * QN + q);
// `full` is std::true_type for whole steps; quad mode ends with one std::false_type step when D % 4 != 0 (some lanes past D).
template <int CH, template <int> class BUF, int QN, class L, class C>
__device__ __forceinline__ void for_chunks_pipelined(int D, L load, C compute) {
    const int NS = D / QN;
    const std::true_type full;
    if constexpr (QN == 1) {
        BUF<CH> a, b;
        int s0 = 0;
        if (CH <= NS) load(a, 0, full);
        for (; s0 + 2 * CH <= NS; s0 += 2 * CH) {
            load(b, s0 + CH, full);
            compute(a, s0, full);
            if (s0 + 3 * CH <= NS) load(a, s0 + 2 * CH, full);
            compute(b, s0 + CH, full);
        }
        if (s0 + CH <= NS) { compute(a, s0, full); s0 += CH; }
        for (; s0 < NS; ++s0) { BUF<1> r; load(r, s0, full); compute(r, s0, full); }
    } else {
        // quad mode: a step is a handful of values per lane — a ring of NB one-step buffers keeps NB steps' loads in flight
        constexpr int NB = KLHR_RING_STEPS * CH;
        BUF<1> r[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) if (j < NS) load(r[j], j, full);
        int s0 = 0;
        for (; s0 + NB <= NS; s0 += NB) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
                compute(r[j], s0 + j, full);
                if (s0 + j + NB < NS) load(r[j], s0 + j + NB, full);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; j++) if (s0 + j < NS) compute(r[j], s0 + j, full);
        if (NS * QN < D) { BUF<1> t; load(t, NS, std::false_type()); compute(t, NS, std::false_type()); }
    }
}

// ---- QUAD MODE (16-chain tiles): lane = (chain cl = lane >> 2, phase q = lane & 3) — the four lanes of a DPP quad walk one
// chain's rows four coordinates per step (element i = 4 step + q; a row segment of a tile is still 16 chains x 4 coordinates =
// 512 contiguous bytes).  A 16-chain tile with lane = chain leaves 48 of the wavefront's 64 lanes idle at full instruction
// cost; here the element-wise arithmetic (theta', OnlineMoments, CCIPCA's updates, the direction) costs a quarter of the
// instructions.  The sums whose order NumPy fixes stay sequential in the coordinate index: the four lanes' terms are broadcast
// over the quad (two v_mov_b32 quad_perm per double) and every lane adds them in index order — identical bits in all four
// lanes, and identical to lane = chain.  Per-chain scalars (generator, MH, counters) are replicated in the quad.
template <int Q>
__device__ __forceinline__ double quad_bcast(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, Q * 0x55, 0xf, 0xf, true);          // (every lane of a quad is active or none is)
    hi = __builtin_amdgcn_mov_dpp(hi, Q * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane q <- lane q - 1 of the quad (lane 0 keeps its own value: the caller substitutes the previous step's last element)
__device__ __forceinline__ double quad_prev(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x90, 0xf, 0xf, true);      // quad_perm:[0,0,1,2]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x90, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// f(i, v...) for the elements of one step in index order: lane = chain calls it once with the lane's own values; quad mode four
// times, with the values of quad lane 0, 1, 2, 3 (elements past D are skipped: `step` and D are wavefront-uniform)
template <int QN, bool FULL, class F, class... T>
__device__ __forceinline__ void seq_each(int step, int D, F f, T... v) {
    if constexpr (QN == 1) f(step, v...);
    else {
        const int i0 = step * 4;
        f(i0, quad_bcast<0>(v)...);
        if (FULL || i0 + 1 < D) f(i0 + 1, quad_bcast<1>(v)...);
        if (FULL || i0 + 2 < D) f(i0 + 2, quad_bcast<2>(v)...);
        if (FULL || i0 + 3 < D) f(i0 + 3, quad_bcast<3>(v)...);
    }
}
template <int N> struct Buf1 { static constexpr int n = N; double v[N]; };
template <int N> struct Buf2 { static constexpr int n = N; double v[N], q[N]; };
template <int N> struct BufT1 { static constexpr int n = N; double th[N], rr[N], mm[N], vv[N], mn[N]; };
template <int N> struct BufDir { static constexpr int n = N; double e0[N], e1[N], e2[N], cv[N], th[N], is[N]; int rk[N]; };
template <int N> struct BufDirC { static constexpr int n = N; double m[N], s[N], th[N], is[N]; int rk[N]; };

#pragma clang fp contract(off)      // NumPy's element-wise roundings and summation orders (bit-exact pieces)

// onlinepca.py:13-26 for the chain of this lane (quad).  u = theta - mean in Z (LDS column of the chain, stride W), V = the tile's
// pca_v rows.  Per component: sums (s, u.v) -> v' = w v + (1-w) u (u.v)/(|v|+tol), sums on v' -> u -= (u.v') v'/(|v'|^2+tol);
// the deflation pass of component j also gathers the first sums of component j+1.  q: the lane's phase in quad mode (0 otherwise).
template <int W, int CHDIV>
__device__ inline void split_pca_update(const KParams &p, double *V, long long n_new, double *Z, const int q) {
    constexpr int QN = W == 16 ? 4 : 1;
    const int D = p.D, K = p.J;
    const long long lim = n_new < K ? n_new : K;
    const long long vs = (long long)K * W;
    double s = 0, udv = 0;
    bool have_sums = false;
    for (int col = 0; col < lim; col++) {
        double *Vc = V + (long long)col * W;
        if (col == n_new - 1) {
            for (int i = q; i < D; i += QN) Vc[i * vs] = Z[i * W];
            have_sums = false;
        } else {
            const double w = ((double)(n_new - 1) - p.l) / (double)n_new;
            if (!have_sums) {
                s = 0; udv = 0;
                for_chunks_pipelined<ST_CH1, Buf1, QN>(D,
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) { const int i = (s0 + k) * QN + q; b.v[k] = Vc[(full || i < D ? i : D - 1) * vs]; }
                    },
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) {
                            const int i = (s0 + k) * QN + q;
                            const double z = Z[(full || i < D ? i : D - 1) * W];
                            seq_each<QN, decltype(full)::value>(s0 + k, D, [&](int, double vv, double zv) { s += vv; udv += zv; }, b.v[k] * b.v[k], z * b.v[k]);
                        }
                    });
            }
            const double nv = sqrt(s);
            double s2 = 0, udv2 = 0;
            for_chunks_pipelined<ST_CH1, Buf1, QN>(D,
                [&](auto &b, int s0, auto full) {
#pragma unroll
                    for (int k = 0; k < b.n; k++) { const int i = (s0 + k) * QN + q; b.v[k] = Vc[(full || i < D ? i : D - 1) * vs]; }
                },
                [&](auto &b, int s0, auto full) {
#pragma unroll
                    for (int k = 0; k < b.n; k++) {
                        const int i = (s0 + k) * QN + q;
                        const bool valid = full || i < D;
                        const double z = Z[(valid ? i : D - 1) * W];
                        const double vn = w * b.v[k] + (1 - w) * z * udv / (nv + PCA_TOL);
                        if (valid) Vc[i * vs] = vn;
                        seq_each<QN, decltype(full)::value>(s0 + k, D, [&](int, double vv, double zv) { s2 += vv; udv2 += zv; }, vn * vn, z * vn);
                    }
                });
            have_sums = false;
            if (col + 1 < lim) {          // (the deflation after the last component is never read: onlinepca.py:24-25)
                const double nv2 = sqrt(s2);
                const bool next_sums = col + 1 != n_new - 1;       // the next component is updated, not assigned
                const double *Vn = V + (long long)(col + 1) * W;
                s = 0; udv = 0;
                for_chunks_pipelined<ST_CH, Buf2, QN>(D,
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) {
                            const int i = (s0 + k) * QN + q;
                            const int ic = full || i < D ? i : D - 1;
                            b.v[k] = Vc[ic * vs]; b.q[k] = next_sums ? Vn[ic * vs] : 0.0;
                        }
                    },
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) {
                            const int i = (s0 + k) * QN + q;
                            const bool valid = full || i < D;
                            const double z = Z[(valid ? i : D - 1) * W] - udv2 * b.v[k] / (nv2 * nv2 + PCA_TOL);
                            if (valid) Z[i * W] = z;
                            if (next_sums) seq_each<QN, decltype(full)::value>(s0 + k, D, [&](int, double vv, double zv) { s += vv; udv += zv; }, b.q[k] * b.q[k], z * b.q[k]);
                        }
                    });
                have_sums = next_sums;
            }
        }
    }
}

// klhr_sinh.py:268-280 (+ the rank permutation multivariate_normal's SVD applies to the following windows' normals).
// Quad mode: lane q of the chain's quad closes the coordinates i = q, q + 4, ...; the column norms are summed in index order.
template <int W>
__device__ inline void split_close_window(const KParams &p, long long c, long long tb, long long tbe, long long tbv, long long omN, double *Z, const int q) {
    constexpr int QN = W == 16 ? 4 : 1;
    const int D = p.D, K = p.J, nc = p.ncol;
    for (int i = q; i < D; i += QN) {
        const long long e = tb + (long long)i * W;
        p.mean[e] = p.om_m[e];
        double cv = omN > 2 ? p.om_v[e] * (double)omN / (double)(omN - 1) : 1.0;
        if (p.flags & FLAG_SCALE_DIR_COV) {
            const double vd = omN > 2 ? p.omd_v[e] * (double)omN / (double)(omN - 1) : 1.0;
            cv = cv / (p.tol + vd);
            p.omd_m[e] = 0.0; p.omd_v[e] = 0.0;
        }
        p.cov[e] = cv;
        Z[i * W] = cv;
        p.om_m[e] = 0.0; p.om_v[e] = 0.0;
    }
    if (QN > 1) KLHR_WAVE_SYNC();          // the ranks read every coordinate's cov, written by the quad's four lanes
    for (int i = q; i < D; i += QN) {
        const double ci = Z[i * W];
        int r = 0;
        for (int j = 0; j < D; j++) { const double cj = Z[j * W]; r += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
        p.rank[tb + (long long)i * W] = r;
    }
    double vals[4] = {0, 0, 0, 0};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < K) {
            double s = 0;
            for (int s0 = 0; s0 * QN < D; s0++) {
                const int i = s0 * QN + q;
                const double vi = p.pca_v[tbv + ((long long)(i < D ? i : D - 1) * K + j) * W];
                seq_each<QN, false>(s0, D, [&](int, double vv) { s += vv; }, vi * vi);
            }
            vals[j] = sqrt(s);
            if (!isfinite(vals[j])) bad = true;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { if (bad) vals[j] = 0.0; vals[j] += PCA_TOL; }
    // (the direction cache of the next window: eigenvalue slots beyond K keep their previous values, as in the reference)
    double evn[4] = {0, 0, 0, 0}, wj[4];
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < nc) evn[j] = j < K ? vals[j] : p.eigvals[c * nc + j];
    dir_weights(p, evn, wj);
    const bool cache = p.dir_m != nullptr && !(p.flags & FLAG_EIGEN_ONE);
    for (int i = q; i < D; i += QN) {
        double e[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) {
            const long long ev = tbv + ((long long)i * K + j) * W;
            e[j] = p.pca_v[ev] / vals[j];
            p.eigvecs[tbe + ((long long)i * nc + j) * W] = e[j];
            p.pca_v[ev] = 0.0;
        }
        if (cache) {
            p.dir_m[tb + (long long)i * W] = dir_mean_elem(nc, wj, e[0], e[1], e[2]);
            p.dir_s[tb + (long long)i * W] = sqrt(Z[i * W]);          // Z still holds this closure's cov
        }
    }
    if (QN > 1) KLHR_WAVE_SYNC();          // (eigvals of slots >= K were read above by every lane of the quad)
    if (q == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) p.eigvals[c * nc + j] = vals[j];
    }
}

// (still contract(off): the streaming kernel's own arithmetic — accept/reject blend, OnlineMoments, direction draw — is
//  NumPy's element-wise arithmetic; the closed-form and proposal helpers it calls carry their own setting)
// element i of the gradient at theta (for OnlineMoments of the gradient, scale_dir_cov): needs the chain's sum of alpha^2
template <int MODEL>
__device__ __forceinline__ double grad_elem(const KParams &p, int i, double x, double ev, double g0) {
    if constexpr (MODEL == MODEL_FUNNEL) return i == 0 ? g0 : -ev * x;
    else if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = p.md.aux[i]; return -(x * is) * is; }
    else return -x;
}

// line sums of (theta_i, r_i) for coordinate i (r still unnormalised): A, B', C', max|a|, max|r| (a, r scaled by 1/s_i for
// ill-normal; the maxima by 1/s_i^2, since |g_i| = |x_i| / s_i^2); the funnel's coordinate 0 (v) is kept aside
template <int MODEL>
__device__ __forceinline__ void line_accumulate(const KParams &p, int i, double th, double r, LineStat &st, double &Bs, double &Cs, double &m1,
                                                double &pth, double &pr) {
    double a = th, rs = r;
    if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = p.md.aux[i]; a = a * is; rs = rs * is; }
    if constexpr (MODEL == MODEL_AR1) {
        // ar1 (stan/ar1.stan): log p = -0.5 x_0^2 - 0.5 sum_i z_i^2, z_i = (x_i - 0.9 x_{i-1}) / b — on a line z_i is linear in t, so the
        // sums run over (u_i, w_i) = the z-transforms of theta and of the direction (i = 0: the coordinate itself)
        const double ca = 0.9, cb = sqrt(1 - ca * ca);
        if (i > 0) { a = (th - ca * pth) / cb; rs = (r - ca * pr) / cb; }
        pth = th; pr = r;
    }
    if (MODEL == MODEL_FUNNEL && i == 0) { st.v0 = a; st.r0 = r; }
    else {
        st.A = fma(a, a, st.A); Bs = fma(a, rs, Bs); Cs = fma(rs, rs, Cs);
        if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = p.md.aux[i]; a = a * is; rs = rs * is; }
        // ar1: |g_i| = |-z_i + 0.9 z_{i+1}| / b  (|g_0| <= |x_0| + 0.9 |z_1| / b)  <=  (1.9 / b) max_j |z_j|
        if constexpr (MODEL == MODEL_AR1) { const double kap = 1.9 / sqrt(1 - 0.81); a = a * kap; rs = rs * kap; }
        st.m0 = fmax(st.m0, fabs(a)); m1 = fmax(m1, fabs(rs));
        if (a != a || rs != rs) st.m0 = NAN;
    }
}

// Quad mode: the element-wise half of line_accumulate for the lane's own coordinate i (pth, pr: coordinate i - 1, ar1 only) —
// a, rs enter the three fma chains in index order (seq_each), am, rm the maxima (order-free: a per-lane maximum, reduced over
// the quad after the pass; a NaN coordinate makes m0 NaN).  is: ill-normal's 1 / s_i (the lanes of a quad need four different
// ones: a vector load, fetched with the chunk's other loads)
template <int MODEL>
__device__ __forceinline__ void line_elem(int i, double is, double th, double r, double pth, double pr, double &a, double &rs, double &am, double &rm) {
    a = th; rs = r;
    if constexpr (MODEL == MODEL_ILL_NORMAL) { a = a * is; rs = rs * is; }
    if constexpr (MODEL == MODEL_AR1) {
        const double ca = 0.9, cb = sqrt(1 - ca * ca);
        if (i > 0) { a = (th - ca * pth) / cb; rs = (r - ca * pr) / cb; }
    }
    double a2 = a, r2 = rs;
    if constexpr (MODEL == MODEL_ILL_NORMAL) { a2 = a2 * is; r2 = r2 * is; }
    if constexpr (MODEL == MODEL_AR1) { const double kap = 1.9 / sqrt(1 - 0.81); a2 = a2 * kap; r2 = r2 * kap; }
    am = fabs(a2); rm = fabs(r2);
}
__device__ __forceinline__ double quad_max(double x) {
    return fmax(fmax(quad_bcast<0>(x), quad_bcast<1>(x)), fmax(quad_bcast<2>(x), quad_bcast<3>(x)));
}

// A workgroup is p.stream_waves wavefronts (1-4), each with its own tile and its own LDS column block; the workgroup shares
// one LDS copy of the ziggurat's WI / KI tables for the direction draw (4 KB; D = 100 x 64 lanes: three 51 200 B column
// blocks + the tables = 157.7 KB, one workgroup per CU).
// PAIR mode (p.stream_pair: narrow tiles, i.e. few chains, in the synchronous rounds): TWO wavefronts per tile.  With few chains
// a launch lasts as long as one wavefront's walk through its phases, and half of them do not depend on the other half: the
// next direction's normals need only the RNG state after the MH draws, the direction pass only theta' (and, at a window
// closure, the new mean / cov / eigenvectors).  Wavefront A runs the tail (MH scalars | T1 | CCIPCA or closure), wavefront B
// the head (normals | direction + line sums), each with its own LDS column, meeting at three workgroup barriers:
//   b1 after the MH step's draws (A hands the RNG state over in LDS; its scalars follow), a flag after T1 (theta' is in HBM: B waits, A goes on), b3 after the tail —
// B's direction pass goes before b3 unless a chain of the workgroup closes a window.  Every chain's arithmetic is the
// single-wavefront form's, operation for operation.
template <int MODEL, bool OVERRELAX, int VARIANT, int W, int CHDIV = 4>
__device__ __forceinline__ void stream_body(const KParams &p, const int do_tail, const int do_head, const int draw_slot, const unsigned bid,
                                            const unsigned tid) {      // bid / tid: blockIdx.x / threadIdx.x (klhr_tile_kernel passes them through an optimisation barrier)
    extern __shared__ double lds[];            // [waves][D][W] column scratch (normals, then theta - mean), then the ziggurat tables
    constexpr int QN = W == 16 ? 4 : 1;                          // 16-chain tiles: quad mode (four lanes per chain)
    const int lane = tid & 63, wv = tid >> 6;
    const int cl = QN == 4 ? lane >> 2 : lane;                  // the lane's chain of the tile
    const int q = QN == 4 ? lane & 3 : 0;                       // its phase: coordinates q, q + QN, ...
    const bool own = q == 0;                                    // the quad's lane that stores the per-chain scalars
    const bool pair = p.stream_pair != 0;
    const int role = pair ? (wv & 1) : 2;                       // 0: the tile's tail, 1: the tile's head, 2: both (one wavefront per tile)
    const int tslot = pair ? wv >> 1 : wv;                      // the workgroup's tile this wavefront works on
    const int tiles_wg = pair ? p.stream_waves >> 1 : p.stream_waves;
    const long long tile = (long long)bid * tiles_wg + tslot;
    const long long c = tile * W + cl;
    const int D = p.D, nc = p.ncol, K = p.J;
    double *Z = lds + (size_t)tslot * (pair ? 2 : 1) * D * W + cl;         // tail: u = theta' - mean
    double *Zn = pair ? Z + (size_t)D * W : Z;                              // head: the normals (pair mode: a column of its own)
    double *zwi = lds + (size_t)p.stream_waves * D * W;
    uint64_t *zki = reinterpret_cast<uint64_t *>(zwi + 256);
    // pair mode: per tile 4 x W words for the RNG hand-over, then four ints per workgroup: a chain closes a window in this launch | - |
    // the T1 flags of its (up to two) tiles
    unsigned long long *hand = reinterpret_cast<unsigned long long *>(zwi + 512) + (size_t)tslot * 4 * W + cl;
    int *closing = reinterpret_cast<int *>(zwi + 512 + (size_t)tiles_wg * 4 * W);
    int *t1_done = closing + 2 + tslot;                         // per tile: the tail's wavefront has stored theta' (pass T1)
    if ((do_head && (p.phases & PH_DIR)) || pair) {
        for (int i = tid; i < 256; i += blockDim.x) { zwi[i] = KLHR_ZIG_WI[i]; zki[i] = KLHR_ZIG_KI[i]; }
        if (pair && tid < 4) closing[tid] = 0;                 // (closing, -, t1_done of the workgroup's two tiles)
        __syncthreads();
    }
    // (a wavefront whose tile lies beyond the last chain ends here; s_barrier waits for the surviving wavefronts only)
    if (cl >= W || c >= p.C) return;
    const bool run_tail = do_tail != 0, run_head = do_head != 0;
    const long long tb = tile * D * W + cl;
    const long long tbe = tile * D * nc * W + cl;
    const long long tbv = tile * D * K * W + cl;
    const bool diag = (p.flags & FLAG_DIAG) != 0;
    double *const theta = p.theta + tb, *const rho = p.d_rho + tb;
#ifdef KLHR_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter(), st_state = 0, st_mh = 0, st_t1 = 0, st_pca = 0, st_norm = 0, st_dir = 0, st_fin = 0;
#endif

    Pcg64 rng;
    rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
    long long draw = p.counters[c * 4 + 0], omN = p.counters[c * 4 + 1], pcan = p.counters[c * 4 + 2];
    int cidx = (int)p.counters[c * 4 + 3];
    double acc = p.acc[c];
    double sm_x = 0, sm_cnt = 0, moved_f = 0;
    int Kc = p.K;
    if (VARIANT == VAR_GAUSS) { sm_x = p.kstate[c * 4 + 0]; sm_cnt = p.kstate[c * 4 + 1]; Kc = (int)p.kstate[c * 4 + 2]; moved_f = p.kstate[c * 4 + 3]; }

    KLHR_STAMP(st_state);
    // =================================================================== tail of iteration k (klhr_sinh.py:266-287)
    const bool tail_here = run_tail && role != 1, head_here = run_head && role != 0;
    const bool mh = (p.phases & PH_MH) != 0, adapt = (p.phases & PH_ADAPT) != 0;
    double zp = 0.0, ad = 0.0, na = 1.0, nn = 1.0, w = 1.0, z0 = 0.0, u01 = 0.0;
    double eta[4] = {0, 0, 0, 0};
    bool closed = false, upd = false, closing_any = false;
    if (tail_here) {
        // grad_evals += nfev1 + 8 nfev2 of the fit that preceded this tail (klhr_sinh.py:188,200)
        if (own && (p.phases & PH_FIT)) p.grad_evals[c] += (long long)KLHR_FB(p, F_NFEV1, c) + 8 * (long long)KLHR_FB(p, F_NFEV2, c);
        // the generator first: the proposal's variate(s) and the accept test's uniform are consecutive in the chain's stream (nothing between
        // them draws), so the head's wavefront can take the generator over before the MH scalars are worked out
        if (mh) {
#pragma unroll
            for (int k = 0; k < 4; k++) eta[k] = p.d_eta[c * 4 + k];
            draw += 1;
            if constexpr (VARIANT == VAR_GAUSS && OVERRELAX) { const ZpOut zo = gauss_overrelaxed_zp_call(rng, eta[0], eta[1], Kc, p.scale_clip); zp = zo.zp; rng = zo.rng; }
            else if constexpr (VARIANT == VAR_GAUSS) zp = gauss_zp<false>(rng, eta, Kc, p.scale_clip);
            else if constexpr (OVERRELAX) { const ZpOut zo = overrelaxed_zp_call(rng, eta[0], eta[1], eta[2], eta[3], p.K, p.tol, p.scale_clip, VARIANT == VAR_SUB); zp = zo.zp; rng = zo.rng; }
            else z0 = rng_std_normal_call(rng);
            u01 = rng_uniform(rng);
        }
        if (adapt && !(p.warmup < p.windowsize) && p.nclos > 0) {
            closed = draw == p.closures[cidx];
            if (closed && cidx < p.nclos - 1) cidx++;
        }
        upd = adapt && !closed;
        if (upd) omN += 1;
        w = 1.0 / (double)omN;
    }
    if (pair) {          // b1: the RNG state after the MH draws goes to the head's wavefront
        if (role == 0) {
            hand[0] = rng.sh; hand[W] = rng.sl; hand[2 * W] = rng.ih; hand[3 * W] = rng.il;
            if (closed) *closing = 1;
        }
        __syncthreads();
        if (role == 1) {
            rng.sh = hand[0]; rng.sl = hand[W]; rng.ih = hand[2 * W]; rng.il = hand[3 * W];
            closing_any = *closing != 0;
        }
    }
    if (tail_here) {
        if (mh) {
            const LineStat st = load_stat(p, c);
            nn = KLHR_FB(p, F_NN, c);
            if constexpr (VARIANT != VAR_GAUSS && !OVERRELAX) zp = sa_T(z0, eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
            double lp_p, lp_c, gd_;
            (void)line_point<MODEL, false>(p, st, zp, INFINITY, lp_p, gd_);
            (void)line_point<MODEL, false>(p, st, 0.0, INFINITY, lp_c, gd_);
            double r = lp_p;
            r -= lp_c;
            if constexpr (VARIANT == VAR_GAUSS) {
                const double gm = eta[0], gs = exp(clipc(eta[1], p.scale_clip));
                r += gauss_logq(0.0, gm, gs);
                r -= gauss_logq(zp, gm, gs);
            } else {
                r += sa_log_q(0.0, eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
                r -= sa_log_q(zp, eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
            }
            const double logu = log(u01);
            const double mn = (r != r) ? r : (r < 0 ? r : 0.0);      // np.minimum(0, r)
            const int a = logu < mn ? 1 : 0;
            ad = (double)a; na = (double)(1 - a);
            acc += (ad - acc) / (double)draw;
            if (diag && own) { double *dd = p.d_diag + c * KLHR_NDIAG; dd[KLHR_D_ZP] = zp; dd[KLHR_D_R] = r; dd[KLHR_D_LOGU] = logu; dd[KLHR_D_ACCEPT] = a; }
        }
        KLHR_STAMP(st_mh);
    }
    double u_choice = 0.0, ssq = 0.0;
    if (tail_here) {
        double *out = p.draws ? p.draws + (long long)draw_slot * p.draw_stride + c * D : nullptr;
        // ---- pass T1: theta' = a (theta + zp rho) + (1 - a) theta; OnlineMoments(theta'); Z = theta' - mean
        double d2 = 0.0;
        bool fin = true;
        double *const om_m = p.om_m + tb, *const om_v = p.om_v + tb;
        const double *const mean = p.mean + tb;
        for_chunks_pipelined<ST_CH, BufT1, QN>(D,
            [&](auto &b, int s0, auto full) {
#pragma unroll
                for (int k = 0; k < b.n; k++) {
                    const int i = (s0 + k) * QN + q;
                    const int ic = full || i < D ? i : D - 1;
                    b.th[k] = theta[ic * W];
                    b.rr[k] = mh ? rho[ic * W] : 0.0;
                    if (upd) { b.mm[k] = om_m[ic * W]; b.vv[k] = om_v[ic * W]; b.mn[k] = mean[ic * W]; }
                }
            },
            [&](auto &b, int s0, auto full) {
#pragma unroll
                for (int k = 0; k < b.n; k++) {
                    const int i = (s0 + k) * QN + q;
                    const bool valid = full || i < D;
                    double nx = b.th[k], dd = 0.0;
                    if (mh) {
                        const double rho_i = b.rr[k] / nn;
                        const double thp = add_rn(mul_rn(zp, rho_i), b.th[k]);
                        nx = add_rn(mul_rn(ad, thp), mul_rn(na, b.th[k]));     // a*thetap + (1-a)*theta
                        if (VARIANT == VAR_GAUSS) dd = (nx - b.th[k]) * (nx - b.th[k]);
                        if (valid) theta[i * W] = nx;
                    }
                    if (valid && !isfinite(nx)) fin = false;
                    if (out && valid) out[i] = nx;
                    if constexpr (VARIANT == VAR_GAUSS)
                        seq_each<QN, decltype(full)::value>(s0 + k, D, [&](int ii, double sq, double dv) { if (mh) d2 += dv; if (ii > 0 || MODEL != MODEL_FUNNEL) ssq += sq; }, nx * nx, dd);
                    else
                        seq_each<QN, decltype(full)::value>(s0 + k, D, [&](int ii, double sq) { if (ii > 0 || MODEL != MODEL_FUNNEL) ssq += sq; }, nx * nx);
                    if (upd) {
                        double mi = b.mm[k], vi = b.vv[k];
                        const double d = nx - mi;
                        mi += d * w;
                        vi += -vi * w + d * d * w * (1 - w);
                        if (valid) {
                            om_m[i * W] = mi;
                            om_v[i * W] = vi;
                            Z[i * W] = nx - b.mn[k];
                        }
                    }
                }
            });
        if (mh && VARIANT == VAR_GAUSS) moved_f = (sqrt(d2) > 0.0) ? 1.0 : 0.0;
        if (!fin) atomicOr(&p.status[c], 2);
        KLHR_STAMP(st_t1);
    }
    // the D normals of the next direction (klhr_sinh.py:207-212; eigen_method_one: rng.choice() consumes one uniform BEFORE them).
    // (64 lanes: a lane in the ziggurat's slow branch parks while the others go on; quad mode: the chain's four lanes draw four
    //  normals per round from the generator's stride-4 recurrence — device_rng.hpp)
    auto draw_normals = [&]() {
        u_choice = (p.flags & FLAG_EIGEN_ONE) ? pcg_next_double(rng) : 0.0;
        if constexpr (W >= 32) rng_std_normal_column(rng, Zn, W, D, zwi, zki);
        else rng_std_normal_quad(rng, Zn, W, D, lane, zwi, zki);
    };
    if (pair) {
        // the head's wavefront draws the normals while the tail's runs T1.  The head's direction pass needs theta'; the tail needs
        // nothing from the head here, so instead of a barrier (the tail would wait out the normals before its CCIPCA) the tail
        // raises a flag in LDS once theta' is stored (release: its stores are complete) and the head waits for it (acquire)
        if (role == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(t1_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (run_head && (p.phases & PH_DIR)) draw_normals();
            while (__hip_atomic_load(t1_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    // pass 0: the rest of the tail; the head follows in pass 1 — in pair mode its wavefront runs it in pass 0 already, beside
    // the other wavefront's CCIPCA, unless a chain of the workgroup closes a window (mean / cov / eigenvectors / ranks change)
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
    if (pass == 0 && tail_here) {
        if (closed) {
            split_close_window<W>(p, c, tb, tbe, tbv, omN, Z, q);
            omN = 0;
            pcan = 0;
            if (VARIANT == VAR_GAUSS) {      // klhr.py:212-214
                Kc = (int)(sm_x < 1.0 ? 1.0 : (sm_x > 50.0 ? 50.0 : sm_x));
                sm_x = (double)p.K; sm_cnt = 0.0;
            }
        } else if (upd) {
            if (VARIANT == VAR_GAUSS) smoother_update(sm_x, sm_cnt, moved_f != 0.0);
            if (p.flags & FLAG_SCALE_DIR_COV) {
                // OnlineMoments of _logp_grad(theta') (klhr_sinh.py:282-285): clipped at scale_clip (KLHRSINH), grad_clip
                // (SUBKLHRSINH), raw for KLHR (klhr.py:216)
                const double clipb = VARIANT == VAR_GAUSS ? INFINITY : (VARIANT == VAR_SUB ? p.grad_clip : p.scale_clip);
                double ev = 0.0, g0 = 0.0;
                if constexpr (MODEL == MODEL_FUNNEL) {
                    const double v = theta[0];
                    ev = exp(-v);
                    g0 = -v * (1.0 / 9.0) - 0.5 * (double)p.md.n + 0.5 * ev * ssq;
                }
                double *const omd_m = p.omd_m + tb, *const omd_v = p.omd_v + tb;
                for_chunks<ST_CH, QN>(D, q, [&](auto n, int i0) {
                    constexpr int N = decltype(n)::value;
                    double th[N], mm[N], vv[N];
#pragma unroll
                    for (int k = 0; k < N; k++) { th[k] = theta[(i0 + k) * W]; mm[k] = omd_m[(i0 + k) * W]; vv[k] = omd_v[(i0 + k) * W]; }
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const int i = i0 + k;
                        const double x = clipc(grad_elem<MODEL>(p, i, th[k], ev, g0), clipb);
                        double mi = mm[k], vi = vv[k];
                        const double d = x - mi;
                        mi += d * w;
                        vi += -vi * w + d * d * w * (1 - w);
                        omd_m[i * W] = mi;
                        omd_v[i * W] = vi;
                    }
                });
            }
            pcan += 1;
            split_pca_update<W, CHDIV>(p, p.pca_v + tbv, pcan, Z, q);
        }
        KLHR_STAMP(st_pca);
    }

    // =================================================================== head of iteration k+1 (klhr_sinh.py:264-265)
    if (head_here && pass == (role == 2 || closing_any ? 1 : 0)) {
        double nn = 1.0;
        LineStat st = {0, 0, 0, 0, 0, 0, 0};
        double Bs = 0, Cs = 0, m1 = 0;       // direction sums, still on the unnormalised r
        double prev_th = 0, prev_r = 0;      // (ar1: the previous coordinate)
        double n2 = 0;
        double m0l = 0, m1l = 0, nanl = 0;   // (quad mode: the lane's maxima)
        // one coordinate of the line sums (and of the direction's norm): lane = chain in index order as it comes; quad mode
        // element-wise on the lane's own coordinate, then the quad's four terms in index order
        auto line_step = [&](auto full, auto with_n2, int step, int i, bool valid, double th, double r, double is) {
            if constexpr (QN == 1) {
                if constexpr (decltype(with_n2)::value) n2 += (r + p.tol) * (r + p.tol);
                line_accumulate<MODEL>(p, i, th, r, st, Bs, Cs, m1, prev_th, prev_r);
            } else {
                double pth = 0, pr = 0;
                if constexpr (MODEL == MODEL_AR1) {
                    pth = quad_prev(th); pr = quad_prev(r);
                    if (q == 0) { pth = prev_th; pr = prev_r; }
                    prev_th = quad_bcast<3>(th); prev_r = quad_bcast<3>(r);
                }
                double a, rs, am, rm;
                line_elem<MODEL>(i, is, th, r, pth, pr, a, rs, am, rm);
                if (valid && !(MODEL == MODEL_FUNNEL && i == 0)) {
                    m0l = fmax(m0l, am); m1l = fmax(m1l, rm);
                    if (am != am || rm != rm) nanl = 1.0;
                }
                auto sums = [&](int ii, double ab, double rb) {
                    if (MODEL == MODEL_FUNNEL && ii == 0) { st.v0 = ab; st.r0 = rb; }
                    else { st.A = fma(ab, ab, st.A); Bs = fma(ab, rb, Bs); Cs = fma(rb, rb, Cs); }
                };
                if constexpr (decltype(with_n2)::value)
                    seq_each<QN, decltype(full)::value>(step, D, [&](int ii, double ab, double rb, double nb) { n2 += nb; sums(ii, ab, rb); }, a, rs, (r + p.tol) * (r + p.tol));
                else
                    seq_each<QN, decltype(full)::value>(step, D, sums, a, rs);
            }
        };
        if (p.phases & PH_DIR) {
            if (role == 2) { draw_normals(); if (QN > 1) KLHR_WAVE_SYNC(); }      // (the direction gathers normals written by the quad's other lanes)
            KLHR_STAMP(st_norm);
            const int *const rank = p.rank + tb;
            if (p.dir_m != nullptr && !(p.flags & FLAG_EIGEN_ONE)) {
                // the window-constant mean and sqrt(cov) come from the cache (written at the closure / by the state setters: the same
                // operations in the same order): two rows to read instead of eigvecs' columns + cov, no sqrt and no weights per draw
                const double *const dm = p.dir_m + tb, *const ds = p.dir_s + tb;
                for_chunks_pipelined<ST_CH, BufDirC, QN>(D,
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) {
                            const int i = (s0 + k) * QN + q;
                            const int ic = full || i < D ? i : D - 1;
                            b.m[k] = dm[ic * W]; b.s[k] = ds[ic * W];
                            b.rk[k] = rank[ic * W]; b.th[k] = theta[ic * W];
                            b.is[k] = (QN > 1 && MODEL == MODEL_ILL_NORMAL) ? p.md.aux[ic] : 1.0;
                        }
                    },
                    [&](auto &b, int s0, auto full) {
#pragma unroll
                        for (int k = 0; k < b.n; k++) {
                            const int i = (s0 + k) * QN + q;
                            const bool valid = full || i < D;
                            const double r = b.m[k] + b.s[k] * Zn[b.rk[k] * W];
                            if (valid) rho[i * W] = r;
                            line_step(full, std::true_type(), s0 + k, i, valid, b.th[k], r, b.is[k]);
                        }
                    });
            } else {
            const double *ev = p.eigvals + c * nc;
            double pj[4] = {0, 0, 0, 0};
            double sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (j < nc) sum += ev[j];
#pragma unroll
            for (int j = 0; j < 4; j++) if (j < nc) pj[j] = ev[j] / sum;
            int pick = -1;
            if (p.flags & FLAG_EIGEN_ONE) {
                // rng.choice(n, p=p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, random(), side='right')
                double cdf[4], accu = 0, last = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { accu += pj[j]; cdf[j] = accu; if (j == nc - 1) last = accu; }
                pick = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) if (j < nc && cdf[j] / last <= u_choice) pick = j + 1;
                if (pick >= nc) pick = nc - 1;
            }
            // weights of the eigenvector columns in the direction's mean (klhr.py:150 weights by raw eigenvalues)
            double wj[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wj[j] = pick >= 0 ? (j == pick ? 1.0 : 0.0) : ((VARIANT == VAR_GAUSS || VARIANT == VAR_SLICE) ? (j < nc ? ev[j] : 0.0) : pj[j]);
            const double *const evec = p.eigvecs + tbe, *const cov = p.cov + tb;
            const long long es = (long long)nc * W;
            for_chunks_pipelined<ST_CH, BufDir, QN>(D,
                [&](auto &b, int s0, auto full) {
#pragma unroll
                    for (int k = 0; k < b.n; k++) {
                        const int i = (s0 + k) * QN + q;
                        const int ic = full || i < D ? i : D - 1;
                        b.e0[k] = evec[ic * es];
                        b.e1[k] = nc > 1 ? evec[ic * es + W] : 0.0;
                        b.e2[k] = nc > 2 ? evec[ic * es + 2 * W] : 0.0;
                        b.cv[k] = cov[ic * W]; b.rk[k] = rank[ic * W]; b.th[k] = theta[ic * W];
                        b.is[k] = (QN > 1 && MODEL == MODEL_ILL_NORMAL) ? p.md.aux[ic] : 1.0;
                    }
                },
                [&](auto &b, int s0, auto full) {
#pragma unroll
                    for (int k = 0; k < b.n; k++) {
                        const int i = (s0 + k) * QN + q;
                        const bool valid = full || i < D;
                        const int ic = valid ? i : D - 1;
                        double m = 0;
                        if (pick >= 0) m = pick == 0 ? b.e0[k] : (pick == 1 ? b.e1[k] : (pick == 2 ? b.e2[k] : evec[ic * es + 3 * W]));
                        else {
                            m += wj[0] * b.e0[k];
                            if (nc > 1) m += wj[1] * b.e1[k];
                            if (nc > 2) m += wj[2] * b.e2[k];
                            if (nc > 3) m += wj[3] * evec[ic * es + 3 * W];
                        }
                        const double r = m + sqrt(b.cv[k]) * Zn[b.rk[k] * W];
                        if (valid) rho[i * W] = r;
                        line_step(full, std::true_type(), s0 + k, i, valid, b.th[k], r, b.is[k]);
                    }
                });
            }
            nn = sqrt(n2);
        } else {
            // direction given in d_rho (klhr_fit / klhr_metropolis / probes of the C ABI): nn = 1
            if constexpr (QN == 1) {
                for_chunks<ST_CH>(D, 0, [&](auto n, int i0) {
                    constexpr int N = decltype(n)::value;
                    double rr[N], th[N];
#pragma unroll
                    for (int k = 0; k < N; k++) { rr[k] = rho[(i0 + k) * W]; th[k] = theta[(i0 + k) * W]; }
#pragma unroll
                    for (int k = 0; k < N; k++) line_accumulate<MODEL>(p, i0 + k, th[k], rr[k], st, Bs, Cs, m1, prev_th, prev_r);
                });
            } else {
                for (int s0 = 0; s0 * QN < D; s0++) {
                    const int i = s0 * QN + q;
                    const bool valid = i < D;
                    const int ic = valid ? i : D - 1;
                    line_step(std::false_type(), std::false_type(), s0, i, valid, theta[ic * W], rho[ic * W], MODEL == MODEL_ILL_NORMAL ? p.md.aux[ic] : 1.0);
                }
            }
        }
        if constexpr (QN > 1) {
            st.m0 = quad_max(nanl) > 0.0 ? NAN : quad_max(m0l);
            m1 = quad_max(m1l);
        }
        {
            const double inv = 1.0 / nn;
            st.B = Bs * inv; st.C = Cs * inv * inv; st.m1 = m1 * inv;
            if (MODEL == MODEL_FUNNEL) st.r0 = st.r0 / nn;
        }
        if (diag && (p.phases & PH_DIR)) {
            // diagnostics / klhr_random_direction: leave the normalised direction itself in d_rho (nn becomes 1)
            for (int i = q; i < D; i += QN) rho[i * W] = rho[i * W] / nn;
            nn = 1.0;
        }
        if (own) {
        KLHR_FB(p, F_S0, c) = st.v0; KLHR_FB(p, F_S1, c) = st.r0; KLHR_FB(p, F_S2, c) = st.A; KLHR_FB(p, F_S3, c) = st.B;
        KLHR_FB(p, F_S4, c) = st.C; KLHR_FB(p, F_S5, c) = st.m0; KLHR_FB(p, F_S6, c) = st.m1; KLHR_FB(p, F_NN, c) = nn;
        }
        if (p.phases & PH_FIT) {
            // fit() consumes five normals — x0, then size-4 init of which the first two are overwritten (klhr_sinh.py:184,
            // 191-193); klhr.py's fit draws only x0 (klhr.py:134-139); SUBKLHRSINH draws size 3 (sub_klhr_sinh.py:184)
            const double x0v = rng_std_normal_call(rng) * p.initscale;
            double n2v = 0, n3v = 0;
            if (VARIANT != VAR_GAUSS) {
                (void)rng_std_normal_call(rng);
                (void)rng_std_normal_call(rng);
                n2v = rng_std_normal_call(rng) * p.initscale;
                if (VARIANT == VAR_SINH) n3v = rng_std_normal_call(rng) * p.initscale;
            }
            if (own) { KLHR_FB(p, F_X0, c) = x0v; KLHR_FB(p, F_N2, c) = n2v; KLHR_FB(p, F_N3, c) = n3v; }
        }
        KLHR_STAMP(st_dir);
    }
    if (pass == 0 && pair) __syncthreads();          // b3: the tail is complete
    }

    // ---- store the per-chain scalars (pair mode: the RNG state by the head's wavefront, the rest by the tail's)
    if (role != 0 && own) { p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il; }
    if (role != 1 && own) {
        p.counters[c * 4 + 0] = draw; p.counters[c * 4 + 1] = omN; p.counters[c * 4 + 2] = pcan; p.counters[c * 4 + 3] = cidx;
        p.acc[c] = acc;
        if (VARIANT == VAR_GAUSS) { p.kstate[c * 4 + 0] = sm_x; p.kstate[c * 4 + 1] = sm_cnt; p.kstate[c * 4 + 2] = (double)Kc; p.kstate[c * 4 + 3] = moved_f; }
    }
#ifdef KLHR_STAMPS
    KLHR_STAMP(st_fin);
    if (lane == 0) {      // diagnostic build: cycle totals of this launch's phases, accumulated in the tile's first diag row (pair mode: the head's wavefront in the second)
        double *ds = p.d_diag + (c + (role == 1 ? 1 : 0)) * KLHR_NDIAG;
        ds[0] += (double)st_state; ds[1] += (double)st_mh; ds[2] += (double)st_t1; ds[3] += (double)st_pca; ds[4] += (double)st_norm;
        ds[5] += (double)st_dir; ds[6] += (double)st_fin; ds[7] += 1.0;
    }
#endif
}
// the two entry points: 64-chain tiles (one wavefront per SIMD: 452 registers, the LDS column bounds the occupancy anyway) and
// 16-chain tiles (two wavefronts per SIMD)
template <int MODEL, bool OVERRELAX, int VARIANT>
__global__ __launch_bounds__(256) void klhr_stream_kernel(const KParams p, const int do_tail, const int do_head) {
    stream_body<MODEL, OVERRELAX, VARIANT, 64>(p, do_tail, do_head, p.draw_slot, blockIdx.x, threadIdx.x);
}
template <int MODEL, bool OVERRELAX, int VARIANT>
__global__ __launch_bounds__(256, 2) void klhr_stream16_kernel(const KParams p, const int do_tail, const int do_head) {
    stream_body<MODEL, OVERRELAX, VARIANT, 16>(p, do_tail, do_head, p.draw_slot, blockIdx.x, threadIdx.x);
}
// Few chains (up to 16 384): WHOLE ITERATIONS of a tile in one workgroup, all the draws of a call in one launch.  A tile of 16
// chains is two wavefronts in the streaming part (pair mode: tail | head, each in quad mode) and two wavefronts in the fit (8
// lanes per chain, 8 chains per wavefront), and nothing of an iteration crosses tiles — so the workgroup loops over the
// iterations on its own: no launch boundaries, no wait for the slowest chain of the whole launch (a tile waits for the slowest
// of its 16), and tiles drift apart, so that some stream (memory) while others fit (ALU).  Streaming and fit code are the
// kernels' own bodies.  Two builds: WAVES = 1 wavefront per SIMD (360-380 registers, up to 8 192 chains = 512 tiles = 1 024
// wavefronts) and WAVES = 2 (256 registers: the streaming body with half the chunk, 64-220 bytes of scratch; 8 193-16 384 chains).
// ill-normal-100, us per iteration, round 2's kernel sequence | this kernel: 2 048 chains 137 | 66, 8 192: 154 | 70, 16 384: 160 | 97
// (the kernel sequence with quad mode: 122).
template <int MODEL, bool OVERRELAX, int VARIANT, int WAVES>
__global__ __launch_bounds__(128, WAVES) void klhr_tile_kernel(const KParams p) {
#pragma unroll 1
    for (int it = 0; it <= p.n_iters; it++) {
        // (the indices go through an empty asm per iteration: everything derived from them — a dozen row addresses, the node
        //  constants — is then recomputed per iteration instead of being kept alive across both bodies)
        unsigned bid = blockIdx.x, tid = threadIdx.x;
        asm volatile("" : "+s"(bid));
        asm volatile("" : "+v"(tid));
        stream_body<MODEL, OVERRELAX, VARIANT, 16, WAVES == 2 ? 8 : 4>(p, it > 0, it < p.n_iters, it - 1, bid, tid);
        __syncthreads();                 // line sums, norm and the fit's normals of every chain of the tile are in HBM
        asm volatile("" : "+s"(bid));
        asm volatile("" : "+v"(tid));
        const long long task = (long long)bid * 16 + (tid >> 6) * 8 + ((tid & 63) >> 3);
#ifdef KLHR_STAMPS
        const unsigned long long tf0 = __builtin_readcyclecounter();
#endif
        if (it < p.n_iters) fitn_body<MODEL, VARIANT>(p, task, tid & 7);
#ifdef KLHR_STAMPS
        const unsigned long long tf1 = __builtin_readcyclecounter();
#endif
        __syncthreads();                 // eta and the evaluation counts are in HBM
#ifdef KLHR_STAMPS
        if ((tid & 63) == 0) {           // diagnostic build: the fit and the wait for the tile's other wavefront, fields 8 / 9 of the wavefront's row
            double *ds = p.d_diag + ((long long)bid * 16 + (tid >> 6)) * KLHR_NDIAG;
            ds[8] += (double)(tf1 - tf0); ds[9] += (double)(__builtin_readcyclecounter() - tf1);
        }
#endif
    }
}
#pragma clang fp contract(fast)

}  // namespace klhr
