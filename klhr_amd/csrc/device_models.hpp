// device_models.hpp — the log densities behind BSModel (bsmodel.py:15-30) as gfx950 device functions.
//
// BridgeStan defaults propto=True, jacobian=True (bsmodel.py:18,27 pass no kwargs): constants dropped,
// change-of-variables terms included.  Closed forms are derived from the reference's Stan programs:
//   FUNNEL        stan/funnel.stan:4-10       theta = (v, alpha_1..alpha_Da)
//                 lp = -v^2/18 - Da*v/2 - exp(-v)*sum(alpha^2)/2
//   ILL_NORMAL    stan/ill-normal.stan:4-11   s_i = i/sqrt(D);  lp = -sum((y_i/s_i)^2)/2
//   GLMM_POISSON  stan/glmm-poisson.stan:6-29 theta = (alpha, beta_1..3, log sigma, eps~_1..n)
//                 lp = -alpha^2/50 - sum(beta^2)/50 - l^2/2 - sum(eps~^2)/2 + sum(C_i*eta_i - exp(eta_i)),
//                 eta_i = alpha + x_i.beta + exp(l)*eps~_i
//   NORMAL        stan/normal.stan:4-8        lp = -sum(y^2)/2
//   AR1           stan/ar1.stan:1-14          y_1 ~ N(0,1), y_i ~ N(0.9 y_{i-1}, sqrt(1 - 0.81))
//   EARNINGS      stan/earnings.stan:1-17     theta = (beta_1, beta_2, log sigma, log s); Gaussian likelihood through
//                                             centred sufficient statistics aux = [N, mean(earn), mean(height), See, Seh, Shh]
//
// One generic evaluator per model:  model_eval<MODEL>(md, xr, cl, sink, logp, gdot)
//   xr(i, xi, ri)   supplies coordinate i of the evaluation point and of the direction rho
//   cl              element-wise gradient clip bound (+inf = no clip): KLHRSINH._logp_grad clips at
//                   +-scale_clip (klhr_sinh.py:158-161), logp_grad_rho does not (klhr_sinh.py:178-180)
//   sink(i, g_i)    receives the UNclipped gradient component (no-op functor for line evaluations)
//   gdot            = sum_i clip(g_i) * rho_i   — the only thing the hot loop needs from the gradient
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace klhr {

enum { MODEL_FUNNEL = 0, MODEL_ILL_NORMAL = 1, MODEL_GLMM_POISSON = 2, MODEL_NORMAL = 3, MODEL_AR1 = 4, MODEL_EARNINGS = 5 };

struct ModelData {
    const double *aux;   // ILL_NORMAL: [1/s_i (D)] ; GLMM: [x1(n), x2(n), x3(n), C(n)]
    int D;               // unconstrained dimension
    int n;               // GLMM: years ; FUNNEL: Da
};

// np.clip semantics: NaN passes through
__device__ __forceinline__ double clipc(double x, double c) { return x < -c ? -c : (x > c ? c : x); }

struct NoSink {
    __device__ __forceinline__ void operator()(int, double) const {}
};

template <int MODEL, class XR, class SINK>
__device__ __forceinline__ void model_eval(const ModelData &md, XR xr, double cl, SINK sink, double &logp, double &gdot) {
    const int D = md.D;
    if constexpr (MODEL == MODEL_FUNNEL) {
        double v, r0;
        xr(0, v, r0);
        const double Da = (double)md.n;
        const double ev = exp(-v);
        double ss = 0.0, gd = 0.0;
        if (std::is_same<SINK, NoSink>::value && ev > 0.0 && ev < INFINITY) {
            // hot path (no gradient sink): clip with v_max/v_min; with exp(-v) finite and positive g_i is NaN exactly
            // when xi is, i.e. exactly when ss ends up NaN, so np.clip's NaN propagation is restored after the loop
            for (int i = 1; i < D; i++) {
                double xi, ri;
                xr(i, xi, ri);
                ss += xi * xi;
                gd += fmin(fmax(-ev * xi, -cl), cl) * ri;
            }
            if (ss != ss) gd = NAN;
        } else {
            for (int i = 1; i < D; i++) {
                double xi, ri;
                xr(i, xi, ri);
                ss += xi * xi;
                double gi = -ev * xi;
                sink(i, gi);
                gd += clipc(gi, cl) * ri;
            }
        }
        double g0 = -v * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev * ss;
        sink(0, g0);
        gdot = clipc(g0, cl) * r0 + gd;
        logp = -v * v * (1.0 / 18.0) - 0.5 * Da * v - 0.5 * ev * ss;
    } else if constexpr (MODEL == MODEL_ILL_NORMAL || MODEL == MODEL_NORMAL) {
        double acc = 0.0, gd = 0.0;
        if (std::is_same<SINK, NoSink>::value) {
            // hot path: g_i = -xi / s_i^2 is NaN exactly when acc ends up NaN (s_i finite, non-zero)
            for (int i = 0; i < D; i++) {
                double xi, ri;
                xr(i, xi, ri);
                double gi;
                if constexpr (MODEL == MODEL_ILL_NORMAL) { double is = md.aux[i]; double z = xi * is; acc += z * z; gi = -z * is; }
                else { acc += xi * xi; gi = -xi; }
                gd += fmin(fmax(gi, -cl), cl) * ri;
            }
            if (acc != acc) gd = NAN;
        } else {
            for (int i = 0; i < D; i++) {
                double xi, ri;
                xr(i, xi, ri);
                double gi;
                if constexpr (MODEL == MODEL_ILL_NORMAL) { double is = md.aux[i]; double z = xi * is; acc += z * z; gi = -z * is; }
                else { acc += xi * xi; gi = -xi; }
                sink(i, gi);
                gd += clipc(gi, cl) * ri;
            }
        }
        gdot = gd;
        logp = -0.5 * acc;
    } else if constexpr (MODEL == MODEL_AR1) {
        // one pass with a one-step delay: g_{i-1} = -(z_{i-1} [/beta]) + alpha z_i / beta is complete once z_i is known
        const double a = 0.9, b = sqrt(1 - a * a);
        double xp, rp;
        xr(0, xp, rp);
        double lp = -0.5 * xp * xp;
        double gprev = -xp;              // partial gradient of coordinate i-1
        double gd = 0.0;
        for (int i = 1; i < D; i++) {
            double xi, ri;
            xr(i, xi, ri);
            double z = (xi - a * xp) / b;
            lp -= 0.5 * z * z;
            double gfull = gprev + a * z / b;
            sink(i - 1, gfull);
            gd += clipc(gfull, cl) * rp;
            gprev = -z / b;
            xp = xi; rp = ri;
        }
        sink(D - 1, gprev);
        gd += clipc(gprev, cl) * rp;
        gdot = gd;
        logp = lp;
    } else if constexpr (MODEL == MODEL_EARNINGS) {
        const double N = md.aux[0], eb = md.aux[1], hb = md.aux[2], See = md.aux[3], Seh = md.aux[4], Shh = md.aux[5];
        double b1, b2, lsig, ls, r1, r2, r3, r4;
        xr(0, b1, r1); xr(1, b2, r2); xr(2, lsig, r3); xr(3, ls, r4);
        const double sigma = exp(lsig), sc = exp(ls);
        const double delta = eb - b1 - b2 * hb;
        const double SS = See - 2 * b2 * Seh + b2 * b2 * Shh + N * delta * delta;
        const double q1 = b1 * b1 / (5 * sc * sc), q2 = b2 * b2 / (5 * sc * sc);
        double lp = -0.01 * sc + ls;
        lp += -3 * log1p(q1) - ls;
        lp += -3 * log1p(q2) - ls;
        lp += -0.1 * sigma + lsig;
        lp += -N * lsig - SS / (2 * sigma * sigma);
        const double i2 = 1 / (sigma * sigma);
        const double g0 = -6 * b1 / (5 * sc * sc + b1 * b1) + N * delta * i2;
        const double g1 = -6 * b2 / (5 * sc * sc + b2 * b2) + (Seh - b2 * Shh + N * delta * hb) * i2;
        const double g2 = -0.1 * sigma + 1 - N + SS * i2;
        const double g3 = -0.01 * sc + 1 + (6 * q1 / (1 + q1) - 1) + (6 * q2 / (1 + q2) - 1);
        sink(0, g0); sink(1, g1); sink(2, g2); sink(3, g3);
        double gd = clipc(g0, cl) * r1;
        gd += clipc(g1, cl) * r2; gd += clipc(g2, cl) * r3; gd += clipc(g3, cl) * r4;
        gdot = gd;
        logp = lp;
    } else {   // MODEL_GLMM_POISSON
        const int n = md.n;
        const double *x1 = md.aux, *x2 = md.aux + n, *x3 = md.aux + 2 * n, *Cn = md.aux + 3 * n;
        double alpha, b0, b1, b2, ell, ra, rb0, rb1, rb2, rl;
        xr(0, alpha, ra); xr(1, b0, rb0); xr(2, b1, rb1); xr(3, b2, rb2); xr(4, ell, rl);
        const double sigma = exp(ell);
        double lp = -alpha * alpha * (1.0 / 50.0);
        lp -= b0 * b0 * (1.0 / 50.0); lp -= b1 * b1 * (1.0 / 50.0); lp -= b2 * b2 * (1.0 / 50.0);
        lp -= 0.5 * ell * ell;
        double sr = 0, s1 = 0, s2 = 0, s3 = 0, sre = 0, see = 0, lik = 0, gd = 0;
        for (int i = 0; i < n; i++) {
            double e, re;
            xr(5 + i, e, re);
            double eta = alpha + x1[i] * b0 + x2[i] * b1 + x3[i] * b2 + sigma * e;
            double mu = exp(eta);
            double r = Cn[i] - mu;
            lik += Cn[i] * eta - mu;
            see += e * e;
            sr += r; s1 += x1[i] * r; s2 += x2[i] * r; s3 += x3[i] * r;
            sre += r * e;
            double ge = -e + sigma * r;
            sink(5 + i, ge);
            gd += clipc(ge, cl) * re;
        }
        lp += -0.5 * see + lik;
        double g0 = -alpha * (1.0 / 25.0) + sr, g1 = -b0 * (1.0 / 25.0) + s1, g2 = -b1 * (1.0 / 25.0) + s2, g3 = -b2 * (1.0 / 25.0) + s3;
        double g4 = -ell + sigma * sre;
        sink(0, g0); sink(1, g1); sink(2, g2); sink(3, g3); sink(4, g4);
        gd += clipc(g0, cl) * ra; gd += clipc(g1, cl) * rb0; gd += clipc(g2, cl) * rb1; gd += clipc(g3, cl) * rb2;
        gd += clipc(g4, cl) * rl;
        gdot = gd;
        logp = lp;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// model_eval for a point theta + t*rho whose (theta_i, rho_i) pairs sit in LDS, without gradient sink (the octet kernel's
// KL nodes).  Same arithmetic in the same order as model_eval; the difference is the LDS schedule: coordinates are read
// in blocks of TR_BLK with the NEXT block's reads issued before the current block is consumed (two register buffers,
// ping-pong), because a wavefront alone on its SIMD otherwise waits out every block's LDS latency.
#ifndef KLHR_TR_BLK
#define KLHR_TR_BLK 8
#endif
#ifndef KLHR_GLMM_BLK
#define KLHR_GLMM_BLK 4          // years of glmm-poisson evaluated together (independent exp chains)
#endif
// exp() of U values in lock step: ocml's double-precision exp (argument reduction by ln2 in two pieces, degree-11 polynomial,
// ldexp, the two range selects), operation for operation — the results are bit-identical to exp() — but written over the
// block, so that every polynomial coefficient is ONE operand shared by the U chains (a scalar-register pair or a single
// register pair), instead of being re-materialised per call the way the inlined library code was inside the glmm year
// loop (15 v_mov per exp on top of its 27 instructions).
template <int U>
__device__ __forceinline__ void exp_block(const double (&x)[U], double (&y)[U]) {
    double dn[U], r[U], q[U];
#pragma unroll
    for (int k = 0; k < U; k++) dn[k] = __builtin_rint(x[k] * 0x1.71547652b82fep+0);
#pragma unroll
    for (int k = 0; k < U; k++) r[k] = __builtin_fma(dn[k], -0x1.62e42fefa39efp-1, x[k]);
#pragma unroll
    for (int k = 0; k < U; k++) r[k] = __builtin_fma(dn[k], -0x1.abc9e3b39803fp-56, r[k]);
#pragma unroll
    for (int k = 0; k < U; k++) q[k] = __builtin_fma(r[k], 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
#define KLHR_EXP_STEP(C) _Pragma("unroll") for (int k = 0; k < U; k++) q[k] = __builtin_fma(r[k], q[k], C);
    KLHR_EXP_STEP(0x1.71dee623fde64p-19) KLHR_EXP_STEP(0x1.a01997c89e6b0p-16) KLHR_EXP_STEP(0x1.a01a014761f6ep-13)
    KLHR_EXP_STEP(0x1.6c16c1852b7b0p-10) KLHR_EXP_STEP(0x1.1111111122322p-7) KLHR_EXP_STEP(0x1.55555555502a1p-5)
    KLHR_EXP_STEP(0x1.5555555555511p-3) KLHR_EXP_STEP(0x1.000000000000bp-1) KLHR_EXP_STEP(1.0) KLHR_EXP_STEP(1.0)
#undef KLHR_EXP_STEP
#pragma unroll
    for (int k = 0; k < U; k++) {
        double z = __builtin_amdgcn_ldexp(q[k], (int)dn[k]);
        z = x[k] > 1024.0 ? __builtin_inf() : z;
        y[k] = x[k] < -1075.0 ? 0.0 : z;
    }
}

// U years of glmm-poisson (indices first, first + step, ...) at the point theta + t rho: model_eval's arithmetic per year,
// accumulated in index order; no bounds checks — callers take whole blocks, then single years, and set gd to NaN when
// sawnan (np.clip passes a NaN gradient component through, and one NaN term makes the sum NaN).
struct GlmmSums { double sr, s1, s2, s3, sre, see, lik, gd; bool sawnan; };
template <int U>
__device__ __forceinline__ void glmm_years(const double2 *TR, const double *x1, const double *x2, const double *x3, const double *Cn,
                                           int first, int step, double t, double alpha, double b0, double b1, double b2,
                                           double sigma, double cl, GlmmSums &a) {
    double2 e2[U];
    double xa[U], xb[U], xc[U], cc[U], e[U], eta[U], mu[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        const int i = first + k * step;
        e2[k] = TR[5 + i]; xa[k] = x1[i]; xb[k] = x2[i]; xc[k] = x3[i]; cc[k] = Cn[i];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
        e[k] = fma(t, e2[k].y, e2[k].x);
        eta[k] = alpha + xa[k] * b0 + xb[k] * b1 + xc[k] * b2 + sigma * e[k];
    }
    exp_block<U>(eta, mu);
#pragma unroll
    for (int k = 0; k < U; k++) {
        const double r = cc[k] - mu[k];
        a.lik += cc[k] * eta[k] - mu[k];
        a.see += e[k] * e[k];
        a.sr += r; a.s1 += xa[k] * r; a.s2 += xb[k] * r; a.s3 += xc[k] * r;
        a.sre += r * e[k];
        const double ge = -e[k] + sigma * r;
        a.sawnan = a.sawnan || (ge != ge);                 // v_max / v_min drop NaNs: np.clip's NaN is restored by the callers
        a.gd += fmin(fmax(ge, -cl), cl) * e2[k].y;
    }
}

template <int B>
__device__ __forceinline__ void tr_load_block(const double2 *TR, int i0, int D, double2 (&r)[B]) {
    if (i0 + B <= D) {          // whole block inside the row: immediate offsets, no index arithmetic
#pragma unroll
        for (int k = 0; k < B; k++) r[k] = TR[i0 + k];
    } else {
#pragma unroll
        for (int k = 0; k < B; k++) r[k] = TR[i0 + k < D ? i0 + k : D - 1];
    }
}
template <class BODY>
__device__ __forceinline__ void tr_pipelined(const double2 *TR, int i_begin, int D, BODY body) {
    constexpr int B = KLHR_TR_BLK;
    double2 a[B], b[B];
    tr_load_block<B>(TR, i_begin, D, a);
    for (int i0 = i_begin; i0 < D; i0 += 2 * B) {
        if (i0 + B < D) tr_load_block<B>(TR, i0 + B, D, b);
        if (i0 + B <= D) {
#pragma unroll
            for (int k = 0; k < B; k++) body(i0 + k, a[k]);
        } else {
#pragma unroll
            for (int k = 0; k < B; k++) if (i0 + k < D) body(i0 + k, a[k]);
        }
        if (i0 + 2 * B < D) tr_load_block<B>(TR, i0 + 2 * B, D, a);
        if (i0 + 2 * B <= D) {
#pragma unroll
            for (int k = 0; k < B; k++) body(i0 + B + k, b[k]);
        } else if (i0 + B < D) {
#pragma unroll
            for (int k = 0; k < B; k++) if (i0 + B + k < D) body(i0 + B + k, b[k]);
        }
    }
}

template <int MODEL>
__device__ __forceinline__ void model_eval_tr(const ModelData &md, const double2 *TR, double t, double cl, double &logp, double &gdot) {
    const int D = md.D;
    if constexpr (MODEL == MODEL_FUNNEL) {
        const double2 e0 = TR[0];
        const double v = fma(t, e0.y, e0.x);
        const double ev = exp(-v);
        if (ev > 0.0 && ev < INFINITY) {
            const double Da = (double)md.n;
            double ss = 0.0, gd = 0.0;
            tr_pipelined(TR, 1, D, [&](int, double2 e) {
                const double xi = fma(t, e.y, e.x);
                ss += xi * xi;
                gd += fmin(fmax(-ev * xi, -cl), cl) * e.y;
            });
            if (ss != ss) gd = NAN;
            const double g0 = -v * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev * ss;
            gdot = clipc(g0, cl) * e0.y + gd;
            logp = -v * v * (1.0 / 18.0) - 0.5 * Da * v - 0.5 * ev * ss;
            return;
        }
    } else if constexpr (MODEL == MODEL_ILL_NORMAL || MODEL == MODEL_NORMAL) {
        double acc = 0.0, gd = 0.0;
        tr_pipelined(TR, 0, D, [&](int i, double2 e) {
            const double xi = fma(t, e.y, e.x);
            double gi;
            if constexpr (MODEL == MODEL_ILL_NORMAL) { const double is = md.aux[i]; const double z = xi * is; acc += z * z; gi = -z * is; }
            else { acc += xi * xi; gi = -xi; }
            gd += fmin(fmax(gi, -cl), cl) * e.y;
        });
        if (acc != acc) gd = NAN;
        gdot = gd;
        logp = -0.5 * acc;
        return;
    } else if constexpr (MODEL == MODEL_GLMM_POISSON) {
        // model_eval's arithmetic in model_eval's order; the years are taken KLHR_GLMM_BLK at a time so that their
        // exp() calls (the long dependent chains of this target) overlap — the wavefront is alone on its SIMD — and
        // the accumulations then run over the block in index order.
        const int n = md.n;
        const double *x1 = md.aux, *x2 = md.aux + n, *x3 = md.aux + 2 * n, *Cn = md.aux + 3 * n;
        const double2 ea = TR[0], eb0 = TR[1], eb1 = TR[2], eb2 = TR[3], el = TR[4];
        const double alpha = fma(t, ea.y, ea.x), b0 = fma(t, eb0.y, eb0.x), b1 = fma(t, eb1.y, eb1.x), b2 = fma(t, eb2.y, eb2.x);
        const double ell = fma(t, el.y, el.x), sigma = exp(ell);
        double lp = -alpha * alpha * (1.0 / 50.0);
        lp -= b0 * b0 * (1.0 / 50.0); lp -= b1 * b1 * (1.0 / 50.0); lp -= b2 * b2 * (1.0 / 50.0);
        lp -= 0.5 * ell * ell;
        GlmmSums a = {0, 0, 0, 0, 0, 0, 0, 0, false};
        constexpr int U = KLHR_GLMM_BLK;
        int i0 = 0;
        for (; i0 + U <= n; i0 += U) glmm_years<U>(TR, x1, x2, x3, Cn, i0, 1, t, alpha, b0, b1, b2, sigma, cl, a);
        for (; i0 < n; i0++) glmm_years<1>(TR, x1, x2, x3, Cn, i0, 1, t, alpha, b0, b1, b2, sigma, cl, a);
        const double sr = a.sr, s1 = a.s1, s2 = a.s2, s3 = a.s3, sre = a.sre, see = a.see, lik = a.lik;
        double gd = a.sawnan ? NAN : a.gd;
        lp += -0.5 * see + lik;
        const double g0 = -alpha * (1.0 / 25.0) + sr, g1 = -b0 * (1.0 / 25.0) + s1, g2 = -b1 * (1.0 / 25.0) + s2, g3 = -b2 * (1.0 / 25.0) + s3;
        const double g4 = -ell + sigma * sre;
        gd += clipc(g0, cl) * ea.y; gd += clipc(g1, cl) * eb0.y; gd += clipc(g2, cl) * eb1.y; gd += clipc(g3, cl) * eb2.y;
        gd += clipc(g4, cl) * el.y;
        gdot = gd;
        logp = lp;
        return;
    }
    model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; ri = v.y; xi = fma(t, v.y, v.x); }, cl, NoSink(), logp, gdot);
}

// ---------------------------------------------------------------------------------------------------------------
// Octet-cooperative evaluation of ONE point theta + t*rho: lane l8 of the chain's octet visits coordinates
// i = l8, l8+8, ... and the partial sums are combined with the commutative butterfly `sum8` (all 8 lanes end with the
// same bits).  Used by the octet kernel for the 1-D line objective and the MH step, where all 8 lanes would otherwise
// repeat the same D-long loop.  Summation order differs from model_eval (strided partials), values agree to rounding.
template <int MODEL, class SUM8>
__device__ __forceinline__ void model_eval_split(const ModelData &md, const double2 *TR, double t, double cl, int l8, SUM8 sum8,
                                                 double &logp, double &gdot) {
    const int D = md.D;
    if constexpr (MODEL == MODEL_AR1 || MODEL == MODEL_EARNINGS) {
        // chained (AR1) / 4-dimensional (earnings) targets: every lane of the octet evaluates the whole point
        model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; ri = v.y; xi = fma(t, v.y, v.x); }, cl,
                          NoSink(), logp, gdot);
    } else if constexpr (MODEL == MODEL_FUNNEL) {
        const double2 e0 = TR[0];
        const double v = fma(t, e0.y, e0.x);
        const double Da = (double)md.n;
        const double ev = exp(-v);
        double ss = 0.0, gd = 0.0;
        for (int i = 1 + l8; i < D; i += 8) {
            const double2 e = TR[i];
            double xi = fma(t, e.y, e.x);
            ss += xi * xi;
            gd += clipc(-ev * xi, cl) * e.y;
        }
        ss = sum8(ss);
        gd = sum8(gd);
        double g0 = -v * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev * ss;
        gdot = clipc(g0, cl) * e0.y + gd;
        logp = -v * v * (1.0 / 18.0) - 0.5 * Da * v - 0.5 * ev * ss;
    } else if constexpr (MODEL == MODEL_ILL_NORMAL || MODEL == MODEL_NORMAL) {
        double acc = 0.0, gd = 0.0;
        for (int i = l8; i < D; i += 8) {
            const double2 e = TR[i];
            double xi = fma(t, e.y, e.x);
            double gi;
            if constexpr (MODEL == MODEL_ILL_NORMAL) { double is = md.aux[i]; double z = xi * is; acc += z * z; gi = -z * is; }
            else { acc += xi * xi; gi = -xi; }
            gd += clipc(gi, cl) * e.y;
        }
        gdot = sum8(gd);
        logp = -0.5 * sum8(acc);
    } else {   // MODEL_GLMM_POISSON
        const int n = md.n;
        const double *x1 = md.aux, *x2 = md.aux + n, *x3 = md.aux + 2 * n, *Cn = md.aux + 3 * n;
        const double2 ea = TR[0], eb0 = TR[1], eb1 = TR[2], eb2 = TR[3], el = TR[4];
        const double alpha = fma(t, ea.y, ea.x), b0 = fma(t, eb0.y, eb0.x), b1 = fma(t, eb1.y, eb1.x), b2 = fma(t, eb2.y, eb2.x);
        const double ell = fma(t, el.y, el.x), sigma = exp(ell);
        GlmmSums a = {0, 0, 0, 0, 0, 0, 0, 0, false};
        constexpr int U = KLHR_GLMM_BLK;
        int i0 = l8;                                   // this lane's years l8, l8 + 8, ...: whole blocks, then single years
        for (; i0 + 8 * (U - 1) < n; i0 += 8 * U) glmm_years<U>(TR, x1, x2, x3, Cn, i0, 8, t, alpha, b0, b1, b2, sigma, cl, a);
        for (; i0 < n; i0 += 8) glmm_years<1>(TR, x1, x2, x3, Cn, i0, 8, t, alpha, b0, b1, b2, sigma, cl, a);
        double sr = a.sr, s1 = a.s1, s2 = a.s2, s3 = a.s3, sre = a.sre, see = a.see, lik = a.lik, gd = a.sawnan ? NAN : a.gd;
        sr = sum8(sr); s1 = sum8(s1); s2 = sum8(s2); s3 = sum8(s3); sre = sum8(sre); see = sum8(see); lik = sum8(lik); gd = sum8(gd);
        double lp = -alpha * alpha * (1.0 / 50.0);
        lp -= b0 * b0 * (1.0 / 50.0); lp -= b1 * b1 * (1.0 / 50.0); lp -= b2 * b2 * (1.0 / 50.0);
        lp -= 0.5 * ell * ell;
        lp += -0.5 * see + lik;
        double g0 = -alpha * (1.0 / 25.0) + sr, g1 = -b0 * (1.0 / 25.0) + s1, g2 = -b1 * (1.0 / 25.0) + s2, g3 = -b2 * (1.0 / 25.0) + s3, g4 = -ell + sigma * sre;
        gd += clipc(g0, cl) * ea.y; gd += clipc(g1, cl) * eb0.y; gd += clipc(g2, cl) * eb1.y; gd += clipc(g3, cl) * eb2.y;
        gd += clipc(g4, cl) * el.y;
        gdot = gd;
        logp = lp;
    }
}

}  // namespace klhr

// ---------------------------------------------------------------------------------------------------------------
// NV-way evaluator for the lane-per-chain kernel: ONE pass over a chain's (theta_i, rho_i) pairs evaluates the log
// density and clip(grad).rho at NV points theta + t[n]*rho simultaneously.  The NV accumulator chains are independent
// (instruction-level parallelism for a wavefront that is alone on its SIMD) and each LDS read is reused NV times.
// Per point the arithmetic and its order are exactly those of model_eval above.
namespace klhr {

// clip for accumulation loops: v_max/v_min (2 instructions instead of 6) drop NaNs, so NaN-ness is tracked on the side
// and re-applied to the sum — the result equals sum(np.clip(g) * rho) including NaN propagation.
__device__ __forceinline__ double clip_acc(double x, double c, bool &sawnan) {
    sawnan = sawnan || (x != x);
    return fmin(fmax(x, -c), c);
}

#ifndef KLHR_LDS_CHUNK
#define KLHR_LDS_CHUNK 3
#endif
template <int MODEL, int NV>
__device__ __forceinline__ void model_eval_nv(const ModelData &md, const double2 *TR, const double (&t)[NV], double cl,
                                              double (&logp)[NV], double (&gdot)[NV]) {
    const int D = md.D;
    if constexpr (MODEL == MODEL_AR1 || MODEL == MODEL_EARNINGS) {
#pragma unroll
        for (int n = 0; n < NV; n++) {
            const double tn = t[n];
            model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; ri = v.y; xi = fma(tn, v.y, v.x); }, cl,
                              NoSink(), logp[n], gdot[n]);
        }
    } else if constexpr (MODEL == MODEL_FUNNEL) {
        const double2 e0 = TR[0];
        const double Da = (double)md.n;
        double v[NV], ev[NV], ss[NV], gd[NV];
        bool odd = false;
#pragma unroll
        for (int n = 0; n < NV; n++) {
            v[n] = fma(t[n], e0.y, e0.x); ev[n] = exp(-v[n]); ss[n] = 0.0; gd[n] = 0.0;
            odd = odd || !(ev[n] > 0.0 && ev[n] < INFINITY);
        }
        // Hot loop: clip with v_max/v_min (which drop NaNs).  While exp(-v) is finite and positive, g_i = -exp(-v) xi is
        // NaN exactly when xi is NaN, i.e. exactly when ss ends up NaN, so NaN-ness is restored after the loop; the
        // (rare) lanes with exp(-v) in {0, inf, NaN} take the careful loop below.  Either way the value equals
        // sum_i np.clip(g_i) rho_i accumulated in index order.
        // coordinates are fetched from LDS KLHR_LDS_CHUNK at a time (one wait per chunk instead of one per coordinate)
        constexpr int CH = KLHR_LDS_CHUNK;
        for (int i = 1; i < D; i += CH) {
            double2 ec[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) ec[k] = TR[i + k < D ? i + k : i];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                if (i + k < D) {
                    const double2 e = ec[k];
#pragma unroll
                    for (int n = 0; n < NV; n++) {
                        double xi = fma(t[n], e.y, e.x);
                        ss[n] += xi * xi;
                        double gi = -ev[n] * xi;
                        gd[n] += fmin(fmax(gi, -cl), cl) * e.y;
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NV; n++) if (ss[n] != ss[n]) gd[n] = NAN;
        if (__any(odd)) {
            bool nanf[NV];
#pragma unroll
            for (int n = 0; n < NV; n++) { gd[n] = 0.0; nanf[n] = false; }
            for (int i = 1; i < D; i++) {
                const double2 ee = TR[i];
#pragma unroll
                for (int n = 0; n < NV; n++) {
                    double xi = fma(t[n], ee.y, ee.x);
                    double gi = -ev[n] * xi;
                    gd[n] += clip_acc(gi, cl, nanf[n]) * ee.y;
                }
            }
#pragma unroll
            for (int n = 0; n < NV; n++) if (nanf[n]) gd[n] = NAN;
        }
#pragma unroll
        for (int n = 0; n < NV; n++) {
            double g0 = -v[n] * (1.0 / 9.0) - 0.5 * Da + 0.5 * ev[n] * ss[n];
            gdot[n] = clipc(g0, cl) * e0.y + gd[n];
            logp[n] = -v[n] * v[n] * (1.0 / 18.0) - 0.5 * Da * v[n] - 0.5 * ev[n] * ss[n];
        }
    } else if constexpr (MODEL == MODEL_ILL_NORMAL || MODEL == MODEL_NORMAL) {
        double acc[NV], gd[NV];
        bool nanf[NV];
#pragma unroll
        for (int n = 0; n < NV; n++) { acc[n] = 0.0; gd[n] = 0.0; nanf[n] = false; }
        for (int i = 0; i < D; i++) {
            const double2 e = TR[i];
            const double is = (MODEL == MODEL_ILL_NORMAL) ? md.aux[i] : 1.0;
#pragma unroll
            for (int n = 0; n < NV; n++) {
                double xi = fma(t[n], e.y, e.x);
                double gi;
                if constexpr (MODEL == MODEL_ILL_NORMAL) { double z = xi * is; acc[n] += z * z; gi = -z * is; }
                else { acc[n] += xi * xi; gi = -xi; }
                gd[n] += clip_acc(gi, cl, nanf[n]) * e.y;
            }
        }
#pragma unroll
        for (int n = 0; n < NV; n++) { gdot[n] = nanf[n] ? NAN : gd[n]; logp[n] = -0.5 * acc[n]; }
    } else {   // MODEL_GLMM_POISSON
        const int nn = md.n;
        const double *x1 = md.aux, *x2 = md.aux + nn, *x3 = md.aux + 2 * nn, *Cn = md.aux + 3 * nn;
        const double2 ea = TR[0], eb0 = TR[1], eb1 = TR[2], eb2 = TR[3], el = TR[4];
        double alpha[NV], b0[NV], b1[NV], b2[NV], ell[NV], sigma[NV], sr[NV], s1[NV], s2[NV], s3[NV], sre[NV], see[NV], lik[NV], gd[NV];
#pragma unroll
        for (int n = 0; n < NV; n++) {
            alpha[n] = fma(t[n], ea.y, ea.x); b0[n] = fma(t[n], eb0.y, eb0.x); b1[n] = fma(t[n], eb1.y, eb1.x);
            b2[n] = fma(t[n], eb2.y, eb2.x); ell[n] = fma(t[n], el.y, el.x); sigma[n] = exp(ell[n]);
            sr[n] = s1[n] = s2[n] = s3[n] = sre[n] = see[n] = lik[n] = gd[n] = 0.0;
        }
        for (int i = 0; i < nn; i++) {
            const double2 e2 = TR[5 + i];
#pragma unroll
            for (int n = 0; n < NV; n++) {
                double e = fma(t[n], e2.y, e2.x);
                double eta = alpha[n] + x1[i] * b0[n] + x2[i] * b1[n] + x3[i] * b2[n] + sigma[n] * e;
                double mu = exp(eta);
                double r = Cn[i] - mu;
                lik[n] += Cn[i] * eta - mu;
                see[n] += e * e;
                sr[n] += r; s1[n] += x1[i] * r; s2[n] += x2[i] * r; s3[n] += x3[i] * r;
                sre[n] += r * e;
                double ge = -e + sigma[n] * r;
                gd[n] += clipc(ge, cl) * e2.y;
            }
        }
#pragma unroll
        for (int n = 0; n < NV; n++) {
            double lp = -alpha[n] * alpha[n] * (1.0 / 50.0);
            lp -= b0[n] * b0[n] * (1.0 / 50.0); lp -= b1[n] * b1[n] * (1.0 / 50.0); lp -= b2[n] * b2[n] * (1.0 / 50.0);
            lp -= 0.5 * ell[n] * ell[n];
            lp += -0.5 * see[n] + lik[n];
            double g0 = -alpha[n] * (1.0 / 25.0) + sr[n], g1 = -b0[n] * (1.0 / 25.0) + s1[n], g2 = -b1[n] * (1.0 / 25.0) + s2[n], g3 = -b2[n] * (1.0 / 25.0) + s3[n];
            double g4 = -ell[n] + sigma[n] * sre[n];
            double g = gd[n];
            g += clipc(g0, cl) * ea.y; g += clipc(g1, cl) * eb0.y; g += clipc(g2, cl) * eb1.y; g += clipc(g3, cl) * eb2.y;
            g += clipc(g4, cl) * el.y;
            gdot[n] = g;
            logp[n] = lp;
        }
    }
}

}  // namespace klhr
