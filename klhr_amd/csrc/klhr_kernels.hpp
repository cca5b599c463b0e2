// klhr_kernels.hpp — gfx950 kernels of the many-chain KL Hit-and-Run sampler.
//
// Mapping of work to the machine ("octet per chain"):
//   * one chain = 8 adjacent lanes (one lane per Gauss-Hermite node, N = 8); one 64-lane wavefront = 8 chains;
//     one workgroup = one wavefront, so there is no inter-wave synchronisation anywhere;
//   * a chain's (theta_i, rho_i) pairs live in LDS as double2 for the whole launch (one ds_read_b128 per
//     coordinate per node evaluation, broadcast inside the octet, bank-staggered across octets);
//   * RNG state, optimiser state and all scalars live in registers, replicated across the octet
//     (identical bits in all 8 lanes: every reduction is a commutative butterfly);
//   * the adaptation state (OnlineMoments, CCIPCA, published mean/cov/eigvecs) stays in HBM, touched once per
//     iteration with lane-strided, chain-contiguous accesses;
//   * a launch runs n_iters whole iterations; between two launches only HBM state survives.
//
// Control flow per iteration (klhr_sinh.py:262-289): direction -> [evaluation passes] -> MH -> adapt.
// All 8 chains of a wavefront share every evaluation pass; SciPy's data-dependent optimiser control flow
// happens between passes in opt_advance() (device_opt.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_models.hpp"
#include "device_opt.hpp"
#include "device_rng.hpp"
#include "../../include/klhr.h"

// Diagnostic build only (-DKLHR_STAMPS): per-wave cycle totals of each phase, written to d_diag[c][0..7] instead of
// the usual diagnostics.  The production library never contains a stamp.
#ifdef KLHR_STAMPS
#define KLHR_STAMP(acc) do { unsigned long long _t = __builtin_readcyclecounter(); (acc) += _t - stamp_t; stamp_t = _t; } while (0)
#else
#define KLHR_STAMP(acc) do { } while (0)
#endif

namespace klhr {

enum { PH_DIR = 1, PH_FIT = 2, PH_MH = 4, PH_ADAPT = 8, PH_ALL = 15, PH_KL = 16, PH_LINE = 32 };
enum { VAR_SINH = 0, VAR_GAUSS = 1, VAR_SUB = 2, VAR_SLICE = 3 };   // klhr_sinh.py / klhr.py / sub_klhr_sinh.py / slice.py
// OnlinePCA's own tolerance: both samplers construct OnlinePCA(D, K=J, l=l) with its default tol = 1e-10
// (klhr_sinh.py:62, klhr.py:63, onlinepca.py:5), independently of the sampler's `tol`.
#define PCA_TOL 1e-10
// Safety valve of the fit loops: SciPy's own limits (n*200 BFGS iterations, <= 100 DCSRCH + 10 wolfe2 + 100 zoom
// evaluations each) bound the two fits of a draw by ~2e5 passes; a wavefront that exceeds this leaves the loop and flags
// its unfinished chains (status value 8), so that a kernel can never spin on the device.
#define KLHR_MAX_PASSES 400000
enum { FLAG_SCALE_DIR_COV = 1, FLAG_EIGEN_ONE = 2, FLAG_DIAG = 4, FLAG_DRAWS = 8, FLAG_OVERRELAXED = 16 };

struct KParams {
    int C, D, S, ncol, J, K, flags, phases, n_iters, variant;   // variant: 0 KLHRSINH (klhr_sinh.py), 1 KLHR (klhr.py)
    ModelData md;
    double tol, scale_clip, grad_clip, initscale, l;
    double sl_w, sl_lower, sl_upper; int sl_maxiter;      // Slice (slice.py:19-22): width, bounds; evaluation cap per draw
    double xq[8], wq[8], asx[8];   // quadrature nodes, weights, asinh(node)
    double *theta;
    uint64_t *rng;
    double *mean, *cov, *eigvecs, *eigvals, *om_m, *om_v, *omd_m, *omd_v, *pca_v, *acc;
    int *rank;
    long long *counters;      // C×4: draw, om_N, pca_n, cidx
    long long *grad_evals;
    int *status;
    double *d_rho, *d_eta, *d_diag;
    double *kstate;           // C×4 (KLHR variant): Smoother x, Smoother count, current K, moved-in-last-MH-step flag
    double *draws;            // [n_iters][C][D] or nullptr
    long long warmup, windowsize;
    int nclos;
    const long long *closures;   // device array of window-closure draw indices
    // split-phase pipeline (klhr_split.hpp): tiled = 1 -> every [C][D...] state array is stored [tile][coordinate][64 lanes]
    int tiled, tw, draw_slot;    // tw: tile width (chains per tile: 16, 32 or 64)
    long long Cpad;              // chains of this launch rounded up to whole tiles
    long long fstride;           // field stride of fitbuf (the handle's padded chain count)
    long long draw_stride;       // doubles between two recorded iterations in `draws` (handle's chains x D)
    double *fitbuf;              // per-chain fit records, SoA [F_NFIELDS][fstride]
    // window-constant pieces of the direction draw (klhr_sinh.py:207-211), refreshed wherever mean-direction inputs change (window
    // closure, state setters, pooled adaptation): dir_m[i] = sum_j p_j eigvecs[i][j], dir_s[i] = sqrt(cov[i]) — the same operations in
    // the same order as the per-draw code they replace, so the direction stays bit-exact.  Not used with eigen_method_one (the
    // eigenvector is drawn anew for every direction).  nullptr for handles of the monolithic kernels.
    double *dir_m, *dir_s;
    int stream_pair;             // streaming kernel: two wavefronts per tile (tail | head), narrow tiles (klhr_split.hpp)
    int stream_waves;            // streaming kernel: wavefronts (tiles) per workgroup, sharing one LDS copy of the ziggurat tables
};

// per-chain fit record of the split pipeline, SoA: field f of chain c at fitbuf[f * Cpad + c]
enum { F_S0 = 0, F_S1, F_S2, F_S3, F_S4, F_S5, F_S6, F_NN, F_X0, F_N2, F_N3, F_MODE, F_INIT1, F_PASS, F_NFEV1, F_NFEV2, F_NFIELDS };

// index of coordinate i (and column j of ncol) of chain c in a [C][D] / [C][D][ncol] state array, either layout
__device__ __forceinline__ long long kix(const KParams &p, long long c, int i) {
    return p.tiled ? ((c / p.tw) * (long long)p.D + i) * p.tw + (c % p.tw) : c * (long long)p.D + i;
}
__device__ __forceinline__ long long kix2(const KParams &p, long long c, int i, int j, int ncol) {
    return p.tiled ? ((c / p.tw) * (long long)p.D * ncol + (long long)i * ncol + j) * p.tw + (c % p.tw) : (c * (long long)p.D + i) * ncol + j;
}

// 8-lane all-reduce with DPP cross-lane moves (no LDS traffic): xor 1 and xor 2 inside each quad (quad_perm), then the
// mirrored lane of the other quad (row_half_mirror).  Every step adds two values in commutative order, so all 8 lanes
// of an octet end with bit-identical sums.
__device__ __forceinline__ double dpp_move(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_sel == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); }
    else if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); }
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); }
    return __hiloint2double(hi, lo);
}
// contract(off) is load-bearing: if the caller's product w*x were fused into the first add, lane i would compute
// fma(w_i, x_i, round(w_j x_j)) and lane j fma(w_j, x_j, round(w_i x_i)) — sums that differ in the last bit, after which
// the lanes of an octet can take different branches of the optimiser and lose lockstep.
#pragma clang fp contract(off)
__device__ __forceinline__ double grp_sum8(double v) {
    v += dpp_move(v, 0);     // quad_perm [1,0,3,2]
    v += dpp_move(v, 1);     // quad_perm [2,3,0,1]
    v += dpp_move(v, 2);     // row_half_mirror: lane i <-> 7-i of its 8-lane half row
    return v;
}
// (contract(on): FMAs formed per expression in the front end, the same way in every kernel these helpers are inlined into;
//  under contract(fast) the back end's fusion depends on the inlining context — see device_opt.hpp)
#pragma clang fp contract(on)

// sinh-arcsinh helpers (klhr_sinh.py:78-84): s = exp(clip(eta1)) + tol, d = exp(clip(eta2)) + tol
// sub == true: SUBKLHRSINH's 3-parameter form, d == 1 and e = eta[2] (sub_klhr_sinh.py:92-97)
__device__ __forceinline__ void sa_unpack(const double *eta, double tol, double sc, double &m, double &s, double &d, double &e,
                                          bool sub = false) {
    m = eta[0];
    s = exp(clipc(eta[1], sc)) + tol;
    if (sub) { d = 1.0; e = eta[2]; return; }
    d = exp(clipc(eta[2], sc)) + tol;
    e = eta[3];
}
// klhr_sinh.py:112-114
__device__ inline double sa_T(double x, const double *eta, double tol, double sc, bool sub = false) {
    double m, s, d, e;
    sa_unpack(eta, tol, sc, m, s, d, e, sub);
    return add_rn(m, mul_rn(s, sinh(clipc((asinh(x) + e) / d, sc))));
}
// klhr_sinh.py:126-129
__device__ inline double sa_T_inv(double x, const double *eta, double tol, double sc, bool sub = false) {
    double m, s, d, e;
    sa_unpack(eta, tol, sc, m, s, d, e, sub);
    double z = (x - m) / s;
    return sinh(clipc(d * asinh(z) - e, sc));
}
// klhr_sinh.py:230-240
__device__ inline double sa_log_q(double x, const double *eta, double tol, double sc, bool sub = false) {
    double m, s, d, e;
    sa_unpack(eta, tol, sc, m, s, d, e, sub);
    double z = (x - m) / s;
    double u = clipc(d * asinh(z) - e, sc);
    double ti = sinh(u);
    double ld = -0.5 * ti * ti;
    ld += log(cosh(u));
    if (sub) ld -= eta[1]; else ld += eta[2] - eta[1];
    ld -= 0.5 * log1p(z * z);
    return ld;
}

// One objective evaluation for one chain, spread over its octet.
//   n == 4: KL(eta = xt) value and gradient (klhr_sinh.py:163-176): lane j evaluates node j
//   n == 1: (-logp, -grad.rho) at theta + xt[0]*rho (klhr_sinh.py:178-180), computed redundantly by the octet
// Over-relaxed proposal (klhr_sinh.py:215-228, Neal's ordered over-relaxation on the fitted proposal's CDF):
//   u = Phi(T_inv(0)); r ~ Binomial(K, u); r > K-r: u' = u * Beta(K-r+1, 2r-K); r < K-r: u' = 1 - (1-u) * Beta(r+1, K-2r);
//   r == K-r: u' = u; z' = T(Phi^-1(u')).
// The reference draws r and the Beta variate from SciPy's *global, unseeded* generator, so its stream cannot be matched;
// here they come from the chain's own PCG64 stream with a fixed consumption (documented in DESIGN.md §2, row a13):
//   K uniforms for the binomial (count of U_k < u), then a + b uniforms for Beta(a, b) = Ga / (Ga + Gb) with
//   Ga = -sum log(1 - U_i) over a uniforms (integer-shape gamma).  Distributions are exactly those of scipy.stats.
// Phi^-1 by Wichura's algorithm AS 241 (PPND16, relative accuracy ~1e-16; checked against scipy.special.ndtri to 1e-15
// over (1e-300, 1 - 1e-16)).  ocml's normcdfinv, inlined, cost the over-relaxed kernels ~1.4 KB of scratch per lane.
#define KLHR_H8(c0, c1, c2, c3, c4, c5, c6, c7, r) fma(fma(fma(fma(fma(fma(fma(c7, r, c6), r, c5), r, c4), r, c3), r, c2), r, c1), r, c0)
__device__ __forceinline__ double ndtri_as241(double p) {
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * KLHR_H8(3.3871328727963666080e0, 1.3314166789178437745e2, 1.9715909503065514427e3, 1.3731693765509461125e4,
                           4.5921953931549871457e4, 6.7265770927008700853e4, 3.3430575583588128105e4, 2.5090809287301226727e3, r) /
                   KLHR_H8(1.0, 4.2313330701600911252e1, 6.8718700749205790830e2, 5.3941960214247511077e3, 2.1213794301586595867e4,
                           3.9307895800092710610e4, 2.8729085735721942674e4, 5.2264952788528545610e3, r);
    }
    double r = q < 0 ? p : 1.0 - p;
    if (!(r > 0.0)) return r == 0.0 ? (q < 0 ? -INFINITY : INFINITY) : NAN;       // p <= 0, p >= 1, NaN
    r = sqrt(-log(r));
    double v;
    if (r <= 5.0) {
        r -= 1.6;
        v = KLHR_H8(1.42343711074968357734e0, 4.63033784615654529590e0, 5.76949722146069140550e0, 3.64784832476320460504e0,
                    1.27045825245236838258e0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4, r) /
            KLHR_H8(1.0, 2.05319162663775882187e0, 1.67638483018380384940e0, 6.89767334985100004550e-1, 1.48103976427480074590e-1,
                    1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9, r);
    } else {
        r -= 5.0;
        v = KLHR_H8(6.65790464350110377720e0, 5.46378491116411436990e0, 1.78482653991729133580e0, 2.96560571828504891230e-1,
                    2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7, r) /
            KLHR_H8(1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2, 7.86869131145613259100e-4,
                    1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15, r);
    }
    return q < 0 ? -v : v;
}
__device__ __forceinline__ double ndtr_erfc(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

__device__ inline double overrelaxed_zp(Pcg64 &rng, const double *eta, int K, double tol, double sc, bool sub = false) {
    const double u = ndtr_erfc(sa_T_inv(0.0, eta, tol, sc, sub));
    int r = 0;
    for (int k = 0; k < K; k++) r += (pcg_next_double(rng) < u) ? 1 : 0;
    double up = u;
    int a = 0, b = 0;
    if (r > K - r) { a = K - r + 1; b = 2 * r - K; }
    else if (r < K - r) { a = r + 1; b = K - 2 * r; }
    if (a > 0) {
        double ga = 0.0, gb = 0.0;
        for (int i = 0; i < a; i++) ga -= log(1.0 - pcg_next_double(rng));
        for (int i = 0; i < b; i++) gb -= log(1.0 - pcg_next_double(rng));
        const double v = ga / (ga + gb);
        up = (r > K - r) ? u * v : 1.0 - (1.0 - u) * v;
    }
    return sa_T(ndtri_as241(up), eta, tol, sc, sub);
}

// ---- KLHR (klhr.py): Gaussian proposal N(m, s), s = exp(clip(eta1, +-scale_clip))  (klhr.py:81-85,152-174)
__device__ __forceinline__ double gauss_logq(double x, double m, double s) {
    const double z = (x - m) / s;
    return -log(s) - 0.5 * z * z;
}
template <bool OVERRELAX>
__device__ inline double gauss_zp(Pcg64 &rng, const double *eta, int K, double sc) {
    const double m = eta[0], s = exp(clipc(eta[1], sc));
    if constexpr (!OVERRELAX) {
        return add_rn(m, mul_rn(s, rng_std_normal(rng)));          // rng.normal(loc=m, scale=s)
    } else {
        const double u = ndtr_erfc((0.0 - m) / s);
        int r = 0;
        for (int k = 0; k < K; k++) r += (pcg_next_double(rng) < u) ? 1 : 0;
        double up = u;
        int a = 0, b = 0;
        if (r > K - r) { a = K - r + 1; b = 2 * r - K; }
        else if (r < K - r) { a = r + 1; b = K - 2 * r; }
        if (a > 0) {
            double ga = 0.0, gb = 0.0;
            for (int i = 0; i < a; i++) ga -= log(1.0 - pcg_next_double(rng));
            for (int i = 0; i < b; i++) gb -= log(1.0 - pcg_next_double(rng));
            const double v = ga / (ga + gb);
            up = (r > K - r) ? u * v : 1.0 - (1.0 - u) * v;
        }
        return m + s * ndtri_as241(up);
    }
}
// The over-relaxed proposals run once per iteration but, inlined, their loops and the Phi / Phi^-1 expansions sit in the
// register allocation of the whole kernel (~1.4 KB of scratch per lane, -10 % throughput with overrelaxed=True).
// They are therefore out-of-line by-value calls (-2 %).  An earlier build of that form faulted in
// klhr_iter_kernel<glmm-poisson, over-relaxed, SUBKLHRSINH>; the cause was the misplaced AGPR copies that
// tools/exec_lint.py now catches at build time (that instantiation is flagged and rebuilt by klhr_amd/build.py).
// -DKLHR_OR_INLINE restores the inlined form.
#ifdef KLHR_OR_INLINE
#define KLHR_OR_ATTR __forceinline__
#else
#define KLHR_OR_ATTR __attribute__((noinline))
#endif
struct ZpOut { double zp; Pcg64 rng; };
__device__ KLHR_OR_ATTR ZpOut overrelaxed_zp_call(Pcg64 rng, double e0, double e1, double e2, double e3, int K, double tol,
                                                               double sc, bool sub) {
    const double eta[4] = {e0, e1, e2, e3};
    ZpOut o;
    o.zp = overrelaxed_zp(rng, eta, K, tol, sc, sub);
    o.rng = rng;
    return o;
}
__device__ KLHR_OR_ATTR ZpOut gauss_overrelaxed_zp_call(Pcg64 rng, double e0, double e1, int K, double sc) {
    const double eta[4] = {e0, e1, 0.0, 0.0};
    ZpOut o;
    o.zp = gauss_zp<true>(rng, eta, K, sc);
    o.rng = rng;
    return o;
}

// Smoother.update(d) (smoother.py:10-14) with d = +1 when the chain moved, -1 otherwise (klhr.py:219-221: _msjd stays 0)
__device__ __forceinline__ void smoother_update(double &x, double &count, bool moved) {
    count += 1.0;
    const double kk = pow(count, -0.75);
    x = kk * (x + (moved ? 1.0 : -1.0)) + (1 - kk) * x;
}

// log(cosh(a)) from m = exp(-|a|): |a| - ln 2 + log1p(m^2), log1p(u) = 2 atanh(u/(2+u)) by its odd series (s <= 1/3,
// truncation < 2e-18).  Absolute error ~2e-16, at ~40 instead of ~100 instructions for log(cosh(.)).
#pragma clang fp contract(off)   // explicit fma only: rounds identically wherever it is inlined
// 1/x for normal, finite x > 0: v_rcp_f64 (about 2^-23 relative) + two Newton steps -> within ~1 ulp, 5 instructions
// against ~12 for the IEEE division sequence.  Used where the quotient feeds rounding-level quantities only.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double log_cosh_from_exp(double absa, double m) {
    const double u = m * m;
    const double s = u * fast_rcp(2.0 + u);
    const double z = s * s;
    double q = 1.0 / 35.0;
    q = fma(q, z, 1.0 / 33.0); q = fma(q, z, 1.0 / 31.0); q = fma(q, z, 1.0 / 29.0); q = fma(q, z, 1.0 / 27.0);
    q = fma(q, z, 1.0 / 25.0); q = fma(q, z, 1.0 / 23.0); q = fma(q, z, 1.0 / 21.0); q = fma(q, z, 1.0 / 19.0);
    q = fma(q, z, 1.0 / 17.0); q = fma(q, z, 1.0 / 15.0); q = fma(q, z, 1.0 / 13.0); q = fma(q, z, 1.0 / 11.0);
    q = fma(q, z, 1.0 / 9.0);  q = fma(q, z, 1.0 / 7.0);  q = fma(q, z, 1.0 / 5.0);  q = fma(q, z, 1.0 / 3.0);
    q = fma(q, z, 1.0);
    const double lp = 2.0 * s * q;
    return absa + (lp - 0.6931471805599453094);
}
#pragma clang fp contract(fast)

template <int MODEL, int VARIANT>
__device__ __forceinline__ void eval_objective(const KParams &p, const ModelData &md, const double2 *TR, int n, const double *xt,
                                               double asx_l, double w_l, int l8, double &f, double *g) {
    if (VARIANT == VAR_GAUSS && n != 1) {
        // klhr.py:109-124: y_n = s x_n + m, raw gradient, KL = -(sum_n w_n logp_n + eta1)
        const double s_ = exp(clipc(xt[1], p.scale_clip));
        const double xn = p.xq[l8];
        const double t = s_ * xn + xt[0];
        double logp, gdot;
        model_eval_tr<MODEL>(md, TR, t, INFINITY, logp, gdot);
        const double wgr = w_l * gdot;
        const double s0 = grp_sum8(w_l * logp), s1 = grp_sum8(wgr), s2 = grp_sum8(wgr * xn * s_);
        f = -(s0 + xt[1]);
        g[0] = -s1;
        g[1] = -(s2 + 1.0);
        g[2] = g[3] = 0.0;
    } else if (n != 1) {
        constexpr bool SUB = VARIANT == VAR_SUB;          // sub_klhr_sinh.py: d == 1, e = xt[2], clip at grad_clip
        const double s_ = exp(clipc(xt[1], p.scale_clip)) + p.tol;
        const double invd = SUB ? 1.0 : 1.0 / (exp(clipc(xt[2], p.scale_clip)) + p.tol);
        const double a = (asx_l + (SUB ? xt[2] : xt[3])) * invd;
        const double ac = clipc(a, p.scale_clip);
        const double E = exp(ac), Ei = fast_rcp(E);          // one exp serves sinh, cosh and log cosh
        const double sh = 0.5 * (E - Ei), ch = 0.5 * (E + Ei);
        const double t = xt[0] + s_ * sh;
        double logp, gdot;
        model_eval_tr<MODEL>(md, TR, t, SUB ? p.grad_clip : p.scale_clip, logp, gdot);
        const double w = w_l;
        const double th = sh * fast_rcp(ch);
        const double laj = (SUB ? -xt[1] : xt[2] - xt[1]) - log_cosh_from_exp(fabs(ac), fmin(E, Ei));
        const double wg = w * gdot;
        double c0 = w * (laj - logp);
        double c1 = -wg;
        double c2 = -(wg * (s_ * sh)) - w;
        double c3 = SUB ? -(wg * (s_ * ch)) - w * th : wg * (s_ * ch * a) + w * (1.0 + th * a);
        double c4 = SUB ? 0.0 : -(wg * (s_ * ch * invd)) - w * (th * invd);
        f = grp_sum8(c0);
        g[0] = grp_sum8(c1);
        g[1] = grp_sum8(c2);
        g[2] = grp_sum8(c3);
        g[3] = grp_sum8(c4);
    } else {
        double logp, gdot;
        model_eval_split<MODEL>(md, TR, xt[0], INFINITY, l8, [](double v) { return grp_sum8(v); }, logp, gdot);
        f = -logp;
        g[0] = -gdot;
        g[1] = g[2] = g[3] = 0.0;
    }
}

// workgroup == one wavefront: LDS hand-offs between the lanes of an octet only need the wave's own LDS traffic
// ordered (fence) and the compiler kept from moving code across (wave_barrier); no s_barrier is executed.
#define KLHR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); } while (0)

// ---- bit-exact (unfused, sequential) pieces: must round exactly like the NumPy expressions they restate
#pragma clang fp contract(off)

// The window-constant pieces of the direction draw for coordinate i of one chain (KParams::dir_m / dir_s): the mean
// sum_j w_j eigvecs[i][j] (w = eigenvalue shares; klhr.py:150 weights by the raw eigenvalues) accumulated from 0 in column order, and
// sqrt(cov[i]) — exactly what the direction pass computed per draw.  ev: the chain's eigenvalues; e: its eigenvector row.
__device__ __forceinline__ void dir_weights(const KParams &p, const double *ev, double (&wj)[4]) {
    const int nc = p.ncol;
    double pj[4] = {0, 0, 0, 0};
    double sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < nc) sum += ev[j];
#pragma unroll
    for (int j = 0; j < 4; j++) if (j < nc) pj[j] = ev[j] / sum;
#pragma unroll
    for (int j = 0; j < 4; j++) wj[j] = (p.variant == VAR_GAUSS || p.variant == VAR_SLICE) ? (j < nc ? ev[j] : 0.0) : pj[j];
}
__device__ __forceinline__ double dir_mean_elem(int nc, const double (&wj)[4], double e0, double e1, double e2) {
    double m = 0;
    m += wj[0] * e0;
    if (nc > 1) m += wj[1] * e1;
    if (nc > 2) m += wj[2] * e2;
    return m;
}
// Lane-strided rows of the chain state: lane l8 of the octet owns elements l8, l8+8, ...  A row is fetched from HBM
// OCT_CH elements at a time with all loads of the chunk in flight at once (a wavefront of this kernel is alone on its
// SIMD most of the time: nothing else hides a load's latency).
constexpr int OCT_CH = 13;         // 13 * 8 = 104 coordinates per chunk

template <class T>
__device__ __forceinline__ void oct_load_chunk(const T *src, int stride, int l8, int k0, int D, T (&r)[OCT_CH]) {
#pragma unroll
    for (int k = 0; k < OCT_CH; k++) {
        int i = l8 + 8 * (k0 + k);
        i = i < D ? i : D - 1;
        r[k] = src[(long long)i * stride];
    }
}

// klhr_sinh.py:203-213 + Generator.multivariate_normal(method='svd') rank rule.  Z must hold the D standard
// normals in stream order.  Writes rho into TR[i].y.
// Two phases: the element-wise part r_i = m_i + sqrt(cov_i) z_rank(i) is lane-strided (each lane 1/8 of the
// coordinates, row loads batched); only the norm, whose summation order NumPy fixes, is accumulated over all D by every
// lane — from LDS.
__device__ inline void direction_from_normals(const KParams &p, long long c, double2 *TR, const double *Z, int l8, double u_choice) {
    const int D = p.D, nc = p.ncol;
    const double *ev = p.eigvals + c * nc;
    const double *evec = p.eigvecs + c * (long long)D * nc;
    const double *cov = p.cov + c * (long long)D;
    const int *rank = p.rank + c * (long long)D;
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
    const int nk = (D + 7) >> 3;
    for (int k0 = 0; k0 < nk; k0 += OCT_CH) {
        double e0[OCT_CH], e1[OCT_CH], cv[OCT_CH];
        int rk[OCT_CH];
        oct_load_chunk(evec, nc, l8, k0, D, e0);
        if (nc > 1) oct_load_chunk(evec + 1, nc, l8, k0, D, e1);
        oct_load_chunk(cov, 1, l8, k0, D, cv);
        oct_load_chunk(rank, 1, l8, k0, D, rk);
#pragma unroll
        for (int k = 0; k < OCT_CH; k++) {
            const int i = l8 + 8 * (k0 + k);
            if (i < D) {
                double m = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (j < nc) {
                        double e = j == 0 ? e0[k] : (j == 1 ? e1[k] : evec[i * nc + j]);
                        if (pick >= 0) { if (j == pick) m = e; }
                        else m += ((p.variant == VAR_GAUSS || p.variant == VAR_SLICE) ? ev[j] : pj[j]) * e;      // klhr.py:150 weights by raw eigenvalues
                    }
                }
                TR[i].y = m + sqrt(cv[k]) * Z[rk[k]];
            }
        }
    }
    KLHR_WAVE_SYNC();
    double nn = 0;
#pragma unroll 4
    for (int i = 0; i < D; i++) { const double r = TR[i].y; nn += (r + p.tol) * (r + p.tol); }
    nn = sqrt(nn);
    KLHR_WAVE_SYNC();
    for (int i = l8; i < D; i += 8) TR[i].y = TR[i].y / nn;
    KLHR_WAVE_SYNC();
}

// onlinemoments.py:10-18, lane-strided (element-wise: bit-exact)
__device__ inline void om_update_strided(double *m, double *v, long long N, int D, int l8, const double2 *TR, const double *Z, bool fromZ, double clipb) {
    const double w = 1.0 / (double)N;
    const int nk = (D + 7) >> 3;
    for (int k0 = 0; k0 < nk; k0 += OCT_CH) {
        double mr[OCT_CH], vr[OCT_CH];
        oct_load_chunk(m, 1, l8, k0, D, mr);
        oct_load_chunk(v, 1, l8, k0, D, vr);
#pragma unroll
        for (int k = 0; k < OCT_CH; k++) {
            const int i = l8 + 8 * (k0 + k);
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

// onlinepca.py:13-26 for one chain.  u (theta - mean) must be in Z; TR[i].y is used as column scratch.
__device__ inline void pca_update(const KParams &p, long long c, long long n_new, double2 *TR, double *Z, int l8) {
    const int D = p.D, K = p.J;
    double *V = p.pca_v + c * (long long)D * K;
    const long long lim = n_new < K ? n_new : K;
    for (int col = 0; col < lim; col++) {
        if (col == n_new - 1) {
            for (int i = l8; i < D; i += 8) V[i * K + col] = Z[i];
        } else {
            const double w = ((double)(n_new - 1) - p.l) / (double)n_new;
            {
                const int nk = (D + 7) >> 3;
                for (int k0 = 0; k0 < nk; k0 += OCT_CH) {
                    double vc[OCT_CH];
                    oct_load_chunk(V + col, K, l8, k0, D, vc);
#pragma unroll
                    for (int k = 0; k < OCT_CH; k++) { const int i = l8 + 8 * (k0 + k); if (i < D) TR[i].y = vc[k]; }
                }
            }
            KLHR_WAVE_SYNC();
            // (the two order-exact sums of a pass share one walk over the LDS column: each keeps its own order)
            double s = 0, udv = 0;
#pragma unroll 4
            for (int i = 0; i < D; i++) { double vi = TR[i].y; s += vi * vi; udv += Z[i] * vi; }
            double nv = sqrt(s);
            KLHR_WAVE_SYNC();
            for (int i = l8; i < D; i += 8) TR[i].y = w * TR[i].y + (1 - w) * Z[i] * udv / (nv + PCA_TOL);
            KLHR_WAVE_SYNC();
            s = 0; udv = 0;
#pragma unroll 4
            for (int i = 0; i < D; i++) { double vi = TR[i].y; s += vi * vi; udv += Z[i] * vi; }
            nv = sqrt(s);
            KLHR_WAVE_SYNC();
            for (int i = l8; i < D; i += 8) {
                double vi = TR[i].y;
                V[i * K + col] = vi;
                Z[i] = Z[i] - udv * vi / (nv * nv + PCA_TOL);
            }
            KLHR_WAVE_SYNC();
        }
    }
}

// window closure: klhr_sinh.py:268-280 (+ rank permutation for the next windows' direction draws)
__device__ inline void close_window(const KParams &p, long long c, long long omN, double2 *TR, double *Z, int l8) {
    const int D = p.D, K = p.J, nc = p.ncol;
    double *mean = p.mean + c * (long long)D, *cov = p.cov + c * (long long)D;
    double *om_m = p.om_m + c * (long long)D, *om_v = p.om_v + c * (long long)D;
    double *V = p.pca_v + c * (long long)D * K;
    double *evec = p.eigvecs + c * (long long)D * nc, *ev = p.eigvals + c * nc;
    int *rank = p.rank + c * (long long)D;
    for (int i = l8; i < D; i += 8) {
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
    KLHR_WAVE_SYNC();
    // rank_i = position of i in the stable descending order of cov
    for (int i = l8; i < D; i += 8) {
        double ci = Z[i];
        int r = 0;
        for (int j = 0; j < D; j++) { double cj = Z[j]; r += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
        rank[i] = r;
    }
    KLHR_WAVE_SYNC();
    // OnlinePCA.values()/vectors() (onlinepca.py:28-35)
    double vals[4] = {0, 0, 0, 0};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < K) {
            for (int i = l8; i < D; i += 8) TR[i].y = V[i * K + j];
            KLHR_WAVE_SYNC();
            double s = 0;
            for (int i = 0; i < D; i++) { double vi = TR[i].y; s += vi * vi; }
            vals[j] = sqrt(s);
            if (!isfinite(vals[j])) bad = true;
            KLHR_WAVE_SYNC();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { if (bad) vals[j] = 0.0; vals[j] += PCA_TOL; }
    for (int i = l8; i < D; i += 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) { evec[i * nc + j] = V[i * K + j] / vals[j]; V[i * K + j] = 0.0; }
    }
    if (l8 == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) ev[j] = vals[j];
    }
    // publish to the other lanes of this wave (they read mean/cov/eigvecs/rank through the vector L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
template <int MODEL, bool OVERRELAX, int VARIANT>
#ifdef KLHR_OCTET_LB2
__global__ __launch_bounds__(64, 2) void klhr_iter_kernel(const KParams p)
#else
__global__ __launch_bounds__(64) void klhr_iter_kernel(const KParams p)
#endif
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 3, l8 = lane & 7;
    const long long c = (long long)blockIdx.x * 8 + grp;
    const int D = p.D;
    // model data (ill-normal: 1/s_i table, glmm-poisson: covariates and counts) is staged in LDS behind the chain rows:
    // the log-density loops then read it next to (theta_i, rho_i) instead of going to the vector cache per element
    // (all 64 lanes copy, also those of octets beyond the last chain, which leave right after)
    ModelData md = p.md;
    const double *zwi = nullptr;          // LDS copies of the ziggurat's WI / KI tables (direction draw), when they fit
    const uint64_t *zki = nullptr;
    {
        const int naux = MODEL == MODEL_GLMM_POISSON ? 4 * p.md.n : (MODEL == MODEL_ILL_NORMAL ? D : 0);
        // (unconditional for the two targets that have model data: a pointer that may be global OR LDS is a generic pointer —
        //  every read of the staged table was a flat_load going through the vector-memory path, 17 k per wavefront-iteration
        //  on glmm-poisson; derived from `lds` on every path it is provably LDS and the reads are ds_read)
        if constexpr (MODEL == MODEL_GLMM_POISSON || MODEL == MODEL_ILL_NORMAL) {
            double *aux_l = lds + (size_t)8 * p.S;
            for (int i = lane; i < naux; i += 64) aux_l[i] = p.md.aux[i];
            md.aux = aux_l;
        }
        // (pays from D ~ 64 on: +5 % at D = 100; below that the copy costs more than the look-ups save)
        if (D >= 64 && ((size_t)8 * p.S + naux + KLHR_ZTAB_DOUBLES) * sizeof(double) <= 64 * 1024) {      // LDS: same test as launch_hot (klhr_hot.hip)
            double *w_l = lds + (size_t)8 * p.S + naux;
            uint64_t *k_l = reinterpret_cast<uint64_t *>(w_l + 256);
            for (int i = lane; i < 256; i += 64) { w_l[i] = KLHR_ZIG_WI[i]; k_l[i] = KLHR_ZIG_KI[i]; }
            zwi = w_l; zki = k_l;
        }
        KLHR_WAVE_SYNC();
    }
    if (c >= p.C) return;
    double2 *TR = reinterpret_cast<double2 *>(lds + (size_t)grp * p.S);
    double *Z = lds + (size_t)grp * p.S + 2 * (size_t)D;

    // ---- load chain state
    double *theta_g = p.theta + c * (long long)D;
    for (int i = l8; i < D; i += 8) { double2 v; v.x = theta_g[i]; v.y = 0.0; TR[i] = v; }
    Pcg64 rng;
    rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
    long long draw = p.counters[c * 4 + 0], omN = p.counters[c * 4 + 1], pcan = p.counters[c * 4 + 2];
    int cidx = (int)p.counters[c * 4 + 3];
    double acc = p.acc[c];
    long long gevals = p.grad_evals[c];
    double sm_x = 0, sm_cnt = 0, moved_f = 0;
    int Kc = p.K;
    if (VARIANT == VAR_GAUSS) { sm_x = p.kstate[c * 4 + 0]; sm_cnt = p.kstate[c * 4 + 1]; Kc = (int)p.kstate[c * 4 + 2]; moved_f = p.kstate[c * 4 + 3]; }
    const double asx_l = p.asx[l8], w_l = p.wq[l8];
    KLHR_WAVE_SYNC();

    double eta[4] = {0, 0, 0, 0};
    double dg_x0 = 0, dg_mode = 0, dg_hinv1 = 0, dg_fun2 = 0, dg_init[4] = {0, 0, 0, 0}, dg_zp = 0, dg_r = 0, dg_logu = 0;
    int dg_nfev1 = 0, dg_nit1 = 0, dg_st1 = 0, dg_nfev2 = 0, dg_nit2 = 0, dg_st2 = 0, dg_acc = 0, dg_fb = 0, dg_pass = 0;
    bool sl_capped = false, fit_capped = false;

#ifdef KLHR_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter(), st_dir = 0, st_eval = 0, st_adv = 0, st_mh = 0, st_adapt = 0, st_out = 0, st_idle = 0;
    unsigned long long st_npass = 0;
#endif
    for (int iter = 0; iter < p.n_iters; ++iter) {
        // =============================================================== direction (klhr_sinh.py:264)
        if (p.phases & PH_DIR) {
            // eigen_method_one: rng.choice() consumes one uniform BEFORE the D normals (klhr_sinh.py:207,212)
            const double u_choice = (p.flags & FLAG_EIGEN_ONE) ? pcg_next_double(rng) : 0.0;
            if (zwi) {
                for (int k = 0; k < D; k++) {
                    double z = rng_std_normal_tab(rng, zwi, zki);
                    if ((k & 7) == l8) Z[k] = z;
                }
            } else {
                for (int k = 0; k < D; k++) {
                    double z = rng_std_normal(rng);
                    if ((k & 7) == l8) Z[k] = z;
                }
            }
            KLHR_WAVE_SYNC();
            direction_from_normals(p, c, TR, Z, l8, u_choice);
        } else {
            const double *rg = p.d_rho + c * (long long)D;
            for (int i = l8; i < D; i += 8) TR[i].y = rg[i];
            KLHR_WAVE_SYNC();
        }
        if ((p.flags & FLAG_DIAG) && (p.phases & PH_DIR)) {
            double *rg = p.d_rho + c * (long long)D;
            for (int i = l8; i < D; i += 8) rg[i] = TR[i].y;
        }

        KLHR_STAMP(st_dir);
        // =============================================================== fit (klhr_sinh.py:182-201)
        if constexpr (VARIANT != VAR_SLICE) {
        if (p.phases & PH_FIT) {
            Opt o;
            {
                double x0v[4] = {rng_std_normal_call(rng) * p.initscale, 0, 0, 0};
                dg_x0 = x0v[0];
                opt_begin(o, 1, x0v, 1e-5);
            }
            bool done = false;
            int trips = 0;
            dg_pass = 0;
#ifndef KLHR_OCTET_ONE_LOOP
            // Two loops, as in the lane kernel: every octet of the wavefront finishes its mode search (cheap passes: the line
            // objective over the octet, opt_advance<1>) before the KL fits start (opt_advance<N>).  In one mixed loop a pass costs
            // a full KL evaluation as soon as one octet is in stage 2, so the wavefront paid max(n1 + n2) full passes; now
            // max(n1) cheap ones + max(n2), and the octets of a pass are all in the same stage of the optimiser.
            // Values are those of the mixed loop (the dimension-specialised opt_advance adds the same numbers without the zeros).
            constexpr int NFIT = VARIANT == VAR_GAUSS ? 2 : (VARIANT == VAR_SUB ? 3 : 4);
            bool done1 = false;
            for (;;) {
                double f = 0, g[4] = {0, 0, 0, 0};
                KLHR_STAMP(st_idle);
#ifdef KLHR_STAMPS
                st_npass++;
#endif
                if (!done1) {
                    if (opt_cache_hit(o)) { f = o.lf; g[0] = o.lg[0]; }
                    else {
                        eval_objective<MODEL, VARIANT>(p, md, TR, 1, o.xt, asx_l, w_l, l8, f, g);
                        opt_cache_store(o, f, g);
                        dg_pass++;
                    }
                    KLHR_STAMP(st_eval);
                    opt_advance<false, 1>(o, f, g);
                    if (o.pc == PC_DONE) done1 = true;
                    KLHR_STAMP(st_adv);
                }
                if (__all(done1) || ++trips > KLHR_MAX_PASSES) break;
            }
            if (done1) {
                gevals += o.nfev;
                dg_mode = o.xk[0]; dg_hinv1 = o.H[0]; dg_nfev1 = o.nfev; dg_nit1 = o.k; dg_st1 = o.status; dg_fb = o.fallbacks;
                double s = o.H[0];
                s = (s > 0 ? 1.0 : 0.0) * 0.5 * log(s);
                if (VARIANT == VAR_GAUSS) {          // klhr.py:134-139: no extra normals, default gtol
                    double init[4] = {o.xk[0], s, 0.0, 0.0};
                    for (int k = 0; k < 4; k++) dg_init[k] = init[k];
                    opt_begin(o, 2, init, 1e-5);
                } else {
                    // init = rng.normal(size=4) * initscale, then [0], [1] overwritten (klhr_sinh.py:191-193):
                    // four normals are consumed, the first two discarded; SUBKLHRSINH draws size=3 (sub_klhr_sinh.py:184)
                    (void)rng_std_normal(rng);
                    (void)rng_std_normal(rng);
                    const double n2 = rng_std_normal(rng) * p.initscale;
                    const double n3 = VARIANT == VAR_SUB ? 0.0 : rng_std_normal(rng) * p.initscale;
                    double init[4] = {o.xk[0], s, n2, n3};
                    for (int k = 0; k < 4; k++) dg_init[k] = init[k];
                    opt_begin(o, VARIANT == VAR_SUB ? 3 : 4, init, 1e-3);
                }
            }
            done = !done1;          // (a capped mode search: no KL fit; `done` is cleared again below)
            for (;;) {
                double f = 0, g[4] = {0, 0, 0, 0};
                KLHR_STAMP(st_idle);
#ifdef KLHR_STAMPS
                st_npass++;
#endif
                if (!done) {
                    if (opt_cache_hit(o)) {
                        f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3];
                    } else {
                        eval_objective<MODEL, VARIANT>(p, md, TR, NFIT, o.xt, asx_l, w_l, l8, f, g);
                        opt_cache_store(o, f, g);
                        dg_pass++;
                    }
#ifdef KLHR_FIT_PRINTF
                    if (c == 0 && l8 == 0)
                        printf("pass %d pc %d x %.17g %.17g %.17g %.17g f %.17g g %.17g %.17g %.17g %.17g oldf %.17g\n", dg_pass, (int)o.pc,
                               o.xt[0], o.xt[1], o.xt[2], o.xt[3], f, g[0], g[1], g[2], g[3], o.old_fval);
#endif
                    KLHR_STAMP(st_eval);
                    opt_advance<false, NFIT>(o, f, g);
                    if (o.pc == PC_DONE) {
                        gevals += (long long)o.nfev * 8;
                        for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                        dg_nfev2 = o.nfev; dg_nit2 = o.k; dg_st2 = o.status; dg_fun2 = o.old_fval; dg_fb += o.fallbacks;
                        done = true;
                    }
                    KLHR_STAMP(st_adv);
                }
                if (__all(done) || ++trips > KLHR_MAX_PASSES) break;
            }
            if (!done1) done = false;
#else
            for (;;) {
                double f = 0, g[4] = {0, 0, 0, 0};
                KLHR_STAMP(st_idle);
#ifdef KLHR_STAMPS
                st_npass++;
#endif
                if (!done) {
                    if (opt_cache_hit(o)) {
                        f = o.lf; g[0] = o.lg[0]; g[1] = o.lg[1]; g[2] = o.lg[2]; g[3] = o.lg[3];
                    } else {
                        eval_objective<MODEL, VARIANT>(p, md, TR, o.n, o.xt, asx_l, w_l, l8, f, g);
                        opt_cache_store(o, f, g);
                        dg_pass++;
                    }
#ifdef KLHR_FIT_PRINTF
                    if (c == 0 && l8 == 0 && o.n != 1)
                        printf("pass %d pc %d x %.17g %.17g %.17g %.17g f %.17g g %.17g %.17g %.17g %.17g oldf %.17g\n", dg_pass, (int)o.pc,
                               o.xt[0], o.xt[1], o.xt[2], o.xt[3], f, g[0], g[1], g[2], g[3], o.old_fval);
#endif
                    KLHR_STAMP(st_eval);
#ifdef KLHR_OCTET_OUTLINE
                    opt_advance<true>(o, f, g);
#else
                    opt_advance<false>(o, f, g);
#endif
                    if (o.pc == PC_DONE) {
                        if (o.n == 1) {
                            gevals += o.nfev;
                            dg_mode = o.xk[0]; dg_hinv1 = o.H[0]; dg_nfev1 = o.nfev; dg_nit1 = o.k; dg_st1 = o.status; dg_fb = o.fallbacks;
                            double s = o.H[0];
                            s = (s > 0 ? 1.0 : 0.0) * 0.5 * log(s);
                            if (VARIANT == VAR_GAUSS) {          // klhr.py:134-139: no extra normals, default gtol
                                double init[4] = {o.xk[0], s, 0.0, 0.0};
                                for (int k = 0; k < 4; k++) dg_init[k] = init[k];
                                opt_begin(o, 2, init, 1e-5);
                            } else {
                                // init = rng.normal(size=4) * initscale, then [0], [1] overwritten (klhr_sinh.py:191-193):
                                // four normals are consumed, the first two discarded (no loop: keeps init[] in registers);
                                // SUBKLHRSINH draws size=3 (sub_klhr_sinh.py:184)
                                (void)rng_std_normal(rng);
                                (void)rng_std_normal(rng);
                                const double n2 = rng_std_normal(rng) * p.initscale;
                                const double n3 = VARIANT == VAR_SUB ? 0.0 : rng_std_normal(rng) * p.initscale;
                                double init[4] = {o.xk[0], s, n2, n3};
                                for (int k = 0; k < 4; k++) dg_init[k] = init[k];
                                opt_begin(o, VARIANT == VAR_SUB ? 3 : 4, init, 1e-3);
                            }
                        } else {
                            gevals += (long long)o.nfev * 8;
                            for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                            dg_nfev2 = o.nfev; dg_nit2 = o.k; dg_st2 = o.status; dg_fun2 = o.old_fval; dg_fb += o.fallbacks;
                            done = true;
                        }
                    }
                    KLHR_STAMP(st_adv);
                }
                if (__all(done) || ++trips > KLHR_MAX_PASSES) break;
            }
#endif
            if (!done) { fit_capped = true; for (int k = 0; k < 4; k++) eta[k] = o.xk[k]; }
            if (p.flags & FLAG_DIAG) {
                const double e_sel = l8 == 0 ? eta[0] : (l8 == 1 ? eta[1] : (l8 == 2 ? eta[2] : eta[3]));
                if (l8 < 4) p.d_eta[c * 4 + l8] = e_sel;
            }
        } else {
            for (int k = 0; k < 4; k++) eta[k] = p.d_eta[c * 4 + k];
        }
        }

        // =============================================================== KL / line probes (API + tests)
        if (VARIANT != VAR_SLICE && (p.phases & (PH_KL | PH_LINE))) {
            double f, g[4];
            eval_objective<MODEL, VARIANT>(p, md, TR, (p.phases & PH_KL) ? (VARIANT == VAR_GAUSS ? 2 : (VARIANT == VAR_SUB ? 3 : 4)) : 1, eta, asx_l, w_l, l8, f, g);
            if (l8 == 0) {
                double *dd = p.d_diag + c * KLHR_NDIAG;
                dd[0] = f; dd[1] = g[0]; dd[2] = g[1]; dd[3] = g[2]; dd[4] = g[3];
            }
        }

        KLHR_STAMP(st_idle);
        // =============================================================== slice step (slice.py:86-141) instead of fit + MH
        // Neal's univariate slice sampler along rho: log y = g(0) - Exp(1); bracket (L, R) of width w placed uniformly
        // around 0 and stepped out while g > log y (m = inf); then shrinkage towards 0 until a point inside the slice is
        // drawn.  One evaluation site serves the three loops: the chains of a wavefront sit in different loops and the
        // log density g(x) = logp(theta + x rho) is the octet-cooperative evaluation of the line objective.
        if constexpr (VARIANT == VAR_SLICE) {
            if (p.phases & PH_MH) {
                draw += 1;
                int st = 0, evals = 0;             // 0: g(0), 1: stepping L out, 2: stepping R out, 3: shrinking, 4: done
                double x = 0.0, logy = 0.0, L = 0.0, R = 0.0, x1 = 0.0;
                const double w = p.sl_w;
                for (;;) {
                    if (st != 4) {
                        double lp, gd;
                        model_eval_split<MODEL>(md, TR, x, INFINITY, l8, [](double v) { return grp_sum8(v); }, lp, gd);
                        evals++;
                        const bool capped = evals > p.sl_maxiter;
                        if (st == 0) {
                            logy = lp - rng_std_exponential(rng);
                            const double u = rng_uniform_range(rng, 0.0, w);
                            L = 0.0 - u; R = 0.0 + (w - u);
                            st = 1;
                        } else if (st == 1) {
                            if (lp <= logy || capped) st = 2; else L -= w;
                        } else if (st == 2) {
                            if (lp <= logy || capped) st = 3; else R += w;
                        } else {
                            if (lp >= logy || capped) st = 4;
                            else if (x1 > 0.0) R = x1; else L = x1;
                        }
                        if (capped) { dg_st2 = 1; sl_capped = true; }
                        // next request (the bound checks of slice.py:102,108 come before the evaluation)
                        if (st == 1) { if (L <= p.sl_lower) st = 2; else x = L; }
                        if (st == 2) { if (R >= p.sl_upper) st = 3; else x = R; }
                        if (st == 3) {
                            if (L < p.sl_lower) L = p.sl_lower;
                            if (R > p.sl_upper) R = p.sl_upper;
                            x1 = rng_uniform_range(rng, L, R);
                            x = x1;
                        }
                    }
                    if (__all(st == 4)) break;
                }
                KLHR_WAVE_SYNC();
                for (int i = l8; i < D; i += 8) {
                    double2 v = TR[i];
                    v.x = add_rn(mul_rn(v.y, x1), v.x);          // theta = rho * x1 + theta
                    TR[i] = v;
                }
                acc += (1.0 - acc) / (double)draw;
                gevals += evals;
                dg_zp = x1; dg_r = logy; dg_logu = L; dg_fun2 = R; dg_nfev1 = evals; dg_acc = 1;
                KLHR_WAVE_SYNC();
            }
        } else
        // =============================================================== MH step (klhr_sinh.py:242-260)
        if (p.phases & PH_MH) {
            draw += 1;
            double zp;
            if constexpr (VARIANT == VAR_GAUSS && OVERRELAX) { const ZpOut zo = gauss_overrelaxed_zp_call(rng, eta[0], eta[1], Kc, p.scale_clip); zp = zo.zp; rng = zo.rng; }
            else if constexpr (VARIANT == VAR_GAUSS) zp = gauss_zp<false>(rng, eta, Kc, p.scale_clip);
            else if constexpr (OVERRELAX) { const ZpOut zo = overrelaxed_zp_call(rng, eta[0], eta[1], eta[2], eta[3], p.K, p.tol, p.scale_clip, VARIANT == VAR_SUB); zp = zo.zp; rng = zo.rng; }
            else zp = sa_T(rng_std_normal_call(rng), eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
            double logp, gdot;
            const double t = (l8 == 0) ? zp : 0.0;
            model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; ri = v.y; xi = fma(t, v.y, v.x); },
                              INFINITY, NoSink(), logp, gdot);
            const double lp_p = __shfl(logp, (grp << 3) + 0);
            const double lp_c = __shfl(logp, (grp << 3) + 1);
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
            const double logu = log(rng_uniform(rng));
            const double mn = (r != r) ? r : (r < 0 ? r : 0.0);      // np.minimum(0, r)
            const int a = logu < mn ? 1 : 0;
            const double ad = (double)a, na = (double)(1 - a);
            double d2 = 0.0;
            for (int i = l8; i < D; i += 8) {
                double2 v = TR[i];
                double thp = add_rn(mul_rn(zp, v.y), v.x);
                double nx = add_rn(mul_rn(ad, thp), mul_rn(na, v.x));     // a*thetap + (1-a)*theta
                d2 += (nx - v.x) * (nx - v.x);
                v.x = nx;
                TR[i] = v;
            }
            if (VARIANT == VAR_GAUSS) moved_f = (sqrt(grp_sum8(d2)) > 0.0) ? 1.0 : 0.0;   // ||theta - prev_theta|| > _msjd = 0
            acc += (ad - acc) / (double)draw;
            dg_zp = zp; dg_r = r; dg_logu = logu; dg_acc = a;
            KLHR_WAVE_SYNC();
        }

        KLHR_STAMP(st_mh);
        // =============================================================== closure-or-update (klhr_sinh.py:268-287)
        if (p.phases & PH_ADAPT) {
            bool closed = false;
            if (!(p.warmup < p.windowsize) && p.nclos > 0) {
                closed = draw == p.closures[cidx];
                if (closed && cidx < p.nclos - 1) cidx++;
            }
            if (closed) {
                close_window(p, c, omN, TR, Z, l8);
                omN = 0;
                pcan = 0;
                if (VARIANT == VAR_GAUSS) {      // klhr.py:212-214: K = int(clip(Smoother.optimum(), 1, 50)); reset
                    Kc = (int)(sm_x < 1.0 ? 1.0 : (sm_x > 50.0 ? 50.0 : sm_x));
                    sm_x = (double)p.K; sm_cnt = 0.0;
                }
            } else {
                omN += 1;
                if (VARIANT == VAR_GAUSS) smoother_update(sm_x, sm_cnt, moved_f != 0.0);
                if (p.flags & FLAG_SCALE_DIR_COV) {
                    // _logp_grad(theta): gradient clipped at +-scale_clip feeds the density moments
                    double lp_, gd_;
                    model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; xi = v.x; ri = 0.0; }, INFINITY,
                                      [&](int i, double gi) { if ((i & 7) == l8) Z[i] = gi; }, lp_, gd_);
                    KLHR_WAVE_SYNC();
                    om_update_strided(p.omd_m + c * (long long)D, p.omd_v + c * (long long)D, omN, D, l8, TR, Z, true,
                                      (VARIANT == VAR_GAUSS || VARIANT == VAR_SLICE) ? INFINITY : (VARIANT == VAR_SUB ? p.grad_clip : p.scale_clip));   // klhr.py:216, slice.py:179: raw gradient
                    KLHR_WAVE_SYNC();
                }
                om_update_strided(p.om_m + c * (long long)D, p.om_v + c * (long long)D, omN, D, l8, TR, Z, false, 0.0);
                {
                    const int nk = (D + 7) >> 3;
                    for (int k0 = 0; k0 < nk; k0 += OCT_CH) {
                        double mr[OCT_CH];
                        oct_load_chunk(p.mean + c * (long long)D, 1, l8, k0, D, mr);
#pragma unroll
                        for (int k = 0; k < OCT_CH; k++) { const int i = l8 + 8 * (k0 + k); if (i < D) Z[i] = sub_rn(TR[i].x, mr[k]); }
                    }
                }
                KLHR_WAVE_SYNC();
                pcan += 1;
                pca_update(p, c, pcan, TR, Z, l8);
            }
        }

        KLHR_STAMP(st_adapt);
        // =============================================================== outputs of this iteration
        if (p.draws) {
            double *out = p.draws + ((long long)iter * p.C + c) * D;
            for (int i = l8; i < D; i += 8) out[i] = TR[i].x;
        }
        KLHR_STAMP(st_out);
    }

    // ---- store chain state
    bool finite_theta = true;
    for (int i = l8; i < D; i += 8) { double v = TR[i].x; theta_g[i] = v; if (!isfinite(v)) finite_theta = false; }
    {
        unsigned long long b = __ballot(!finite_theta);
        if (((b >> (grp << 3)) & 0xffULL) && l8 == 0) atomicOr(&p.status[c], 2);
        if (VARIANT == VAR_SLICE && sl_capped && l8 == 0) atomicOr(&p.status[c], 4);
        if (fit_capped && l8 == 0) atomicOr(&p.status[c], 8);
    }
    if (l8 == 0) {
        p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il;
        p.counters[c * 4 + 0] = draw; p.counters[c * 4 + 1] = omN; p.counters[c * 4 + 2] = pcan; p.counters[c * 4 + 3] = cidx;
        p.acc[c] = acc;
        p.grad_evals[c] = gevals;
        if (VARIANT == VAR_GAUSS) { p.kstate[c * 4 + 0] = sm_x; p.kstate[c * 4 + 1] = sm_cnt; p.kstate[c * 4 + 2] = (double)Kc; p.kstate[c * 4 + 3] = moved_f; }
        if ((p.flags & FLAG_DIAG) && !(p.phases & (PH_KL | PH_LINE))) {
            double *dd = p.d_diag + c * KLHR_NDIAG;
            dd[KLHR_D_X0] = dg_x0; dd[KLHR_D_MODE] = dg_mode; dd[KLHR_D_HINV1] = dg_hinv1; dd[KLHR_D_NFEV1] = dg_nfev1;
            dd[KLHR_D_NIT1] = dg_nit1; dd[KLHR_D_STATUS1] = dg_st1; dd[KLHR_D_INIT0] = dg_init[0]; dd[KLHR_D_INIT1] = dg_init[1];
            dd[KLHR_D_INIT2] = dg_init[2]; dd[KLHR_D_INIT3] = dg_init[3]; dd[KLHR_D_NFEV2] = dg_nfev2; dd[KLHR_D_NIT2] = dg_nit2;
            dd[KLHR_D_STATUS2] = dg_st2; dd[KLHR_D_FUN2] = dg_fun2; dd[KLHR_D_ZP] = dg_zp; dd[KLHR_D_R] = dg_r;
            dd[KLHR_D_LOGU] = dg_logu; dd[KLHR_D_ACCEPT] = dg_acc; dd[KLHR_D_FALLBACKS] = dg_fb; dd[KLHR_D_PASSES] = dg_pass;
#ifdef KLHR_STAMPS
            dd[0] = (double)st_dir; dd[1] = (double)st_eval; dd[2] = (double)st_adv; dd[3] = (double)st_mh; dd[4] = (double)st_adapt;
            dd[5] = (double)st_out; dd[6] = (double)st_idle; dd[7] = (double)st_npass;
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MCMCBase.__init__ (mcmc.py:15) + KLHRSINH._initialize (klhr_sinh.py:86-98), one thread per chain.
template <int MODEL>
__global__ void klhr_init_kernel(const KParams p, int theta_given, int max_tries) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    const int D = p.D;
    Pcg64 rng;
    rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
    double *th = p.theta;
    if (!theta_given)
        for (int i = 0; i < D; i++) (void)rng_std_normal(rng);     // rng.normal(scale=0.1, size=D), overwritten below
    int tries = 0, st = 0;
    for (;;) {
        tries++;
        for (int i = 0; i < D; i++) th[kix(p, c, i)] = rng_std_normal(rng) * p.initscale;
        double lp, gd, n2 = 0;
        model_eval<MODEL>(p.md, [&](int i, double &xi, double &ri) { xi = th[kix(p, c, i)]; ri = 0.0; }, INFINITY,
                          [&](int, double gi) { n2 += gi * gi; }, lp, gd);
        if (isfinite(lp) && isfinite(sqrt(n2))) break;
        if (tries >= max_tries) { st = 1; break; }
    }
    p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il;
    p.status[c] = st;
}

// BSModel.log_density / log_density_gradient (bsmodel.py:15-30) for n points, one thread per point.
template <int MODEL>
__global__ void klhr_logp_kernel(ModelData md, const double *thetas, long long n, double *lp_out, double *grad_out) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double *th = thetas + k * md.D;
    double lp, gd;
    if (grad_out) {
        double *go = grad_out + k * md.D;
        model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { xi = th[i]; ri = 0.0; }, INFINITY,
                          [&](int i, double gi) { go[i] = gi; }, lp, gd);
    } else {
        model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { xi = th[i]; ri = 0.0; }, INFINITY, NoSink(), lp, gd);
    }
    lp_out[k] = lp;
}

#ifndef KLHR_HOT_TU   // plain (non-template) kernels live in the API translation unit only
// sinh-arcsinh primitives for n (x, eta) pairs (klhr_sinh.py:112-156,230-240)
__global__ void klhr_prim_kernel(const double *x, const double *eta, long long n, double tol, double sc, double *out, int sub) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double *e4 = eta + k * 4;
    const double xv = x[k];
    double m, s, d, e;
    sa_unpack(e4, tol, sc, m, s, d, e, sub != 0);
    double *o = out + k * 12;
    o[0] = sa_T(xv, e4, tol, sc, sub != 0);
    o[1] = sa_T_inv(xv, e4, tol, sc, sub != 0);
    const double a_div = (asinh(xv) + e) / d;
    o[2] = (sub ? -e4[1] : e4[2] - e4[1]) - log(cosh(clipc(a_div, sc)));
    o[3] = sa_log_q(xv, e4, tol, sc, sub != 0);
    const double invd = 1 / d, a = (asinh(xv) + e) * invd, ac = clipc(a, sc);
    const double t = tanh(ac);
    if (sub) {      // sub_klhr_sinh.py:115-121,143-149
        o[4] = 1.0; o[5] = s * sinh(ac); o[6] = s * cosh(ac); o[7] = 0.0;
        o[8] = 0.0; o[9] = -1.0; o[10] = -t; o[11] = 0.0;
    } else {
        o[4] = 1.0; o[5] = s * sinh(ac); o[6] = -s * cosh(ac) * a; o[7] = s * cosh(ac) * invd;
        o[8] = 0.0; o[9] = -1.0; o[10] = 1 + t * a; o[11] = -t * invd;
    }
}

#endif  // KLHR_HOT_TU

// MH.draw() (mh.py:18-37), one thread per chain; proposal scratch in p.d_rho (C×D)
template <int MODEL>
__global__ void klhr_mh_kernel(const KParams p, double stepsize, int seed_only) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    const int D = p.D;
    Pcg64 rng;
    rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
    double *th = p.theta, *pr = p.d_rho;          // element i of this chain: [kix(p, c, i)]
#define KI(i) kix(p, c, (i))
    long long draw = p.counters[c * 4 + 0];
    double acc = p.acc[c];
    if (seed_only) {
        for (int i = 0; i < D; i++) th[KI(i)] = 0.0 + 0.1 * rng_std_normal(rng);     // mcmc.py:15
        draw = 0; acc = 0;
    }
    for (int it = 0; it < (seed_only ? 0 : p.n_iters); ++it) {
        draw += 1;
        for (int i = 0; i < D; i++) pr[KI(i)] = rng_std_normal(rng);
        for (int i = 0; i < D; i++) pr[KI(i)] = add_rn(th[KI(i)], mul_rn(pr[KI(i)], stepsize));
        double lpp, lpc, gd;
        model_eval<MODEL>(p.md, [&](int i, double &xi, double &ri) { xi = pr[KI(i)]; ri = 0.0; }, INFINITY, NoSink(), lpp, gd);
        model_eval<MODEL>(p.md, [&](int i, double &xi, double &ri) { xi = th[KI(i)]; ri = 0.0; }, INFINITY, NoSink(), lpc, gd);
        double r = lpp;
        r -= lpc;
        double q1 = 0, q2 = 0;
        for (int i = 0; i < D; i++) { double z = (th[KI(i)] - pr[KI(i)]) / stepsize; q1 += z * z; }
        for (int i = 0; i < D; i++) { double z = (pr[KI(i)] - th[KI(i)]) / stepsize; q2 += z * z; }
        r += -0.5 * q1;
        r -= -0.5 * q2;
        const double logu = log(rng_uniform(rng));
        const double mn = (r != r) ? r : (r < 0 ? r : 0.0);
        const int a = logu < mn ? 1 : 0;
        const double ad = (double)a, na = (double)(1 - a);
        for (int i = 0; i < D; i++) th[KI(i)] = add_rn(mul_rn(ad, pr[KI(i)]), mul_rn(na, th[KI(i)]));
        acc += (ad - acc) / (double)draw;
        if (p.draws) {
            double *out = p.draws + ((long long)it * p.C + c) * D;
            for (int i = 0; i < D; i++) out[i] = th[KI(i)];
        }
    }
    p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il;
    p.counters[c * 4 + 0] = draw;
    p.acc[c] = acc;
#undef KI
}

#ifndef KLHR_HOT_TU
// refresh of the direction cache (KParams::dir_m / dir_s) from eigvecs / eigvals / cov, one thread per chain: after the state
// setters, the pooled-adaptation setters and klhr_seed (the closure inside the streaming kernel writes it itself)
#pragma clang fp contract(off)
__global__ void klhr_dircache_kernel(const KParams p) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C || !p.dir_m) return;
    const int D = p.D, nc = p.ncol;
    double ev[4] = {0, 0, 0, 0}, wj[4];
    for (int j = 0; j < 4; j++) if (j < nc) ev[j] = p.eigvals[c * nc + j];
    dir_weights(p, ev, wj);
    for (int i = 0; i < D; i++) {
        const double e0 = p.eigvecs[kix2(p, c, i, 0, nc)], e1 = nc > 1 ? p.eigvecs[kix2(p, c, i, 1, nc)] : 0.0, e2 = nc > 2 ? p.eigvecs[kix2(p, c, i, 2, nc)] : 0.0;
        p.dir_m[kix(p, c, i)] = dir_mean_elem(nc, wj, e0, e1, e2);
        p.dir_s[kix(p, c, i)] = sqrt(p.cov[kix(p, c, i)]);
    }
}
#pragma clang fp contract(fast)

// rank permutation of cov after klhr_set_state(KLHR_F_COV)
__global__ void klhr_rank_kernel(const KParams p) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    const int D = p.D;
    for (int i = 0; i < D; i++) {
        int r = 0;
        const double ci = p.cov[kix(p, c, i)];
        for (int j = 0; j < D; j++) { const double cj = p.cov[kix(p, c, j)]; r += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
        p.rank[kix(p, c, i)] = r;
    }
}

// pooled-adaptation sufficient statistics over this GPU's chains: out = [n, sum m_i (D), sum (v_i + m_i^2) (D)]
// (chains whose state has gone non-finite — status bit 1, the reference's own failure mode — are left out of the pool)
__global__ void klhr_pool_kernel(const KParams p, double *out) {
    const int i = blockIdx.x;        // coordinate
    const int C = p.C, D = p.D;
    const long long *counters = p.counters;
    const int *status = p.status;
    double sm = 0, sq = 0, sn = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (status[c] & 2) continue;
        double n = (double)counters[c * 4 + 1];
        double m = p.om_m[kix(p, c, i)], v = p.om_v[kix(p, c, i)];
        sm += n * m;
        sq += n * (v + m * m);
        sn += n;
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = sm; red[1][threadIdx.x] = sq; red[2][threadIdx.x] = sn;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; red[2][threadIdx.x] += red[2][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[1 + i] = red[0][0]; out[1 + D + i] = red[1][0]; if (i == 0) out[0] = red[2][0]; }
}

// Pooled FULL covariance (extension): Gram matrix of the recorded draws X [(n_iters*C) x D] on the matrix
// cores.  G = Xa^T Xa with Xa = [X | 1] (the appended ones column yields the column sums and the row count for free).
// One wavefront per (16x16 output tile, slab of rows); K = rows, 4 per v_mfma_f64_16x16x4_f64:
//   A operand: lane l supplies Xa[r0 + l/16][ti*16 + l%16]   (A is 16(M) x 4(K): M index = l%16, K index = l/16)
//   B operand: lane l supplies Xa[r0 + l/16][tj*16 + l%16]   (B is 4(K) x 16(N): N index = l%16, K index = l/16)
//   D result : lane l holds D[4*r + l/16][l%16], r = 0..3   (verified against numpy in tests/test_gpu_api.py)
// Slab partials are combined with fp64 atomic adds (order-dependent in the last bits: statistics, not parity data).
typedef double klhr_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void klhr_gram_mfma_kernel(const double *X, long long nrows, int D, int Dp, long long rows_per_slab,
                                                            const int *status, int C, double *G /* Dp x Dp, zero-initialised */) {
    const int lane = threadIdx.x & 63;
    const int tiles = Dp / 16;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
    if (tj < ti) return;                                    // symmetric: upper triangle of tiles only
    const long long r_begin = (long long)blockIdx.y * rows_per_slab;
    long long r_end = r_begin + rows_per_slab;
    if (r_end > nrows) r_end = nrows;
    const int ca = ti * 16 + (lane & 15), cb = tj * 16 + (lane & 15), kk = lane >> 4;
    klhr_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (long long r0 = r_begin; r0 < r_end; r0 += 4) {
        const long long r = r0 + kk;
        double a = 0.0, b = 0.0;
        if (r < r_end && !(status[r % C] & 2)) {          // rows are [iteration][chain]; non-finite chains contribute nothing
            const double *row = X + r * (long long)D;
            a = ca < D ? row[ca] : (ca == D ? 1.0 : 0.0);
            b = cb < D ? row[cb] : (cb == D ? 1.0 : 0.0);
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const int j = tj * 16 + (lane & 15);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int i = ti * 16 + 4 * rr + (lane >> 4);
        atomicAdd(&G[(long long)i * Dp + j], acc[rr]);
    }
}

// LDS-tiled form: X is read from HBM exactly once.  A workgroup of 4 wavefronts owns a slab of rows; chunks of GR rows x Dp
// columns (ones column at D, zero padding beyond) are staged in LDS by coalesced copies, every wavefront accumulates its
// share of the T(T+1)/2 upper-triangular 16x16 tiles (pair p goes to wavefront p % 4) with one v_mfma_f64_16x16x4_f64 per
// tile and 4 rows: per step a wavefront reads its T column fragments from LDS once and issues one MFMA per pair.
// Slab partials are combined with fp64 atomic adds into G (Dp x Dp, upper-triangle tiles; zeroed by the caller).
constexpr int GRAM_ROWS = 32;
template <int T, int WID>
__device__ __forceinline__ void gram_wave(const double *Xs, int DpS, int lane, int nrows_chunk, klhr_d4 (&acc)[(T * (T + 1) / 2 + 3) / 4]) {
    const int cl = lane & 15, kk = lane >> 4;
    for (int k0 = 0; k0 < nrows_chunk; k0 += 4) {
        double frag[T];
#pragma unroll
        for (int t = 0; t < T; t++) frag[t] = Xs[(k0 + kk) * DpS + t * 16 + cl];
        int p = 0, slot = 0;
#pragma unroll
        for (int ti = 0; ti < T; ti++)
#pragma unroll
            for (int tj = ti; tj < T; tj++) {
                if (p % 4 == WID) { acc[slot] = __builtin_amdgcn_mfma_f64_16x16x4f64(frag[ti], frag[tj], acc[slot], 0, 0, 0); slot++; }
                p++;
            }
    }
}
template <int T, int WID>
__device__ __forceinline__ void gram_flush(double *G, int Dp, int lane, const klhr_d4 (&acc)[(T * (T + 1) / 2 + 3) / 4]) {
    int p = 0, slot = 0;
#pragma unroll
    for (int ti = 0; ti < T; ti++)
#pragma unroll
        for (int tj = ti; tj < T; tj++) {
            if (p % 4 == WID) {
                const int j = tj * 16 + (lane & 15);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) atomicAdd(&G[(long long)(ti * 16 + 4 * rr + (lane >> 4)) * Dp + j], acc[slot][rr]);
                slot++;
            }
            p++;
        }
}
template <int T>
__global__ __launch_bounds__(256) void klhr_gram_lds_kernel(const double *X, long long nrows, int D, long long rows_per_slab, const int *status, int C, double *G) {
    extern __shared__ double Xs[];            // [GRAM_ROWS][DpS]
    constexpr int Dp = T * 16;
    // LDS row stride: consecutive rows of a ds_read_b64's 32-lane group must sit 32 banks apart (stride = 32 dwords mod 64)
    constexpr int DpS = Dp + ((T % 2 == 0) ? 16 : 0);
    constexpr int NACC = (T * (T + 1) / 2 + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long r_begin = (long long)blockIdx.x * rows_per_slab;
    long long r_end = r_begin + rows_per_slab;
    if (r_end > nrows) r_end = nrows;
    klhr_d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = klhr_d4{0.0, 0.0, 0.0, 0.0};
    for (long long r0 = r_begin; r0 < r_end; r0 += GRAM_ROWS) {
        const int nr = (int)((r_end - r0) < GRAM_ROWS ? (r_end - r0) : GRAM_ROWS);
        __syncthreads();
        // stage: rows r0 .. r0+nr-1 (contiguous nr*D doubles in X), ones column, zero padding; rows of non-finite chains zeroed
        for (int idx = tid; idx < GRAM_ROWS * Dp; idx += 256) {
            const int r = idx / Dp, col = idx - r * Dp;
            double v = 0.0;
            if (r < nr && !(status[(r0 + r) % C] & 2)) v = col < D ? X[(r0 + r) * (long long)D + col] : (col == D ? 1.0 : 0.0);
            Xs[r * DpS + col] = v;
        }
        __syncthreads();
        const int nr4 = (nr + 3) & ~3;        // (rows beyond nr are zero)
        if (wid == 0) gram_wave<T, 0>(Xs, DpS, lane, nr4, acc);
        else if (wid == 1) gram_wave<T, 1>(Xs, DpS, lane, nr4, acc);
        else if (wid == 2) gram_wave<T, 2>(Xs, DpS, lane, nr4, acc);
        else gram_wave<T, 3>(Xs, DpS, lane, nr4, acc);
    }
    if (wid == 0) gram_flush<T, 0>(G, Dp, lane, acc);
    else if (wid == 1) gram_flush<T, 1>(G, Dp, lane, acc);
    else if (wid == 2) gram_flush<T, 2>(G, Dp, lane, acc);
    else gram_flush<T, 3>(G, Dp, lane, acc);
}

// G (Dp x Dp, upper-triangle tiles) -> packed statistics += [count, sum theta (D), sum theta theta^T (D x D, symmetric)]
__global__ void klhr_gram_pack_kernel(const double *G, int D, int Dp, double *pool) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 1 + D + D * D;
    if (idx >= n) return;
    auto g = [&](int i, int j) { return i <= j ? G[(long long)i * Dp + j] : G[(long long)j * Dp + i]; };
    double v;
    if (idx == 0) v = g(D, D);
    else if (idx <= D) v = g(idx - 1, D);
    else { const int k = idx - 1 - D, i = k / D, j = k - i * D; v = g(i, j); }
    pool[idx] = v;
}

__global__ void klhr_pool_apply_full_kernel(const KParams p, const double *mean_in, const double *cov_in, const double *evec_in, const double *eval_in) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    const int D = p.D, ncol = p.ncol;
    for (int i = 0; i < D; i++) {
        p.mean[kix(p, c, i)] = mean_in[i];
        p.cov[kix(p, c, i)] = cov_in[i];
        for (int j = 0; j < ncol; j++) p.eigvecs[kix2(p, c, i, j, ncol)] = evec_in[i * ncol + j];
    }
    for (int j = 0; j < ncol; j++) p.eigvals[c * (long long)ncol + j] = eval_in[j];
    for (int i = 0; i < D; i++) {
        int r = 0;
        for (int j = 0; j < D; j++) r += (cov_in[j] > cov_in[i] || (cov_in[j] == cov_in[i] && j < i)) ? 1 : 0;
        p.rank[kix(p, c, i)] = r;
    }
}

__global__ void klhr_pool_apply_kernel(const KParams p, const double *mean_in, const double *cov_in) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    const int D = p.D;
    for (int i = 0; i < D; i++) { p.mean[kix(p, c, i)] = mean_in[i]; p.cov[kix(p, c, i)] = cov_in[i]; }
    for (int i = 0; i < D; i++) {
        int r = 0;
        for (int j = 0; j < D; j++) r += (cov_in[j] > cov_in[i] || (cov_in[j] == cov_in[i] && j < i)) ? 1 : 0;
        p.rank[kix(p, c, i)] = r;
    }
}

#endif  // KLHR_HOT_TU

}  // namespace klhr
