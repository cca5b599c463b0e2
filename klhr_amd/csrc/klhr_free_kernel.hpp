// klhr_free_kernel.hpp — the general hot path: KLHRSINH.draw() (klhr_sinh.py:262-289) for targets whose line restriction has NO
// closed form (glmm-poisson: 40 exp per quadrature node; any log density a Stan user brings), octet per chain, with octets
// that do not wait for each other.
//
// Why.  klhr_iter_kernel (klhr_kernels.hpp) runs the 8 chains of a wavefront in lock step: direction -> fit passes until the
// slowest of the 8 fits is done -> MH -> update.  Its cycle split on glmm-poisson-45 x 8 192 (profiles/r02_stamps): 112
// wavefront-level passes per iteration for ~72 per chain, 31.7 % of the cycles spent executing evaluation passes on behalf of
// octets that had already finished, 46 % of the lanes active on issued VALU instructions.  A chain's pass count does not predict
// its next one (tools/pass_corr.py: r = 0.15-0.2), so regrouping chains cannot help; what helps is to let every octet run its
// own sequence of iterations, so that a wavefront's time is the SUM of its chains' passes over the launch (which concentrates
// like 1/sqrt(iterations)) instead of the sum of per-iteration maxima.
//
// How.  Every octet carries its own phase and iteration counter.  One trip of the kernel's loop:
//   site L   the cheap line objective (-logp, -grad.rho at theta + t rho, octet-cooperative: 1/8 of the terms per lane) for the
//            octets in the mode search — and for the two log-density values of the MH step, which are line values too;
//   site K   the KL objective (lane l8 = quadrature node l8) for the octets in the KL fit;
//   advance  opt_advance<1> / opt_advance<N> for those octets; stage transitions (the fit's normals, the MH proposal);
//   service  every octet whose fit and MH values are complete is served, ONE CHAIN AT A TIME, BY THE WHOLE WAVEFRONT: lane =
//            coordinate for the accept/reject blend, OnlineMoments, CCIPCA (or the window closure), the draw output, the next
//            direction's normals and the direction itself.  In klhr_iter_kernel these O(D) phases run on the 8 lanes of each
//            octet, all octets at once; free-running octets reach them at different trips, and an 8-lane pass would then cost
//            the wavefront 8 x as much per iteration.  On 64 lanes one chain's phases take about a sixth of the time, rows are
//            read from HBM as contiguous segments (chain-major state: a chain's row is D consecutive doubles), and the
//            order-exact sums (NumPy's summation order) are the same sequential loops over the LDS row as before.
// Arithmetic per chain is that of klhr_iter_kernel operation for operation (the same device functions for the objective, the
// optimiser, the generator; the same unfused element-wise code for the bit-exact pieces), with one exception: the two log-density
// values of the MH step come from the octet-cooperative line evaluation (strided partial sums + the butterfly) instead of one
// lane's index-order loop — the same terms in another order, i.e. rounding-level differences in log r, inside the stated tolerance
// (tests/test_gpu_parity.py: r 1e-9).  -DKLHR_FREE_MH_EXACT restores the index-order loop (then the two kernels return the same bits:
// tests/test_gpu_parity.py::test_free_running_octets_equal_lockstep_octets builds on KLHR_FREE_MH=exact at run time).
//
// Scope: whole draws (klhr_step).  The pieces of the C ABI (klhr_random_direction / klhr_fit / klhr_metropolis / klhr_adapt, the
// KL / line probes), the slice sampler and scale_dir_cov stay on klhr_iter_kernel (klhr_api.hip chooses).
#pragma once
#include "klhr_kernels.hpp"

namespace klhr {

enum { FP_HEAD0 = 0,     // needs its first direction of this launch (service, head only)
       FP_FIT1 = 1,      // mode search (klhr_sinh.py:183-190)
       FP_FIT2 = 2,      // KL fit (klhr_sinh.py:191-201)
       FP_MHP = 3,       // waiting for logp(theta + zp rho)
       FP_MHC = 4,       // waiting for logp(theta)
       FP_TAIL = 5,      // fit and MH values complete: needs the service (tail of this iteration, head of the next)
       FP_DONE = 6 };

// wave-uniform reads of another lane's registers (src_lane must be wave-uniform: v_readlane_b32 with a scalar lane select)
__device__ __forceinline__ double rl_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int src_lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src_lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ long long rl_i64(long long v, int src_lane) { return (long long)rl_u64((uint64_t)v, src_lane); }
__device__ __forceinline__ int rl_i32(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }

// ---- bit-exact (unfused, sequential) pieces on 64 lanes: lane = coordinate, one chain at a time.  They restate
// direction_from_normals / om_update_strided / pca_update / close_window of klhr_kernels.hpp element for element.
#pragma clang fp contract(off)

// klhr_sinh.py:203-213 + Generator.multivariate_normal(method='svd') rank rule; Z holds the D normals in stream order
__device__ inline void wave_direction(const KParams &p, long long c, double2 *TR, const double *Z, int lane, double u_choice) {
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
        double cdf[4], accu = 0, last = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { accu += pj[j]; cdf[j] = accu; if (j == nc - 1) last = accu; }
        pick = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < nc && cdf[j] / last <= u_choice) pick = j + 1;
        if (pick >= nc) pick = nc - 1;
    }
    for (int i = lane; i < D; i += 64) {
        double e[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < nc) e[j] = evec[i * nc + j];
        const double cv = cov[i];
        const int rk = rank[i];
        double m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < nc) {
                if (pick >= 0) { if (j == pick) m = e[j]; }
                else m += ((p.variant == VAR_GAUSS || p.variant == VAR_SLICE) ? ev[j] : pj[j]) * e[j];      // klhr.py:150 weights by raw eigenvalues
            }
        }
        TR[i].y = m + sqrt(cv) * Z[rk];
    }
    KLHR_WAVE_SYNC();
    double nn = 0;
#pragma unroll 4
    for (int i = 0; i < D; i++) { const double r = TR[i].y; nn += (r + p.tol) * (r + p.tol); }
    nn = sqrt(nn);
    KLHR_WAVE_SYNC();
    for (int i = lane; i < D; i += 64) TR[i].y = TR[i].y / nn;
    KLHR_WAVE_SYNC();
}

// onlinemoments.py:10-18 on theta' (TR[i].x)
__device__ inline void wave_om_update(double *m, double *v, long long N, int D, int lane, const double2 *TR) {
    const double w = 1.0 / (double)N;
    for (int i = lane; i < D; i += 64) {
        const double x = TR[i].x;
        double mi = m[i], vi = v[i];
        const double d = x - mi;
        mi += d * w;
        vi += -vi * w + d * d * w * (1 - w);
        m[i] = mi;
        v[i] = vi;
    }
}

// onlinepca.py:13-26; u = theta' - mean in Z, TR[i].y is the column scratch (the direction is no longer needed)
__device__ inline void wave_pca_update(const KParams &p, long long c, long long n_new, double2 *TR, double *Z, int lane) {
    const int D = p.D, K = p.J;
    double *V = p.pca_v + c * (long long)D * K;
    const long long lim = n_new < K ? n_new : K;
    for (int col = 0; col < lim; col++) {
        if (col == n_new - 1) {
            for (int i = lane; i < D; i += 64) V[i * K + col] = Z[i];
        } else {
            const double w = ((double)(n_new - 1) - p.l) / (double)n_new;
            for (int i = lane; i < D; i += 64) TR[i].y = V[i * K + col];
            KLHR_WAVE_SYNC();
            double s = 0, udv = 0;
#pragma unroll 4
            for (int i = 0; i < D; i++) { const double vi = TR[i].y; s += vi * vi; udv += Z[i] * vi; }
            double nv = sqrt(s);
            KLHR_WAVE_SYNC();
            for (int i = lane; i < D; i += 64) TR[i].y = w * TR[i].y + (1 - w) * Z[i] * udv / (nv + PCA_TOL);
            KLHR_WAVE_SYNC();
            s = 0; udv = 0;
#pragma unroll 4
            for (int i = 0; i < D; i++) { const double vi = TR[i].y; s += vi * vi; udv += Z[i] * vi; }
            nv = sqrt(s);
            KLHR_WAVE_SYNC();
            for (int i = lane; i < D; i += 64) {
                const double vi = TR[i].y;
                V[i * K + col] = vi;
                Z[i] = Z[i] - udv * vi / (nv * nv + PCA_TOL);
            }
            KLHR_WAVE_SYNC();
        }
    }
}

// window closure: klhr_sinh.py:268-280 (+ the rank permutation of the following windows' direction draws); no scale_dir_cov here
__device__ inline void wave_close_window(const KParams &p, long long c, long long omN, double2 *TR, double *Z, int lane) {
    const int D = p.D, K = p.J, nc = p.ncol;
    double *mean = p.mean + c * (long long)D, *cov = p.cov + c * (long long)D;
    double *om_m = p.om_m + c * (long long)D, *om_v = p.om_v + c * (long long)D;
    double *V = p.pca_v + c * (long long)D * K;
    double *evec = p.eigvecs + c * (long long)D * nc, *ev = p.eigvals + c * nc;
    int *rank = p.rank + c * (long long)D;
    for (int i = lane; i < D; i += 64) {
        mean[i] = om_m[i];
        const double cv = omN > 2 ? om_v[i] * (double)omN / (double)(omN - 1) : 1.0;
        cov[i] = cv;
        Z[i] = cv;
        om_m[i] = 0.0; om_v[i] = 0.0;
    }
    KLHR_WAVE_SYNC();
    for (int i = lane; i < D; i += 64) {
        const double ci = Z[i];
        int r = 0;
        for (int j = 0; j < D; j++) { const double cj = Z[j]; r += (cj > ci || (cj == ci && j < i)) ? 1 : 0; }
        rank[i] = r;
    }
    KLHR_WAVE_SYNC();
    double vals[4] = {0, 0, 0, 0};
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (j < K) {
            for (int i = lane; i < D; i += 64) TR[i].y = V[i * K + j];
            KLHR_WAVE_SYNC();
            double s = 0;
            for (int i = 0; i < D; i++) { const double vi = TR[i].y; s += vi * vi; }
            vals[j] = sqrt(s);
            if (!isfinite(vals[j])) bad = true;
            KLHR_WAVE_SYNC();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { if (bad) vals[j] = 0.0; vals[j] += PCA_TOL; }
    for (int i = lane; i < D; i += 64) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) { evec[i * nc + j] = V[i * K + j] / vals[j]; V[i * K + j] = 0.0; }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (j < K) ev[j] = vals[j];
    }
    // publish to the other lanes of this wave (the next direction reads mean / cov / eigvecs / rank through the vector L1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
template <int MODEL, bool OVERRELAX, int VARIANT>
__global__ __launch_bounds__(64) void klhr_free_kernel(const KParams p, const int mh_exact) {
    static_assert(VARIANT != VAR_SLICE, "the slice sampler runs on klhr_iter_kernel");
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 3, l8 = lane & 7;
    const long long c0 = (long long)blockIdx.x * 8;
    const long long c = c0 + grp;
    const int D = p.D;
    constexpr int NFIT = VARIANT == VAR_GAUSS ? 2 : (VARIANT == VAR_SUB ? 3 : 4);
    // model data and the ziggurat's WI / KI tables staged in LDS behind the chain rows (as klhr_iter_kernel does)
    ModelData md = p.md;
    const double *zwi = nullptr;
    const uint64_t *zki = nullptr;
    {
        const int naux = MODEL == MODEL_GLMM_POISSON ? 4 * p.md.n : (MODEL == MODEL_ILL_NORMAL ? D : 0);
        if constexpr (MODEL == MODEL_GLMM_POISSON || MODEL == MODEL_ILL_NORMAL) {
            double *aux_l = lds + (size_t)8 * p.S;
            for (int i = lane; i < naux; i += 64) aux_l[i] = p.md.aux[i];
            md.aux = aux_l;
        }
        if (((size_t)8 * p.S + naux + KLHR_ZTAB_DOUBLES) * sizeof(double) <= 64 * 1024) {      // (same test as the launcher, klhr_free.hip)
            double *w_l = lds + (size_t)8 * p.S + naux;
            uint64_t *k_l = reinterpret_cast<uint64_t *>(w_l + 256);
            for (int i = lane; i < 256; i += 64) { w_l[i] = KLHR_ZIG_WI[i]; k_l[i] = KLHR_ZIG_KI[i]; }
            zwi = w_l; zki = k_l;
        }
        KLHR_WAVE_SYNC();
    }
    const bool active = c < p.C;
    double2 *TR = reinterpret_cast<double2 *>(lds + (size_t)grp * p.S);          // this octet's (theta_i, rho_i) row

    // ---- load chain state: theta rows cooperatively (a chain's row is contiguous), scalars replicated over the octet
    for (int g = 0; g < 8; g++) {
        if (c0 + g >= p.C) break;
        const double *src = p.theta + (c0 + g) * (long long)D;
        double2 *row = reinterpret_cast<double2 *>(lds + (size_t)g * p.S);
        for (int i = lane; i < D; i += 64) { double2 v; v.x = src[i]; v.y = 0.0; row[i] = v; }
    }
    Pcg64 rng = {0, 0, 0, 0};
    long long draw = 0, omN = 0, pcan = 0, gevals = 0;
    int cidx = 0;
    double acc = 0, sm_x = 0, sm_cnt = 0, moved_f = 0;
    int Kc = p.K;
    if (active) {
        rng.sh = p.rng[c * 4 + 0]; rng.sl = p.rng[c * 4 + 1]; rng.ih = p.rng[c * 4 + 2]; rng.il = p.rng[c * 4 + 3];
        draw = p.counters[c * 4 + 0]; omN = p.counters[c * 4 + 1]; pcan = p.counters[c * 4 + 2]; cidx = (int)p.counters[c * 4 + 3];
        acc = p.acc[c];
        gevals = p.grad_evals[c];
        if (VARIANT == VAR_GAUSS) { sm_x = p.kstate[c * 4 + 0]; sm_cnt = p.kstate[c * 4 + 1]; Kc = (int)p.kstate[c * 4 + 2]; moved_f = p.kstate[c * 4 + 3]; }
    }
    const double asx_l = p.asx[l8], w_l = p.wq[l8];
    KLHR_WAVE_SYNC();

    const bool diag = (p.flags & FLAG_DIAG) != 0;
    double *const dd = p.d_diag + (active ? c : 0) * KLHR_NDIAG;
    Opt o;
    {
        const double z4[4] = {0, 0, 0, 0};
        opt_begin(o, 1, z4, 1e-5);
    }
    double eta[4] = {0, 0, 0, 0};
    double zp = 0, lp_p = 0, lp_c = 0;
    int ph = active && p.n_iters > 0 ? FP_HEAD0 : FP_DONE;
    int iter = 0, npass = 0, fb = 0;
    bool fit_capped = false;
#ifdef KLHR_STAMPS
    unsigned long long stamp_t = __builtin_readcyclecounter(), st_l = 0, st_k = 0, st_adv = 0, st_srv = 0, st_trips = 0, st_ktrips = 0, st_ltrips = 0, st_mine = 0;
#endif

    for (;;) {
        // =============================================================== service: one chain at a time, the whole wavefront
        for (;;) {
            const unsigned long long want = __ballot(ph == FP_TAIL || ph == FP_HEAD0);
            if (!want) break;
            const int ol = (int)(__ffsll((long long)want) - 1);         // first lane of the owner's octet (wave-uniform)
            const int g = ol >> 3;
            const long long cg = c0 + g;
            double2 *TRg = reinterpret_cast<double2 *>(lds + (size_t)g * p.S);
            double *Zg = lds + (size_t)g * p.S + 2 * (size_t)D;
            // the owner's scalars, wave-uniform
            Pcg64 r;
            r.sh = rl_u64(rng.sh, ol); r.sl = rl_u64(rng.sl, ol); r.ih = rl_u64(rng.ih, ol); r.il = rl_u64(rng.il, ol);
            const int ph_o = rl_i32(ph, ol);
            int it_o = rl_i32(iter, ol);
            long long draw_o = rl_i64(draw, ol), omN_o = rl_i64(omN, ol), pcan_o = rl_i64(pcan, ol);
            int cidx_o = rl_i32(cidx, ol), Kc_o = rl_i32(Kc, ol);
            double acc_o = rl_f64(acc, ol), smx_o = rl_f64(sm_x, ol), smc_o = rl_f64(sm_cnt, ol), mov_o = rl_f64(moved_f, ol);
            if (ph_o == FP_TAIL) {
                // ------------------------------------------------------- MH step (klhr_sinh.py:242-260)
                double e4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) e4[k] = rl_f64(eta[k], ol);
                const double zp_o = rl_f64(zp, ol), lpp = rl_f64(lp_p, ol), lpc = rl_f64(lp_c, ol);
                double rr = lpp;
                rr -= lpc;
                if constexpr (VARIANT == VAR_GAUSS) {
                    const double gm = e4[0], gs = exp(clipc(e4[1], p.scale_clip));
                    rr += gauss_logq(0.0, gm, gs);
                    rr -= gauss_logq(zp_o, gm, gs);
                } else {
                    rr += sa_log_q(0.0, e4, p.tol, p.scale_clip, VARIANT == VAR_SUB);
                    rr -= sa_log_q(zp_o, e4, p.tol, p.scale_clip, VARIANT == VAR_SUB);
                }
                const double logu = log(rng_uniform(r));
                const double mn = (rr != rr) ? rr : (rr < 0 ? rr : 0.0);      // np.minimum(0, r)
                const int a = logu < mn ? 1 : 0;
                const double ad = (double)a, na = (double)(1 - a);
                bool moved_any = false, nan_any = false;
                for (int i = lane; i < D; i += 64) {
                    double2 v = TRg[i];
                    const double thp = add_rn(mul_rn(zp_o, v.y), v.x);
                    const double nx = add_rn(mul_rn(ad, thp), mul_rn(na, v.x));     // a*thetap + (1-a)*theta
                    const double dq = (nx - v.x) * (nx - v.x);
                    moved_any = moved_any || dq > 0.0;
                    nan_any = nan_any || dq != dq;
                    v.x = nx;
                    TRg[i] = v;
                }
                // ||theta - prev_theta|| > _msjd = 0 (klhr.py:219-221): a sum of squares is positive iff a term is (NaN: not > 0)
                if (VARIANT == VAR_GAUSS) mov_o = (__any(moved_any) && !__any(nan_any)) ? 1.0 : 0.0;
                acc_o += (ad - acc_o) / (double)draw_o;
                if (diag && lane == 0) {
                    double *dg = p.d_diag + cg * KLHR_NDIAG;
                    dg[KLHR_D_ZP] = zp_o; dg[KLHR_D_R] = rr; dg[KLHR_D_LOGU] = logu; dg[KLHR_D_ACCEPT] = a;
                }
                KLHR_WAVE_SYNC();
                // ------------------------------------------------------- closure-or-update (klhr_sinh.py:268-287)
                bool closed = false;
                if (!(p.warmup < p.windowsize) && p.nclos > 0) {
                    closed = draw_o == p.closures[cidx_o];
                    if (closed && cidx_o < p.nclos - 1) cidx_o++;
                }
                if (closed) {
                    wave_close_window(p, cg, omN_o, TRg, Zg, lane);
                    omN_o = 0;
                    pcan_o = 0;
                    if (VARIANT == VAR_GAUSS) {      // klhr.py:212-214
                        Kc_o = (int)(smx_o < 1.0 ? 1.0 : (smx_o > 50.0 ? 50.0 : smx_o));
                        smx_o = (double)p.K; smc_o = 0.0;
                    }
                } else {
                    omN_o += 1;
                    if (VARIANT == VAR_GAUSS) smoother_update(smx_o, smc_o, mov_o != 0.0);
                    wave_om_update(p.om_m + cg * (long long)D, p.om_v + cg * (long long)D, omN_o, D, lane, TRg);
                    {
                        const double *mean = p.mean + cg * (long long)D;
                        for (int i = lane; i < D; i += 64) Zg[i] = sub_rn(TRg[i].x, mean[i]);
                    }
                    KLHR_WAVE_SYNC();
                    pcan_o += 1;
                    wave_pca_update(p, cg, pcan_o, TRg, Zg, lane);
                }
                // ------------------------------------------------------- the draw
                if (p.draws) {
                    double *out = p.draws + ((long long)it_o * p.C + cg) * D;
                    for (int i = lane; i < D; i += 64) out[i] = TRg[i].x;
                }
                it_o += 1;
            }
            // ----------------------------------------------------------- head of the next iteration (klhr_sinh.py:264, 183-184)
            int ph_n = FP_DONE;
            double x0 = 0.0;
            if (it_o < p.n_iters) {
                const double u_choice = (p.flags & FLAG_EIGEN_ONE) ? pcg_next_double(r) : 0.0;      // rng.choice() consumes one uniform BEFORE the D normals
                if (zwi) {
                    for (int k = 0; k < D; k++) { const double z = rng_std_normal_tab(r, zwi, zki); if ((k & 63) == lane) Zg[k] = z; }
                } else {
                    for (int k = 0; k < D; k++) { const double z = rng_std_normal(r); if ((k & 63) == lane) Zg[k] = z; }
                }
                KLHR_WAVE_SYNC();
                wave_direction(p, cg, TRg, Zg, lane, u_choice);
                if (diag) {
                    double *rg = p.d_rho + cg * (long long)D;
                    for (int i = lane; i < D; i += 64) rg[i] = TRg[i].y;
                }
                x0 = rng_std_normal_call(r) * p.initscale;
                if (diag && lane == 0) p.d_diag[cg * KLHR_NDIAG + KLHR_D_X0] = x0;
                ph_n = FP_FIT1;
            }
            // ----------------------------------------------------------- hand the chain back to its octet
            if (grp == g) {
                rng = r;
                draw = draw_o; omN = omN_o; pcan = pcan_o; cidx = cidx_o; Kc = Kc_o;
                acc = acc_o; sm_x = smx_o; sm_cnt = smc_o; moved_f = mov_o;
                iter = it_o;
                ph = ph_n;
                npass = 0;
                if (ph_n == FP_FIT1) {
                    const double x0v[4] = {x0, 0, 0, 0};
                    opt_begin(o, 1, x0v, 1e-5);
                }
            }
        }
        KLHR_STAMP(st_srv);
        if (__all(ph == FP_DONE)) break;
#ifdef KLHR_STAMPS
        st_trips++;
        if (lane == 0 && ph != FP_DONE) st_mine++;
#endif

        // =============================================================== site L: the line objective / the MH step's log densities
        double f = 0, gr[4] = {0, 0, 0, 0};
        const bool hit = (ph == FP_FIT1 || ph == FP_FIT2) && opt_cache_hit(o);
        const bool needL = (ph == FP_FIT1 && !hit) || ph == FP_MHP || ph == FP_MHC;
        if (__any(needL)) {
            if (needL) {
                if (mh_exact && ph == FP_MHP) {
                    // the index-order loop of klhr_iter_kernel's MH step: lane 0 of the octet at zp, the others at 0
                    double logp, gdot;
                    const double t = (l8 == 0) ? zp : 0.0;
                    model_eval<MODEL>(md, [&](int i, double &xi, double &ri) { double2 v = TR[i]; ri = v.y; xi = fma(t, v.y, v.x); },
                                      INFINITY, NoSink(), logp, gdot);
                    lp_p = __shfl(logp, (grp << 3) + 0);
                    lp_c = __shfl(logp, (grp << 3) + 1);
                } else {
                    const double tt = ph == FP_FIT1 ? o.xt[0] : (ph == FP_MHP ? zp : 0.0);
                    const double xt1[4] = {tt, 0, 0, 0};
                    eval_objective<MODEL, VARIANT>(p, md, TR, 1, xt1, asx_l, w_l, l8, f, gr);
                }
            }
#ifdef KLHR_STAMPS
            st_ltrips++;
#endif
        }
        KLHR_STAMP(st_l);
        // =============================================================== site K: the KL objective (lane l8 = node l8)
        const bool needK = ph == FP_FIT2 && !hit;
        if (__any(needK)) {
            if (needK) eval_objective<MODEL, VARIANT>(p, md, TR, NFIT, o.xt, asx_l, w_l, l8, f, gr);
#ifdef KLHR_STAMPS
            st_ktrips++;
#endif
        }
        KLHR_STAMP(st_k);
        // =============================================================== advance
        if (ph == FP_FIT1) {
            if (hit) { f = o.lf; gr[0] = o.lg[0]; }
            else { opt_cache_store(o, f, gr); npass++; }
            opt_advance<false, 1>(o, f, gr);
            const bool capped = npass > KLHR_MAX_PASSES;
            if (o.pc == PC_DONE || capped) {
                if (capped && o.pc != PC_DONE) fit_capped = true;
                gevals += o.nfev;
                fb = o.fallbacks;
                double s1 = o.H[0];
                if (diag && l8 == 0) {
                    dd[KLHR_D_MODE] = o.xk[0]; dd[KLHR_D_HINV1] = s1; dd[KLHR_D_NFEV1] = o.nfev; dd[KLHR_D_NIT1] = o.k; dd[KLHR_D_STATUS1] = o.status;
                }
                s1 = (s1 > 0 ? 1.0 : 0.0) * 0.5 * log(s1);
                if (VARIANT == VAR_GAUSS) {          // klhr.py:134-139: no extra normals, default gtol
                    const double init[4] = {o.xk[0], s1, 0.0, 0.0};
                    if (diag && l8 == 0) { dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = 0.0; dd[KLHR_D_INIT3] = 0.0; }
                    opt_begin(o, 2, init, 1e-5);
                } else {
                    // init = rng.normal(size=4) * initscale, then [0], [1] overwritten (klhr_sinh.py:191-193); SUBKLHRSINH: size=3
                    (void)rng_std_normal(rng);
                    (void)rng_std_normal(rng);
                    const double n2 = rng_std_normal(rng) * p.initscale;
                    const double n3 = VARIANT == VAR_SUB ? 0.0 : rng_std_normal(rng) * p.initscale;
                    const double init[4] = {o.xk[0], s1, n2, n3};
                    if (diag && l8 == 0) { dd[KLHR_D_INIT0] = init[0]; dd[KLHR_D_INIT1] = init[1]; dd[KLHR_D_INIT2] = init[2]; dd[KLHR_D_INIT3] = init[3]; }
                    opt_begin(o, VARIANT == VAR_SUB ? 3 : 4, init, 1e-3);
                }
                ph = capped ? FP_MHP : FP_FIT2;
                if (capped) {           // (safety valve: no KL fit; the proposal below uses the mode-search state as eta)
#pragma unroll
                    for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                }
            }
        } else if (ph == FP_FIT2) {
            if (hit) { f = o.lf; gr[0] = o.lg[0]; gr[1] = o.lg[1]; gr[2] = o.lg[2]; gr[3] = o.lg[3]; }
            else { opt_cache_store(o, f, gr); npass++; }
            opt_advance<false, NFIT>(o, f, gr);
            const bool capped = npass > KLHR_MAX_PASSES;
            if (o.pc == PC_DONE || capped) {
                if (capped && o.pc != PC_DONE) fit_capped = true;
                else gevals += (long long)o.nfev * 8;
#pragma unroll
                for (int k = 0; k < 4; k++) eta[k] = o.xk[k];
                if (diag) {
                    if (l8 == 0) {
                        dd[KLHR_D_NFEV2] = o.nfev; dd[KLHR_D_NIT2] = o.k; dd[KLHR_D_STATUS2] = o.status; dd[KLHR_D_FUN2] = o.old_fval;
                        dd[KLHR_D_FALLBACKS] = fb + o.fallbacks; dd[KLHR_D_PASSES] = npass;
                    }
                    const double e_sel = l8 == 0 ? eta[0] : (l8 == 1 ? eta[1] : (l8 == 2 ? eta[2] : eta[3]));
                    if (l8 < 4) p.d_eta[c * 4 + l8] = e_sel;
                }
                // the MH proposal (klhr_sinh.py:244-247): the chain's own generator, replicated over the octet
                draw += 1;
                if constexpr (VARIANT == VAR_GAUSS && OVERRELAX) { const ZpOut zo = gauss_overrelaxed_zp_call(rng, eta[0], eta[1], Kc, p.scale_clip); zp = zo.zp; rng = zo.rng; }
                else if constexpr (VARIANT == VAR_GAUSS) zp = gauss_zp<false>(rng, eta, Kc, p.scale_clip);
                else if constexpr (OVERRELAX) { const ZpOut zo = overrelaxed_zp_call(rng, eta[0], eta[1], eta[2], eta[3], p.K, p.tol, p.scale_clip, VARIANT == VAR_SUB); zp = zo.zp; rng = zo.rng; }
                else zp = sa_T(rng_std_normal_call(rng), eta, p.tol, p.scale_clip, VARIANT == VAR_SUB);
                ph = FP_MHP;
            }
        } else if (ph == FP_MHP) {
            if (mh_exact) ph = FP_TAIL;
            else { lp_p = -f; ph = FP_MHC; }
        } else if (ph == FP_MHC) {
            lp_c = -f;
            ph = FP_TAIL;
        }
        KLHR_STAMP(st_adv);
    }

    // ---- store chain state
    KLHR_WAVE_SYNC();
    bool finite_theta = true;
    for (int g = 0; g < 8; g++) {
        if (c0 + g >= p.C) break;
        double *dst = p.theta + (c0 + g) * (long long)D;
        const double2 *row = reinterpret_cast<const double2 *>(lds + (size_t)g * p.S);
        bool fin = true;
        for (int i = lane; i < D; i += 64) { const double v = row[i].x; dst[i] = v; if (!isfinite(v)) fin = false; }
        if (__any(!fin) && grp == g) finite_theta = false;
    }
    if (active && l8 == 0) {
        if (!finite_theta) atomicOr(&p.status[c], 2);
        if (fit_capped) atomicOr(&p.status[c], 8);
        p.rng[c * 4 + 0] = rng.sh; p.rng[c * 4 + 1] = rng.sl; p.rng[c * 4 + 2] = rng.ih; p.rng[c * 4 + 3] = rng.il;
        p.counters[c * 4 + 0] = draw; p.counters[c * 4 + 1] = omN; p.counters[c * 4 + 2] = pcan; p.counters[c * 4 + 3] = cidx;
        p.acc[c] = acc;
        p.grad_evals[c] = gevals;
        if (VARIANT == VAR_GAUSS) { p.kstate[c * 4 + 0] = sm_x; p.kstate[c * 4 + 1] = sm_cnt; p.kstate[c * 4 + 2] = (double)Kc; p.kstate[c * 4 + 3] = moved_f; }
    }
#ifdef KLHR_STAMPS
    if (lane == 0 && diag) {       // diagnostic build: cycle totals of the wavefront in its first chain's diag row
        double *ds = p.d_diag + c0 * KLHR_NDIAG;
        ds[0] = (double)st_srv; ds[1] = (double)st_l; ds[2] = (double)st_k; ds[3] = (double)st_adv; ds[4] = (double)st_trips;
        ds[5] = (double)st_ltrips; ds[6] = (double)st_ktrips; ds[7] = (double)st_mine;
    }
#endif
}

}  // namespace klhr
