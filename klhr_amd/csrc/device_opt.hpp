// device_opt.hpp — SciPy-faithful BFGS as a resumable per-chain state machine.
//
// The reference fits the proposal with scipy.optimize.minimize(method="BFGS") twice per draw
// (klhr_sinh.py:183-199).  The number of objective evaluations is data dependent (4-90 per fit), so on a
// 64-wide wavefront the optimiser is written "inside out": the ONLY place an objective is evaluated is
// the kernel's single convergence point, and everything SciPy does between two evaluations
// (DCSRCH step, fallback line search, zoom, inverse-Hessian update, termination tests) is the cheap
// scalar function opt_advance().  Chains of one wavefront that are in different places of SciPy's
// control flow therefore still share every expensive evaluation pass.
//
// Restated algorithms (SciPy 1.15.3, third-party to the reference; cited for parity checking):
//   _minimize_bfgs            optimize/_optimize.py:1391-1502   (norm=inf, xrtol=0, c1=1e-4, c2=0.9, H0=I)
//   line_search_wolfe1/DCSRCH optimize/_linesearch.py:37-180, optimize/_dcsrch.py:201-500 (amin=1e-100,
//                             amax=1e100, xtol=1e-14, <=100 iterations), dcstep _dcsrch.py:502-728
//   line_search_wolfe2/_zoom  optimize/_linesearch.py:186-640 (fallback when DCSRCH fails)
//   ScalarFunction/MemoizeJac optimize/_differentiable_functions.py:323-334, _optimize.py:60-85
//                             (a repeat evaluation at the most recent x is served from a 1-entry cache)
//
// Problems of dimension n=1 (mode search) run through the same 4-wide code with zero padding: padded
// coordinates have zero gradient, so H stays block-diagonal and every padded term adds an exact 0.
#pragma once
#ifndef KLHR_HOST_HARNESS
#include <hip/hip_runtime.h>
#ifndef KLHR_BFGS_LITERAL
#define KLHR_BFGS_RANK2 1
#endif
#endif
#include <math.h>

namespace klhr {

// Separately rounded a+b, a*b, a-b, as NumPy's element-wise operations and SciPy's `xk + alpha * pk` produce them.
// (HIP's __dadd_rn / __dmul_rn / __dsub_rn are plain operators that the compiler contracts into FMAs with their
// neighbours; these carry no contract flag.  The host-compiled parity harness is built with -ffp-contract=off.)
#ifndef KLHR_HOST_HARNESS
__device__ __forceinline__ double add_rn(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double sub_rn(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
#else
static inline double add_rn(double a, double b) { return a + b; }
static inline double mul_rn(double a, double b) { return a * b; }
static inline double sub_rn(double a, double b) { return a - b; }
#endif

// The optimiser is inlined into several kernels that must agree bit for bit (the split pipeline's lane = chain and node-parallel
// fit kernels: a chain's draws may not depend on the launch size).  Under the default -ffp-contract=fast the backend fuses a
// multiplication with an addition wherever it finds the pair after inlining — a decision that depends on the surrounding code
// (seen: the same inputs gave a request point one ulp apart in klhr_fit2_kernel and klhr_fitn_kernel, 1 chain-draw in ~2 000).
// contract(on) forms the FMAs in the front end, within an expression, the same way wherever the function is inlined.
// (The host-compiled parity harness keeps its -ffp-contract=off.)
#ifndef KLHR_HOST_HARNESS
#pragma clang fp contract(on)
#endif

enum { PC_INIT = 0, PC_W1 = 1, PC_W2 = 2, PC_ZOOM = 3, PC_DONE = 4 };

// line-search scratch, overlaid: DCSRCH state and wolfe2/zoom state are never live together
enum { L_FINIT = 0, L_GINIT, L_GTEST, L_WIDTH, L_WIDTH1, L_STX, L_FX, L_GX, L_STY, L_FY, L_GY, L_STMIN, L_STMAX, L_N };
enum { Z_ALPHA0 = 0, Z_ALPHA1, Z_PHI_A0, Z_DERPHI_A0, Z_ALO, Z_AHI, Z_PHILO, Z_PHIHI, Z_DERPHILO, Z_PHIREC, Z_AREC };

struct Opt {
    double xk[4], gfk[4], H[16], pk[4];
    double old_fval, old_old_fval, derphi0, gtol;
    double xt[4];      // point whose evaluation is requested next
    double areq;       // its step length along pk
    double lx[4], lf, lg[4];   // most recent evaluation (MemoizeJac cache)
    double ls[L_N];
    int n, k, nfev, status, fallbacks, pc, it, i2;
    bool have, brackt, stage2;
};

__device__ __forceinline__ double pymin2(double a, double b) { return b < a ? b : a; }   // Python min(a, b)
__device__ __forceinline__ double pymax2(double a, double b) { return b > a ? b : a; }   // Python max(a, b)
__device__ __forceinline__ double pymax3(double a, double b, double c) { double m = a; if (b > m) m = b; if (c > m) m = c; return m; }
__device__ __forceinline__ double npsign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : (x == 0 ? 0.0 : NAN)); }
__device__ __forceinline__ double npclip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ bool finite_(double x) { return isfinite(x); }

// np.amax(np.abs(g)) over N entries: NaN propagates (N = 4 covers the zero-padded entries: |0| never raises the maximum)
template <int N = 4>
__device__ __forceinline__ double amax_abs4(const double *g) {
    double m = fabs(g[0]);
#pragma unroll
    for (int i = 1; i < N; i++) {
        double a = fabs(g[i]);
        if (!(m != m) && (a != a || a > m)) m = a;
    }
    return m;
}

// xk + a*pk, unfused like NumPy's element-wise loops
template <int N = 4>
__device__ __forceinline__ void opt_request(Opt &o, double a) {
    o.areq = a;
#pragma unroll
    for (int i = 0; i < N; i++) o.xt[i] = add_rn(o.xk[i], mul_rn(a, o.pk[i]));
}

// _dcsrch.py:502-728
// The body is force-inlined into two wrappers (below): dcstep_inl for the lane-per-chain kernel, which runs alone on
// its SIMD and can afford the registers, and the out-of-line dcstep_call for the octet kernel, where an inlined copy
// costs a wavefront of occupancy (its by-reference state then round-trips through scratch memory, 8 doubles per call).
__device__ __forceinline__ void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy, double &stp,
                                       double fp, double dp, bool &brackt, double stpmin, double stpmax) {
    // The four cases of MINPACK's dcstep evaluate the same cubic-step expressions on different operands.  Lanes of
    // one wavefront sit in different cases, so the expressions are written ONCE on case-selected operands: every lane
    // performs exactly the operations (and roundings) of its own case, but the wavefront executes one copy of the
    // divisions and the square root instead of up to four.
    const double sgnd = npsign(dp) * npsign(dx);
    const int cs = fp > fx ? 1 : (sgnd < 0.0 ? 2 : (fabs(dp) < fabs(dx) ? 3 : 4));
    const bool c4 = cs == 4;
    // theta = 3 (A - B) / (C - D) + E + dp ; s = max(|theta|, |E|, |dp|) ; gamma = s sqrt((theta/s)^2 - (E/s)(dp/s))
    const double A = c4 ? fp : fx, B = c4 ? fy : fp, C = c4 ? sty : stp, Dd = c4 ? stp : stx, E = c4 ? dy : dx;
    const double theta = 3.0 * (A - B) / (C - Dd) + E + dp;
    const double s = pymax3(fabs(theta), fabs(E), fabs(dp));
    const double rad = (theta / s) * (theta / s) - (E / s) * (dp / s);
    double gamma = s * sqrt(cs == 3 ? pymax2(0, rad) : rad);
    const bool flip = cs == 1 ? stp < stx : (c4 ? stp > sty : stp > stx);
    if (flip) gamma = -gamma;
    const double P1 = cs == 1 ? dx : dp;
    const double p = (gamma - P1) + theta;
    const double q = cs == 3 ? (gamma + (dx - dp)) + gamma : ((gamma - P1) + gamma) + (cs == 1 ? dp : (cs == 2 ? dx : dy));
    const double r = p / q;
    const double base = cs == 1 ? stx : stp, oth = cs == 1 ? stp : (c4 ? sty : stx);
    const double cub = base + r * (oth - base);
    double stpc, stpq, stpf;
    if (cs == 1) {
        stpc = cub;
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        if (fabs(stpc - stx) <= fabs(stpq - stx)) stpf = stpc; else stpf = stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (cs == 2 || cs == 3) {
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (cs == 2) {
            stpc = cub;
            if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc; else stpf = stpq;
            brackt = true;
        } else {
            if (r < 0 && gamma != 0) stpc = cub;
            else if (stp > stx) stpc = stpmax;
            else stpc = stpmin;
            if (brackt) {
                if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc; else stpf = stpq;
                if (stp > stx) stpf = pymin2(stp + 0.66 * (sty - stp), stpf);
                else stpf = pymax2(stp + 0.66 * (sty - stp), stpf);
            } else {
                if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc; else stpf = stpq;
                stpf = npclip(stpf, stpmin, stpmax);
            }
        }
    } else {
        if (brackt) stpf = cub;
        else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    // interval update (_dcsrch.py:713-726), written as value selects so that the by-reference state stays in registers
    const bool up = fp > fx, sw = sgnd < 0;
    const double n_sty = up ? stp : (sw ? stx : sty), n_fy = up ? fp : (sw ? fx : fy), n_dy = up ? dp : (sw ? dx : dy);
    const double n_stx = up ? stx : stp, n_fx = up ? fx : fp, n_dx = up ? dx : dp;
    stx = n_stx; fx = n_fx; dx = n_dx; sty = n_sty; fy = n_fy; dy = n_dy;
    stp = stpf;
}

#ifndef KLHR_HOST_HARNESS
__device__ __attribute__((noinline)) void dcstep_call(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
                                                      double &stp, double fp, double dp, bool &brackt, double stpmin, double stpmax) {
    dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax);
}
#endif

// _linesearch.py:503-549 — "None" (errstate raise / non-finite) is returned as false
__device__ inline bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double &xmin) {
    double C = fpa, db = b - a, dc = c - a;
    double denom = (db * dc) * (db * dc) * (db - dc);
    double d00 = dc * dc, d01 = -(db * db), d10 = -(dc * dc * dc), d11 = db * db * db;
    double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
    double A = d00 * v0 + d01 * v1, B = d10 * v0 + d11 * v1;
    if (!finite_(denom) || !finite_(A) || !finite_(B) || denom == 0.0) return false;
    A /= denom; B /= denom;
    double radical = B * B - 3 * A * C;
    if (!finite_(A) || !finite_(B) || !finite_(radical) || radical < 0 || 3 * A == 0.0) return false;
    double x = a + (-B + sqrt(radical)) / (3 * A);
    if (!finite_(x)) return false;
    xmin = x;
    return true;
}
__device__ inline bool quadmin(double a, double fa, double fpa, double b, double fb, double &xmin) {
    double D = fa, C = fpa, db = b - a * 1.0;
    if (db * db == 0.0) return false;
    double B = (fb - D - C * db) / (db * db);
    if (!finite_(B) || 2.0 * B == 0.0) return false;
    double x = a - C / (2.0 * B);
    if (!finite_(x)) return false;
    xmin = x;
    return true;
}

// choose the next trial point of _zoom (_linesearch.py:575-595); iz = zoom iteration counter (o.it)
__device__ inline double zoom_pick(const Opt &o) {
    const double a_lo = o.ls[Z_ALO], a_hi = o.ls[Z_AHI];
    const double dalpha = a_hi - a_lo;
    double a, b;
    if (dalpha < 0) { a = a_hi; b = a_lo; } else { a = a_lo; b = a_hi; }
    bool have = false;
    double cchk = 0, a_j = 0;
    if (o.it > 0) {
        cchk = 0.2 * dalpha;
        have = cubicmin(a_lo, o.ls[Z_PHILO], o.ls[Z_DERPHILO], a_hi, o.ls[Z_PHIHI], o.ls[Z_AREC], o.ls[Z_PHIREC], a_j);
    }
    if (o.it == 0 || !have || a_j > b - cchk || a_j < a + cchk) {
        double qchk = 0.1 * dalpha;
        have = quadmin(a_lo, o.ls[Z_PHILO], o.ls[Z_DERPHILO], a_hi, o.ls[Z_PHIHI], a_j);
        if (!have || a_j > b - qchk || a_j < a + qchk) a_j = a_lo + 0.5 * dalpha;
    }
    return a_j;
}

__device__ __forceinline__ void opt_begin(Opt &o, int n, const double *x0, double gtol) {
    o.n = n; o.gtol = gtol; o.k = 0; o.nfev = 0; o.status = 0; o.fallbacks = 0; o.pc = PC_INIT; o.have = false;
    o.it = 0; o.i2 = 0; o.brackt = false; o.stage2 = false;
    o.old_fval = 0; o.old_old_fval = 0; o.derphi0 = 0; o.areq = 0; o.lf = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { o.xk[i] = x0[i]; o.xt[i] = x0[i]; o.pk[i] = 0; o.gfk[i] = 0; o.lx[i] = 0; o.lg[i] = 0; }
#pragma unroll
    for (int i = 0; i < 16; i++) o.H[i] = (i % 5 == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < L_N; i++) o.ls[i] = 0;
}

// 1-entry evaluation cache.  Returns true (and f,g) when xt equals the most recently evaluated point.
__device__ __forceinline__ bool opt_cache_hit(const Opt &o) {
    return o.have && o.xt[0] == o.lx[0] && o.xt[1] == o.lx[1] && o.xt[2] == o.lx[2] && o.xt[3] == o.lx[3];
}
__device__ __forceinline__ void opt_cache_store(Opt &o, double f, const double *g) {
    o.have = true; o.lf = f; o.nfev += 1;
#pragma unroll
    for (int i = 0; i < 4; i++) { o.lx[i] = o.xt[i]; o.lg[i] = g[i]; }
}

// Consume the objective value f and gradient g at o.xt; leave either the next request in o.xt or pc == PC_DONE.
// OUTLINE selects the out-of-line dcstep (octet kernel); the arithmetic is identical.
// N = number of live coordinates (the problem's dimension, as SciPy sees it).  The monolithic kernels run every problem
// through N = 4 with zero padding (padded terms add exact zeros); the split pipeline's fit kernels (klhr_split.hpp)
// instantiate N = 1 for the mode search and N = 2 / 3 / 4 for the KL fits: same values, a fraction of the instructions.
template <bool OUTLINE = false, int N = 4>
__device__ inline void opt_advance(Opt &o, double f, const double *g) {
    const double c1 = 1e-4, c2 = 0.9, stpmin = 1e-100, stpmax = 1e100, xtol = 1e-14;
    const double phi0 = o.old_fval;
    enum { A_NONE, A_SUCCESS, A_FAIL, A_W2START, A_LOOPHEAD };
    int act = A_NONE;
    double alpha_k = 0.0;
    double dphi = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) dphi += g[i] * o.pk[i];

    if (o.pc == PC_INIT) {
        o.old_fval = f;
        double n2 = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { o.gfk[i] = g[i]; n2 += g[i] * g[i]; }
        o.old_old_fval = f + sqrt(n2) / 2;
        act = A_LOOPHEAD;
    } else if (o.pc == PC_W1) {
        // DCSRCH.__call__ loop body + _iterate (non-START)
        o.it += 1;
        if (o.it >= 100) act = A_W2START;        // "dcsrch did not converge within max iterations"
        else {
            double stp = o.areq;
            const double finit = o.ls[L_FINIT], ginit = o.ls[L_GINIT], gtest = o.ls[L_GTEST];
            int task = 0;   // 0 FG, 1 CONV, 2 WARN
            double ftest = finit + stp * gtest;
            if (!o.stage2 && f <= ftest && dphi >= 0) o.stage2 = true;
            if (o.brackt && (stp <= o.ls[L_STMIN] || stp >= o.ls[L_STMAX])) task = 2;
            if (o.brackt && o.ls[L_STMAX] - o.ls[L_STMIN] <= xtol * o.ls[L_STMAX]) task = 2;
            if (stp == stpmax && f <= ftest && dphi <= gtest) task = 2;
            if (stp == stpmin && (f > ftest || dphi >= gtest)) task = 2;
            if (f <= ftest && fabs(dphi) <= c2 * -ginit) task = 1;
            if (task == 1) { act = A_SUCCESS; alpha_k = stp; }
            else if (task == 2) act = A_W2START;
            else {
                double stx = o.ls[L_STX], fx = o.ls[L_FX], gx = o.ls[L_GX], sty = o.ls[L_STY], fy = o.ls[L_FY], gy = o.ls[L_GY];
                // stage 1 works on the modified function psi(stp) = f(stp) - f(0) - ftol*stp*f'(0) (_dcsrch.py:401-441);
                // one dcstep call serves both branches, with the modification applied / removed around it
                const bool mod = !o.stage2 && f <= fx && f > ftest;
                double fpa = f, dpa = dphi;
                if (mod) {
                    fpa = f - stp * gtest; fx = fx - stx * gtest; fy = fy - sty * gtest;
                    dpa = dphi - gtest; gx = gx - gtest; gy = gy - gtest;
                }
#ifndef KLHR_HOST_HARNESS
                if constexpr (OUTLINE) dcstep_call(stx, fx, gx, sty, fy, gy, stp, fpa, dpa, o.brackt, o.ls[L_STMIN], o.ls[L_STMAX]);
                else
#endif
                    dcstep(stx, fx, gx, sty, fy, gy, stp, fpa, dpa, o.brackt, o.ls[L_STMIN], o.ls[L_STMAX]);
                if (mod) { fx = fx + stx * gtest; fy = fy + sty * gtest; gx = gx + gtest; gy = gy + gtest; }
                if (o.brackt) {
                    if (fabs(sty - stx) >= 0.66 * o.ls[L_WIDTH1]) stp = stx + 0.5 * (sty - stx);
                    o.ls[L_WIDTH1] = o.ls[L_WIDTH];
                    o.ls[L_WIDTH] = fabs(sty - stx);
                }
                double stmin, stmax;
                if (o.brackt) { stmin = pymin2(stx, sty); stmax = pymax2(stx, sty); }
                else { stmin = stp + 1.1 * (stp - stx); stmax = stp + 4.0 * (stp - stx); }
                stp = npclip(stp, stpmin, stpmax);
                if ((o.brackt && (stp <= stmin || stp >= stmax)) || (o.brackt && stmax - stmin <= xtol * stmax)) stp = stx;
                o.ls[L_STX] = stx; o.ls[L_FX] = fx; o.ls[L_GX] = gx; o.ls[L_STY] = sty; o.ls[L_FY] = fy; o.ls[L_GY] = gy;
                o.ls[L_STMIN] = stmin; o.ls[L_STMAX] = stmax;
                if (!finite_(stp)) act = A_W2START;
                else opt_request<N>(o, stp);
            }
        }
    } else if (o.pc == PC_W2) {
        // scalar_search_wolfe2 loop body, iteration i2 (phi_a1 = f, derphi_a1 = dphi at alpha1)
        const double alpha0 = o.ls[Z_ALPHA0], alpha1 = o.ls[Z_ALPHA1], phi_a0 = o.ls[Z_PHI_A0];
        if (o.i2 >= 10) { act = A_SUCCESS; alpha_k = alpha1; }          // for-else: not converged, alpha_star = alpha1
        else if (alpha1 == 0 || alpha0 > stpmax) act = A_FAIL;
        else if (f > phi0 + c1 * alpha1 * o.derphi0 || (f >= phi_a0 && o.i2 > 0)) {
            o.ls[Z_ALO] = alpha0; o.ls[Z_AHI] = alpha1; o.ls[Z_PHILO] = phi_a0; o.ls[Z_PHIHI] = f; o.ls[Z_DERPHILO] = o.ls[Z_DERPHI_A0];
            o.ls[Z_PHIREC] = phi0; o.ls[Z_AREC] = 0; o.it = 0; o.pc = PC_ZOOM;
            opt_request<N>(o, zoom_pick(o));
        } else if (fabs(dphi) <= -c2 * o.derphi0) { act = A_SUCCESS; alpha_k = alpha1; }
        else if (dphi >= 0) {
            o.ls[Z_ALO] = alpha1; o.ls[Z_AHI] = alpha0; o.ls[Z_PHILO] = f; o.ls[Z_PHIHI] = phi_a0; o.ls[Z_DERPHILO] = dphi;
            o.ls[Z_PHIREC] = phi0; o.ls[Z_AREC] = 0; o.it = 0; o.pc = PC_ZOOM;
            opt_request<N>(o, zoom_pick(o));
        } else {
            double alpha2 = pymin2(2 * alpha1, stpmax);
            o.ls[Z_ALPHA0] = alpha1; o.ls[Z_ALPHA1] = alpha2; o.ls[Z_PHI_A0] = f; o.ls[Z_DERPHI_A0] = dphi;
            o.i2 += 1;
            opt_request<N>(o, alpha2);
        }
    } else if (o.pc == PC_ZOOM) {
        const double a_j = o.areq;
        if (f > phi0 + c1 * a_j * o.derphi0 || f >= o.ls[Z_PHILO]) {
            o.ls[Z_PHIREC] = o.ls[Z_PHIHI]; o.ls[Z_AREC] = o.ls[Z_AHI]; o.ls[Z_AHI] = a_j; o.ls[Z_PHIHI] = f;
        } else {
            if (fabs(dphi) <= -c2 * o.derphi0) { act = A_SUCCESS; alpha_k = a_j; }
            else {
                if (dphi * (o.ls[Z_AHI] - o.ls[Z_ALO]) >= 0) {
                    o.ls[Z_PHIREC] = o.ls[Z_PHIHI]; o.ls[Z_AREC] = o.ls[Z_AHI]; o.ls[Z_AHI] = o.ls[Z_ALO]; o.ls[Z_PHIHI] = o.ls[Z_PHILO];
                } else { o.ls[Z_PHIREC] = o.ls[Z_PHILO]; o.ls[Z_AREC] = o.ls[Z_ALO]; }
                o.ls[Z_ALO] = a_j; o.ls[Z_PHILO] = f; o.ls[Z_DERPHILO] = dphi;
            }
        }
        if (act == A_NONE) {
            o.it += 1;
            if (o.it > 10) act = A_FAIL;
            else opt_request<N>(o, zoom_pick(o));
        }
    }

    double gnorm = 0.0;
    if (act == A_SUCCESS) {
        // back in _minimize_bfgs after a successful line search (_optimize.py:1431-1474)
        o.old_old_fval = o.old_fval;
        o.old_fval = f;
        double sk[4], yk[4];
#pragma unroll
        for (int i = 0; i < N; i++) {
            sk[i] = mul_rn(alpha_k, o.pk[i]);
            o.xk[i] = add_rn(o.xk[i], sk[i]);
            yk[i] = g[i] - o.gfk[i];
            o.gfk[i] = g[i];
        }
        o.k += 1;
        gnorm = amax_abs4<N>(o.gfk);
        double s2 = 0;
#pragma unroll
        for (int i = 0; i < N; i++) s2 += fabs(o.pk[i]) * fabs(o.pk[i]);
        if (gnorm <= o.gtol) o.pc = PC_DONE;
        else if (alpha_k * sqrt(s2) <= 0.0) o.pc = PC_DONE;
        else if (!finite_(o.old_fval)) { o.status = 2; o.pc = PC_DONE; }
        else {
            double rhok_inv = 0;
#pragma unroll
            for (int i = 0; i < N; i++) rhok_inv += yk[i] * sk[i];
            const double rhok = rhok_inv == 0.0 ? 1000.0 : 1.0 / rhok_inv;
#ifdef KLHR_BFGS_RANK2
            // Hk+1 = (I - rho s y^T) Hk (I - rho y s^T) + rho s s^T expanded to its rank-two form
            //   Hk - (rho s)(y^T Hk) - (Hk y)(rho s)^T + rho (1 + rho y^T Hk y) s s^T
            // (about 90 instead of 270 operations; differs from the two matrix products SciPy forms — kept below and
            // used by the host-compiled parity harness — by rounding only: tests/test_host_opt.py)
            double wl[4], wr[4], rs[4], cs_[4];
#pragma unroll
            for (int j = 0; j < N; j++) {
                double a = 0, b = 0;
#pragma unroll
                for (int m = 0; m < N; m++) { a += yk[m] * o.H[m * 4 + j]; b += o.H[j * 4 + m] * yk[m]; }
                wl[j] = a; wr[j] = b;
            }
            double yHy = 0;
#pragma unroll
            for (int m = 0; m < N; m++) yHy += yk[m] * wr[m];
            const double cc = rhok * (1.0 + rhok * yHy);
#pragma unroll
            for (int i = 0; i < N; i++) { rs[i] = rhok * sk[i]; cs_[i] = cc * sk[i]; }
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int j = 0; j < N; j++)
                    o.H[i * 4 + j] = ((o.H[i * 4 + j] - rs[i] * wl[j]) - wr[i] * rs[j]) + cs_[i] * sk[j];
#else
            double T1[16];
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int j = 0; j < N; j++) {
                    double s = 0;
#pragma unroll
                    for (int m = 0; m < N; m++) s += o.H[i * 4 + m] * ((m == j ? 1.0 : 0.0) - yk[m] * sk[j] * rhok);
                    T1[i * 4 + j] = s;
                }
#pragma unroll
            for (int i = 0; i < N; i++)
#pragma unroll
                for (int j = 0; j < N; j++) {
                    double s = 0;
#pragma unroll
                    for (int m = 0; m < N; m++) s += ((i == m ? 1.0 : 0.0) - sk[i] * yk[m] * rhok) * T1[m * 4 + j];
                    o.H[i * 4 + j] = s + rhok * sk[i] * sk[j];
                }
#endif
            act = A_LOOPHEAD;
        }
    } else if (act == A_FAIL) {
        o.status = 2;              // _LineSearchError -> warnflag 2
        o.pc = PC_DONE;
    }

    if (act == A_LOOPHEAD) {
        gnorm = amax_abs4<N>(o.gfk);
        if (gnorm > o.gtol && o.k < o.n * 200) {
            double d0 = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                double s = 0;
#pragma unroll
                for (int j = 0; j < N; j++) s += o.H[i * 4 + j] * o.gfk[j];
                o.pk[i] = -s;
            }
#pragma unroll
            for (int i = 0; i < N; i++) d0 += o.gfk[i] * o.pk[i];
            o.derphi0 = d0;
            // scalar_search_wolfe1 + DCSRCH START
            double alpha1;
            if (d0 != 0) {
                alpha1 = pymin2(1.0, 1.01 * 2 * (o.old_fval - o.old_old_fval) / d0);
                if (alpha1 < 0) alpha1 = 1.0;
            } else alpha1 = 1.0;
            bool err = (alpha1 < stpmin) || (alpha1 > stpmax) || (d0 >= 0);
            if (err) act = A_W2START;
            else {
                o.brackt = false; o.stage2 = false; o.it = 0;
                o.ls[L_FINIT] = o.old_fval; o.ls[L_GINIT] = d0; o.ls[L_GTEST] = c1 * d0;
                o.ls[L_WIDTH] = stpmax - stpmin; o.ls[L_WIDTH1] = (stpmax - stpmin) / 0.5;
                o.ls[L_STX] = 0.0; o.ls[L_FX] = o.old_fval; o.ls[L_GX] = d0;
                o.ls[L_STY] = 0.0; o.ls[L_FY] = o.old_fval; o.ls[L_GY] = d0;
                o.ls[L_STMIN] = 0; o.ls[L_STMAX] = alpha1 + 4.0 * alpha1;
                if (!finite_(alpha1)) act = A_W2START;
                else { o.pc = PC_W1; opt_request<N>(o, alpha1); }
            }
        } else {
            o.pc = PC_DONE;
        }
    }

    if (act == A_W2START) {
        // _line_search_wolfe12 falls back to line_search_wolfe2 (c1, c2, amax carried over)
        o.fallbacks += 1;
        double alpha1;
        if (o.derphi0 != 0) alpha1 = pymin2(1.0, 1.01 * 2 * (o.old_fval - o.old_old_fval) / o.derphi0);
        else alpha1 = 1.0;
        if (alpha1 < 0) alpha1 = 1.0;
        alpha1 = pymin2(alpha1, stpmax);
        o.ls[Z_ALPHA0] = 0; o.ls[Z_ALPHA1] = alpha1; o.ls[Z_PHI_A0] = o.old_fval; o.ls[Z_DERPHI_A0] = o.derphi0;
        o.i2 = 0; o.pc = PC_W2;
        opt_request<N>(o, alpha1);
    }

    if (o.pc == PC_DONE && o.status == 0) {
        // final status (_optimize.py:1478-1487)
        gnorm = amax_abs4<N>(o.gfk);
        if (o.k >= o.n * 200) o.status = 1;
        else if (gnorm != gnorm || o.old_fval != o.old_fval || o.xk[0] != o.xk[0] || o.xk[1] != o.xk[1] ||
                 o.xk[2] != o.xk[2] || o.xk[3] != o.xk[3]) o.status = 3;
    }
}

}  // namespace klhr
#ifndef KLHR_HOST_HARNESS
#pragma clang fp contract(fast)
#endif
