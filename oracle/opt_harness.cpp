// opt_harness.cpp — TEST INFRASTRUCTURE.  Compiles the PRODUCT's optimiser state machine
// (klhr_amd/csrc/device_opt.hpp) for the host and drives it with the oracle's objective functions, so that
// tests/test_host_opt.py can require exact agreement with the oracle's straight-line SciPy restatement.
#include <cmath>
#include <cstring>
#define __device__
#define __forceinline__ inline
#define KLHR_HOST_HARNESS 1
namespace hip_stub {}
#include "../klhr_amd/csrc/device_opt.hpp"

extern "C" {
double klhr_oracle_kl(void *c, const double *eta, const double *rho, double *grad);
void klhr_oracle_line(void *c, double t, const double *rho, double *out2);

__attribute__((visibility("default")))
void klhr_harness_bfgs(void *chain, const double *rho, int n, const double *x0, double gtol, double *out) {
    using namespace klhr;
    Opt o;
    double x[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) x[i] = x0[i];
    opt_begin(o, n, x, gtol);
    long passes = 0;
    while (o.pc != PC_DONE) {
        double f, g[4] = {0, 0, 0, 0};
        if (opt_cache_hit(o)) {
            f = o.lf;
            for (int i = 0; i < 4; i++) g[i] = o.lg[i];
        } else {
            if (n == 1) {
                double o2[2];
                klhr_oracle_line(chain, o.xt[0], rho, o2);
                f = o2[0]; g[0] = o2[1];
            } else {
                f = klhr_oracle_kl(chain, o.xt, rho, g);
            }
            opt_cache_store(o, f, g);
            passes++;
        }
        opt_advance(o, f, g);
        if (passes > 2000000) break;
    }
    for (int i = 0; i < 4; i++) out[i] = o.xk[i];
    out[4] = o.old_fval; out[5] = o.H[0]; out[6] = o.nfev; out[7] = o.k; out[8] = o.status; out[9] = o.fallbacks;
}

}   // extern "C"

// the same drive with opt_advance specialised on the problem's dimension (opt_advance<OUTLINE, N = n>): what the lane kernel's
// two-loop fit and the split pipeline's fit kernels instantiate.  Must equal the zero-padded N = 4 form bit for bit.
template <int N>
static void bfgs_n(void *chain, const double *rho, const double *x0, double gtol, double *out) {
    using namespace klhr;
    Opt o;
    double x[4] = {0, 0, 0, 0};
    for (int i = 0; i < N; i++) x[i] = x0[i];
    opt_begin(o, N, x, gtol);
    long passes = 0;
    while (o.pc != PC_DONE) {
        double f, g[4] = {0, 0, 0, 0};
        if (opt_cache_hit(o)) {
            f = o.lf;
            for (int i = 0; i < 4; i++) g[i] = o.lg[i];
        } else {
            if (N == 1) {
                double o2[2];
                klhr_oracle_line(chain, o.xt[0], rho, o2);
                f = o2[0]; g[0] = o2[1];
            } else {
                f = klhr_oracle_kl(chain, o.xt, rho, g);
            }
            opt_cache_store(o, f, g);
            passes++;
        }
        opt_advance<false, N>(o, f, g);
        if (passes > 2000000) break;
    }
    for (int i = 0; i < 4; i++) out[i] = o.xk[i];
    out[4] = o.old_fval; out[5] = o.H[0]; out[6] = o.nfev; out[7] = o.k; out[8] = o.status; out[9] = o.fallbacks;
}
extern "C" {
__attribute__((visibility("default")))
void klhr_harness_bfgs_n(void *chain, const double *rho, int n, const double *x0, double gtol, double *out) {
    if (n == 1) bfgs_n<1>(chain, rho, x0, gtol, out);
    else bfgs_n<4>(chain, rho, x0, gtol, out);
}
}
