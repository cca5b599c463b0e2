// klhr_free.hip — one translation unit per (sampler variant, over-relaxation) pair of the free-running octet kernel
// (klhr_free_kernel.hpp): the general hot path for targets without a closed-form line restriction.
// Compiled by klhr_amd/build.py with -DKLHR_VARIANT={0,1,2} -DKLHR_OVERRELAX={0,1}, like klhr_hot.hip.
#define KLHR_HOT_TU 1
#include "klhr_free_kernel.hpp"

#ifndef KLHR_VARIANT
#error "compile with -DKLHR_VARIANT=0|1|2 -DKLHR_OVERRELAX=0|1"
#endif
#define KLHR_CAT_(a, b, c, d) a##b##c##d
#define KLHR_CAT(a, b, c, d) KLHR_CAT_(a, b, c, d)

namespace klhr {

void KLHR_CAT(launch_free_v, KLHR_VARIANT, _o, KLHR_OVERRELAX)(const KParams &kp, int model_id, int mh_exact, hipStream_t stream) {
    constexpr int V = KLHR_VARIANT;
    constexpr bool O = KLHR_OVERRELAX != 0;
    const dim3 grid((kp.C + 7) / 8), block(64);
    // chain rows + staged model data (ill-normal D doubles, glmm-poisson 4n doubles) + the ziggurat's WI / KI copies when they fit
    const size_t naux = model_id == KLHR_MODEL_GLMM_POISSON ? (size_t)4 * kp.md.n : (model_id == KLHR_MODEL_ILL_NORMAL ? (size_t)kp.D : 0);
    size_t lds = ((size_t)8 * kp.S + naux) * sizeof(double);
    if (lds + KLHR_ZTAB_DOUBLES * sizeof(double) <= 64 * 1024) lds += KLHR_ZTAB_DOUBLES * sizeof(double);
    switch (model_id) {
    case KLHR_MODEL_FUNNEL: hipLaunchKernelGGL((klhr_free_kernel<MODEL_FUNNEL, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    case KLHR_MODEL_ILL_NORMAL: hipLaunchKernelGGL((klhr_free_kernel<MODEL_ILL_NORMAL, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    case KLHR_MODEL_GLMM_POISSON: hipLaunchKernelGGL((klhr_free_kernel<MODEL_GLMM_POISSON, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    case KLHR_MODEL_AR1: hipLaunchKernelGGL((klhr_free_kernel<MODEL_AR1, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    case KLHR_MODEL_EARNINGS: hipLaunchKernelGGL((klhr_free_kernel<MODEL_EARNINGS, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    default: hipLaunchKernelGGL((klhr_free_kernel<MODEL_NORMAL, O, V>), grid, block, lds, stream, kp, mh_exact); break;
    }
}

}  // namespace klhr
