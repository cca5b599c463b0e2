// klhr_hot.hip — one translation unit per (sampler variant, over-relaxation) pair of the hot-path kernels.
// Compiled six times by klhr_amd/build.py with -DKLHR_VARIANT={0,1,2} -DKLHR_OVERRELAX={0,1}: the variants are
// compile-time template arguments so that the code of one never weighs on the register allocation of another
// (measured: a runtime-flagged over-relaxed branch cost the default path 8 % and 1.5 KB of scratch per lane).
#define KLHR_HOT_TU 1
#include "klhr_kernels.hpp"
#include "klhr_lane_kernel.hpp"

#ifndef KLHR_VARIANT
#error "compile with -DKLHR_VARIANT=0|1|2|3 -DKLHR_OVERRELAX=0|1"
#endif
#define KLHR_CAT_(a, b, c, d) a##b##c##d
#define KLHR_CAT(a, b, c, d) KLHR_CAT_(a, b, c, d)

namespace klhr {

void KLHR_CAT(launch_hot_v, KLHR_VARIANT, _o, KLHR_OVERRELAX)(const KParams &kp, int model_id, bool lane, hipStream_t stream) {
    constexpr int V = KLHR_VARIANT;
    constexpr bool O = KLHR_OVERRELAX != 0;
    if constexpr (V == VAR_SLICE) { (void)lane; }        // slice.py runs on the octet kernel only (klhr_api.hip)
    if (lane && V != VAR_SLICE) {
      if constexpr (V != VAR_SLICE) {
        const dim3 grid((kp.C + 63) / 64), block(64);
        const size_t lds = ((size_t)64 * kp.S + LANE_LDS_EXTRA) * sizeof(double);
        switch (model_id) {
        case KLHR_MODEL_FUNNEL: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_FUNNEL, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_ILL_NORMAL: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_ILL_NORMAL, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_GLMM_POISSON: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_GLMM_POISSON, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_AR1: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_AR1, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_EARNINGS: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_EARNINGS, O, V>), grid, block, lds, stream, kp); break;
        default: hipLaunchKernelGGL((klhr_lane_kernel<MODEL_NORMAL, O, V>), grid, block, lds, stream, kp); break;
        }
      }
    } else {
        const dim3 grid((kp.C + 7) / 8), block(64);
        // chain rows + staged model data (klhr_iter_kernel: ill-normal D doubles, glmm-poisson 4n doubles)
        const size_t naux = model_id == KLHR_MODEL_GLMM_POISSON ? (size_t)4 * kp.md.n : (model_id == KLHR_MODEL_ILL_NORMAL ? (size_t)kp.D : 0);
        size_t lds = ((size_t)8 * kp.S + naux) * sizeof(double);
        if (lds + KLHR_ZTAB_DOUBLES * sizeof(double) <= 64 * 1024) lds += KLHR_ZTAB_DOUBLES * sizeof(double);      // ziggurat WI / KI copies
        switch (model_id) {
        case KLHR_MODEL_FUNNEL: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_FUNNEL, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_ILL_NORMAL: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_ILL_NORMAL, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_GLMM_POISSON: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_GLMM_POISSON, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_AR1: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_AR1, O, V>), grid, block, lds, stream, kp); break;
        case KLHR_MODEL_EARNINGS: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_EARNINGS, O, V>), grid, block, lds, stream, kp); break;
        default: hipLaunchKernelGGL((klhr_iter_kernel<MODEL_NORMAL, O, V>), grid, block, lds, stream, kp); break;
        }
    }
}

}  // namespace klhr
