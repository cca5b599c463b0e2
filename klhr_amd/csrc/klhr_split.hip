// klhr_split.hip — one translation unit per sampler variant of the split-phase pipeline (klhr_split.hpp).
// Compiled by klhr_amd/build.py with -DKLHR_VARIANT={0,1,2}.  The kernels are far smaller than the monolithic ones of
// klhr_hot.hip (the fit kernels hold the optimiser state and seven line sums, nothing else), so over-relaxation and the
// three separable models share a unit.
#define KLHR_HOT_TU 1
#include "klhr_split.hpp"

#ifndef KLHR_VARIANT
#error "compile with -DKLHR_VARIANT=0|1|2"
#endif
#define KLHR_CAT2_(a, b) a##b
#define KLHR_CAT2(a, b) KLHR_CAT2_(a, b)

namespace klhr {

template <int MODEL>
static void launch_split_model(const KParams &kp0, int what, bool overrelaxed, int do_tail, int do_head, hipStream_t stream) {
    constexpr int V = KLHR_VARIANT;
    // fit / probe kernels: lane = chain, 64 chains per wavefront (few chains go to the node-parallel kernel, what == 4)
    KParams kp = kp0;
    const unsigned tiles = (unsigned)((kp.C + 63) / 64);
    const unsigned stiles = (unsigned)(kp.Cpad / kp.tw);          // streaming kernel: one wavefront per tile of tw chains,
    // pair mode (two wavefronts per tile, klhr_split.hpp): a launch that runs a tail AND a head on narrow tiles, synchronous rounds
    if (kp.stream_pair && kp.tw == 16 && do_tail && do_head && kp.stream_waves >= 2) kp.stream_waves &= ~1;
    else kp.stream_pair = 0;
    const unsigned tiles_wg = kp.stream_pair ? kp.stream_waves / 2 : kp.stream_waves;
    const unsigned sgroups = (stiles + tiles_wg - 1) / tiles_wg;                    // tiles_wg tiles per workgroup
    const dim3 block(64), sblock(64 * kp.stream_waves);
    const unsigned fitgrid = tiles;
    if (what == 8) {
        hipLaunchKernelGGL((klhr_fit12_kernel<MODEL, V>), dim3(fitgrid), block, 0, stream, kp);
    } else if (what == 7) {
        // whole iterations per tile (klhr_tile_kernel): one workgroup of two wavefronts per 16-chain tile, kp.n_iters draws
        kp.stream_pair = 1; kp.stream_waves = 2;
        const size_t lds = ((size_t)2 * kp.D * 16 + KLHR_ZTAB_DOUBLES + 4 * 16 + 2) * sizeof(double);
        // up to 8 192 chains one wavefront per SIMD holds every tile; beyond, the 256-register build (two per SIMD)
#define KLHR_LAUNCH_TILE(O, WV)                                                                                                        \
    do {                                                                                                                               \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)klhr_tile_kernel<MODEL, O, V, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((klhr_tile_kernel<MODEL, O, V, WV>), dim3(stiles), dim3(128), lds, stream, kp);                              \
    } while (0)
        if (stiles <= 512) { if (overrelaxed) KLHR_LAUNCH_TILE(true, 1); else KLHR_LAUNCH_TILE(false, 1); }
        else { if (overrelaxed) KLHR_LAUNCH_TILE(true, 2); else KLHR_LAUNCH_TILE(false, 2); }
#undef KLHR_LAUNCH_TILE
    } else if (what == 4) {
        hipLaunchKernelGGL((klhr_fitn_kernel<MODEL, V>), dim3((unsigned)((kp.C + 7) / 8)), block, 0, stream, kp);
    } else if (what == 3) {
        hipLaunchKernelGGL((klhr_probe_kernel<MODEL, V>), dim3(tiles), block, 0, stream, kp);
    } else {
        const size_t lds = ((size_t)kp.stream_waves * kp.D * kp.tw + KLHR_ZTAB_DOUBLES + (kp.stream_pair ? (size_t)tiles_wg * 4 * kp.tw + 2 : 0)) * sizeof(double);
#define KLHR_LAUNCH_STREAM(KERNEL, O)                                                                                                  \
    do {                                                                                                                               \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)KERNEL<MODEL, O, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((KERNEL<MODEL, O, V>), dim3(sgroups), sblock, lds, stream, kp, do_tail, do_head);                            \
    } while (0)
        if (kp.tw == 16) { if (overrelaxed) KLHR_LAUNCH_STREAM(klhr_stream16_kernel, true); else KLHR_LAUNCH_STREAM(klhr_stream16_kernel, false); }
        else { if (overrelaxed) KLHR_LAUNCH_STREAM(klhr_stream_kernel, true); else KLHR_LAUNCH_STREAM(klhr_stream_kernel, false); }
#undef KLHR_LAUNCH_STREAM
    }
}

// what: 0 streaming kernel (do_tail / do_head), 3 KL / line probe, 4 both fit stages node-parallel (klhr_fitn_kernel),
// 7 whole iterations per tile (kp.n_iters draws in one launch), 8 both fit stages lane = chain in one launch (klhr_fit12_kernel)
void KLHR_CAT2(launch_split_v, KLHR_VARIANT)(const KParams &kp, int model_id, int what, bool overrelaxed, int do_tail, int do_head, hipStream_t stream) {
    switch (model_id) {
    case KLHR_MODEL_FUNNEL: launch_split_model<MODEL_FUNNEL>(kp, what, overrelaxed, do_tail, do_head, stream); break;
    case KLHR_MODEL_ILL_NORMAL: launch_split_model<MODEL_ILL_NORMAL>(kp, what, overrelaxed, do_tail, do_head, stream); break;
    case KLHR_MODEL_AR1: launch_split_model<MODEL_AR1>(kp, what, overrelaxed, do_tail, do_head, stream); break;
    default: launch_split_model<MODEL_NORMAL>(kp, what, overrelaxed, do_tail, do_head, stream); break;
    }
}

}  // namespace klhr
