// execz_copy.hip — stand-alone reproducer of the code-generation hazard of DESIGN.md ("Code-generation hazard").
//
// Compiler: HIP 7.2.26015-fc0010cf6a, AMD clang 22.0.0git (roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5),
//           hipcc --offload-arch=gfx950 -O3 -c execz_copy.hip -save-temps=obj      (no GPU needed: inspect the .s)
// Check:    python tools/exec_lint.py execz_copy-hip-amdgcn-amd-amdhsa-gfx950.s     -> block .LBB0_15 is flagged
//
// 225 doubles live across a divergent region (an `if` around a loop with a lane-dependent trip count) at 256 VGPRs: the
// allocator parks values in AGPRs around the region.  The region's join block .LBB0_15 is entered by s_cbranch_execz; the
// SGPR spills (v_writelane) emitted at its top are not "block prologue" instructions, so the allocator's insertion point for the
// AGPR save `v_accvgpr_write_b32 a1, v227` lands AHEAD of `s_or_b64 exec, exec, s[54:55]`: the value is saved only for the
// lanes that went through the region and every other lane later reads a stale a1.  Source and semantics are fine; smaller N /
// NS (less VGPR / SGPR pressure) or dropping the `if` make the placement correct again.  This is the pattern rocgdb showed
// behind the round-1 memory fault (stale window index after the ziggurat loop) and that the rebuilt unit of commit e4cb1ae
// still shows (36 copies ahead of the restore in klhr_iter_kernel<glmm-poisson, over-relaxed, SUBKLHRSINH>).
#include <hip/hip_runtime.h>
#define N 225
#define NS 48
struct P { double s[NS]; const double *in; double *out; const int *trips; };
__global__ __launch_bounds__(64) void execz_copy(const P p) {
    const int lane = threadIdx.x;
    double a[N];
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = p.in[i * 64 + lane];
    double x = p.in[lane], acc = 0.0;
    for (int it = 0; it < 4; ++it) {
        int t = p.trips[lane + it];
        if (t & 1)
        while (t-- > 0) {                       // divergent: every lane its own trip count
            double y = x;
            y = fma(y, y, 0.25) - exp(-y * y) + log1p(y * y);
            if (y > 3.0) { acc += sin(y) * p.s[t % NS]; x = y * 0.5; break; } else x = y + 0.125;
        }
#pragma unroll
        for (int i = 0; i < N; i++) { acc = fma(acc, p.s[i % NS], a[i] * x); a[i] = fma(a[i], 1.0000001, acc * 1e-9); }
    }
    p.out[lane] = acc;
}
int main() { return 0; }
