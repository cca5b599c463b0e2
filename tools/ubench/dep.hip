// micro-benchmark: issue cadence of dependent vs independent v_fma_f64 with ONE wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64, 1) void k(double *out, int n, double a, double b) {
    double x[ILP];
    for (int j = 0; j < ILP; j++) x[j] = threadIdx.x * 1e-3 + j;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int j = 0; j < ILP; j++) x[j] = __builtin_fma(x[j], a, b);
    }
    double s = 0; for (int j = 0; j < ILP; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int ILP> void run(double *d, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int n = 20000;
    hipLaunchKernelGGL(k<ILP>, dim3(waves), dim3(64), 0, 0, d, 10, 0.999, 1e-3);
    hipEventRecord(e0); hipLaunchKernelGGL(k<ILP>, dim3(waves), dim3(64), 0, 0, d, n, 0.999, 1e-3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)n * 16 * ILP;
    printf("ILP %d waves %d: %.3f ms, %.2f ns per wave-instruction (%.2f cycles @2.4GHz)\n", ILP, waves, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
}
int main() {
    double *d; hipMalloc(&d, 1 << 24);
    for (int waves : {1024, 2048}) { run<1>(d, waves); run<2>(d, waves); run<4>(d, waves); run<8>(d, waves); }
    return 0;
}
