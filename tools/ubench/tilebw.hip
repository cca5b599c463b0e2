// Streaming bandwidth of the split pipeline's tiled walk at the streaming kernel's occupancy (one wavefront per workgroup with a
// [D][64] LDS column -> three wavefronts per CU at D = 100): 8 bytes per lane and row (the present layout) against 16 bytes per
// lane (two coordinates interleaved), 5 arrays read + 4 written per coordinate, chunks of 8 rows in flight.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tilebw.hip -o tools/ubench/tilebw && tools/ubench/tilebw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <class T, int CH>
__global__ __launch_bounds__(64) void walk(const T *a0, const T *a1, const T *a2, const T *a3, const T *a4, T *o0, T *o1, T *o2, T *o3, int rows) {
    extern __shared__ double lds[];
    const long long base = (long long)blockIdx.x * rows * 64 + threadIdx.x;
    double acc = 0;
    for (int i0 = 0; i0 + CH <= rows; i0 += CH) {
        T v0[CH], v1[CH], v2[CH], v3[CH], v4[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const long long e = base + (long long)(i0 + k) * 64;
            v0[k] = a0[e]; v1[k] = a1[e]; v2[k] = a2[e]; v3[k] = a3[e]; v4[k] = a4[e];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const long long e = base + (long long)(i0 + k) * 64;
            if constexpr (sizeof(T) == 8) {
                const double s = v0[k] + v1[k] * v2[k] + v3[k] - v4[k];
                o0[e] = s; o1[e] = s * 0.5; o2[e] = s + 1.0; o3[e] = s - 1.0; acc += s;
            } else {
                T s; s.x = v0[k].x + v1[k].x * v2[k].x + v3[k].x - v4[k].x; s.y = v0[k].y + v1[k].y * v2[k].y + v3[k].y - v4[k].y;
                T h = s; h.x *= 0.5; h.y *= 0.5;
                o0[e] = s; o1[e] = h; o2[e] = s; o3[e] = h; acc += s.x + s.y;
            }
        }
    }
    if (acc == 12345.678) lds[threadIdx.x] = acc;
}
template <class T, int CH>
static double run(size_t tiles, int rows, size_t ldsbytes, const char *name) {
    const size_t n = tiles * rows * 64;
    T *a[5], *o[4];
    for (auto &p : a) { hipMalloc(&p, n * sizeof(T)); hipMemset(p, 0, n * sizeof(T)); }
    for (auto &p : o) hipMalloc(&p, n * sizeof(T));
    hipFuncSetAttribute((const void *)walk<T, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((walk<T, CH>), dim3(tiles), dim3(64), ldsbytes, 0, a[0], a[1], a[2], a[3], a[4], o[0], o[1], o[2], o[3], rows);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = 9.0 * n * sizeof(T);
    printf("%-34s tiles %zu rows %d LDS %zu KB: %.3f ms, %.2f TB/s\n", name, tiles, rows, ldsbytes / 1024, best, bytes / best / 1e9);
    for (auto &p : a) hipFree(p);
    for (auto &p : o) hipFree(p);
    return best;
}
int main() {
    for (size_t lds : {(size_t)51200, (size_t)25600, (size_t)0}) {
        run<double, 8>(2048, 100, lds, "8 B per lane, chunks of 8");
        run<double, 16>(2048, 100 - 100 % 16, lds, "8 B per lane, chunks of 16");
        run<double2, 4>(2048, 50 - 50 % 4, lds, "16 B per lane, chunks of 4 (= 8 coords)");
        run<double2, 8>(2048, 50 - 50 % 8, lds, "16 B per lane, chunks of 8 (= 16 coords)");
    }
    return 0;
}
