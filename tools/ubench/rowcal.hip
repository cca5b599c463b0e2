// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for this project's access pattern (MI355X_MICROARCH.md, HBM: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern"): every lane reads /
// writes its OWN row of D doubles with 8-byte accesses (rows are D*8 bytes apart), the pattern of the lane kernel's
// adaptation state.  Arrays are far larger than L2 + Infinity Cache, each byte is touched exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void row_read(const double *a, double *out, long rows, int D) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s = 0;
    for (int i = 0; i < D; i++) s += a[r * D + i];
    if (s == 1234.5) out[0] = s;
}
__global__ void row_write(double *a, long rows, int D, double v) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    for (int i = 0; i < D; i++) a[r * D + i] = v + i;
}
__global__ void coalesced_read(const double2 *a, double *out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (long)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1234.5) out[0] = s;
}
// The split pipeline's pattern (klhr_split.hpp): a wavefront walks the D rows of its tile, row i = W lanes x 8 bytes
// contiguous (512 B at W = 64, 128 B at W = 16), tiles D*W*8 bytes apart; each byte touched once.
__global__ void tile_read(const double *a, double *out, long tiles, int D, int W) {
    const long t = blockIdx.x;
    const int lane = threadIdx.x;
    if (t >= tiles || lane >= W) return;
    double s = 0;
    for (int i = 0; i < D; i++) s += a[(t * D + i) * W + lane];
    if (s == 1234.5) out[0] = s;
}
__global__ void tile_write(double *a, long tiles, int D, int W, double v) {
    const long t = blockIdx.x;
    const int lane = threadIdx.x;
    if (t >= tiles || lane >= W) return;
    for (int i = 0; i < D; i++) a[(t * D + i) * W + lane] = v + i;
}
int main(int argc, char **argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 10;
    const long rows = 1L << 25;                       // 33.5 M rows * D * 8 B = 2.7 GB at D = 10
    double *a, *out;
    if (hipMalloc(&a, rows * D * sizeof(double)) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, rows * D * sizeof(double));
    hipDeviceSynchronize();
    hipLaunchKernelGGL(row_write, dim3((rows + 63) / 64), dim3(64), 0, 0, a, rows, D, 1.0);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(row_read, dim3((rows + 63) / 64), dim3(64), 0, 0, a, out, rows, D);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(coalesced_read, dim3(4096), dim3(256), 0, 0, (const double2 *)a, out, rows * D / 2);
    hipDeviceSynchronize();
    for (int W = 64; W >= 16; W /= 4) {
        hipLaunchKernelGGL(tile_write, dim3(rows / W), dim3(64), 0, 0, a, rows / W, D, W, 2.0);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(tile_read, dim3(rows / W), dim3(64), 0, 0, a, out, rows / W, D, W);
        hipDeviceSynchronize();
    }
    printf("bytes per kernel: %ld (rows %ld, D %d)\n", rows * D * 8, rows, D);
    return 0;
}
