// fp64 MFMA vs fp64 VALU FMA on gfx950: the measurement behind "MFMA is not used" for the filter's GEMM-shaped phases
// (DESIGN.md 3b).  hipcc --offload-arch=gfx950 -O3 mfma_f64_microbench.hip -o mfma_f64_microbench && ./mfma_f64_microbench
// One wavefront per SIMD (256-thread workgroups, 4 per CU would be 4 waves/SIMD: here grid = 256 CUs x 4 workgroups),
// operands in registers, independent accumulators, random-ish data.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0)
{
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        a += 1e-9;
    }
    double s = 0;
    for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0)
{
    double acc[16];
    for (int u = 0; u < 16; ++u) acc[u] = u;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = __builtin_fma(a, b, acc[u]);
        a += 1e-9;
    }
    double s = 0;
    for (int u = 0; u < 16; ++u) s += acc[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    const int blocks = 256 * 4, iters = 20000;
    double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        // one v_mfma_f64_16x16x4 = 16*16*4 MACs = 2048 flops per wavefront-instruction
        const double tf_m = (double)blocks * 4 * iters * 4 * 2048.0 / (ms * 1e-3) / 1e12;
        hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double tf_v = (double)blocks * 256 * iters * 16 * 2.0 / (ms * 1e-3) / 1e12;
        if (rep) printf("{\"v_mfma_f64_16x16x4_TFLOPs\": %.1f, \"v_fma_f64_TFLOPs\": %.1f, \"waves_per_simd\": 4}\n", tf_m, tf_v);
    }
    return 0;
}
