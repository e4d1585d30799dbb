// mfma_f64_probe.hip -- what v_mfma_f64_16x16x4_f64 offers the filter's small dense products on gfx950 (round 5).
//
// The filter's update back end (upd_tt / upd_s / upd_chol / upd_fsolve / upd_p) multiplies matrices of <= 144 x 144 doubles per
// stream with v_fma_f64 on 4 x 4 register tiles out of LDS.  Round 3 dismissed the fp64 matrix instruction because its peak equals
// the vector peak on this part (78.6 TFLOP/s).  That overlooks what the products are really bound by: a 4 x 4 fp64 register tile
// reads 64 B of LDS per lane and panel step for 16 FMAs (LDS-bandwidth bound at half the FMA rate), every FMA is a half-rate VALU
// instruction in the pipe the front-end's kernels saturate, and a 64 x 64 tile per workgroup wastes 30 % of a 141-row product.
// One MFMA does 1,024 multiply-adds for ONE issue slot and 16 B of operands per lane.  This program checks, on the hardware:
//   1. the lane maps (A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[row = (l >> 4) + 4 reg][col = l & 15]);
//   2. that register `s` of a result tile IS the B operand of k-step s of a following product  (E = A2 x D,   no lane movement),
//      and, used as the A operand, stands for the TRANSPOSED tile                                (F = D^T x B2, no lane movement);
//   3. the issue cost: ns per MFMA and SIMD with 1 .. 4 waves per SIMD, independent accumulators and one dependent chain;
//   4. whether a VALU-bound wave and an MFMA-bound wave on the same SIMD overlap (the front-end's kernels are VALU bound).
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe > mfma_f64_probe.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- 1, 2: lane maps.  A [16][4], B [4][16], A2 [16][16], B2 [16][16] row-major; D, E, F [16][16] row-major out.
__global__ void layout_kernel(const double* A, const double* B, const double* A2, const double* B2, double* D, double* E, double* F)
{
    const int l = threadIdx.x, lo = l & 15, hi = l >> 4;
    d4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[lo * 4 + hi], B[hi * 16 + lo], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(hi + 4 * r) * 16 + lo] = d[r];
    d4 e = {0, 0, 0, 0}, f = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[lo * 16 + 4 * s + hi], d[s], e, 0, 0, 0);      // E = A2 x D
        f = __builtin_amdgcn_mfma_f64_16x16x4f64(d[s], B2[(4 * s + hi) * 16 + lo], f, 0, 0, 0);    // F = D^T x B2
    }
    for (int r = 0; r < 4; ++r) { E[(hi + 4 * r) * 16 + lo] = e[r]; F[(hi + 4 * r) * 16 + lo] = f[r]; }
}

// ---- 3: issue cost.  One workgroup of 256 W threads per CU; every wave: n x 8 MFMAs on NACC accumulators.
template <int NACC>
__global__ void mfma_rate_kernel(double* out, int n)
{
    extern __shared__ double pin[];                      // 100 KB: one workgroup per CU
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NACC], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[threadIdx.x] = s + pin[0];
}

// ---- 4: overlap.  Workgroup of 512 threads = two waves per SIMD: waves 0-3 run `na` x 8 MFMAs, waves 4-7 `nv` x 32 v_fma_f64.
template <int KIND>   // 0: v_fma_f64, 1: v_mad_i32_i24 (the half-rate integer class of the LK kernel), 2: v_add_u32 + v_fma_f32 (full rate)
__global__ void overlap_kernel(double* out, int na, int nv)
{
    extern __shared__ double pin[];
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        d4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
        for (int it = 0; it < na; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else if (KIND == 0) {
        double x[32];
        const double m = 1.0 + threadIdx.x * 1e-12, c = 1e-9;
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = i;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = __builtin_fma(x[i], m, c);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
    } else if (KIND == 1) {
        int x[32];
        const int m = 3 + (threadIdx.x & 1), c = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = i + threadIdx.x;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
    } else {
        float x[32];
        const float m = 1.0f + threadIdx.x * 1e-7f, c = 1e-5f;
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = i;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) s += x[i];
    }
    if (s == 12345.678) out[threadIdx.x] = s + pin[0];
}

template <typename F> static double time_ms(F launch, int reps = 5)
{
    launch(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        launch(); CK(hipDeviceSynchronize());
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    // ---- layout
    std::vector<double> A(64), B(64), A2(256), B2(256), D(256), E(256), F(256);
    srand(5);
    for (auto& v : A) v = rand() % 17 - 8;
    for (auto& v : B) v = rand() % 17 - 8;
    for (auto& v : A2) v = rand() % 9 - 4;
    for (auto& v : B2) v = rand() % 9 - 4;
    double *dA, *dB, *dA2, *dB2, *dD, *dE, *dF, *dout;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dA2, 2048)); CK(hipMalloc(&dB2, 2048));
    CK(hipMalloc(&dD, 2048)); CK(hipMalloc(&dE, 2048)); CK(hipMalloc(&dF, 2048)); CK(hipMalloc(&dout, 8192));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, A2.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, B2.data(), 2048, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dA2, dB2, dD, dE, dF);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost)); CK(hipMemcpy(E.data(), dE, 2048, hipMemcpyDeviceToHost)); CK(hipMemcpy(F.data(), dF, 2048, hipMemcpyDeviceToHost));
    int badD = 0, badE = 0, badF = 0;
    std::vector<double> Dh(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; Dh[i * 16 + j] = s; if (s != D[i * 16 + j]) ++badD; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double e = 0, f = 0;
        for (int k = 0; k < 16; ++k) { e += A2[i * 16 + k] * Dh[k * 16 + j]; f += Dh[k * 16 + i] * B2[k * 16 + j]; }
        if (e != E[i * 16 + j]) ++badE;
        if (f != F[i * 16 + j]) ++badF;
    }
    printf("{\"layout\": {\"D_mismatches\": %d, \"E_A2xD_result_as_B_mismatches\": %d, \"F_DTxB2_result_as_A_mismatches\": %d},\n", badD, badE, badF);

    // ---- rate
    const size_t dyn = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(overlap_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(overlap_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(overlap_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    const int n = 4000;
    printf(" \"rate\": [\n");
    for (int W = 1; W <= 4; ++W) {
        const double t8 = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<8>, dim3(256), dim3(256 * W), dyn, 0, dout, n); });
        const double t2 = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(256), dim3(256 * W), dyn, 0, dout, n); });
        const double t1 = time_ms([&] { hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(256), dim3(256 * W), dyn, 0, dout, n); });
        const double per = 1e6 / ((double)n * 8 * W);
        const double tf = 256.0 * 4 * W * n * 8 * 2048.0 / (t8 * 1e-3) / 1e12;
        printf("  {\"waves_per_simd\": %d, \"ns_per_mfma_and_simd_8acc\": %.2f, \"ns_2acc\": %.2f, \"ns_dependent_chain\": %.2f, \"chip_tflops_8acc\": %.1f}%s\n",
               W, t8 * per, t2 * per, t1 * per, tf, W < 4 ? "," : "");
    }
    printf(" ],\n");
    // ---- overlap: na MFMA blocks beside nv blocks of 32 VALU instructions of one class, on the same SIMDs
    const int na = 2000, nv = 8000;
    printf(" \"overlap\": [\n");
    auto run = [&](auto kern, const char* name, bool last) {
        const double ta = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), dyn, 0, dout, na, 0); });
        const double tv = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), dyn, 0, dout, 0, nv); });
        const double tb = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), dyn, 0, dout, na, nv); });
        printf("  {\"valu\": \"%s\", \"mfma_wave_alone_ms\": %.3f, \"valu_wave_alone_ms\": %.3f, \"both_on_one_simd_ms\": %.3f}%s\n", name, ta, tv, tb, last ? "" : ",");
    };
    run(overlap_kernel<0>, "v_fma_f64", false);
    run(overlap_kernel<1>, "v_mad_i32_i24", false);
    run(overlap_kernel<2>, "v_fma_f32", true);
    printf(" ],\n \"overlap_note\": \"both = max(alone) means the matrix pipe and the VALU run side by side; both = sum means they share the unit\"}\n");
    return 0;
}
