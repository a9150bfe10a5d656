// microbenchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 and 32x32x2 on gfx950, 1/2 waves per SIMD,
// with 0 / N independent VALU ops interleaved per MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int CHAINS>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters, float seed) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8] = {a, b, a, b, a, b, a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < NV; ++x) v[(c + x) & 7] = v[(c + x) & 7] * 1.0001f + 0.5f;
            }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int x = 0; x < 8; ++x) s += v[x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int CHAINS>
float run(int wgs, int iters) {
    float* d; hipMalloc(&d, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k16<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}
template <int NV, int CHAINS>
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters, float seed) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = seed;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8] = {a, b, a, b, a, b, a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < NV; ++x) v[(c + x) & 7] = v[(c + x) & 7] * 1.0001f + 0.5f;
            }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    for (int x = 0; x < 8; ++x) s += v[x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int CHAINS>
float run32(int wgs, int iters) {
    float* d; hipMalloc(&d, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k32<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same with packed VALU operations (v_pk_fma_f32: two floats per lane and instruction)
template <int NV, int CHAINS>
__global__ __launch_bounds__(256, 2) void k16pk(float* out, int iters, float seed) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    f32x2 v[8];
    for (int x = 0; x < 8; ++x) v[x] = f32x2{a + x, b - x};
    const f32x2 m = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < NV; ++x) v[(c + x) & 7] = __builtin_elementwise_fma(v[(c + x) & 7], m, d);
            }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int x = 0; x < 8; ++x) s += v[x][0] + v[x][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int CHAINS>
float runpk(int wgs, int iters) {
    float* d; hipMalloc(&d, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k16pk<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16pk<NV, CHAINS>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms;
}
int main() {
    int dev = 0, cus = 0, clk = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
    printf("CUs %d clock %d kHz\n", cus, clk);
    const int iters = 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {           // workgroups per CU (= waves per SIMD)
        const int wgs = cus * wpc;
        const double mf = (double)iters * 4 * 5;    // MFMAs per wave
#define R(NV) { float ms = run<NV, 5>(wgs, iters); double tf = mf * 2048.0 * wgs * 4 / (ms * 1e-3) / 1e12; \
        printf("waves/SIMD %d valu/mfma %d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", wpc, NV, ms, tf, ms * 1e6 / (mf * wpc)); }
        R(0) R(1) R(2) R(4) R(6)
#define RPK(NV) { float ms = runpk<NV, 5>(wgs, iters); \
        printf("16x16x4 + v_pk_fma_f32: waves/SIMD %d pk/mfma %d: %.3f ms, %.1f ns per MFMA per SIMD\n", wpc, NV, ms, ms * 1e6 / (mf * wpc)); }
        RPK(1) RPK(2) RPK(4)
        const double mf32 = (double)iters / 2 * 4 * 3;    // 32x32x2 MFMAs per wave (2 048 MACs each)
#define R32(NV) { float ms = run32<NV, 3>(wgs, iters / 2); double tf = mf32 * 4096.0 * wgs * 4 / (ms * 1e-3) / 1e12; \
        printf("32x32x2: waves/SIMD %d valu/mfma %d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", wpc, NV, ms, tf, ms * 1e6 / (mf32 * wpc)); }
        R32(0) R32(1) R32(2) R32(4) R32(8) R32(12)
    }
    return 0;
}
