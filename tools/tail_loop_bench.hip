// Microbenchmark of the planned LDS-resident expert tail: one workgroup of 8 waves per CU keeps 10 items x 9 rows x
// 256 channels in LDS (rows of 260 floats: consecutive rows shift by one bank quad; one shared zero row), wave w owns output channels
// 32 w .. 32 w + 31 of all 30 tiles (v_mfma_f32_32x32x2_f32, F(3,3): 5 accumulators), its weights come straight from global memory in
// lane order one chunk ahead, a layer = 32 chunks of 8 input channels, then barrier, in-place epilogue (12 ds_write_b128), barrier.
// Reports ns per MFMA per SIMD against the pipe's 26.67 ns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int G = 10, RSTR = 260, ROWS = 9 * G + 1;

template <bool TRANSFORM, bool EPILOGUE>
__global__ __launch_bounds__(512, 2) void tail_kernel(const float* __restrict__ w, float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lj = lane & 31, lh = lane >> 5;
    for (int i = t; i < ROWS * RSTR; i += 512) img[i] = i < RSTR ? 0.f : 1.0f + 1e-4f * (i % 977);
    __syncthreads();
    int addr[5];
    const int a = lj / 3, b = lj % 3;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int pos = 3 * b + c - 1;
        addr[c] = (lj < 3 * G && pos >= 0 && pos < 9) ? (9 * a + pos + 1) * RSTR + 4 * lh : 4 * lh;
    }
    float sum = 0.f;
    for (int layer = 0; layer < layers; ++layer) {
        const float* wl = w + ((size_t)(layer % 5) * 8 + wave) * (32 * 1280) + lane * 4;
        f32x16 acc[5];
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
        f32x4 wa[2][5];
#pragma unroll
        for (int c = 0; c < 5; ++c) wa[0][c] = *(const f32x4*)(wl + c * 256);
#pragma unroll 1
        for (int kb = 0; kb < 32; kb += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kb + h;
#pragma unroll
                for (int c = 0; c < 5; ++c) wa[h ^ 1][c] = *(const f32x4*)(wl + (size_t)((k + 1) & 31) * 1280 + c * 256);
                f32x4 d[5], v[5];
#pragma unroll
                for (int c = 0; c < 5; ++c) d[c] = *(const f32x4*)&img[addr[c] + k * 8];
                if constexpr (TRANSFORM) {
                    const f32x4 s31 = d[3] - d[1];
                    v[0] = 2.f * (d[0] - d[2]) + s31;
                    v[1] = s31 - (d[1] + d[2]);
                    v[2] = 3.f * (d[1] - d[2]) + s31;
                    v[3] = s31;
                    v[4] = (d[4] - d[2]) - 2.f * s31;
                } else {
#pragma unroll
                    for (int c = 0; c < 5; ++c) v[c] = d[c];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[h][c][e], v[c][e], acc[c], 0, 0, 0);
            }
        }
        __syncthreads();                      // every wave has read the image
        if constexpr (EPILOGUE) {
            if (lj < 3 * G) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float m0 = acc[0][4 * q + e], m1 = acc[1][4 * q + e], m2 = acc[2][4 * q + e], m3 = acc[3][4 * q + e], m4 = acc[4][4 * q + e];
                            const float x = u == 0 ? (m0 + (m1 + m2)) + m3 : (u == 1 ? __builtin_fmaf(m3, 2.f, m1 - m2) : __builtin_fmaf(m3, 4.f, m1 + m2) + m4);
                            y[e] = fminf(fmaxf(x, 0.f), 2.f);
                        }
                        *(f32x4*)&img[(3 * lj + u + 1) * RSTR + 32 * wave + 8 * q + 4 * lh] = y;
                    }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 5; ++c) sum += acc[c][0];
        }
        __syncthreads();
    }
    out[(size_t)blockIdx.x * 512 + t] = sum + img[RSTR + t];
}

template <bool TRANSFORM, bool EPILOGUE>
void run(const char* what, int cus, const float* w, float* out) {
    const size_t lds = (size_t)ROWS * RSTR * 4;
    hipFuncSetAttribute((const void*)tail_kernel<TRANSFORM, EPILOGUE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int layers = 50;
    hipLaunchKernelGGL((tail_kernel<TRANSFORM, EPILOGUE>), dim3(cus), dim3(512), lds, 0, w, out, 5);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tail_kernel<TRANSFORM, EPILOGUE>), dim3(cus), dim3(512), lds, 0, w, out, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)layers * 32 * 20 * 2);      // two waves per SIMD
    printf("%-72s %6.2f ns/MFMA/SIMD (%4.1f %% of the pipe), %.1f us per layer\n", what, per, 100.0 * 26.67 / per, 1e3 * ms / layers);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *w, *out;
    const size_t wf = (size_t)5 * 8 * 32 * 1280 + 4096;
    hipMalloc(&w, wf * 4);
    hipMemset(w, 0, wf * 4);
    hipMalloc(&out, (size_t)cus * 512 * 4);
    run<false, false>("LDS-resident loop: 5 ds_read_b128 + 5 weight loads + 20 MFMA per chunk", cus, w, out);
    run<true, false>("+ F(3,3) input transform", cus, w, out);
    run<true, true>("+ in-place epilogue (output transform, 12 ds_write_b128 per lane and layer)", cus, w, out);
    return 0;
}
