// Micro-benchmark behind the bf16x3 layers of readconv_kernel (hello_amd/csrc/readconv_fused.hip, bf16x3_layer): the
// steady-state step "request the operands of step u + DEPTH, wait for those of step u, 3 MFMAs per channel block" on the
// layer's own LDS image and occupancy (workgroups of 4 waves, two per CU), without the rest of the kernel around it:
//     DEPTH   operand requests in flight ahead of the MFMAs (1..6)
//     NBLK    16-channel blocks a wave multiplies per operand pair (1: the shipped layer; 2: half the reads per MFMA)
//     READS   2: hi and lo parts (the real layer); 1: hi only; 0: none (the MFMA floor)
// Prints cycles per v_mfma_f32_16x16x32_bf16 per SIMD (16 = the matrix pipe's own rate).
//     hipcc -O3 --offload-arch=gfx950 tools/bf16_step_bench.hip -o tools/_bin/stepbench && tools/_bin/stepbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
__device__ __forceinline__ int split_off(int row, int g, int part) { return row * 256 + 16 * ((g + 8 * part) ^ (2 * (row & 7))); }

constexpr int ROWS = 146, IMG = ROWS * 256, LDS_BYTES = 2 * IMG + 64;        // two split images: one workgroup pair per CU

template <int DEPTH, int NBLK, int READS, bool EPI, bool ACC3 = false>
__global__ __launch_bounds__(256, 2) void step_bench(float* out, int layers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) ((unsigned*)lds)[i] = 0x3f803f80u + i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    bf16x8 wh[NBLK][6], wl[NBLK][6];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wh[b][s][e] = (short)(0x3c00 + lane + s + e + b);
                wl[b][s][e] = (short)(0x3800 + lane + s + e);
            }
    f32x4 total = {0, 0, 0, 0};
    for (int layer = 0; layer < layers; ++layer) {
        const unsigned char* in = lds + (layer & 1) * IMG;
        unsigned char* outimg = lds + ((layer + 1) & 1) * IMG;
        const unsigned char* ph[6];
        const unsigned char* pl[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            ph[s] = in + split_off(j + s / 2, 4 * (s & 1) + q, 0);
            pl[s] = in + split_off(j + s / 2, 4 * (s & 1) + q, 1);
        }
        bf16x8 rh[DEPTH + 1], rl[DEPTH + 1];
        auto issue = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int k = u / 6, s = u % 6;
            if constexpr (READS >= 1) rh[u % (DEPTH + 1)] = *(const bf16x8*)(ph[s] + k * 16 * 256);
            else { bf16x8 t = wh[0][s]; asm volatile("" : "+v"(t)); rh[u % (DEPTH + 1)] = t; }      // opaque: no hoisting of the MFMAs
            if constexpr (READS >= 2) rl[u % (DEPTH + 1)] = *(const bf16x8*)(pl[s] + k * 16 * 256);
            else { bf16x8 t = wl[0][s]; asm volatile("" : "+v"(t)); rl[u % (DEPTH + 1)] = t; }
        };
        const f32x4 zero4 = {0, 0, 0, 0};
        f32x4 acc_a[NBLK][2], acc_b[NBLK][2], acc_c[NBLK][2];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc_a[b][0] = acc_a[b][1] = acc_b[b][0] = acc_b[b][1] = acc_c[b][0] = acc_c[b][1] = zero4;
        auto epilogue = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fmaxf(acc_a[b][k & 1][e] + (ACC3 ? acc_b[b][k & 1][e] + acc_c[b][k & 1][e] : acc_b[b][k & 1][e]), 0.f);
                if constexpr (EPI) {
                    // the split store of the real layer: hi = bf16(y), lo = bf16(y - hi), 8 bytes each
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                    unsigned hi[2], lo[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        bf2 h = {(__bf16)y[2 * p], (__bf16)y[2 * p + 1]};
                        hi[p] = __builtin_bit_cast(unsigned, h);
                        bf2 l = {(__bf16)(y[2 * p] - __uint_as_float(hi[p] << 16)), (__bf16)(y[2 * p + 1] - __uint_as_float(hi[p] & 0xffff0000u))};
                        lo[p] = __builtin_bit_cast(unsigned, l);
                    }
                    const int row = 16 * k + j + 1, ch4 = 4 * ((wave * NBLK + b) & 3) + q;
                    *(uint2*)(outimg + row * 256 + 16 * (((ch4 >> 1)) ^ (2 * (row & 7))) + 8 * (ch4 & 1)) = make_uint2(hi[0], hi[1]);
                    *(uint2*)(outimg + row * 256 + 16 * (((ch4 >> 1) + 8) ^ (2 * (row & 7))) + 8 * (ch4 & 1)) = make_uint2(lo[0], lo[1]);
                } else {
                    total += y;
                }
                acc_a[b][k & 1] = zero4;
                acc_b[b][k & 1] = zero4;
                acc_c[b][k & 1] = zero4;
            }
        };
        constexpr int NT = 9, NU = NT * 6;
        static_for<0, DEPTH>(issue);
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int k = u / 6, s = u % 6;
            if constexpr (u + DEPTH < NU) issue(std::integral_constant<int, u + DEPTH>{});
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 xh = rh[u % (DEPTH + 1)], xl = rl[u % (DEPTH + 1)];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                acc_a[b][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][s], xh, acc_a[b][k & 1], 0, 0, 0);
                acc_b[b][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[b][s], xl, acc_b[b][k & 1], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                if constexpr (ACC3) acc_c[b][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b][s], xh, acc_c[b][k & 1], 0, 0, 0);
                else acc_b[b][k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[b][s], xh, acc_b[b][k & 1], 0, 0, 0);
            }
            if constexpr (s == 1 && k >= 1) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
            if constexpr (u == NU - 1) epilogue(std::integral_constant<int, NT - 1>{});
        });
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3] + (float)lds[threadIdx.x];
}

template <int DEPTH, int NBLK, int READS, bool EPI, bool ACC3 = false>
void run(float* d, double ghz) {
    const int layers = 400;
    auto k = step_bench<DEPTH, NBLK, READS, EPI, ACC3>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), LDS_BYTES, 0, d, layers);
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(512), dim3(256), LDS_BYTES, 0, d, layers);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 3;
    // per SIMD: two waves (one of each workgroup of the CU) x layers x 54 steps x 3 NBLK MFMAs
    const double mfmas = 2.0 * layers * 54 * 3 * NBLK;
    printf("ACC3 %d DEPTH %d  NBLK %d  READS %d  EPI %d : %8.3f ms  %6.2f cycles per MFMA per SIMD (at %.2f GHz)  %6.2f us per layer\n", (int)ACC3, DEPTH, NBLK, READS,
           (int)EPI, ms, ms * 1e-3 * ghz * 1e9 / mfmas, ghz, ms * 1e3 / layers);
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    run<2, 1, 2, true>(d, ghz);
    run<2, 1, 2, false>(d, ghz);
    run<2, 1, 2, true, true>(d, ghz);
    run<2, 1, 2, false, true>(d, ghz);
    run<3, 1, 2, true, true>(d, ghz);
    run<4, 1, 2, true, true>(d, ghz);
    run<2, 2, 2, true>(d, ghz);
    run<2, 2, 2, true, true>(d, ghz);
    run<2, 2, 2, false, true>(d, ghz);
    run<3, 2, 2, true, true>(d, ghz);
    return 0;
}
