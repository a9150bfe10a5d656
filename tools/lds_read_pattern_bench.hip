// Micro-benchmark: cost of one ds_read_b128 wave-instruction for the address patterns the fused kernels use, measured
// (not modelled): N back-to-back reads per wave, 4 or 8 waves per CU.  Prints ns per read per CU and the ratio to the
// linear (conflict-free) pattern.
//     hipcc -O3 --offload-arch=gfx950 tools/lds_read_pattern_bench.hip -o /tmp/ldsbench && /tmp/ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int split_off(int row, int g, int part) { return row * 256 + 16 * ((g + 8 * part) ^ (2 * (row & 7))); }
__device__ __forceinline__ int split_off32(int row, int g, int part) { return row * 128 + 16 * ((g + 4 * part) ^ (2 * ((row >> 1) & 3))); }

template <int P>
__device__ __forceinline__ int pattern(int lane, int it) {
    const int j = lane & 15, q = lane >> 4;
    if constexpr (P == 0) return lane * 16;                                           // linear
    if constexpr (P == 1) return split_off(j + (it % 3), (it & 1) * 4 + q, (it >> 1) & 1);   // bf16x3 64-channel operand
    if constexpr (P == 2) return (j + (it % 3)) * 256 + 16 * q;                        // no swizzle: 16 rows on one column
    if constexpr (P == 3) return split_off32(j + (it % 3), q, it & 1);                // bf16x3 32-channel operand
    if constexpr (P == 4) return ((j + (it % 3)) * 64 + 4 * ((4 * (it & 3) + q) ^ (2 * ((j + (it % 3)) & 7)))) * 4;   // fp32 SW_OLD 64 ch
    if constexpr (P == 5) return (j + (it % 3)) * 256 + 16 * ((((it & 1) * 4 + q) * 2 + ((it >> 1) & 1)) ^ (2 * ((j + it % 3) & 7)));  // hi/lo interleaved per group
    return 0;
}

template <int P>
__global__ __launch_bounds__(512) void bench(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 40960 / 4; i += blockDim.x) ((float*)lds)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    const unsigned char* base = lds + (wave & 1) * 4096 * 4;
    unsigned addr[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) addr[u] = (unsigned)(size_t)(base + pattern<P>(lane, u));
    for (int it = 0; it < iters; it += 12) {
        f32x4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr[u]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 12; ++u) acc += v[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int P>
float run(int waves, int iters, float* d) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(bench<P>, dim3(256), dim3(64 * waves), 40960, 0, d, iters);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench<P>, dim3(256), dim3(64 * waves), 40960, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int iters = 120000;
    const char* names[] = {"linear (lane * 16)", "bf16x3 64-channel operand (split_off)", "no swizzle (16 rows, one column)",
                           "bf16x3 32-channel operand (split_off32)", "fp32 64-channel operand (SW_OLD)", "hi/lo interleaved per group"};
    for (int waves : {4, 8}) {
        float ms[6] = {run<0>(waves, iters, d), run<1>(waves, iters, d), run<2>(waves, iters, d), run<3>(waves, iters, d),
                       run<4>(waves, iters, d), run<5>(waves, iters, d)};
        for (int p = 0; p < 6; ++p)
            printf("%d waves/CU  %-44s %8.3f ms  %6.2f ns per wave-read per CU  x%.2f\n", waves, names[p], ms[p],
                   ms[p] * 1e6 / ((double)iters * waves), ms[p] / ms[0]);
    }
    return 0;
}
