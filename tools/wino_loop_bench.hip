// Microbenchmark: where the main loop of conv1d_wino_kernel loses matrix-pipe time.  One chunk = 10 ds_read_b128 (the lane's A and
// B operands) + NV VALU operations + 20 v_mfma_f32_32x32x2_f32 over 5 accumulators; optional: a barrier per chunk, 6 global b128
// loads per chunk (prefetched a chunk ahead), 6 ds_write_b128 per chunk.  Reported: ns per MFMA per SIMD (27.1 = the pipe's 64 cycles at
// 2.4 GHz... minus DVFS) at 1 / 2 / 3 workgroups of 4 waves per CU.
//     hipcc -O3 --offload-arch=gfx950 tools/wino_loop_bench.hip -o tools/_bin/wino_loop_bench && tools/_bin/wino_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 44, ROWS = 64;

template <int READS, int NV, bool BARRIER, int LOADS, int WRITES>
__global__ __launch_bounds__(256, 3) void loop_kernel(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s_a[2][ROWS * LD];
    __shared__ __attribute__((aligned(16))) float s_w[2][ROWS * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lj = lane & 31, lh = lane >> 5;
    for (int i = t; i < 2 * ROWS * LD; i += 256) { (&s_a[0][0])[i] = 1.0f + 1e-3f * i; (&s_w[0][0])[i] = 0.5f - 1e-3f * i; }
    __syncthreads();
    f32x16 acc[5];
    for (int c = 0; c < 5; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    const int w_lds = ((wave & 1) * 32 + lj) * LD + lh * 4, a_lds = ((wave >> 1) * 32 + lj) * LD + lh * 4;
    const float* g = src + (size_t)blockIdx.x * 4096 + t * 4;
    f32x4 r[LOADS > 0 ? LOADS : 1];
    for (int j = 0; j < LOADS; ++j) r[j] = *(const f32x4*)(g + j * 1024);
    f32x4 wa[5] = {}, d[5] = {};
    for (int it = 0; it < iters; ++it) {
        const int st = it & 1;
        f32x4 rn[LOADS > 0 ? LOADS : 1];
        for (int j = 0; j < LOADS; ++j) rn[j] = *(const f32x4*)(g + ((it + 1) & 7) * 8192 + j * 1024);
        if constexpr (READS >= 5) {
#pragma unroll
            for (int c = 0; c < 5; ++c) d[c] = *(const f32x4*)&s_a[st][a_lds + c * 8];
        }
        if constexpr (READS >= 10) {
#pragma unroll
            for (int c = 0; c < 5; ++c) wa[c] = *(const f32x4*)&s_w[st][w_lds + c * 8];
        }
        f32x4 v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) v[c] = d[c];
        if constexpr (NV > 0) {                       // the F(3,3) input transform: 36 VALU operations (9 per float)
            const f32x4 s31 = d[3] - d[1];
            v[0] = 2.f * (d[0] - d[2]) + s31;
            v[1] = s31 - (d[1] + d[2]);
            v[2] = 3.f * (d[1] - d[2]) + s31;
            v[3] = s31;
            v[4] = (d[4] - d[2]) - 2.f * s31;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][e], v[c][e], acc[c], 0, 0, 0);
        if constexpr (READS == 0) {                   // keep the operands loop-carried so nothing is hoisted
#pragma unroll
            for (int c = 0; c < 5; ++c) { d[c][0] += 1e-9f; }
        }
#pragma unroll
        for (int j = 0; j < WRITES; ++j) {
            const f32x4 val = LOADS > 0 ? r[j % (LOADS > 0 ? LOADS : 1)] : f32x4{1.f, 2.f, 3.f, 4.f};
            *(f32x4*)&(j & 1 ? s_w : s_a)[st ^ 1][((t + 256 * (j >> 1)) % (ROWS * LD / 4)) * 4] = val;
        }
        for (int j = 0; j < LOADS; ++j) r[j] = rn[j];
        if constexpr (BARRIER) __syncthreads();
    }
    float s = 0;
    for (int c = 0; c < 5; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    for (int j = 0; j < LOADS; ++j) s += r[j][0];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

// the staging by LDS-DMA: DMAS x (buffer_load_dwordx4 ... lds) per wave and chunk into the idle image, no VGPR, no ds_write
template <int DMAS>
__global__ __launch_bounds__(256, 3) void dma_kernel(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s_a[2][ROWS * 40];
    __shared__ __attribute__((aligned(16))) float s_w[2][ROWS * 40];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lj = lane & 31, lh = lane >> 5;
    for (int i = t; i < 2 * ROWS * 40; i += 256) { (&s_a[0][0])[i] = 1.0f + 1e-3f * i; (&s_w[0][0])[i] = 0.5f - 1e-3f * i; }
    __syncthreads();
    f32x16 acc[5];
    for (int c = 0; c < 5; ++c)
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    // unpadded 40-float rows, quads rotated by (row >> 3) & 1: conflict-free for the b128 lane groups
    const int rw = (wave & 1) * 32 + lj, ra = (wave >> 1) * 32 + lj;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 4096), 0, 1 << 20, 0x00020000);
    const unsigned voff = (unsigned)(t * 16);
    for (int it = 0; it < iters; ++it) {
        const int st = it & 1;
#pragma unroll
        for (int j = 0; j < DMAS; ++j) {
            float* dst = (j & 1 ? &s_w[st ^ 1][0] : &s_a[st ^ 1][0]) + (wave * 2 + (j >> 1)) * 256;       // a 1 KiB piece per wave-instruction
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, ((it + 1) & 7) * 32768 + j * 4096, 0, 0);
        }
        f32x4 wa[5], d[5], v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) d[c] = *(const f32x4*)&s_a[st][ra * 40 + 4 * ((2 * c + lh + ((ra >> 3) & 1)) % 10)];
#pragma unroll
        for (int c = 0; c < 5; ++c) wa[c] = *(const f32x4*)&s_w[st][rw * 40 + 4 * ((2 * c + lh + ((rw >> 3) & 1)) % 10)];
        const f32x4 s31 = d[3] - d[1];
        v[0] = 2.f * (d[0] - d[2]) + s31;
        v[1] = s31 - (d[1] + d[2]);
        v[2] = 3.f * (d[1] - d[2]) + s31;
        v[3] = s31;
        v[4] = (d[4] - d[2]) - 2.f * s31;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][e], v[c][e], acc[c], 0, 0, 0);
        __syncthreads();
    }
    float s = 0;
    for (int c = 0; c < 5; ++c)
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

// pre-transformed activations (the producing layer wrote V), weights straight to registers, 64 tiles x 128 channels per workgroup:
// a wave = 2 tile blocks x 32 channels (10 accumulators = 160 VGPRs, two workgroups per CU): per chunk 10 ds_read_b128 (V of both tile
// blocks), 5 weight loads, 40 MFMAs, and only the 64 tiles' V chunk staged (2.5 loads + 2.5 ds_write_b128 per thread)
template <bool BARRIER, bool TRANSFORM = false>
__global__ __launch_bounds__(256, 2) void vdirect_kernel(const float* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s_a[2][ROWS * LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lj = lane & 31, lh = lane >> 5;
    for (int i = t; i < 2 * ROWS * LD; i += 256) (&s_a[0][0])[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    f32x16 acc[2][5];
    for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 5; ++c)
            for (int e = 0; e < 16; ++e) acc[b][c][e] = 0.f;
    const float* g = src + (size_t)blockIdx.x * 4096 + t * 4;
    const float* gw = src + (size_t)wave * 65536 + lane * 4;
    f32x4 r[3], wa[2][5];
    for (int j = 0; j < 3; ++j) r[j] = *(const f32x4*)(g + j * 1024);
    for (int c = 0; c < 5; ++c) wa[0][c] = *(const f32x4*)(gw + c * 256);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 rn[3];
            for (int j = 0; j < 3; ++j) rn[j] = *(const f32x4*)(g + ((it + h + 1) & 7) * 8192 + j * 1024);
#pragma unroll
            for (int c = 0; c < 5; ++c) wa[h ^ 1][c] = *(const f32x4*)(gw + ((it + h + 1) & 31) * 1280 + c * 256);
            f32x4 v[2][5];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 5; ++c) v[b][c] = *(const f32x4*)&s_a[h][(b * 32 + lj) * LD + lh * 4 + c * 8];
            if constexpr (TRANSFORM) {                    // natural rows staged: the F(3,3) input transform stays in the loop
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f32x4 d0 = v[b][0], d1 = v[b][1], d2 = v[b][2], d3 = v[b][3], d4 = v[b][4];
                    const f32x4 s31 = d3 - d1;
                    v[b][0] = 2.f * (d0 - d2) + s31;
                    v[b][1] = s31 - (d1 + d2);
                    v[b][2] = 3.f * (d1 - d2) + s31;
                    v[b][3] = s31;
                    v[b][4] = (d4 - d2) - 2.f * s31;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[b][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[h][c][e], v[b][c][e], acc[b][c], 0, 0, 0);
            *(f32x4*)&s_a[h ^ 1][(t % (ROWS * LD / 4)) * 4] = r[0];
            *(f32x4*)&s_a[h ^ 1][((t + 256) % (ROWS * LD / 4)) * 4] = r[1];
            if (t < 128) *(f32x4*)&s_a[h ^ 1][((t + 512) % (ROWS * LD / 4)) * 4] = r[2];
            for (int j = 0; j < 3; ++j) r[j] = rn[j];
            if constexpr (BARRIER) __syncthreads();
        }
    }
    float s = 0;
    for (int b = 0; b < 2; ++b)
        for (int c = 0; c < 5; ++c)
            for (int e = 0; e < 16; ++e) s += acc[b][c][e];
    out[(size_t)blockIdx.x * 256 + t] = s + r[0][0];
}

template <bool TRANSFORM>
void run_vdirect(int cus, const float* src, float* out) {
    printf("%-64s", TRANSFORM ? "natural rows + weights to registers, 64 x 128 per workgroup" : "V from the producer + weights to registers, 64 x 128 per workgroup");
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int wgs = cus * wpc, iters = 2000;
        const size_t dyn = wpc == 2 ? 40000 : 90000;
        hipFuncSetAttribute((const void*)vdirect_kernel<true, TRANSFORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        hipLaunchKernelGGL((vdirect_kernel<true, TRANSFORM>), dim3(wgs), dim3(256), dyn, 0, src, out, 50);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((vdirect_kernel<true, TRANSFORM>), dim3(wgs), dim3(256), dyn, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / ((double)iters * 40 * wpc);
        printf("  %d wg/cu: %6.2f ns/MFMA (%4.1f %%)", wpc, per, 100.0 * 26.67 / per);
    }
    printf("\n");
}

template <int DMAS>
void run_dma(const char* what, int cus, const float* src, float* out) {
    printf("%-64s", what);
    for (int wpc = 1; wpc <= 3; ++wpc) {
        const int wgs = cus * wpc, iters = 4000;
        hipFuncSetAttribute((const void*)dma_kernel<DMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        const size_t dyn = wpc == 3 ? 0 : (wpc == 2 ? 30000 : 60000);
        hipLaunchKernelGGL((dma_kernel<DMAS>), dim3(wgs), dim3(256), dyn, 0, src, out, 50);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((dma_kernel<DMAS>), dim3(wgs), dim3(256), dyn, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / ((double)iters * 20 * wpc);
        printf("  %d wg/cu: %6.2f ns/MFMA (%4.1f %%)", wpc, per, 100.0 * 26.67 / per);
    }
    printf("\n");
}

template <int READS, int NV, bool BARRIER, int LOADS, int WRITES>
void run(const char* what, int cus, const float* src, float* out) {
    printf("%-64s", what);
    for (int wpc = 1; wpc <= 3; ++wpc) {
        const int wgs = cus * wpc, iters = 4000;
        hipFuncSetAttribute((const void*)loop_kernel<READS, NV, BARRIER, LOADS, WRITES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        const size_t dyn = wpc == 3 ? 0 : (wpc == 2 ? 30000 : 60000);      // LDS padding pins the workgroups per CU
        hipLaunchKernelGGL((loop_kernel<READS, NV, BARRIER, LOADS, WRITES>), dim3(wgs), dim3(256), dyn, 0, src, out, 50);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<READS, NV, BARRIER, LOADS, WRITES>), dim3(wgs), dim3(256), dyn, 0, src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / ((double)iters * 20 * wpc);
        printf("  %d wg/cu: %6.2f ns/MFMA (%4.1f %%)", wpc, per, 100.0 * 26.67 / per);
    }
    printf("\n");
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *src, *out;
    hipMalloc(&src, (size_t)cus * 3 * 4096 * 4 + 8 * 8192 * 4 + (1 << 20));
    hipMemset(src, 0, (size_t)cus * 3 * 4096 * 4 + 8 * 8192 * 4 + (1 << 20));
    hipMalloc(&out, (size_t)cus * 3 * 256 * 4);
    printf("CUs %d; %% = 64 cycles at 2.4 GHz (26.67 ns) / measured ns per MFMA per SIMD\n", cus);
    run<0, 0, false, 0, 0>("bare: 20 MFMA per iteration, operands in registers", cus, src, out);
    run<5, 0, false, 0, 0>("+ 5 ds_read_b128 (B operands)", cus, src, out);
    run<10, 0, false, 0, 0>("+ 10 ds_read_b128 (A and B operands)", cus, src, out);
    run<10, 1, false, 0, 0>("+ 10 ds_read_b128 + F(3,3) input transform (36 VALU)", cus, src, out);
    run<10, 1, true, 0, 0>("+ reads + transform + barrier", cus, src, out);
    run<10, 1, true, 0, 6>("+ reads + transform + barrier + 6 ds_write_b128", cus, src, out);
    run<10, 1, true, 6, 0>("+ reads + transform + barrier + 6 global b128 loads", cus, src, out);
    run<10, 1, true, 6, 6>("+ reads + transform + barrier + 6 loads + 6 ds_write_b128 (the kernel's loop)", cus, src, out);
    run<10, 0, true, 6, 6>("the kernel's loop without the transform", cus, src, out);
    run<5, 1, true, 7, 2>("weights straight to registers: 5 reads, 7 loads, 2 writes", cus, src, out);
    run_vdirect<false>(cus, src, out);
    run_vdirect<true>(cus, src, out);
    run_dma<5>("LDS-DMA staging: reads + transform + barrier + 5 buffer_load ... lds", cus, src, out);
    run_dma<4>("LDS-DMA staging with 4 pieces per wave", cus, src, out);
    return 0;
}
