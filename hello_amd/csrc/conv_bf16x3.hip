// Conv1d (+bias, +ReLU / Softplus, +residual) as an implicit GEMM on the gfx950 BF16 matrix cores with 3-term split
// operands -- the allele- / site-level layers of arithmetic mode "bf16x3+allele" (never the default).
//
//   x w ~= xh wh + (xh wl + xl wh),  xh = bf16(x), xl = bf16(x - xh)  (16 of fp32's 24 mantissa bits; the dropped xl wl
//   and the split residues are ~2^-17 of a product), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//
// Same GEMM orientation and tiling as conv_generic.hip (D[channel][position] = W[channel][k] X[k][position], K index
// = tap * cin + c, channels-last activations), direct form for every kernel size: a workgroup of 4 waves owns 128
// positions x 128 channels (2 x 2 waves, each 2 channel blocks x 2 position tiles of 32 x 32), K walked in chunks of 32
// through LDS.  Activations arrive as fp32 from HBM and are split by the staging threads (2 cvt_pk + 4 shifts + 4 subs +
// 2 cvt_pk per float4) into two bf16 planes; weights are split on the host (hello_amd/compiler.py pack_conv_bf16x3:
// [hi | lo][cout][kpad] bf16).  LDS rows are 32 bf16 + 8 of padding (80 bytes): the ds_read_b128 operand reads of a
// 16-lane group then touch 16 distinct bank groups.  One ds_read_b128 per operand and part feeds a whole 16-deep MFMA; the
// next chunk is prefetched global -> registers while the current one is multiplied.  Epilogue as in conv_generic.hip
// (the accumulator layout of the 32 x 32 bf16 and fp32 MFMAs is the same).
#include "kernels.h"

namespace hello {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KC = 32;
constexpr int LDB = 80;                       // bytes per LDS row: 32 bf16 + 16 bytes of padding

__device__ __forceinline__ unsigned short to_bf16(float x) {
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
}  // namespace

__global__ __launch_bounds__(256) void conv1d_bf16x3_kernel(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_act[2][BM * LDB];      // [part][position][k]
    __shared__ __attribute__((aligned(16))) unsigned char s_w[2][BN * LDB];        // [part][channel][k]
    const int t = threadIdx.x;
    const int kq = t & 7;              // which float4 (4 k values) of the chunk this thread stages
    const int lrow = t >> 3;           // 0..31: rows lrow + 32 j
    const long long m0 = (long long)blockIdx.x * BM;
    const int cb0 = blockIdx.y * BN;
    const float* src = (const float*)a.src;
    const unsigned short* wsplit = (const unsigned short*)a.w;
    const long long plane = (long long)a.cout_pad * a.kpad;       // elements between the hi and lo weight planes
    const int kreal = a.k * a.cin;

    long long row_base[4];
    int pos_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long mg = m0 + lrow + 32 * j;
        if (mg < a.m_total) {
            const long long item = mg / a.lout;
            const int p = (int)(mg - item * a.lout);
            row_base[j] = item * a.lin;
            pos_base[j] = p * a.stride - a.pad;
        } else {
            row_base[j] = 0;
            pos_base[j] = -(1 << 28);
        }
    }
    // weights: a chunk is 128 rows x 64 bytes per part = 512 16-byte pieces per part, 2 per thread and part
    const int wrow = t >> 2, wpiece = t & 3;      // rows wrow and wrow + 64, 16-byte piece wpiece (8 bf16)
    f32x4 ra[4];
    bf16x8 rw[2][2];
    auto prefetch = [&](int kbase) {
        const int kk = kbase + kq * 4;
        const int tap = kk / a.cin;                // cin % 32 == 0: a chunk never straddles taps
        const int c = kk - tap * a.cin;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pos = pos_base[j] + tap;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kk < kreal && pos >= 0 && pos < a.lin) v = *(const f32x4*)(src + ((row_base[j] + pos) * a.cin + c));
            ra[j] = v;
        }
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                rw[part][h] = *(const bf16x8*)(wsplit + part * plane + (long long)(cb0 + wrow + 64 * h) * a.kpad + kbase + wpiece * 8);
    };

    const int wave = t >> 6, lane = t & 63;
    const int lj = lane & 31, lh = lane >> 5;
    const int wn = (wave & 1) * 2;                 // first channel block of this wave
    const int ptile0 = (wave >> 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int cw = 0; cw < 2; ++cw)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cw][tp][r] = 0.f;

    prefetch(0);
    for (int kbase = 0; kbase < a.kpad; kbase += KC) {
        __syncthreads();                           // the previous chunk's operand reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned short hi = to_bf16(ra[j][e]);
                h[e] = (short)hi;
                l[e] = (short)to_bf16(ra[j][e] - __uint_as_float((unsigned)hi << 16));
            }
            *(bf16x4*)(s_act[0] + (lrow + 32 * j) * LDB + kq * 8) = h;
            *(bf16x4*)(s_act[1] + (lrow + 32 * j) * LDB + kq * 8) = l;
        }
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int h = 0; h < 2; ++h) *(bf16x8*)(s_w[part] + (wrow + 64 * h) * LDB + wpiece * 16) = rw[part][h];
        __syncthreads();
        if (kbase + KC < a.kpad) prefetch(kbase + KC);
#pragma unroll
        for (int step = 0; step < KC / 16; ++step) {
            bf16x8 wa[2][2], xb[2][2];             // [part][channel block | position tile]
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int cw = 0; cw < 2; ++cw)
                    wa[part][cw] = *(const bf16x8*)(s_w[part] + ((wn + cw) * 32 + lj) * LDB + step * 32 + lh * 16);
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
                    xb[part][tp] = *(const bf16x8*)(s_act[part] + (ptile0 + tp * 32 + lj) * LDB + step * 32 + lh * 16);
            }
#pragma unroll
            for (int cw = 0; cw < 2; ++cw)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    // the small cross terms first, the hh term last
                    acc[cw][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1][cw], xb[0][tp], acc[cw][tp], 0, 0, 0);
                    acc[cw][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][cw], xb[1][tp], acc[cw][tp], 0, 0, 0);
                    acc[cw][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0][cw], xb[0][tp], acc[cw][tp], 0, 0, 0);
                }
        }
    }

    // epilogue: accumulator register r of lane (lj, lh) is channel (r & 3) + 8 (r >> 2) + 4 lh, position lj
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        const long long mg = m0 + ptile0 + tp * 32 + lj;
        if (mg >= a.m_total) continue;
#pragma unroll
        for (int cw = 0; cw < 2; ++cw) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = cb0 + (wn + cw) * 32 + 8 * q + 4 * lh;
                if (ch >= a.cout) continue;
                const f32x4 b4 = *(const f32x4*)(a.bias + ch);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[cw][tp][4 * q + e] + b4[e];
                    v[e] = a.relu == 1 ? fmaxf(x, 0.f) : (a.relu == 2 ? (x > 20.f ? x : __logf(1.f + __expf(x))) : x);
                }
                const long long o = mg * a.cout + ch;
                if (a.res) {
                    const f32x4 r4 = *(const f32x4*)(a.res + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                }
                *(f32x4*)(a.dst + o) = v;
            }
        }
    }
}

// float input, cin a multiple of 32 (a 32-deep chunk stays inside one tap), cout a multiple of 4; weights padded to 128
// output channels and 32-deep chunks (cout_pad / kpad of the split block)
bool conv1d_bf16x3_supported(const ConvArgs& a) {
    return !a.src_u8 && a.cin % 32 == 0 && a.cout % 4 == 0 && a.kpad == a.k * a.cin && a.cout_pad % BN == 0 && a.cout_pad >= a.cout;
}

hipError_t launch_conv1d_bf16x3(const ConvArgs& a, hipStream_t stream) {
    if (a.m_total <= 0) return hipSuccess;
    if (!conv1d_bf16x3_supported(a)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.m_total + BM - 1) / BM), (unsigned)(a.cout_pad / BN));
    hipLaunchKernelGGL(conv1d_bf16x3_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hello
