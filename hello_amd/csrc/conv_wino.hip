// Conv1d, kernel 3 / stride 1 / padding 1, in Winograd F(2,3) form on the gfx950 FP32 matrix cores.
//
// Two neighbouring outputs of a row share four inputs d0..d3 (positions 2p-1 .. 2p+2):
//     V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
//     Mc = Uc . Vc over the input channels, c = 0..3, with U = G g precomputed on the host
//     y(2p) = M0 + M1 + M2,  y(2p+1) = M1 - M2 - M3
// i.e. 4 contractions per pair of positions instead of 6 (1.5x fewer MFMAs; 1.35x at the 9-position layers,
// whose fifth pair is half empty).  Same fp32 arithmetic; results differ from the direct form by float
// re-association only.  Used for the allele- / site-level residual convolutions (compressor, experts, meta,
// combiners) and for the layer-by-layer read convolvers; everything else stays in conv_generic.hip.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32): workgroup = 4 waves = 64 PAIRS of positions x 64 channels; a wave
// owns 32 pairs x 32 channels x 4 Winograd components (4 accumulator tiles; three waves per SIMD).  The gather is that of a
// kernel-4 / stride-2 / padding-1 convolution whose K index is ordered (channel group, tap, channel): a chunk
// of KC floats per pair holds taps d0..d3 of KC/4 channels, so a lane reads its four taps with four
// ds_read_b128, forms V in registers (8 packed VALU operations per 16 MFMAs) and never stores V.  The weights
// are packed in the same order: [cout][channel group][component][KC/4 channels].
#include "kernels.h"

namespace hello {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TPW = 1;          // 32-pair tiles per wave
constexpr int BMP = 64 * TPW;   // pairs per workgroup
constexpr int BN = 64;     // channels per workgroup
}  // namespace

template <int KC>
__global__ __launch_bounds__(256, 3) void conv1d_wino_kernel(ConvArgs a) {
    constexpr int CPC = KC / 4;               // input channels per chunk
    constexpr int LD = KC + 4;                // LDS row stride (floats): ds_read_b128 of 16 consecutive rows is conflict-free
    constexpr int QPR = KC / 4;               // float4 per row of a chunk
    constexpr int RPP = 256 / QPR;            // rows covered per pass of the 256 threads
    constexpr int NA = BMP / RPP;             // activation float4 per thread per chunk
    constexpr int NWV = BN / RPP;             // weight float4 per thread per chunk
    __shared__ __attribute__((aligned(16))) float s_act[BMP * LD];
    __shared__ __attribute__((aligned(16))) float s_w[BN * LD];

    const int t = threadIdx.x;
    const int kq = t % QPR, lrow = t / QPR;
    const int L = a.lin;                       // == a.lout
    const int PP = (L + 1) / 2;                // pairs per row
    const long long items = a.m_total / L;
    const long long mp_total = items * PP;
    // One-dimensional grid, XCD-aware: workgroup ids round-robin over the 8 XCDs (each with its own L2), so
    // the cout/64 channel blocks of one tile of pairs take ids 8 apart: same XCD, back to back -> the
    // activation tile is fetched into that L2 once.
    const int gy = a.cout / BN;
    const long long id = blockIdx.x;
    const long long slot = id >> 3;
    const long long mtile = (slot / gy) * 8 + (id & 7);
    const long long m0 = mtile * BMP;
    const int cb0 = (int)(slot % gy) * BN;
    if (m0 >= mp_total) return;
    const float* src = (const float*)a.src;

    long long row_base[NA];
    int pos_base[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const long long mg = m0 + lrow + RPP * j;
        if (mg < mp_total) {
            const long long item = mg / PP;
            const int p = (int)(mg - item * PP);
            row_base[j] = item * L;
            pos_base[j] = 2 * p - 1;
        } else {
            row_base[j] = 0;
            pos_base[j] = -(1 << 28);
        }
    }

    f32x4 ra[NA], rw[NWV];
    const int tap = kq / (CPC / 4), csub = (kq % (CPC / 4)) * 4;
    auto prefetch = [&](int kb) {
        const int c = kb * CPC + csub;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int pos = pos_base[j] + tap;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos >= 0 && pos < L) v = *(const f32x4*)(src + ((row_base[j] + pos) * a.cin + c));
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NWV; ++j)
            rw[j] = *(const f32x4*)(a.w + (long long)(cb0 + lrow + RPP * j) * a.kpad + kb * KC + kq * 4);
    };

    const int wave = t >> 6, lane = t & 63;
    const int lj = lane & 31, lh = lane >> 5;
    const int wn = wave & 1;                   // 32-channel block of this wave
    const int ptile0 = (wave >> 1) * 32 * TPW;       // first pair of this wave
    f32x16 acc[4][TPW];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][tp][r] = 0.f;

    const int nchunks = a.cin / CPC;
    prefetch(0);
    for (int kb = 0; kb < nchunks; ++kb) {
        __syncthreads();   // previous chunk's operand reads are done
#pragma unroll
        for (int j = 0; j < NA; ++j) *(f32x4*)&s_act[(lrow + RPP * j) * LD + kq * 4] = ra[j];
#pragma unroll
        for (int j = 0; j < NWV; ++j) *(f32x4*)&s_w[(lrow + RPP * j) * LD + kq * 4] = rw[j];
        __syncthreads();
        if (kb + 1 < nchunks) prefetch(kb + 1);
#pragma unroll
        for (int g = 0; g < CPC / 8; ++g) {
            f32x4 wa[4];
#pragma unroll
            // the host packs the taps per 8-channel group ([cin/8][4 components][8]) whatever the chunk size is
            for (int c = 0; c < 4; ++c) wa[c] = *(const f32x4*)&s_w[(wn * 32 + lj) * LD + (g * 4 + c) * 8 + lh * 4];
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
                const float* row = &s_act[(ptile0 + tp * 32 + lj) * LD + g * 8 + lh * 4];
                const f32x4 d0 = *(const f32x4*)(row), d1 = *(const f32x4*)(row + CPC);
                const f32x4 d2 = *(const f32x4*)(row + 2 * CPC), d3 = *(const f32x4*)(row + 3 * CPC);
                f32x4 v[4];
                v[0] = d0 - d2;
                v[1] = d1 + d2;
                v[2] = d2 - d1;
                v[3] = d1 - d3;
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[c][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][e], v[c][e], acc[c][tp], 0, 0, 0);
            }
        }
    }

    // epilogue: accumulator register r of lane (lj, lh) is channel (r&3) + 8*(r>>2) + 4*lh, pair lj.
    // The residual rows of a tile are all requested before the first is used (their latency overlaps the
    // output transform instead of being paid once per 16-byte piece).
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const long long mg = m0 + ptile0 + tp * 32 + lj;
        if (mg >= mp_total) continue;
        const long long item = mg / PP;
        const int p = (int)(mg - item * PP);
        const bool second = 2 * p + 1 < L;
        const int ch0 = cb0 + wn * 32 + 4 * lh;
        const long long o = (item * L + 2 * p) * a.cout + ch0;
        f32x4 r0[4], r1[4];
        if (a.res) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r0[q] = *(const f32x4*)(a.res + o + 8 * q);
                r1[q] = second ? *(const f32x4*)(a.res + o + a.cout + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *(const f32x4*)(a.bias + ch0 + 8 * q);
            f32x4 y0, y1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m0v = acc[0][tp][4 * q + e], m1v = acc[1][tp][4 * q + e];
                const float m2v = acc[2][tp][4 * q + e], m3v = acc[3][tp][4 * q + e];
                const float x0 = ((m0v + m1v) + m2v) + b4[e], x1 = ((m1v - m2v) - m3v) + b4[e];
                y0[e] = a.relu == 1 ? fmaxf(x0, 0.f) : (a.relu == 2 ? (x0 > 20.f ? x0 : __logf(1.f + __expf(x0))) : x0);
                y1[e] = a.relu == 1 ? fmaxf(x1, 0.f) : (a.relu == 2 ? (x1 > 20.f ? x1 : __logf(1.f + __expf(x1))) : x1);
            }
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y0[e] += r0[q][e];
                    y1[e] += r1[q][e];
                }
            }
            *(f32x4*)(a.dst + o + 8 * q) = y0;
            if (second) *(f32x4*)(a.dst + o + a.cout + 8 * q) = y1;
        }
    }
}

bool conv1d_wino_supported(const ConvArgs& a) {
    return a.k == 3 && a.stride == 1 && a.pad == 1 && a.lin == a.lout && !a.src_u8 && (a.cin % 8) == 0 &&
           (a.cout % BN) == 0;
}

hipError_t launch_conv1d_wino(const ConvArgs& a, hipStream_t stream) {
    if (a.m_total <= 0) return hipSuccess;
    if (!conv1d_wino_supported(a) || a.kpad != 4 * a.cin) return hipErrorInvalidValue;
    const long long pairs = (a.m_total / a.lin) * ((a.lin + 1) / 2);
    const long long mtiles8 = ((pairs + BMP - 1) / BMP + 7) / 8 * 8;       // tiles of pairs, a multiple of the XCD count
    const dim3 grid((unsigned)(mtiles8 * (a.cout / BN)));
    // measured on the allele stage (8 192 sites): 64 pairs x 64 channels per workgroup at three waves per SIMD
    // (4 accumulator tiles per wave, 132 VGPRs) 3.41 ms; 128 pairs (8 tiles, two waves per SIMD) 3.60 ms with
    // 16-channel chunks and 3.84 ms with 8-channel chunks; 8 waves x 128 channels with double-buffered LDS 4.26 ms
    hipLaunchKernelGGL((conv1d_wino_kernel<32>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hello
