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
constexpr int BMP = 64;    // Winograd tiles (pairs or triples of positions) per workgroup
constexpr int BN = 64;     // channels per workgroup
constexpr int SMALL_LAUNCH_DIVISOR = 4;   // the small kernel serves launches of at most (CUs / this) 64 x 64 workgroups
}  // namespace

// M = outputs per Winograd tile: 2 -> F(2,3), 4 components; 3 -> F(3,3), 5 components (points 0, 1, -1, 2, inf):
//     V0 = 2(d0-d2) + (d3-d1), V1 = (d3-d1) - (d1+d2), V2 = (d3-d1) + 3(d1-d2), V3 = d3-d1, V4 = (d4-d2) - 2(d3-d1)
//     y(3p) = M0+M1+M2+M3,  y(3p+1) = M1 - M2 + 2 M3,  y(3p+2) = M1 + M2 + 4 M3 + M4
// with U = (g0/2, -(g0+g1+g2)/2, (-g0+g1-g2)/6, (g0+2g1+4g2)/6, g2) from the host: 5 contractions per 3
// positions instead of 9 (1.8x fewer MFMAs; F(2,3) needs 20 for a 9-position row, this 15).  Chosen by the
// launcher whenever the row length is a multiple of 3 (9, 18, 36, 150), and the weights are packed to match.
// ACT: 0 none | 1 ReLU | 2 Softplus; RES: a residual input is added after the activation.
// TWO: the input is the channel-wise concatenation of two tensors that is never materialised (a CONCAT op folded into
// this convolution): chunks below a.split come from a.src (rows of a.split channels), the others from a.src2.
template <int M, int ACT, bool RES, bool TWO = false>
__global__ __launch_bounds__(256, 3) void conv1d_wino_kernel(ConvArgs a) {
    constexpr int NT = M + 2;                 // taps per tile == Winograd components
    constexpr int CPC = 8;                    // input channels per chunk
    constexpr int KC = NT * CPC;              // floats per chunk row: [tap][channel]
    constexpr int LD = KC + 4;                // LDS row stride (floats): ds_read_b128 of 16 consecutive rows is conflict-free
    constexpr int QPR = KC / 4;               // float4 per row of a chunk
    constexpr int NQ = (BMP * QPR + 255) / 256;   // float4 per thread per chunk (activations; same for the weights)
    constexpr int ROWS = (NQ * 256 + QPR - 1) / QPR;   // staged rows incl. the overhang of the last pass (never read)
    static_assert(BMP == BN, "one staging loop shape for both operands");
    __shared__ __attribute__((aligned(16))) float s_act[ROWS * LD];
    __shared__ __attribute__((aligned(16))) float s_w[ROWS * LD];

    const int t = threadIdx.x;
    const int L = a.lin;                       // == a.lout
    const int PP = (L + M - 1) / M;            // tiles per row
    const unsigned items = a.wino_rows;       // the launcher admits fewer than 2^31 tiles: 32-bit tile arithmetic
    const unsigned mp_total = items * (unsigned)PP;
    // One-dimensional grid, XCD-aware: workgroup ids round-robin over the 8 XCDs (each with its own L2), so
    // the cout/64 channel blocks of one tile of pairs take ids 8 apart: same XCD, back to back -> the
    // activation tile is fetched into that L2 once.
    const unsigned gy = (unsigned)(a.cout / BN);
    const unsigned id = blockIdx.x;
    const unsigned slot = id >> 3, srow = slot / gy;
    const unsigned m0 = (srow * 8 + (id & 7)) * BMP;
    const int cb0 = (int)(slot - srow * gy) * BN;
    if (m0 >= mp_total) return;
    // Operand staging goes through buffer descriptors (32-bit per-lane byte offsets, the chunk offset in an
    // SGPR, out-of-range lanes read zeros): the 64 tiles of a workgroup span few rows, so the activation
    // descriptor starts at the first of them and an offset of 2 GiB marks "zero padding / row past the tile".
    const unsigned item0 = m0 / (unsigned)PP;
    const int stride0 = TWO ? a.split : a.cin_stride, stride1 = TWO ? a.cin - a.split : 0;     // channels per position of the source(s)
    const long long left = (long long)(items - item0) * L * stride0 * 4;
    const int goff = a.groups > 1 ? (cb0 / (a.cout / a.groups)) * a.cin : 0;     // first input channel of this block's group
    // A workgroup's tiles are numbered from its first row's first tile: x = p0 + r with p0 < PP and r < 80, so
    // x / PP is one 32-bit multiply-high by a wave-uniform reciprocal (exact for x * PP < 2^32) instead of a
    // 64-bit division per lane.
    const unsigned p0 = m0 - item0 * (unsigned)PP;
    const unsigned tiles_left = mp_total - m0;
    const unsigned magic = 0xffffffffu / (unsigned)PP + 1u;
    const __amdgpu_buffer_rsrc_t act_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const float*)a.src + (long long)item0 * L * stride0), 0, (int)(left < 0x7fffffffLL ? left : 0x7fffffffLL), 0x00020000);
    const long long left1 = (long long)(items - item0) * L * stride1 * 4;
    const __amdgpu_buffer_rsrc_t act_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(TWO ? a.src2 + (long long)item0 * L * stride1 : (const float*)a.src), 0, (int)(left1 < 0x7fffffffLL ? left1 : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.w + (long long)cb0 * a.kpad), 0, BN * a.kpad * 4, 0x00020000);

    // float4 number i = t + 256 j of a chunk is (row i / QPR, tap (i % QPR) / 2, channel half i & 1); the last pass
    // overhangs the tile (rows >= 64): those lanes load zeros into LDS rows nobody reads
    unsigned act_off[NQ], act_off1[NQ], w_off[NQ];
    int lds_off[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int i = t + 256 * j;
        const int row = i / QPR, q = i - row * QPR;
        const int tap = q >> 1, csub = (q & 1) * 4;
        lds_off[j] = row * LD + q * 4;
        w_off[j] = (unsigned)(row * a.kpad + q * 4) * 4u;
        act_off[j] = act_off1[j] = 0x80000000u;
        if (row < BMP && (unsigned)row < tiles_left) {
            const unsigned x = p0 + (unsigned)row;
            const unsigned irow = PP == 1 ? x : __umulhi(x, magic);       // rows past the workgroup's first
            const int pos = M * (int)(x - irow * (unsigned)PP) - 1 + tap;
            if (pos >= 0 && pos < L) {
                act_off[j] = (unsigned)(((int)irow * L + pos) * stride0 + goff + csub) * 4u;
                if constexpr (TWO) act_off1[j] = (unsigned)(((int)irow * L + pos) * stride1 + csub) * 4u;
            }
        }
    }

    f32x4 ra[NQ], rw[NQ];
    auto prefetch = [&](int kb) {
        const bool second = TWO && kb * CPC >= a.split;                      // wave-uniform: which source holds this chunk's channels
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (second)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(act_rsrc1, act_off1[j], (kb * CPC - a.split) * 4, 0));
            else
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(act_rsrc, act_off[j], kb * CPC * 4, 0));
            rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[j], kb * KC * 4, 0));
        }
    };

    const int wave = t >> 6, lane = t & 63;
    const int lj = lane & 31, lh = lane >> 5;
    const int wn = wave & 1;                   // 32-channel block of this wave
    const int ptile0 = (wave >> 1) * 32;       // first tile of this wave
    // accumulator register r of lane (lj, lh) is channel (r&3) + 8*(r>>2) + 4*lh of the wave's block, tile lj.
    // The bias rides in component 1, which enters every output of a tile with coefficient 1.
    const int ch0 = cb0 + wn * 32 + 4 * lh;
    f32x16 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *(const f32x4*)(a.bias + ch0 + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][4 * q + e] = b4[e];
    }

    const int nchunks = a.cin / CPC;
    prefetch(0);
    for (int kb = 0; kb < nchunks; ++kb) {
        __syncthreads();   // previous chunk's operand reads are done
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            *(f32x4*)&s_act[lds_off[j]] = ra[j];
            *(f32x4*)&s_w[lds_off[j]] = rw[j];
        }
        __syncthreads();
        if (kb + 1 < nchunks) prefetch(kb + 1);
        f32x4 wa[NT], d[NT], v[NT];
        // the host packs the taps per 8-channel group ([cin/8][components][8])
#pragma unroll
        for (int c = 0; c < NT; ++c) wa[c] = *(const f32x4*)&s_w[(wn * 32 + lj) * LD + c * CPC + lh * 4];
        const float* row = &s_act[(ptile0 + lj) * LD + lh * 4];
#pragma unroll
        for (int c = 0; c < NT; ++c) d[c] = *(const f32x4*)(row + c * CPC);
        if constexpr (M == 2) {
            v[0] = d[0] - d[2];
            v[1] = d[1] + d[2];
            v[2] = d[2] - d[1];
            v[3] = d[1] - d[3];
        } else {
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
            for (int c = 0; c < NT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[c][e], v[c][e], acc[c], 0, 0, 0);
    }

    // epilogue: accumulator register r of lane (lj, lh) is channel (r&3) + 8*(r>>2) + 4*lh, tile lj.
    // The residual rows of a tile are all requested before the first is used (their latency overlaps the
    // output transform instead of being paid once per 16-byte piece).
    if ((unsigned)(ptile0 + lj) >= tiles_left) return;
    const unsigned local = p0 + (unsigned)(ptile0 + lj);       // tiles past the first row's start: small
    const unsigned irow = PP == 1 ? local : __umulhi(local, magic);
    const int p = (int)(local - irow * (unsigned)PP);
    bool live[M];                              // F(3,3) rows are whole triples
#pragma unroll
    for (int u = 0; u < M; ++u) live[u] = M == 3 || M * p + u < L;
    const long long o = ((long long)(item0 + irow) * L + M * p) * a.cout + ch0;
    f32x4 res[M][4];
    if constexpr (RES) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < M; ++u)
                res[u][q] = live[u] ? *(const f32x4*)(a.res + o + u * a.cout + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 y[M];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x[M];
            if constexpr (M == 2) {
                const float m0v = acc[0][4 * q + e], m1v = acc[1][4 * q + e];
                const float m2v = acc[2][4 * q + e], m3v = acc[3][4 * q + e];
                x[0] = (m0v + m1v) + m2v;
                x[1] = (m1v - m2v) - m3v;
            } else {
                const float m0v = acc[0][4 * q + e], m1v = acc[1][4 * q + e], m2v = acc[2][4 * q + e];
                const float m3v = acc[3][4 * q + e], m4v = acc[4][4 * q + e];
                const float sum = m1v + m2v, dif = m1v - m2v;
                x[0] = (m0v + sum) + m3v;
                x[1] = __builtin_fmaf(m3v, 2.f, dif);
                x[2] = __builtin_fmaf(m3v, 4.f, sum) + m4v;
            }
#pragma unroll
            for (int u = 0; u < M; ++u)
                y[u][e] = ACT == 1 ? fmaxf(x[u], 0.f)
                                   : (ACT == 2 ? (x[u] > 20.f ? x[u] : __logf(1.f + __expf(x[u]))) : x[u]);
        }
#pragma unroll
        for (int u = 0; u < M; ++u) {
            if constexpr (RES) y[u] += res[u][q];
            if (live[u]) *(f32x4*)(a.dst + o + u * a.cout + 8 * q) = y[u];
        }
    }
}

// ---- the same convolution for SMALL launches --------------------------------------------------------------------
// When the tiles of a launch make fewer 64 x 64 workgroups than a quarter of the CUs, what the launch waits for is one
// wave's chain: a 32 x 32 block over K = NT x cin is cin/8 chunks of NT x 4 MFMAs of 64 cycles behind two barriers each
// (36 us at 256 channels, whatever the batch).  Here a WORKGROUP owns 16 tiles x 16 channels (v_mfma_f32_16x16x4_f32)
// and its four waves split the input channels: a sixteenth of the chain, no staging and one barrier -- lane (j, q)
// loads the float4 of channels 16 m + 4 q .. + 3 of its tile's NT taps and of its output channel's NT transformed taps
// straight from global memory (the weights in the layout the large kernel packs: a 16-channel group is two 8-channel
// groups), two groups ahead; one b128 feeds four MFMAs per component.  D[channel][tile]: lane (j, q) ends with
// channels 4 q .. 4 q + 3 of tile j; waves 1..3 hand their accumulators to wave 0 through LDS, added in wave order.
template <int M, int ACT, bool RES, bool TWO = false>
__global__ __launch_bounds__(256) void conv1d_wino_small_kernel(ConvArgs a) {
    constexpr int NT = M + 2;
    __shared__ __attribute__((aligned(16))) f32x4 s_red[3][NT][64];
    const int L = a.lin, PP = (L + M - 1) / M;
    const unsigned gy = (unsigned)(a.cout / 16);
    const unsigned tb = blockIdx.x / gy, cy = blockIdx.x - tb * gy;          // neighbours share their 16 tiles (L2)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int cb = (int)cy * 16;                                             // this workgroup's 16 output channels
    const unsigned mp_total = a.wino_rows * (unsigned)PP;
    const unsigned T = tb * 16u + (unsigned)j;
    const bool live = T < mp_total;
    const unsigned item = live ? T / (unsigned)PP : 0u;
    const int p = live ? (int)(T - item * (unsigned)PP) : 0;
    // activations through a buffer descriptor: taps outside the row (and dead tiles) read zeros
    const __amdgpu_buffer_rsrc_t act_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((long long)a.wino_rows * L * (TWO ? a.split : a.cin_stride) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t act_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(TWO ? (const void*)a.src2 : a.src), 0, (int)((long long)a.wino_rows * L * (TWO ? a.cin - a.split : 0) * 4), 0x00020000);
    const int stride0 = TWO ? a.split : a.cin_stride, stride1 = TWO ? a.cin - a.split : 0;
    const int goff = a.groups > 1 ? (cb / (a.cout / a.groups)) * a.cin : 0;
    unsigned act_off[NT], act_off1[NT];
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int pos = M * p - 1 + tap;
        const bool inside = live && pos >= 0 && pos < L;
        act_off[tap] = inside ? (unsigned)((((int)item * L + pos) * stride0 + goff + 4 * q) * 4) : 0x80000000u;
        act_off1[tap] = (TWO && inside) ? (unsigned)((((int)item * L + pos) * stride1 + 4 * q) * 4) : 0x80000000u;
    }
    const float* wrow = a.w + (long long)(cb + j) * a.kpad + (q >> 1) * (NT * 8) + 4 * (q & 1);
    const int groups = a.cin / 16, per_wave = (groups + 3) / 4;
    const int g0 = wave * per_wave, g1 = g0 + per_wave < groups ? g0 + per_wave : groups;   // this wave's input groups

    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave == 0) acc[1] = *(const f32x4*)(a.bias + cb + 4 * q);            // the bias rides in component 1

    f32x4 d[3][NT], w[3][NT];                                                // operands of groups m, m + 1, m + 2
    auto request = [&](int m, f32x4 (&dd)[NT], f32x4 (&ww)[NT]) {
        const bool second = TWO && m * 16 >= a.split;                        // wave-uniform (split is a multiple of 16)
#pragma unroll
        for (int tap = 0; tap < NT; ++tap)
            dd[tap] = second ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(act_rsrc1, act_off1[tap], (m * 16 - a.split) * 4, 0))
                             : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(act_rsrc, act_off[tap], m * 64, 0));
#pragma unroll
        for (int c = 0; c < NT; ++c) ww[c] = *(const f32x4*)(wrow + (long long)m * (2 * NT * 8) + c * 8);
    };
    if (g0 < g1) request(g0, d[0], w[0]);
    if (g0 + 1 < g1) request(g0 + 1, d[1], w[1]);
    for (int m = g0; m < g1; m += 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {                                        // group m + r sits in slot r
            if (m + r < g1) {
                if (m + r + 2 < g1) request(m + r + 2, d[(r + 2) % 3], w[(r + 2) % 3]);
                const f32x4(&x)[NT] = d[r];
                f32x4 v[NT];
                if constexpr (M == 2) {
                    v[0] = x[0] - x[2];
                    v[1] = x[1] + x[2];
                    v[2] = x[2] - x[1];
                    v[3] = x[1] - x[3];
                } else {
                    const f32x4 s31 = x[3] - x[1];
                    v[0] = 2.f * (x[0] - x[2]) + s31;
                    v[1] = s31 - (x[1] + x[2]);
                    v[2] = 3.f * (x[1] - x[2]) + s31;
                    v[3] = s31;
                    v[4] = (x[4] - x[2]) - 2.f * s31;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r][c][e], v[c][e], acc[c], 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < NT; ++c) s_red[wave - 1][c][lane] = acc[c];
    }
    __syncthreads();
    if (wave > 0 || !live) return;
#pragma unroll
    for (int wv = 0; wv < 3; ++wv)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] += s_red[wv][c][lane];

    f32x4 y[M];
    if constexpr (M == 2) {
        y[0] = (acc[0] + acc[1]) + acc[2];
        y[1] = (acc[1] - acc[2]) - acc[3];
    } else {
        const f32x4 sum = acc[1] + acc[2], dif = acc[1] - acc[2];
        y[0] = (acc[0] + sum) + acc[3];
        y[1] = 2.f * acc[3] + dif;
        y[2] = (4.f * acc[3] + sum) + acc[4];
    }
    const long long o = ((long long)item * L + M * p) * a.cout + cb + 4 * q;
#pragma unroll
    for (int u = 0; u < M; ++u) {
        if (M * p + u >= L) break;                                           // F(2,3) over an odd row length
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = y[u][e];
            out[e] = ACT == 1 ? fmaxf(x, 0.f) : (ACT == 2 ? (x > 20.f ? x : __logf(1.f + __expf(x))) : x);
        }
        if constexpr (RES) out += *(const f32x4*)(a.res + o + u * a.cout);
        *(f32x4*)(a.dst + o + u * a.cout) = out;
    }
}

bool conv1d_wino_supported(const ConvArgs& a) {
    return a.k == 3 && a.stride == 1 && a.pad == 1 && a.lin == a.lout && !a.src_u8 && (a.cin % 8) == 0 &&
           (a.cout % BN) == 0;
}

int conv1d_wino_outputs_per_tile(int length) { return length % 3 == 0 ? 3 : 2; }

hipError_t launch_conv1d_wino(const ConvArgs& args, hipStream_t stream) {
    ConvArgs a = args;
    if (a.m_total <= 0) return hipSuccess;
    const int m = conv1d_wino_outputs_per_tile(a.lin);
    if (!conv1d_wino_supported(a) || a.kpad != (m + 2) * a.cin) return hipErrorInvalidValue;
    const long long tiles = (a.m_total / a.lin) * ((a.lin + m - 1) / m);
    if (tiles >= (1LL << 31) - 8 * BMP) return hipErrorInvalidValue;      // the kernel's tile arithmetic is 32-bit
    a.wino_rows = (unsigned)(a.m_total / a.lin);
    const long long mtiles8 = ((tiles + BMP - 1) / BMP + 7) / 8 * 8;       // workgroup rows, a multiple of the XCD count
    const dim3 grid((unsigned)(mtiles8 * (a.cout / BN)));
    // measured on the allele stage (8 192 sites), F(2,3): 64 pairs x 64 channels per workgroup at three waves per
    // SIMD (4 accumulator tiles per wave, 132 VGPRs) 3.41 ms; 128 pairs (8 tiles, two waves per SIMD) 3.60 ms with
    // 16-channel chunks and 3.84 ms with 8-channel chunks; 8 waves x 128 channels with double-buffered LDS 4.26 ms
    const int variant = (m == 3 ? 6 : 0) + (a.relu < 0 || a.relu > 2 ? 0 : a.relu) * 2 + (a.res ? 1 : 0);
    // small launches: 16 x 16 blocks per wave, no LDS (conv1d_wino_small_kernel)
    const long long cus = device_cus();
    const long long big_wgs = (tiles + BMP - 1) / BMP * (a.cout / BN);
    if (a.src2) {
        // two-source form: ReLU, no residual, no groups, both parts whole 16-channel groups (what the compiler folds)
        if (a.relu != 1 || a.res || a.groups > 1 || a.split <= 0 || a.split >= a.cin || (a.split % 16) || ((a.cin - a.split) % 16))
            return hipErrorInvalidValue;
        if (big_wgs * SMALL_LAUNCH_DIVISOR <= cus && (long long)a.m_total * a.cin * 4 < (1LL << 31)) {
            const dim3 sgrid((unsigned)((tiles + 15) / 16 * (a.cout / 16)));
            if (m == 3) hipLaunchKernelGGL((conv1d_wino_small_kernel<3, 1, false, true>), sgrid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((conv1d_wino_small_kernel<2, 1, false, true>), sgrid, dim3(256), 0, stream, a);
        } else {
            if (m == 3) hipLaunchKernelGGL((conv1d_wino_kernel<3, 1, false, true>), grid, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((conv1d_wino_kernel<2, 1, false, true>), grid, dim3(256), 0, stream, a);
        }
        return hipGetLastError();
    }
    if (big_wgs * SMALL_LAUNCH_DIVISOR <= cus && (a.cin % 16) == 0 && (long long)a.m_total * a.cin_stride * 4 < (1LL << 31)) {
        const dim3 sgrid((unsigned)((tiles + 15) / 16 * (a.cout / 16)));
        switch (variant) {
#define HELLO_WINO_SMALL_CASE(V, MM, ACT, RES) \
    case V: hipLaunchKernelGGL((conv1d_wino_small_kernel<MM, ACT, RES>), sgrid, dim3(256), 0, stream, a); break;
            HELLO_WINO_SMALL_CASE(0, 2, 0, false) HELLO_WINO_SMALL_CASE(1, 2, 0, true) HELLO_WINO_SMALL_CASE(2, 2, 1, false)
            HELLO_WINO_SMALL_CASE(3, 2, 1, true) HELLO_WINO_SMALL_CASE(4, 2, 2, false) HELLO_WINO_SMALL_CASE(5, 2, 2, true)
            HELLO_WINO_SMALL_CASE(6, 3, 0, false) HELLO_WINO_SMALL_CASE(7, 3, 0, true) HELLO_WINO_SMALL_CASE(8, 3, 1, false)
            HELLO_WINO_SMALL_CASE(9, 3, 1, true) HELLO_WINO_SMALL_CASE(10, 3, 2, false) HELLO_WINO_SMALL_CASE(11, 3, 2, true)
#undef HELLO_WINO_SMALL_CASE
        }
        return hipGetLastError();
    }
    switch (variant) {
#define HELLO_WINO_CASE(V, MM, ACT, RES) \
    case V: hipLaunchKernelGGL((conv1d_wino_kernel<MM, ACT, RES>), grid, dim3(256), 0, stream, a); break;
        HELLO_WINO_CASE(0, 2, 0, false) HELLO_WINO_CASE(1, 2, 0, true) HELLO_WINO_CASE(2, 2, 1, false)
        HELLO_WINO_CASE(3, 2, 1, true) HELLO_WINO_CASE(4, 2, 2, false) HELLO_WINO_CASE(5, 2, 2, true)
        HELLO_WINO_CASE(6, 3, 0, false) HELLO_WINO_CASE(7, 3, 0, true) HELLO_WINO_CASE(8, 3, 1, false)
        HELLO_WINO_CASE(9, 3, 1, true) HELLO_WINO_CASE(10, 3, 2, false) HELLO_WINO_CASE(11, 3, 2, true)
#undef HELLO_WINO_CASE
    }
    return hipGetLastError();
}

}  // namespace hello
