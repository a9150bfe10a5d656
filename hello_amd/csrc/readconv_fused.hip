// Fused read convolver + reads->alleles segment sum (gfx950): pileup bytes in, per-allele sums out.
//
// Reference semantics: architectures/read_convolver.py:9-144 -- the stem (6|7 -> 16 -> 16 -> 32, kernel 3,
// no padding, ReLU each, MaxPool1d(3,2)), 3 x ResidualBlock(32), the strided 32->64 block with its 1x1
// shortcut, 3 (or 5: transfer-learning models) x ResidualBlock(64) -- followed by reduceSlots over the reads of
// each allele (MixtureOfExpertsAdvanced.py:23-34,163).  That is 5 076 096 MAC per read, 83 % of a 30x site.
//
// One workgroup (256 threads = 4 waves) carries G = 4 reads through all 18 convolutions with every
// activation resident in LDS; nothing but the bytes is read and nothing but per-allele partial sums is
// written.  Two workgroups share a CU (2 waves per SIMD): one workgroup's barriers, prologue and epilogues
// hide behind the other's MFMAs.
//
//   LDS     two ping-pong images, float32 channels-last [row][channel].  At 32 channels the reads of the
//           group are stacked along the row axis with ONE shared zero row between neighbours (row stride 72
//           at 71 positions), so the k=3/pad=1 convolutions need no edge handling.  At 64 channels the reads
//           are stacked with no rows between them (144 rows) and the taps that would cross a read boundary
//           are zeroed in registers.  16-byte chunks of a row are XOR-swizzled with the row index so that a
//           ds_read_b128 of 16 lanes hits 16 distinct bank groups: SW_OLD for images walked one row per lane,
//           SW_W for images walked two rows per lane (Winograd layers, stride-2 convs); see img_off.
//   MFMA    v_mfma_f32_16x16x4_f32 (exact fp32), D[channel][position] = W[channel][k] X[k][position]:
//           a wave owns one 16-channel block; lane (j, q) of a tile ends with channels 4q..4q+3 of column j
//           -> bias / ReLU / residual / store are float4.  k is ordered so that lane-quarter q supplies
//           channels 16m+4q+t at step (tap, m, t) for BOTH operands: one ds_read_b128 feeds four MFMAs.
//   form    WINO (default): the k=3 / stride-1 / pad-1 convolutions run in Winograd form.  The residual trunk
//           (13 of the 18 convs, 94 % of the MACs outside the stem) in F(3,3) -- 5 instead of 9 channel
//           contractions per 3 positions, whole tiles of 16 triples: wino3_layer -- where the geometry is
//           whole triples (150 bp), else in F(2,3) (4 instead of 6 per pair of positions: wino_layer, which
//           also runs the stem's conv2 and conv3 + pool).  Everything else, and the whole kernel with
//           WINO = false, runs the direct form: conv_layer (tile pairs, two accumulation chains per tile,
//           operands two steps ahead, deferred epilogue).
//   weights each wave keeps ONLY its own 16-channel slice of ONE layer in registers (<= 64 VGPRs; F(3,3)
//           layers: of one 16-channel input group, 20 VGPRs + the next group's), loaded straight from L2 in
//           lane order (pre-packed by hello_amd/readconv_pack.py); the next slice is rolled in behind it.
//   stem    runs first in the same kernel over the stacked reads (natural stride 150 rows): conv1 reads the
//           bytes themselves (channels-last bytes: the im2col index k = tap*C + c is the byte offset from
//           the row start), conv2 runs over the stacked rows as one sequence (rows whose window straddles
//           two reads are garbage nothing valid consumes), conv3 + ReLU + MaxPool: stem_conv3_pool.
//   sum     reads of an allele are contiguous, so the group adds its reads per allele in order and
//           writes one partial [36][64] slot per (group, allele) incidence; a tiny finalize kernel adds
//           an allele's slots in order.  No atomics: results are bit-reproducible.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace hello {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace rc {

// packed weight block (floats): per conv [COUT/16][KT][CIN/16][64 lanes][4], then bias [COUT].  With WINO the
// convolutions of the identity-shortcut residual blocks are stored in their Winograd F(2,3) form: KT = 4
// transformed taps U = G g (computed on the host in float64) instead of 3.
constexpr int S1_STEPS = 6;                                // stem conv1: K = 3*C <= 21 -> 6 MFMA steps of 4
// With F33 the k3/s1 convolutions of the residual trunk (the blocks' convs and the strided block's second conv)
// are stored in F(3,3) form: 5 transformed taps, ordered [COUT/16][CIN/16][5][64 lanes][4] (wino3_layer walks the
// input groups in its outer loop).
template <bool WINO, bool F33 = false>
struct Offs {
    static constexpr int KT = WINO ? 4 : 3;
    static constexpr int W3232 = 2 * KT * 2 * 256, W3264 = 4 * 3 * 2 * 256, W3264S = 4 * 1 * 2 * 256;
    static constexpr int W6464 = 4 * KT * 4 * 256;
    static constexpr int W6464D = F33 ? 4 * 5 * 4 * 256 : W6464;      // a convolution of the 64-channel residual blocks
    static constexpr int W3232D = F33 ? 2 * 5 * 2 * 256 : W3232;      // ... of the 32-channel residual blocks
    static constexpr int OFF_B = 0;                                   // 6 convs 32->32 (residual blocks)
    static constexpr int OFF_C1 = OFF_B + 6 * (W3232D + 32);          // 32->64 k3 s2
    static constexpr int OFF_SC = OFF_C1 + W3264 + 64;                // 32->64 k1 s2
    static constexpr int OFF_C2 = OFF_SC + W3264S + 64;               // 64->64 k3 of the strided block
    static constexpr int OFF_D = OFF_C2 + W6464D + 64;                // 6 convs 64->64 (residual blocks)
    static constexpr int W_TRUNK = OFF_D + 6 * (W6464D + 64);
    static constexpr int OFF_S1 = W_TRUNK;                            // [6 steps][64 lanes], bias[16]
    static constexpr int OFF_S2 = OFF_S1 + S1_STEPS * 64 + 16;        // [3 taps | 4 Winograd taps][64 lanes][4], bias[16]
    static constexpr int OFF_S3 = OFF_S2 + KT * 256 + 16;             // [2 blocks][3 taps | 4 Winograd taps][64 lanes][4], bias[32]
    static constexpr int W_TOTAL = OFF_S3 + 2 * KT * 256 + 32;
    // transfer-learning models append identity-shortcut 64-channel blocks (read_convolver_addendum.py); their
    // weights follow the canonical blob.  First conv of 64-channel block `blk`:
    static constexpr int off_d(int blk) {
        return blk < 3 ? OFF_D + 2 * blk * (W6464D + 64) : W_TOTAL + 2 * (blk - 3) * (W6464D + 64);
    }
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int STAMP_SLOTS = 48;                           // uint64 words per (workgroup, wave, group) of a stamped launch: readconv_kernel

// Geometry of a pileup window of WIN_ positions (150 in every shipped model, 250 in the feature-map variant):
//   window -> 3 valid k=3 convs (WIN-6) -> MaxPool(3,2): L1 -> strided block: L2
enum { ACT_RELU = 0, ACT_SOFTPLUS = 1 };
template <int G_, int NW_, int WIN_ = 150, int ACT_ = ACT_RELU>
struct Cfg {
    static constexpr int ACT = ACT_;                   // the network's activation (ReLU; Softplus in ..._layer_norm.py)
    // torch.nn.Softplus(beta 1, threshold 20) with the hardware exp / log (see conv_generic.hip); monotone like
    // ReLU, so it commutes with the stem's max pool
    static __device__ __forceinline__ float act(float x) {
        if constexpr (ACT_ == ACT_RELU) return fmaxf(x, 0.f);
        else return x > 20.f ? x : __logf(1.f + __expf(x));
    }
    static constexpr int G = G_;                       // reads per workgroup
    static constexpr int NW = NW_;                     // waves per workgroup
    static constexpr int THREADS = 64 * NW_;
    static constexpr int WINDOW = WIN_;
    static constexpr int L1 = (WIN_ - 6 - 3) / 2 + 1;  // positions per read at 32 channels (71 | 121)
    static constexpr int RS1 = L1 + 1;                 // row stride: ONE shared zero row between reads (even)
    static constexpr int L2 = (L1 - 1) / 2 + 1;        // positions per read at 64 channels (36 | 61)
    // 64 channels: an even L2 stacks the reads WITHOUT rows between them (the taps that would cross a read
    // boundary are zeroed at the source instead); an odd L2 takes one shared zero row, like the 32-channel image
    static constexpr bool COMPACT = (L2 % 2) == 0;
    static constexpr int RS2 = COMPACT ? L2 : L2 + 1;
    // the compact image is whole tiles of 16 TRIPLES of rows (and a read is whole triples): its residual blocks
    // run in Winograd F(3,3) form (wino3_layer)
    static constexpr bool F33 = COMPACT && (RS2 % 3) == 0 && ((RS2 * G_) % 48) == 0;
    static constexpr bool F33_32 = F33 && (RS1 % 3) == 0 && ((RS1 * G_) % 96) == 0;   // the 32-channel blocks too
    static constexpr int NTT = (L1 + 6) / 7;           // stem pool tiles per read (7 pooled outputs each)
    static constexpr int WPR = NW_ / G_;               // waves sharing a read in the stem pool
    static_assert(RS1 % 2 == 0 && RS2 % 2 == 0 && NW_ % G_ == 0, "pairs of rows must not straddle reads");
    static constexpr int T1 = (RS1 * G_ + 15) / 16;    // position tiles at 32 channels
    static constexpr int T2 = (RS2 * G_ + 15) / 16;    // ... at 64 channels
    static constexpr int ROWS1 = RS1 * G_ + 1;         // rows of the 32-channel image (row 0 = leading zero row)
    static constexpr int ROWS2 = RS2 * G_ + 2;         // leading and trailing zero row
    static constexpr int SB = G_;                      // the stem runs over all reads of the group at once
    static constexpr int SROWS = WIN_ * SB;
    static constexpr int ST12 = ((SROWS + 15) / 16 + 3) / 4 * 4;   // tiles of stem conv1 / conv2 (4 position groups)
    static constexpr int BUF_FLOATS = cmax(cmax(ROWS2 * 64, ROWS1 * 32), SROWS * 16);
    static constexpr int U8_BYTES = ((ST12 * 16 + 8) * 7 + 15) / 16 * 16;   // every conv1 tile reads in bounds

    // shortcut tiles a wave keeps in registers: T2 16-position tiles, or (Winograd form of the block's second
    // conv) two 16-position tiles per tile of 16 pairs
    static constexpr int NSREG = cmax((T2 * 4 + NW_ - 1) / NW_, 2 * ((RS2 * G_ / 2 + 15) / 16));
    // Tiles past the last read of the group are computed and discarded; their operand reads run up to
    // (T2*16 + 2) rows of 64 floats past the start of the SECOND image, so the allocation covers that
    // extent (the bytes read there are never consumed by a stored result).
    static constexpr int LDS_BYTES = cmax(2 * BUF_FLOATS * 4 + 64 + U8_BYTES,
                                          BUF_FLOATS * 4 + (T2 * 16 + 2) * 256 + 64);
};
}  // namespace rc

using Geometry = rc::Cfg<4, 4, 150>;      // 4 reads x 4 waves per workgroup, two workgroups per CU
using Geometry250 = rc::Cfg<2, 4, 250>;   // 250 bp windows: 2 reads per workgroup fill the same LDS
using GeometrySoftplus = rc::Cfg<4, 4, 150, rc::ACT_SOFTPLUS>;
static_assert(Geometry::F33 && GeometrySoftplus::F33 && !Geometry250::F33, "weight packing rule of readconv_pack.py");
static_assert(Geometry::F33_32 && GeometrySoftplus::F33_32, "weight packing rule of readconv_pack.py");
// bf16x3: [1 + 2 (3 + extra) layers][4 blocks][6 steps][hi | lo][64 lanes][8 bf16] = 12 288 floats per layer behind the blob
// and [6 layers 32 -> 32][2 blocks][3 taps][hi | lo][64 lanes][8] = 3 072 floats per layer behind those
int readconv_bf16x3_extra_floats(int extra_blocks) { return (1 + 2 * (3 + extra_blocks)) * 12288 + 6 * 3072; }
int readconv_weight_floats(int extra_blocks, bool winograd, int window) {
    if (!winograd) return rc::Offs<false>::off_d(3 + extra_blocks);
    return window == 150 ? rc::Offs<true, true>::off_d(3 + extra_blocks) : rc::Offs<true>::off_d(3 + extra_blocks);
}
bool readconv_supports_extra_blocks(int extra_blocks) { return extra_blocks == 0 || extra_blocks == 2; }

bool readconv_supports_window(int window) { return window == 150 || window == 250; }
int readconv_reads_per_group(int window) { return window == 250 ? Geometry250::G : Geometry::G; }
int readconv_stamp_slots() { return rc::STAMP_SLOTS; }
int readconv_stamp_waves() { return Geometry::NW; }    // waves per workgroup of the stamped instantiation (the kernel indexes its records with CF::NW)
int readconv_frame_rows(int window) { return window == 250 ? Geometry250::L2 : Geometry::L2; }
// Groups a workgroup walks.  More groups per workgroup = fewer partial-sum slots and one prologue per several
// groups (measured 1.2 %), but fewer, longer workgroups = a coarser tail when the last wave of workgroups does
// not fill the chip (two workgroups per CU run at a time).  Pick the count in 1..8 that minimises
// ceil(workgroups / resident slots) x groups, largest count on ties.
int readconv_groups_per_workgroup(long long n_reads, int window) {
    const long long slots = 2LL * device_cus();      // two workgroups per CU
    const int G = readconv_reads_per_group(window);
    const long long groups = (n_reads + G - 1) / G;
    int best = 1;
    double best_cost = 1e300;
    for (int n = 1; n <= 8; ++n) {
        const long long wgs = (groups + n - 1) / n;
        const double cost = (double)((wgs + slots - 1) / slots) * n * (n > 1 ? 0.988 : 1.0);
        if (cost <= best_cost) {
            best_cost = cost;
            best = n;
        }
    }
    return best;
}

// Two launches of the same kernel instead of one: the bulk in whole rounds of the resident workgroup slots, each
// workgroup walking n groups, and the remainder as one-group workgroups -- the partial last round then costs one
// group's time whatever n is, so n can be large (fewer partial slots, one prologue per n groups) without a coarse tail.
// n = the count in 1..8 minimising  rounds x n (x 0.988 for n > 1: the shared prologue) + ceil(rest / slots);
// largest n on ties.  Batches smaller than one round of n-group workgroups keep the single launch of
// readconv_groups_per_workgroup.
ReadConvPlan readconv_plan(long long n_reads, int window) {
    const long long slots = 2LL * device_cus();      // two workgroups per CU
    const int G = readconv_reads_per_group(window);
    const long long groups = (n_reads + G - 1) / G;
    ReadConvPlan best{1, groups, 0};
    double best_cost = 1e300;
    for (int n = 1; n <= 8; ++n) {
        const long long rounds = (groups / n) / slots;              // whole rounds of whole n-group workgroups
        if (rounds == 0) continue;
        const long long bulk = rounds * slots, rest = groups - bulk * n;
        const double cost = (double)rounds * n * (n > 1 ? 0.988 : 1.0) + (double)((rest + slots - 1) / slots);
        if (cost <= best_cost) {
            best_cost = cost;
            best = ReadConvPlan{n, bulk, rest};
        }
    }
    if (best_cost == 1e300) {                                        // less than one round: the single-launch rule
        const int n = readconv_groups_per_workgroup(n_reads, window);
        best = ReadConvPlan{n, (groups + n - 1) / n, 0};
    }
    return best;
}

// chunk swizzles: 16-byte chunk index of a row XORed with a function of the row
template <int C>
__device__ __forceinline__ int swz(int row) {
    return (C == 64 || C == 128) ? 2 * (row & 7) : (C == 32 ? 2 * ((row >> 1) & 3) : 2 * ((row >> 2) & 1));
}

// Images read by the Winograd layers (and by the stride-2 convolutions) are walked two rows per lane, so
// they carry a swizzle (SW_W) that makes rows r, r+2, ..., r+30 conflict-free instead of r, r+1, ..., r+15:
// at 64 channels chunk ^ 2*((row>>1)&7); at 32 channels (a row is half the banks) rows 4a+1 and 4a+2 also
// trade places and the chunk is XORed with 2*((row>>2)&3).
// SW_3 (64 channels): images walked THREE rows per lane (F(3,3) layers): chunk ^ 2*((row/3)&7); rows 3j..3j+2 share
// their swizzle, and 16 lanes' rows 3j+i hit 16 distinct bank groups.
enum { SW_OLD = 0, SW_W = 1, SW_3 = 2, SW_SPLIT = 3, SW_GLOBAL = 4 };   // SW_GLOBAL: `out` is global memory, rows of COUT floats

// ---- split images (arithmetic mode bf16x3: the 64-channel trunk on the bf16 matrix cores) ------------------------------
// A value x is kept as two bf16: hi = bf16(x), lo = bf16(x - hi): 16 of its 24 mantissa bits.  A 64-channel row is 256
// bytes -- the size of its fp32 row -- cut into sixteen 16-byte chunks: chunk g + 8 * part holds the 8 channels 8g .. 8g+7
// of part 0 (hi) or 1 (lo), so ONE ds_read_b128 is the 8-deep B operand of a v_mfma_f32_16x16x32_bf16 lane.  Chunks are
// XOR-swizzled with 2 * (row & 7): the 16 lanes of a ds_read_b128 group (rows j = 0-3, 12-15 of one lane quarter and
// 4-11 of the next, whose chunks differ in bit 0) then hit 16 distinct bank groups for any first row.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int split_off(int row, int g, int part) { return row * 256 + 16 * ((g + 8 * part) ^ (2 * (row & 7))); }
__device__ __forceinline__ unsigned short to_bf16(float x) {            // v_cvt_pk_bf16_f32: round to nearest even
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}
// 32 channels: a row is 128 bytes = eight 16-byte chunks, chunk g + 4 * part (g = 8-channel group 0..3), XOR-swizzled with
// 2 * ((row >> 1) & 3): two rows share a 256-byte bank span, and the chunks of neighbouring lane quarters differ in bit 0.
__device__ __forceinline__ int split_off32(int row, int g, int part) { return row * 128 + 16 * ((g + 4 * part) ^ (2 * ((row >> 1) & 3))); }
__device__ __forceinline__ void split4(f32x4 v, bf16x4& h, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned short hi = to_bf16(v[e]);
        h[e] = (short)hi;
        l[e] = (short)to_bf16(v[e] - __uint_as_float((unsigned)hi << 16));
    }
}
__device__ __forceinline__ void store_split32(unsigned char* __restrict__ img, int row, int c4, f32x4 v) {
    bf16x4 h, l;
    split4(v, h, l);
    *(bf16x4*)(img + split_off32(row, c4 >> 1, 0) + 8 * (c4 & 1)) = h;
    *(bf16x4*)(img + split_off32(row, c4 >> 1, 1) + 8 * (c4 & 1)) = l;
}
// channels 4 c4 .. 4 c4 + 3 of `row` <- v, as hi and lo parts (two 8-byte stores)
__device__ __forceinline__ void store_split(unsigned char* __restrict__ img, int row, int c4, f32x4 v) {
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned short hi = to_bf16(v[e]);
        h[e] = (short)hi;
        l[e] = (short)to_bf16(v[e] - __uint_as_float((unsigned)hi << 16));
    }
    *(bf16x4*)(img + split_off(row, c4 >> 1, 0) + 8 * (c4 & 1)) = h;
    *(bf16x4*)(img + split_off(row, c4 >> 1, 1) + 8 * (c4 & 1)) = l;
}
template <int C, int SW>
__device__ __forceinline__ int img_off(int row, int chunk) {     // float offset of 16-byte chunk `chunk` of `row`
    if constexpr (SW == SW_OLD) {
        return row * C + 4 * (chunk ^ swz<C>(row));
    } else if constexpr (SW == SW_3) {
        static_assert(C % 64 == 0, "SW_3 images have whole bank rows of channels (64 | 128): the XOR stays inside a row's first 16 chunks");
        return row * C + 4 * (chunk ^ (2 * ((row / 3) & 7)));
    } else if constexpr (C == 64 || C == 128) {
        return row * C + 4 * (chunk ^ (2 * ((row >> 1) & 7)));
    } else {
        static_assert(C == 32, "SW_W images have 32, 64 or 128 channels");
        const int prow = (row & ~3) | ((row & 1) << 1) | ((row >> 1) & 1);
        return prow * 32 + 4 * (chunk ^ (2 * ((row >> 2) & 3)));
    }
}

// img_off of row 48 t + 3 j + i (the i-th row of triple 16 t + j), i a small constant: SW_3's row / 3 without a
// division by 3 of a lane-varying value (16 t + j + i / 3, and 16 t vanishes under & 7)
template <int C, int SW>
__device__ __forceinline__ int img_off_triple(int t, int j, int i, int chunk) {
    if constexpr (SW == SW_3) return (48 * t + 3 * j + i) * C + 4 * (chunk ^ (2 * ((j + i / 3) & 7)));
    else return img_off<C, SW>(48 * t + 3 * j + i, chunk);
}

template <int NV>
__device__ __forceinline__ void load_weights(f32x4 (&w)[NV], const float* __restrict__ base, int cb, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) w[i] = *(const f32x4*)(base + ((cb * NV + i) * 64 + lane) * 4);
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// packed fp32 add / subtract of two lanes' worth of a float4 (v_pk_add_f32: two IEEE adds per instruction).
// ONLY for operands that come from LDS / plain VALU results: the compiler's hazard recogniser does not look
// inside inline asm, so feeding it a fresh MFMA result would read the accumulator before the matrix pipe has
// written it (the software-managed MFMA -> VALU hazard).  Epilogues therefore use ordinary vector arithmetic.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_add2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

enum { MODE_PLAIN = 0, MODE_RESID_INPLACE = 1, MODE_TO_REGS = 2, MODE_ADD_REGS = 3 };
enum { GEOM_TRUNK = 0, GEOM_STEM = 1, GEOM_WPAIR = 2, GEOM_WTRIPLE = 3 };
// WPAIR: tile 2t / 2t+1 = even / odd rows of the pairs of Winograd tile t;  WTRIPLE: tile 3t + u = rows 3T + u of the
// triples T of F(3,3) tile t

// value of lane+n of the same 16-lane row (DPP row_shl); lanes whose source falls outside the row read 0
// (bound_ctrl: no preload of the destination, and the shift can fold into the consuming instruction)
__device__ __forceinline__ float row_shl(float v, int n) {
    const int vi = __float_as_int(v);
    return __int_as_float(n == 1 ? __builtin_amdgcn_update_dpp(0, vi, 0x101, 0xf, 0xf, true)
                                 : __builtin_amdgcn_update_dpp(0, vi, 0x102, 0xf, 0xf, true));
}

// One convolution over the whole group.  `in`/`out` are LDS images with CIN / COUT floats per row.
//   w        this wave's 16-channel weight slice; with ROLL its registers are refilled in place with the
//            NEXT layer's slice (`next_w`, already offset to this wave's block and lane) right after their
//            last use, so only one slice is ever resident
//   padmask  bit k set: row j of the wave's k-th tile is a shared zero row (must be stored as zero)
//   dump     16 spare LDS bytes: stores of rows past the group are redirected there instead of branching
//   GEOM     GEOM_TRUNK: images with a leading zero row and RS_IN / RS_OUT rows per read;  GEOM_STEM: the stem's
//            flat stacks (150 rows per read, no leading row): tile t starts at row TS*t, valid convolution
//   VROWS    output rows >= VROWS are discarded
template <class CF, int CIN, int COUT, int KT, int STRIDE, int PAD, int RS_IN, int RS_OUT, int LOUT, int T, int MODE,
          bool ROLL, int GEOM = GEOM_TRUNK, int TS = 16, int VROWS = RS_OUT * CF::G, bool BMASK = false,
          int SIN = SW_OLD, int SOUT = SW_OLD, bool PARTIAL = false>
__device__ __forceinline__ void conv_layer(const float* __restrict__ in, float* __restrict__ out,
                                           f32x4 (&w)[KT * CIN / 16], const float* __restrict__ next_w,
                                           const float* __restrict__ bias, f32x4 (&sreg)[CF::NSREG],
                                           unsigned padmask, float* __restrict__ dump, int wave, int lane,
                                           int vrows_rt = 0,        // SW_GLOBAL: output rows >= vrows_rt are not written
                                           int live_tiles = 1 << 30) {   // PARTIAL: only the wave's first `live_tiles` tiles hold
                                                                          // items (a partly filled workgroup): later pairs are skipped
    constexpr int M = CIN / 16, NCB = COUT / 16, NPG = CF::NW / NCB, ITER = T / NPG;
    static_assert(NPG >= 1 && NCB * NPG == CF::NW && T % NPG == 0, "waves must tile channel blocks x position groups");
    const int cb = wave % NCB, pg = wave / NCB;
    const int j = lane & 15, q = lane >> 4;
    const f32x4 b4 = *(const f32x4*)(bias + cb * 16 + 4 * q);

    // Stride-1 layers keep the image geometry: output row r reads input rows r + tap (+1 for the leading
    // zero row, -PAD).  A tile then only adds a compile-time constant to the lane's address (16 t + x keeps
    // x's low bits, so the swizzle of row 16 t + x is the swizzle of x): one pointer per (tap, m) serves
    // every tile through the ds_read offset field.
    constexpr bool IDENT = (GEOM == GEOM_STEM) ? (TS == 16) : ((RS_IN == RS_OUT) && (STRIDE == 1));
    constexpr int LEAD = (GEOM == GEOM_STEM) ? 0 : 1;
    // Stride-2 layers over images whose input row stride is exactly twice the output row stride (the compact
    // 150 bp geometry: 72 = 2 x 36): the read index cancels, output row r reads input row 2 r + 1 - PAD + tap,
    // so here too a tile only adds a constant (two pointer sets for the even / odd tiles of GEOM_WPAIR).
    constexpr bool S2LIN = (GEOM != GEOM_STEM) && (STRIDE == 2) && (RS_IN == 2 * RS_OUT);
    static_assert(!S2LIN || GEOM != GEOM_WPAIR || NPG == 1, "paired-row tiles assume one position group");
    static_assert(GEOM != GEOM_WTRIPLE || (S2LIN && NPG == 1), "triple-row tiles: the linear stride-2 geometry, one position group");
    const float* opbase[(S2LIN && GEOM == GEOM_WPAIR) ? 2 * KT * M : ((GEOM == GEOM_WTRIPLE) ? 3 * KT * M : KT * M)];
    if constexpr (GEOM == GEOM_WTRIPLE) {
#pragma unroll
        for (int s = 0; s < 3 * KT * M; ++s) {           // rows 96 (t/3) + 6 j + 2 (t%3) + 1 - PAD + tap
            const int u = s / (KT * M), ss = s % (KT * M);
            opbase[s] = in + img_off<CIN, SIN>(6 * j + 2 * u + 1 - PAD + ss / M, 4 * (ss % M) + q);
        }
    } else if constexpr (S2LIN && GEOM == GEOM_WPAIR) {
#pragma unroll
        for (int s = 0; s < 2 * KT * M; ++s) {           // rows 64 (t/2) + 4 j + 2 (t&1) + 1 - PAD + tap
            const int par = s / (KT * M), ss = s % (KT * M);
            opbase[s] = in + img_off<CIN, SIN>(4 * j + 2 * par + 1 - PAD + ss / M, 4 * (ss % M) + q);
        }
    } else if constexpr (S2LIN) {
#pragma unroll
        for (int s = 0; s < KT * M; ++s)                  // rows 32 t + 2 j + 1 - PAD + tap
            opbase[s] = in + 32 * pg * CIN + img_off<CIN, SIN>(2 * j + 1 - PAD + s / M, 4 * (s % M) + q);
    } else {
#pragma unroll
        for (int s = 0; s < KT * M; ++s) {
            const int x = j + LEAD - PAD + s / M;
            opbase[s] = in + 16 * pg * CIN + img_off<CIN, SIN>(x, 4 * (s % M) + q);
        }
    }
    auto tile_operand = [&](int k, int s) -> f32x4 {     // k-th tile of this wave, step s = tap*M + m
        if (IDENT) return *(const f32x4*)(opbase[s] + k * NPG * 16 * CIN);
        if constexpr (GEOM == GEOM_WTRIPLE) return *(const f32x4*)(opbase[(k % 3) * KT * M + s] + (k / 3) * 96 * CIN);
        if constexpr (S2LIN && GEOM == GEOM_WPAIR) return *(const f32x4*)(opbase[(k & 1) * KT * M + s] + (k / 2) * 64 * CIN);
        if constexpr (S2LIN && GEOM != GEOM_WPAIR) return *(const f32x4*)(opbase[s] + k * NPG * 32 * CIN);
        int row;
        if (GEOM == GEOM_STEM) {
            row = TS * (pg + NPG * k) + j + s / M;
        } else {
            const int kk = pg + NPG * k;
            const int r = (GEOM == GEOM_WPAIR) ? 32 * (kk / 2) + 2 * j + (kk & 1) : kk * 16 + j;
            const int rd = r / RS_OUT;
            row = 1 + rd * RS_IN + (r - rd * RS_OUT) * STRIDE - PAD + s / M;
        }
        return *(const f32x4*)(in + img_off<CIN, SIN>(row, 4 * (s % M) + q));
    };
    // output pointer of the k-th tile: base + constant; rows past the group go to the dump slot
    static_assert(SOUT != SW_SPLIT || (COUT == 64 && MODE == MODE_PLAIN), "split images: the 64-channel trunk's input");
    static_assert(SOUT != SW_GLOBAL || MODE == MODE_PLAIN, "global outputs: plain convolutions");
    float* const outbase = out + 16 * pg * COUT + img_off<COUT, (SOUT == SW_SPLIT || SOUT == SW_GLOBAL) ? SW_OLD : SOUT>(j + LEAD, 4 * cb + q);
    auto out_ptr = [&](int k) -> float* {
        float* ptr = outbase + k * NPG * 16 * COUT;
        if constexpr (SOUT == SW_3)                                     // a swizzle without the tiles' 16-row period
            ptr = out + img_off<COUT, SOUT>(16 * (pg + NPG * k) + j + LEAD, 4 * cb + q);
        if ((NPG - 1 + NPG * k) * 16 + 15 >= VROWS)                   // only the last tile(s) can overrun
            ptr = ((pg + NPG * k) * 16 + j < VROWS) ? ptr : dump;
        return ptr;
    };
    auto epilogue = [&](int k, f32x4 acc, f32x4 res) {
        f32x4 v;
        if (MODE == MODE_TO_REGS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[e] + b4[e];
            sreg[k < CF::NSREG ? k : 0] = v;
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = CF::act(acc[e] + b4[e]);
        if (MODE == MODE_RESID_INPLACE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += res[e];
        }
        if (MODE == MODE_ADD_REGS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += sreg[k < CF::NSREG ? k : 0][e];
        }
        if ((padmask >> k) & 1u) v = f32x4{0.f, 0.f, 0.f, 0.f};      // the shared zero row between reads
        if constexpr (SOUT == SW_SPLIT) {                              // hi / lo bf16 parts for the bf16x3 trunk
            if ((pg + NPG * k) * 16 + j < VROWS) store_split((unsigned char*)out, 16 * (pg + NPG * k) + j + LEAD, 4 * cb + q, v);
        } else if constexpr (SOUT == SW_GLOBAL) {                      // straight to HBM: [flat output row][COUT]
            const int row = (pg + NPG * k) * 16 + j;
            if (row < vrows_rt) *(f32x4*)(out + (long long)row * COUT + 16 * cb + 4 * q) = v;
        } else {
            *(f32x4*)out_ptr(k) = v;
        }
    };
    auto residual = [&](int k) -> f32x4 {
        if (MODE == MODE_RESID_INPLACE) return *(const f32x4*)out_ptr(k);
        return f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- tile pairs, software pipelined over the whole layer --------------------------------------
    // A step = one (tap, m) of a tile pair = 2 ds_read_b128 + 8 MFMAs.  The two waves of a SIMD run the
    // same program and fall into lockstep (they reach their LDS waits together), so a wave hides its own
    // latencies: operands of step u+DEPTH are requested before the MFMAs of step u are issued, across pair
    // boundaries; the finished pair's epilogue (bias, ReLU, residual, store) is deferred into steps 1 and 2
    // of the next pair, where it fills MFMA issue gaps; its residual input is requested at step 0.  The
    // order is pinned with sched_barrier.  Each tile accumulates in two chains (even / odd k steps): with
    // four chains in flight a chain is revisited every 128 cycles, far beyond the 40-cycle dependent latency.
    constexpr int S = KT * M, NP = ITER / 2, NU = NP * S, DEPTH = 2;
    constexpr bool DEFER = (MODE != MODE_TO_REGS) && (S >= 3) && !PARTIAL;   // (a skipped pair could not run its predecessor's epilogue)
    static_assert(!PARTIAL || !ROLL, "partial workgroups: plain layers (live_tiles counts the tiles of THIS wave: pg + NPG k)");
    // An odd share leaves one lone tile.  It runs in the same pipeline as a pseudo-pair: its k steps are
    // split in two halves that play the roles of the two tiles (each with its own weight registers).
    constexpr bool LONE = (ITER % 2) != 0;
    static_assert(!LONE || (S % 2 == 0), "a lone tile needs an even number of k steps");
    constexpr int HS = S / 2, NUT = NU + (LONE ? HS : 0);
    static_assert(!BMASK || NPG == 1, "boundary masks assume one position group (tile index = wave's tile index)");
    // BMASK (64-channel images without zero rows between reads): lane j of tile t must see zero instead of
    // the previous read's last row at tap 0 when 16 t + j is a read's first row, and instead of the next
    // read's first row at the last tap when it is a read's last row.
    f32x4 ring0[DEPTH + 1], ring1[DEPTH + 1];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = zero4, a1 = zero4, b0 = zero4, b1 = zero4;
    f32x4 pend0 = zero4, pend1 = zero4, res0 = zero4, res1 = zero4;

    // every index below is a compile-time constant (static_for hands the step number over as a type)
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int t0 = u < NU ? 2 * (u / S) : ITER - 1, t1 = u < NU ? 2 * (u / S) + 1 : ITER - 1;
        constexpr int s0 = u < NU ? u % S : u - NU, s1 = u < NU ? u % S : u - NU + HS;
        ring0[u % (DEPTH + 1)] = tile_operand(t0, s0);
        ring1[u % (DEPTH + 1)] = tile_operand(t1, s1);
    };
    static_for<0, (DEPTH < NUT ? DEPTH : NUT)>(issue);
    static_for<0, NUT>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int t0 = u < NU ? 2 * (u / S) : ITER - 1, t1 = u < NU ? 2 * (u / S) + 1 : ITER - 1;
        constexpr int s0 = u < NU ? u % S : u - NU, s1 = u < NU ? u % S : u - NU + HS;
        constexpr int ls = u < NU ? u % S : u - NU;                 // step within the current (pseudo-)pair
        constexpr int ip = (u < NU ? u / S : NP) - 1;              // the pair whose epilogue may be pending
        constexpr bool pending = DEFER && ip >= 0 && (u < NU || HS >= 3);
        if (PARTIAL && t0 >= live_tiles) return;                   // wave-uniform: this pair (or lone tile) holds no items
        if constexpr (u + DEPTH < NUT) issue(std::integral_constant<int, u + DEPTH>{});
        if constexpr (pending && ls == 0) {
            res0 = residual(2 * ip);
            res1 = residual(2 * ip + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pending && ls == 1) epilogue(2 * ip, pend0, res0);
        if constexpr (pending && ls == 2) epilogue(2 * ip + 1, pend1, res1);
        const f32x4 w0 = w[s0], w1 = w[s1];
        f32x4 x0 = ring0[u % (DEPTH + 1)], x1 = ring1[u % (DEPTH + 1)];
        if constexpr (BMASK) {
            constexpr int tap0 = s0 / M, tap1 = s1 / M;
            constexpr int first0 = ((16 * t0 + RS_OUT - 1) / RS_OUT) * RS_OUT, last0 = ((16 * t0 + RS_OUT) / RS_OUT) * RS_OUT - 1;
            constexpr int first1 = ((16 * t1 + RS_OUT - 1) / RS_OUT) * RS_OUT, last1 = ((16 * t1 + RS_OUT) / RS_OUT) * RS_OUT - 1;
            if constexpr (RS_OUT < 16) {
                // short items (the allele stage's 9-row outputs): a tile holds up to two item starts
                if constexpr (tap0 == 0) {
                    bool f0 = false;
                    static_for<0, 16>([&](auto jc) {
                        constexpr int jj = decltype(jc)::value;
                        if constexpr ((16 * t0 + jj) % RS_OUT == 0 && 16 * t0 + jj > 0) f0 = f0 || (j == jj);
                    });
                    x0 = f0 ? zero4 : x0;
                }
                if constexpr (tap1 == 0) {
                    bool f1 = false;
                    static_for<0, 16>([&](auto jc) {
                        constexpr int jj = decltype(jc)::value;
                        if constexpr ((16 * t1 + jj) % RS_OUT == 0 && 16 * t1 + jj > 0) f1 = f1 || (j == jj);
                    });
                    x1 = f1 ? zero4 : x1;
                }
                static_assert(RS_OUT >= 16 || STRIDE == 2, "short items: the stride-2 layer (no tap crosses an item's end)");
            } else {
            if constexpr (tap0 == 0 && first0 > 0 && first0 <= 16 * t0 + 15) x0 = (j == first0 - 16 * t0) ? zero4 : x0;
            // (a stride-2 layer over rows of even length never reads past its row's end: 2 (L/2 - 1) + 1 = L - 1)
            if constexpr (STRIDE == 1 && tap0 == KT - 1 && last0 <= 16 * t0 + 15 && last0 < RS_OUT * CF::G - 1) x0 = (j == last0 - 16 * t0) ? zero4 : x0;
            if constexpr (tap1 == 0 && first1 > 0 && first1 <= 16 * t1 + 15) x1 = (j == first1 - 16 * t1) ? zero4 : x1;
            if constexpr (STRIDE == 1 && tap1 == KT - 1 && last1 <= 16 * t1 + 15 && last1 < RS_OUT * CF::G - 1) x1 = (j == last1 - 16 * t1) ? zero4 : x1;
            }
        }
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[0], x0[0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0], x1[0], a1, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[1], x0[1], b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1], x1[1], b1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[2], x0[2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[2], x1[2], a1, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[3], x0[3], b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[3], x1[3], b1, 0, 0, 0);
        if constexpr (ROLL) {
            // refill a weight register with the next layer's value right after its last use
            if constexpr (LONE && u >= NU) {
                w[s0] = *(const f32x4*)(next_w + s0 * 256);
                w[s1] = *(const f32x4*)(next_w + s1 * 256);
            }
            if constexpr (!LONE && u >= NU - S) w[s0] = *(const f32x4*)(next_w + s0 * 256);
        }
        if constexpr (u < NU && u % S == S - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] += b0[e];
                a1[e] += b1[e];
            }
            if constexpr (DEFER && (u + 1 < NU || (LONE && HS >= 3))) {
                pend0 = a0;
                pend1 = a1;
            } else {
                epilogue(t0, a0, residual(t0));
                epilogue(t1, a1, residual(t1));
            }
            a0 = a1 = b0 = b1 = zero4;
        }
        if constexpr (LONE && u == NUT - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a0[e] = (a0[e] + b0[e]) + (a1[e] + b1[e]);
            epilogue(ITER - 1, a0, residual(ITER - 1));
        }
    });
}

// ---- Winograd F(2,3) form of a k = 3, stride 1, pad 1 convolution C -> C over a trunk image ------------
// Two neighbouring outputs share four inputs d0..d3 (image rows 2P .. 2P+3 for the pair P of flat rows 2P, 2P+1):
//     V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3        (per channel, in the lane that read them)
//     Mc = Uc * Vc  over the input channels, c = 0..3                (4 MFMA accumulators = 4 independent chains)
//     y(2P) = M0 + M1 + M2,   y(2P+1) = M1 - M2 - M3
// i.e. 4 contractions per pair of positions instead of 6.  A tile is 16 PAIRS (32 rows): lane (j, q) reads rows
// 32 t + 2 j + i, so the images use the SW_W swizzle.  A step = one 16-channel input group: 4 ds_read_b128
// (requested one step ahead), 16 VALU, 16 MFMAs.  `w` holds U for this wave's 16 output channels:
// w[c * C/16 + m]; with ROLL it is refilled in place with the next layer's registers after their last use.
//   zmask   (32 channels) bit k: this lane's odd row of the wave's k-th tile is a shared zero row
template <class CF, int C, int MODE, bool ROLL, bool FLIP = false, int SOUT = SW_W>
__device__ __forceinline__ void wino_layer(const float* __restrict__ in, float* __restrict__ out,
                                           f32x4 (&w)[4 * C / 16], const float* __restrict__ next_w,
                                           const float* __restrict__ bias, unsigned zmask, float* __restrict__ dump,
                                           int wave, int lane, const f32x4* __restrict__ sreg = nullptr) {
    // MODE_ADD_REGS: the block's shortcut waits in `sreg` (GEOM_WPAIR layout: 2k = even rows, 2k+1 = odd rows)
    static_assert(MODE == MODE_PLAIN || MODE == MODE_RESID_INPLACE || MODE == MODE_ADD_REGS, "block convolutions only");
    constexpr int M = C / 16, NCB = C / 16, NPG = CF::NW / NCB;
    constexpr int PAIRS = (C == 64 ? CF::RS2 : CF::RS1) * CF::G / 2;       // 72 | 144 pairs of rows in the group
    constexpr bool EDGE = (C == 64) && CF::COMPACT;                          // reads stacked without zero rows
    constexpr int PPR = (C == 64 ? CF::RS2 : CF::RS1) / 2;                   // pairs per read
    constexpr int NT = (PAIRS + 15) / 16;                                    // 5 | 9 tiles
    constexpr int ITER = (NT + NPG - 1) / NPG;                               // tiles of position group 0
    constexpr int TOFF = NPG * 32 * C;                                       // floats between a wave's tiles
    constexpr bool HAS_TAIL = (NT % NPG) != 0;                               // the last tile exists for group 0 only
    static_assert(NCB * NPG == CF::NW, "waves must tile channel blocks x position groups");
    // FLIP hands the odd tile to the other position group: alternate layers load the SIMDs evenly
    const int cb = wave % NCB, pg = FLIP ? NPG - 1 - wave / NCB : wave / NCB;
    const int j = lane & 15, q = lane >> 4;
    const f32x4 b4 = *(const f32x4*)(bias + cb * 16 + 4 * q);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // rows 32 pg + 2 j + i, i = 0..3: rows i and i+1 (i even) share their swizzle and sit a constant apart
    // (64 floats in either image), so one pointer per (i / 2, input group) serves both
    const float* opb[2 * M];
#pragma unroll
    for (int s = 0; s < 2 * M; ++s) opb[s] = in + 32 * pg * C + img_off<C, SW_W>(2 * j + 2 * (s / M), 4 * (s % M) + q);
    constexpr int ODD = 64;                                                  // img_off(r + 1, c) - img_off(r, c), r even
    static_assert(SOUT == SW_W || (C == 64 && NPG == 1 && MODE != MODE_RESID_INPLACE), "another output swizzle: one position group, not in place");
    float* const o0 = out + 32 * pg * C + img_off<C, SOUT>(2 * j + 1, 4 * cb + q);   // flat row 2P   = image row 2P + 1
    float* const o1 = out + 32 * pg * C + img_off<C, SOUT>(2 * j + 2, 4 * cb + q);   // flat row 2P+1 = image row 2P + 2
    const bool last_tile_here = !HAS_TAIL || (pg + NPG * (ITER - 1)) < NT;   // wave-uniform

    f32x4 ring[2][4];
    f32x4 acc[2][4];                                                         // accumulator sets of even / odd tiles
    f32x4 res0 = zero4, res1 = zero4, y0 = zero4, y1 = zero4;
    constexpr int NU = ITER * M;
    // At 64 channels the reads are stacked without zero rows: the row before a read's first and the row after
    // its last must read as zero.  The one lane of a tile that sits on such a boundary fetches its d0 (d3) from
    // the image's leading zero row instead: one address select per read instead of four value selects.
    const float* const zrow = in + 4 * q;
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / M, m = u % M;
        constexpr int lo = ((16 * k + PPR - 1) / PPR) * PPR;     // first pair >= 16 k that starts a read
        constexpr int hi = ((16 * k + PPR) / PPR) * PPR - 1;     // first pair >= 16 k that ends a read
        static_for<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float* ptr = opb[(i / 2) * M + m] + (i & 1) * ODD + k * TOFF;
            if constexpr (EDGE && i == 0 && lo > 0 && lo <= 16 * k + 15) ptr = (j == lo - 16 * k) ? zrow : ptr;
            if constexpr (EDGE && i == 3 && hi <= 16 * k + 15 && hi < PAIRS - 1) ptr = (j == hi - 16 * k) ? zrow : ptr;
            ring[u & 1][i] = *(const f32x4*)ptr;
        });
    };
    // output transform, bias, ReLU, residual of element e of tile kk (its accumulators are complete)
    auto epi_store = [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        {
            const f32x4(&a)[4] = acc[kk & 1];
            y0 = (a[0] + a[1]) + a[2];               // the bias rides in a[1]: its chain started from b4
            y1 = (a[1] - a[2]) - a[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y0[e] = CF::act(y0[e]);
                y1[e] = CF::act(y1[e]);
            }
            if constexpr (MODE == MODE_RESID_INPLACE) {
                y0 = y0 + res0;
                y1 = y1 + res1;
            }
            if constexpr (MODE == MODE_ADD_REGS) {
                y0 = y0 + sreg[2 * kk];
                y1 = y1 + sreg[2 * kk + 1];
            }
        }
        if constexpr (!EDGE) {
            if ((zmask >> kk) & 1u) y1 = zero4;                               // the shared zero row between reads
        }
        float* p0 = o0 + kk * TOFF;
        float* p1 = o1 + kk * TOFF;
        if constexpr (SOUT != SW_W) {                                         // a swizzle without the tiles' 32-row period
            p0 = out + img_off<C, SOUT>(32 * kk + 2 * j + 1, 4 * cb + q);
            p1 = out + img_off<C, SOUT>(32 * kk + 2 * j + 2, 4 * cb + q);
        }
        if constexpr (16 * (NT - 1) + 15 >= PAIRS && kk == ITER - 1) {       // only the last tile can overrun
            const bool ok = 16 * (pg + NPG * kk) + j < PAIRS;
            p0 = ok ? p0 : dump;
            p1 = ok ? p1 : dump;
        }
        *(f32x4*)p0 = y0;
        *(f32x4*)p1 = y1;
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / M, m = u % M;
        constexpr bool tail = HAS_TAIL && (k == ITER - 1);                   // a tile only position group 0 owns
        constexpr bool pending = (m == 0) && (k >= 1);                       // tile k-1 awaits its epilogue
        if constexpr (u + 1 < NU) issue(std::integral_constant<int, u + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's epilogue: one block of VALU work + two stores ahead of this step's MFMAs
        if constexpr (pending) epi_store(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
        if (!tail || last_tile_here) {
            const f32x4 d0 = ring[u & 1][0], d1 = ring[u & 1][1], d2 = ring[u & 1][2], d3 = ring[u & 1][3];
            f32x4(&a)[4] = acc[k & 1];
            // The input transform runs as ONE block of (packed) VALU operations ahead of the step's MFMAs: VALU
            // and MFMA instructions share the SIMD's issue port, and a VALU operation in front of every MFMA
            // costs the other wave of the SIMD an issue slot per MFMA (measured: 16 % of the layer).
            const f32x4 t0 = pk_sub(d0, d2), t1 = pk_add(d1, d2), t2 = pk_sub(d2, d1), t3 = pk_sub(d1, d3);
            asm volatile("s_nop 1");         // VALU write -> MFMA source read distance, whatever the MFMA order below
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool first = (m == 0) && (e == 0);
                a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0 * M + m][e], t0[e], first ? zero4 : a[0], 0, 0, 0);
                a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1 * M + m][e], t1[e], first ? b4 : a[1], 0, 0, 0);
                a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2 * M + m][e], t2[e], first ? zero4 : a[2], 0, 0, 0);
                a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3 * M + m][e], t3[e], first ? zero4 : a[3], 0, 0, 0);
            }
        }
        if constexpr (MODE == MODE_RESID_INPLACE && m == M - 1) {            // residual input of this tile's epilogue
            res0 = *(const f32x4*)(o0 + k * TOFF);
            res1 = *(const f32x4*)(o1 + k * TOFF);
        }
        if constexpr (ROLL && k == ITER - 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c * M + m] = *(const f32x4*)(next_w + (c * M + m) * 256);
        }
        if constexpr (u == NU - 1) {
            if (!tail || last_tile_here) epi_store(std::integral_constant<int, ITER - 1>{});
        }
    });
}

// ---- Winograd F(3,3) form of a k = 3, stride 1, pad 1 convolution 64 -> 64 over the compact image ---------
// Three neighbouring outputs share five inputs d0..d4 (image rows 3T .. 3T+4 for the triple T of flat rows 3T..3T+2);
// interpolation points 0, 1, -1, 2, inf:
//     V0 = 2(d0-d2) + (d3-d1), V1 = (d3-d1) - (d1+d2), V2 = 3(d1-d2) + (d3-d1), V3 = d3-d1, V4 = (d4-d2) - 2(d3-d1)
//     Mc = Uc * Vc over the input channels, c = 0..4, U = (g0/2, -(g0+g1+g2)/2, (-g0+g1-g2)/6, (g0+2g1+4g2)/6, g2)
//     y(3T) = M0+M1+M2+M3,  y(3T+1) = M1 - M2 + 2 M3,  y(3T+2) = M1 + M2 + 4 M3 + M4
// i.e. 5 contractions per 3 positions instead of 9 (F(2,3): 6).  The group's 144 rows are exactly 3 tiles of 16
// triples (F(2,3): 72 pairs = 4.5 tiles, the fifth half empty): 240 MFMAs per wave and layer instead of 320.  A
// read is 12 triples, so only d0 of a read's first triple and d4 of its last cross a read boundary: those lanes
// fetch the image's leading zero row instead.  Images use the SW_3 swizzle.
// The INPUT GROUPS are the outer loop and the 3 tiles the inner one: all 15 accumulators (3 tiles x 5 components)
// stay live, but only one input group's weights (5 registers) are resident, with the next group's (or the next
// layer's first) requested a whole group ahead: w[m & 1] serves group m, so a layer starts and ends on w[0].
// Tile k's output transform + bias + activation + residual + stores run ahead of tile k+1's last step.
// At 32 channels (C = 32) the same layer runs on the SW_W image with its shared zero rows: 288 rows = 6 tiles of 16
// triples, 2 channel blocks x 2 position groups of waves, 3 tiles (pg, pg+2, pg+4) and 2 input groups per wave: 120
// MFMAs per wave and layer, evenly (F(2,3): 9 tiles split 5 / 4, 144 on average).  No read boundaries to patch on
// the input side (the zero rows are the padding); a read's 24th triple ends ON the zero row, whose output is
// stored as zero.
// The allele-level compressor kernel below instantiates the same layer at 128 channels over 8 items of 18 rows (8 waves):
// RS (rows per item), the image swizzle and the compact-stacking flag are template parameters defaulting to the read
// convolver's geometry.
// PARTIAL: only the wave's first `live_tiles` tiles hold items (a partly filled workgroup of a small launch): the
// others are neither read, multiplied nor stored.
template <class CF, int C, int MODE, bool LAST, int RS = (C == 64 ? CF::RS2 : CF::RS1), int SW = (C == 32 ? SW_W : SW_3),
          bool EDGE = (C != 32), bool PARTIAL = false>
__device__ __forceinline__ void wino3_layer(const float* __restrict__ in, float* __restrict__ out, f32x4 (&w)[2][5],
                                            const float* __restrict__ wl, const float* __restrict__ next_wl,
                                            const float* __restrict__ bias, int wave, int lane, int live_tiles = 1 << 30) {
    static_assert(MODE == MODE_PLAIN || MODE == MODE_RESID_INPLACE, "block convolutions only (the strided block's second "
                  "conv finds its shortcut in the output image, like a residual)");
    static_assert(RS % 3 == 0 && (RS * CF::G) % 48 == 0, "image of whole tiles of 16 triples");
    constexpr int M = C / 16, NCB = C / 16, NPG = CF::NW / NCB;              // input groups; waves = blocks x position groups
    static_assert(NPG >= 1 && NCB * NPG == CF::NW && (RS * CF::G / 48) % NPG == 0, "waves = channel blocks x position groups");
    constexpr int NT = RS * CF::G / 48 / NPG;                                // 3 tiles of 16 triples per wave
    constexpr int TPR = RS / 3;                                              // triples per read (12 | 24; 6 per allele)
    constexpr int TOFF = NPG * 48 * C;                                       // floats between a wave's tiles (a period of the swizzle)
    constexpr int NU = M * NT;
    const int cb = wave % NCB, pg = wave / NCB, j = lane & 15, q = lane >> 4;
    const f32x4 b4 = *(const f32x4*)(bias + cb * 16 + 4 * q);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // rows 48 pg + 3j + i, i = 0..4, input group m.  SW_3: i = 0..2 share the swizzle of row 3j, i = 3, 4 that of
    // row 3j + 3 (two pointers per group); SW_W (32 channels): one pointer per row
    constexpr int NP = SW == SW_3 ? 2 : 5;
    const float* pin[M][NP];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int i = 0; i < NP; ++i) pin[m][i] = in + img_off_triple<C, SW>(pg, j, (SW == SW_3 ? 3 * i : i), 4 * m + q);
    const float* const zrow = in + 4 * q;                                    // the leading zero row
    // flat rows 3T, 3T+1, 3T+2 = image rows 3T+1, 3T+2, 3T+3
    float* po[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) po[u] = out + img_off_triple<C, SW>(pg, j, 1 + u, 4 * cb + q);

    unsigned zmask = 0;                     // bit k: the third row of this lane's triple in the wave's k-th tile is a zero row
    if constexpr (!EDGE) {
#pragma unroll
        for (int k = 0; k < NT; ++k) zmask |= ((16 * (pg + NPG * k) + j) % TPR == TPR - 1 ? 1u : 0u) << k;
    }
    f32x4 ring[2][5];
    f32x4 acc[NT][5];
    f32x4 res[3] = {zero4, zero4, zero4};
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int m = u / NT, k = u % NT;
        bool first = false, last = false;                                    // this lane's triple opens / closes a read
        if constexpr (EDGE) {
            static_for<0, 16>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                if constexpr ((16 * k + jj) % TPR == 0) first = first || (j == jj);
                if constexpr ((16 * k + jj) % TPR == TPR - 1) last = last || (j == jj);
            });
        }
        static_for<0, 5>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float* ptr = (SW == SW_3 ? pin[m][i / 3] + (i % 3) * C : pin[m][i < NP ? i : 0]) + k * TOFF;
            if constexpr (EDGE && i == 0) ptr = first ? zrow : ptr;
            if constexpr (EDGE && i == 4) ptr = last ? zrow : ptr;
            ring[u & 1][i] = *(const f32x4*)ptr;
        });
    };
    // output transform, bias (rides in M1), activation, residual, stores of tile kk (its accumulators are complete)
    auto epi_store = [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        const f32x4(&a)[5] = acc[kk];
        const f32x4 sum = a[1] + a[2], dif = a[1] - a[2];
        const f32x4 two = {2.f, 2.f, 2.f, 2.f}, four = {4.f, 4.f, 4.f, 4.f};
        f32x4 y0 = (a[0] + sum) + a[3];
        f32x4 y1 = __builtin_elementwise_fma(a[3], two, dif);
        f32x4 y2 = __builtin_elementwise_fma(a[3], four, sum) + a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y0[e] = CF::act(y0[e]);
            y1[e] = CF::act(y1[e]);
            y2[e] = CF::act(y2[e]);
        }
        if constexpr (MODE == MODE_RESID_INPLACE) {
            y0 = y0 + res[0];
            y1 = y1 + res[1];
            y2 = y2 + res[2];
        }
        if constexpr (!EDGE) {                                               // the shared zero row between reads
            if ((zmask >> kk) & 1u) y2 = zero4;
        }
        *(f32x4*)(po[0] + kk * TOFF) = y0;
        *(f32x4*)(po[1] + kk * TOFF) = y1;
        *(f32x4*)(po[2] + kk * TOFF) = y2;
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int m = u / NT, k = u % NT;
        if constexpr (u + 1 < NU) {
            if (!PARTIAL || (u + 1) % NT < live_tiles) issue(std::integral_constant<int, u + 1>{});
        }
        if constexpr (k == 0) {                                              // the next group's weights, a group ahead
            const float* nw = (m + 1 < M) ? wl + (m + 1) * 5 * 256 : next_wl;
            if constexpr (m + 1 < M || !LAST) {
#pragma unroll
                for (int c = 0; c < 5; ++c) w[(m + 1) & 1][c] = *(const f32x4*)(nw + c * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the previous tile's epilogue: one block of VALU work + three stores ahead of this step's MFMAs
        if constexpr (m == M - 1 && k >= 1) {
            if (!PARTIAL || k - 1 < live_tiles) epi_store(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
        }
        if (!PARTIAL || k < live_tiles) {
            const f32x4(&d)[5] = ring[u & 1];
            // one block of packed VALU operations ahead of the step's MFMAs (VALU and MFMA share the issue port)
            const f32x4 s31 = pk_sub(d[3], d[1]);
            const f32x4 t02 = pk_sub(d[0], d[2]), p12 = pk_add(d[1], d[2]), m12 = pk_sub(d[1], d[2]), t42 = pk_sub(d[4], d[2]);
            const f32x4 two = {2.f, 2.f, 2.f, 2.f}, three = {3.f, 3.f, 3.f, 3.f};
            const f32x4 v0 = __builtin_elementwise_fma(t02, two, s31);        // v_pk_fma_f32
            const f32x4 v1 = pk_sub(s31, p12);
            const f32x4 v2 = __builtin_elementwise_fma(m12, three, s31);
            const f32x4 v4 = __builtin_elementwise_fma(s31, -two, t42);
            asm volatile("s_nop 1");         // VALU write -> MFMA source read distance, whatever the MFMA order below
            __builtin_amdgcn_sched_barrier(0);
            f32x4(&a)[5] = acc[k];
            const f32x4(&wm)[5] = w[m & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool first = (m == 0) && (e == 0);
                a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[0][e], v0[e], first ? zero4 : a[0], 0, 0, 0);
                a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[1][e], v1[e], first ? b4 : a[1], 0, 0, 0);
                a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[2][e], v2[e], first ? zero4 : a[2], 0, 0, 0);
                a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[3][e], s31[e], first ? zero4 : a[3], 0, 0, 0);
                a[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(wm[4][e], v4[e], first ? zero4 : a[4], 0, 0, 0);
            }
        }
        if constexpr (MODE == MODE_RESID_INPLACE && m == M - 1) {            // residual input of this tile's epilogue
#pragma unroll
            for (int r = 0; r < 3; ++r) res[r] = *(const f32x4*)(po[r] + k * TOFF);
        }
        if constexpr (u == NU - 1) {
            if (!PARTIAL || NT - 1 < live_tiles) epi_store(std::integral_constant<int, NT - 1>{});
        }
    });
}

// ---- bf16x3: a k = 3, stride 1, pad 1 convolution 64 -> 64 over a split image, on the bf16 matrix cores --------------
// x w ~= xh wh + (xh wl + xl wh): three v_mfma_f32_16x16x32_bf16 (K = 32 in 16 cycles) per 32-deep chunk of the direct
// form's K = 3 taps x 64 channels, against v_mfma_f32_16x16x4_f32's K = 4 in 32 cycles; the dropped xl wl and the split
// residues are ~2^-17 of a product.  The hh terms and the cross terms accumulate in separate registers (bias in the first).
// A wave owns one of the four 16-channel blocks and walks the compact image's 9 tiles of 16 rows; lane (j, q) supplies
// channels 32 h + 8 q .. + 7 of row 16 t + j + tap at step s = 2 tap + h (one ds_read_b128 per part) and ends with
// channels 16 cb + 4 q .. + 3 of row 16 t + j, as in the fp32 layers.  Rows of different reads touch (compact image):
// the lanes whose tap would cross a read boundary feed zeros, like conv_layer's BMASK.
// The RESIDUAL STREAM STAYS IN REGISTERS in fp32 (`xres`, one float4 per tile: every layer maps the same lane to the same
// outputs): a block's output is relu(conv) + xres exactly, and only the convolutions' INPUTS are rounded to 16 bits, so
// the rounding does not compound through the blocks.
//   BF_PLAIN   out = relu(conv(in))                      (a block's first convolution)
//   BF_RESID   xres = relu(conv(in)) + xres; out = xres  (its second one; the strided block's second conv with xres =
//              the 1x1 shortcut).  OUT_F32: the last layer writes fp32 (the SW_3 image the segment sum reads).
// `wh` / `wl`: this wave's weights, [6 steps] x 8 bf16 per lane; unless LAST they are refilled in place with the next
// layer's (`next_w`: this wave's block and lane, [6 steps][hi | lo][64 lanes][8]) after their last use.
enum { BF_PLAIN = 0, BF_RESID = 1 };
// Operand requests run BF16_DEPTH steps ahead of the MFMAs that consume them (measured 1 / 2 / 3 / 4: 8.15 / 7.91 / 7.95 /
// 7.93 ms for the kernel), and a tile's epilogue is deferred behind the first MFMAs of the next tile (7.75 -> 7.68 ms);
// profiles/r03_bf16x3_experiments.txt has the ablations (without its MFMAs the kernel is no faster: the layers are bound by
// their LDS operand stream and VALU epilogues, not by the matrix pipe)
constexpr int BF16_DEPTH = 2;
constexpr bool BF16_DEFER = true;
template <class CF, int MODE, bool LAST, bool OUT_F32>
__device__ __forceinline__ void bf16x3_layer(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                             bf16x8 (&wh)[6], bf16x8 (&wl)[6], const unsigned short* __restrict__ next_w,
                                             const float* __restrict__ bias, f32x4 (&xres)[CF::NSREG], int wave, int lane) {
    constexpr int RS = CF::RS2, NT = RS * CF::G / 16, NU = NT * 6;
    static_assert(CF::COMPACT && (RS * CF::G) % 16 == 0 && CF::NW == 4 && NT <= CF::NSREG, "the compact 150 bp geometry");
    const int cb = wave, j = lane & 15, q = lane >> 4;
    const f32x4 b4 = *(const f32x4*)(bias + cb * 16 + 4 * q);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // image row of (tile k, lane row j, tap) = 16 k + j + tap (flat row + 1 leading zero row - 1 pad): the swizzle term
    // depends on (j + tap) & 7 only, so a tile adds a constant
    const unsigned char* ph[6];
    const unsigned char* pl[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        ph[s] = in + split_off(j + s / 2, 4 * (s & 1) + q, 0);
        pl[s] = in + split_off(j + s / 2, 4 * (s & 1) + q, 1);
    }
    const unsigned char* const zrow = in + 16 * q;          // the image's leading zero row (both parts of it are zeros)
    constexpr int DEPTH = BF16_DEPTH;            // operand requests in flight ahead of the MFMAs that consume them
    bf16x8 rh[DEPTH + 1], rl[DEPTH + 1];
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / 6, s = u % 6, tap = s / 2;
        const unsigned char* a_h = ph[s] + k * 16 * 256;
        const unsigned char* a_l = pl[s] + k * 16 * 256;
        // read boundaries inside the tile: flat row 16 k + jj opens a read -> its tap 0 is padding; closes one -> its tap 2
        // is.  Those lanes fetch the zero row instead (one select per address, not eight per operand).
        constexpr int first = ((16 * k + RS - 1) / RS) * RS - 16 * k, last = ((16 * k + RS) / RS) * RS - 1 - 16 * k;
        if constexpr (tap == 0 && 16 * k + first > 0 && first >= 0 && first < 16) {
            a_h = (j == first) ? zrow : a_h;
            a_l = (j == first) ? zrow : a_l;
        }
        if constexpr (tap == 2 && last < 16 && 16 * k + last < RS * CF::G - 1) {
            a_h = (j == last) ? zrow : a_h;
            a_l = (j == last) ? zrow : a_l;
        }
        rh[u % (DEPTH + 1)] = *(const bf16x8*)a_h;
        rl[u % (DEPTH + 1)] = *(const bf16x8*)a_l;
    };
    // tile k accumulates in acc[k & 1]; its epilogue (sum, ReLU, residual, split, stores) runs behind the first MFMAs of
    // tile k + 1, whose accumulators are the other pair: no wait for the matrix pipe between tiles
    f32x4 acc_a[2] = {b4, b4}, acc_b[2] = {zero4, zero4};
    auto epilogue = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = CF::act(acc_a[k & 1][e] + acc_b[k & 1][e]);
        if constexpr (MODE == BF_RESID) {
            y = y + xres[k];
            xres[k] = y;
        }
        if constexpr (OUT_F32) *(f32x4*)((float*)out + img_off<64, SW_3>(16 * k + j + 1, 4 * cb + q)) = y;
        else store_split(out, 16 * k + j + 1, 4 * cb + q, y);
        acc_a[k & 1] = b4;
        acc_b[k & 1] = zero4;
    };
    static_for<0, DEPTH>(issue);
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / 6, s = u % 6;
        if constexpr (u + DEPTH < NU) issue(std::integral_constant<int, u + DEPTH>{});
        __builtin_amdgcn_sched_barrier(0);                    // requests stay ahead of the MFMAs, step by step
        const bf16x8 xh = rh[u % (DEPTH + 1)], xl = rl[u % (DEPTH + 1)];
        acc_a[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xh, acc_a[k & 1], 0, 0, 0);
        acc_b[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xl, acc_b[k & 1], 0, 0, 0);
        acc_b[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], xh, acc_b[k & 1], 0, 0, 0);
        if constexpr (!LAST && k == NT - 1) {                 // the next layer's weights roll in after their last use
            wh[s] = *(const bf16x8*)(next_w + (s * 2) * 512);
            wl[s] = *(const bf16x8*)(next_w + (s * 2 + 1) * 512);
        }
        if constexpr (BF16_DEFER) {
            if constexpr (s == 1 && k >= 1) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
            if constexpr (u == NU - 1) epilogue(std::integral_constant<int, NT - 1>{});
        } else if constexpr (u % (NU / NT) == NU / NT - 1) {
            epilogue(std::integral_constant<int, k>{});
        }
    });
}

// The same at 32 channels (the three ResidualBlock(32) ahead of the strided block): the image keeps its shared zero rows
// (row stride 72 = 71 + 1), which are the padding -- no boundary masks --, a tap is exactly one 32-deep chunk (3 steps of
// 3 MFMAs per tile), and the 18 tiles are split over 2 channel blocks x 2 position groups of waves (9 tiles each, tile
// pg + 2 k).  An output row that is a shared zero row is stored as zero and stays zero in the residual stream.
// OUT_F32: the last layer writes the fp32 SW_W image the strided convolution and its shortcut read.
template <class CF, int MODE, bool LAST, bool OUT_F32>
__device__ __forceinline__ void bf16x3_layer32(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                               bf16x8 (&wh)[3], bf16x8 (&wl)[3], const unsigned short* __restrict__ next_w,
                                               const float* __restrict__ bias, f32x4 (&xres)[CF::NSREG], int wave, int lane) {
    constexpr int RS = CF::RS1, NT = RS * CF::G / 32, NU = NT * 3;
    static_assert((RS * CF::G) % 32 == 0 && CF::NW == 4 && NT <= CF::NSREG, "18 tiles over 2 position groups");
    const int cb = wave % 2, pg = wave / 2, j = lane & 15, q = lane >> 4;
    const f32x4 b4 = *(const f32x4*)(bias + cb * 16 + 4 * q);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // image row of (tile pg + 2 k, lane row j, tap) = 16 (pg + 2 k) + j + tap; the swizzle has a period of 8 rows
    const unsigned char* ph[3];
    const unsigned char* pl[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        ph[s] = in + 16 * pg * 128 + split_off32(j + s, q, 0);
        pl[s] = in + 16 * pg * 128 + split_off32(j + s, q, 1);
    }
    unsigned zmask = 0;                          // bit k: this lane's row of the wave's k-th tile is a shared zero row
#pragma unroll
    for (int k = 0; k < NT; ++k) zmask |= ((((pg + 2 * k) * 16 + j) % RS) == RS - 1 ? 1u : 0u) << k;
    constexpr int DEPTH = BF16_DEPTH;
    bf16x8 rh[DEPTH + 1], rl[DEPTH + 1];
    auto issue = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / 3, s = u % 3;
        rh[u % (DEPTH + 1)] = *(const bf16x8*)(ph[s] + k * 32 * 128);
        rl[u % (DEPTH + 1)] = *(const bf16x8*)(pl[s] + k * 32 * 128);
    };
    f32x4 acc_a[2] = {b4, b4}, acc_b[2] = {zero4, zero4};
    auto epilogue = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = CF::act(acc_a[k & 1][e] + acc_b[k & 1][e]);
        if constexpr (MODE == BF_RESID) y = y + xres[k];
        if ((zmask >> k) & 1u) y = zero4;
        if constexpr (MODE == BF_RESID) xres[k] = y;
        const int row = 16 * (pg + 2 * k) + j + 1;
        if constexpr (OUT_F32) *(f32x4*)((float*)out + img_off<32, SW_W>(row, 4 * cb + q)) = y;
        else store_split32(out, row, 4 * cb + q, y);
        acc_a[k & 1] = b4;
        acc_b[k & 1] = zero4;
    };
    static_for<0, DEPTH>(issue);
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int k = u / 3, s = u % 3;
        if constexpr (u + DEPTH < NU) issue(std::integral_constant<int, u + DEPTH>{});
        __builtin_amdgcn_sched_barrier(0);                    // requests stay ahead of the MFMAs, step by step
        const bf16x8 xh = rh[u % (DEPTH + 1)], xl = rl[u % (DEPTH + 1)];
        acc_a[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xh, acc_a[k & 1], 0, 0, 0);
        acc_b[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xl, acc_b[k & 1], 0, 0, 0);
        acc_b[k & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], xh, acc_b[k & 1], 0, 0, 0);
        if constexpr (!LAST && k == NT - 1) {
            wh[s] = *(const bf16x8*)(next_w + (s * 2) * 512);
            wl[s] = *(const bf16x8*)(next_w + (s * 2 + 1) * 512);
        }
        if constexpr (BF16_DEFER) {
            if constexpr (s == 1 && k >= 1) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
            if constexpr (u == NU - 1) epilogue(std::integral_constant<int, NT - 1>{});
        } else if constexpr (u % (NU / NT) == NU / NT - 1) {
            epilogue(std::integral_constant<int, k>{});
        }
    });
}

// ---- stem conv1: pileup bytes -> 16 channels (valid convolution over the stacked reads) ---------------
// The bytes are read as they are (no float copy): lane (j, q) of a tile needs k = 4*step + q of row j, and
// k = tap*C + c is the byte offset from the row start.  Tiles of a wave: t = wave + 4*i; pairs of tiles in
// flight; the bytes of the next pair are requested before the MFMAs of the current one.
template <class CF>
__device__ __forceinline__ void stem_conv1(const unsigned char* __restrict__ s_u8, float* __restrict__ c1,
                                           const float* __restrict__ W, int ch, float* __restrict__ dump, int wave,
                                           int lane) {
    using namespace rc;
    constexpr int NPAIR = CF::ST12 / 8;          // 4 waves x 2 tiles per pair
    static_assert(CF::ST12 % 8 == 0, "stem tiles must split into pairs over 4 waves");
    const int j = lane & 15, q = lane >> 4;
    float w1[S1_STEPS];
#pragma unroll
    for (int s = 0; s < S1_STEPS; ++s) w1[s] = W[s * 64 + lane];              // W = the conv1 block of the blob
    const f32x4 b4 = *(const f32x4*)(W + S1_STEPS * 64 + 4 * q);
    const unsigned char* base = s_u8 + (16 * wave + j) * ch + q;       // tile `wave`, row j, byte q
    const int tile_bytes = 16 * ch;
    unsigned char cur0[S1_STEPS], cur1[S1_STEPS], nxt0[S1_STEPS], nxt1[S1_STEPS];
#pragma unroll
    for (int s = 0; s < S1_STEPS; ++s) {
        cur0[s] = base[4 * s];
        cur1[s] = base[4 * tile_bytes + 4 * s];
    }
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        if (i + 1 < NPAIR) {
#pragma unroll
            for (int s = 0; s < S1_STEPS; ++s) {
                nxt0[s] = base[(8 * (i + 1)) * tile_bytes + 4 * s];
                nxt1[s] = base[(8 * (i + 1) + 4) * tile_bytes + 4 * s];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S1_STEPS; ++s) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s], (float)cur0[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s], (float)cur1[s], a1, 0, 0, 0);
        }
        const int r0 = 16 * (wave + 8 * i) + j, r1 = r0 + 64;
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = CF::act(a0[e] + b4[e]);
            v1[e] = CF::act(a1[e] + b4[e]);
        }
        *(f32x4*)(r0 < CF::SROWS ? c1 + r0 * 16 + 4 * (q ^ swz<16>(r0)) : dump) = v0;
        *(f32x4*)(r1 < CF::SROWS ? c1 + r1 * 16 + 4 * (q ^ swz<16>(r1)) : dump) = v1;
#pragma unroll
        for (int s = 0; s < S1_STEPS; ++s) {
            cur0[s] = nxt0[s];
            cur1[s] = nxt1[s];
        }
    }
}

// ---- stem conv3 (16 -> 32, valid) + ReLU + MaxPool1d(3, 2), one read per wave ------------------------------
// A tile is 16 conv3 positions of ONE read, tiles 14 apart (11 per 150 bp read): lane row j holds position 14 t + j, so
// the pooled outputs 7 t .. 7 t + 6 = max over positions (2p, 2p+1, 2p+2) are two DPP row shifts away and only
// pooled values reach LDS (image row 1 + read*72 + p).  The wave computes BOTH 16-channel blocks of its tiles:
// the three operand reads of a tile feed 24 MFMAs (4 chains), and one output-row computation serves both
// blocks.  Everything a tile adds to the lane's addresses is a compile-time constant: the 16-channel image's
// swizzle bit depends on (row + c) mod 8 only, so 8 per-lane base pointers cover every (tile, tap).
// The bias starts one accumulation chain and ReLU is applied AFTER the max (monotone, so the same value).
template <class CF, int SOUT>
__device__ __forceinline__ void stem_conv3_pool(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ W3, float* __restrict__ dump, int wave,
                                                int lane, int n_here) {
    constexpr int WPR = CF::WPR;                              // waves per read: wave (rd, half) takes tiles half + WPR k
    constexpr int NTT = CF::NTT / WPR;                        // tiles of this wave
    static_assert(CF::NTT % WPR == 0, "the pool tiles of a read must split evenly over its waves");
    const int j = lane & 15, q = lane >> 4;
    const int rd = wave / WPR, half = wave % WPR;
    f32x4 w[2][3];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) w[blk][tap] = *(const f32x4*)(W3 + ((blk * 3 + tap) * 64 + lane) * 4);
    f32x4 b4[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) b4[blk] = *(const f32x4*)(W3 + 2 * 3 * 256 + blk * 16 + 4 * q);
    const int lrow = rd * CF::WINDOW + 14 * half + j;
    const float* pin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pin[k] = in + lrow * 16 + 4 * (q ^ (2 * (((lrow + k) >> 2) & 1)));
    const int orow = 1 + rd * CF::RS1 + 7 * half + (j >> 1);   // pooled position 7 * (half + WPR k) + (j >> 1)
    const bool lane_ok = ((j & 1) == 0) && (j <= 12) && (rd < n_here);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 ring[2][3];
    f32x4 acc[2][2][2];                                       // [tile parity][block][chain]
    auto issue = [&](auto tc) {
        constexpr int tt = decltype(tc)::value;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            constexpr int c0 = 14 * WPR * tt;
            ring[tt & 1][tap] = *(const f32x4*)(pin[(c0 + tap) & 7] + (c0 + tap) * 16);
        }
    };
    auto epilogue = [&](auto tc) {
        constexpr int tt = decltype(tc)::value;
        const int row = orow + 7 * WPR * tt;
        const bool ok = lane_ok && (row - 1 - rd * CF::RS1 < CF::L1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4 sum = acc[tt & 1][blk][0] + acc[tt & 1][blk][1];      // bias included: chain 0 started from b4
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // lanes 14 / 15 of a row see zeros shifted in; their results are never stored (even lanes <= 12 are)
                v[e] = CF::act(fmaxf(fmaxf(sum[e], row_shl(sum[e], 1)), row_shl(sum[e], 2)));
            }
            float* ptr = out + img_off<32, SOUT>(row, 4 * blk + q);
            *(f32x4*)(ok ? ptr : dump) = v;
        }
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NTT>([&](auto tc) {
        constexpr int tt = decltype(tc)::value;
        if constexpr (tt + 1 < NTT) issue(std::integral_constant<int, tt + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (tt >= 1) epilogue(std::integral_constant<int, (tt >= 1 ? tt - 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const f32x4 x = ring[tt & 1][tap];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool first = (tap == 0) && (e < 2);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
                    acc[tt & 1][blk][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        w[blk][tap][e], x[e], first ? (e == 0 ? b4[blk] : zero4) : acc[tt & 1][blk][e & 1], 0, 0, 0);
            }
        }
    });
    epilogue(std::integral_constant<int, NTT - 1>{});
}

// ---- stem conv2 (16 -> 16, valid) in Winograd F(2,3) form over the stacked reads ----------------------------
// Tile T = 16 pairs of rows 32 T + 2 j (+1); wave w takes tiles w, w + NW, ...  Pairs whose window straddles two
// reads produce rows nothing valid consumes, exactly as in the direct form.
template <class CF>
__device__ __forceinline__ void stem_conv2_wino(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ W2, float* __restrict__ dump, int wave,
                                                int lane) {
    constexpr int NT = (CF::SROWS / 2 + 15) / 16;             // tiles of 16 pairs
    constexpr int NK = (NT + CF::NW - 1) / CF::NW;            // tiles of wave 0
    const int j = lane & 15, q = lane >> 4;
    f32x4 w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = *(const f32x4*)(W2 + (c * 64 + lane) * 4);
    const f32x4 b4 = *(const f32x4*)(W2 + 4 * 256 + 4 * q);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* pin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pin[i] = in + 32 * wave * 16 + img_off<16, SW_OLD>(2 * j + i, q);
    float* const o0 = out + 32 * wave * 16 + img_off<16, SW_OLD>(2 * j, q);
    float* const o1 = out + 32 * wave * 16 + img_off<16, SW_OLD>(2 * j + 1, q);
    constexpr int TOFF = CF::NW * 32 * 16;                    // floats between a wave's tiles
    const bool last_here = wave + CF::NW * (NK - 1) < NT;     // wave-uniform

    f32x4 ring[2][4], acc[2][4];
    auto issue = [&](auto tc) {
        constexpr int k = decltype(tc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[k & 1][i] = *(const f32x4*)(pin[i] + k * TOFF);
    };
    auto epilogue = [&](auto tc) {
        constexpr int k = decltype(tc)::value;
        const f32x4(&a)[4] = acc[k & 1];
        f32x4 y0 = (a[0] + a[1]) + a[2], y1 = (a[1] - a[2]) - a[3];       // the bias rides in a[1]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y0[e] = CF::act(y0[e]);
            y1[e] = CF::act(y1[e]);
        }
        const int row = 32 * (wave + CF::NW * k) + 2 * j;
        *(f32x4*)(row < CF::SROWS ? o0 + k * TOFF : dump) = y0;
        *(f32x4*)(row + 1 < CF::SROWS ? o1 + k * TOFF : dump) = y1;
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NK>([&](auto tc) {
        constexpr int k = decltype(tc)::value;
        constexpr bool tail = (k == NK - 1) && (NT % CF::NW != 0);
        if constexpr (k + 1 < NK) issue(std::integral_constant<int, k + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k >= 1) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
        if (!tail || last_here) {
            const f32x4 d0 = ring[k & 1][0], d1 = ring[k & 1][1], d2 = ring[k & 1][2], d3 = ring[k & 1][3];
            const f32x4 t0 = pk_sub(d0, d2), t1 = pk_add(d1, d2), t2 = pk_sub(d2, d1), t3 = pk_sub(d1, d3);
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
            f32x4(&a)[4] = acc[k & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][e], t0[e], e == 0 ? zero4 : a[0], 0, 0, 0);
                a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][e], t1[e], e == 0 ? b4 : a[1], 0, 0, 0);
                a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][e], t2[e], e == 0 ? zero4 : a[2], 0, 0, 0);
                a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3][e], t3[e], e == 0 ? zero4 : a[3], 0, 0, 0);
            }
        }
        if constexpr (k == NK - 1) {
            if (!tail || last_here) epilogue(std::integral_constant<int, NK - 1>{});
        }
    });
}

// ---- stem conv3 + pool in Winograd F(2,3) form -------------------------------------------------------------
// Lane row j of a tile holds the PAIR of conv3 positions (2P, 2P+1), P = 15 t + j: y0 = conv(2P), y1 = conv(2P+1)
// come out of four contractions of d0..d3 = rows 2P..2P+3 (valid convolution: no padding), and the pooled
// output P = max(conv(2P), conv(2P+1), conv(2P+2)) = max(y0, y1, y0 of lane j+1) needs ONE lane shift.  Tiles
// are 15 pairs apart (lane 15's pair is recomputed as lane 0 of the next tile): 5 tiles of 32 MFMAs per 150 bp
// read instead of 11 tiles of 24, and 15 of 16 lane rows store a pooled row instead of 7.
template <class CF, int SOUT>
__device__ __forceinline__ void stem_conv3_pool_wino(const float* __restrict__ in, float* __restrict__ out,
                                                     const float* __restrict__ W3, float* __restrict__ dump, int wave,
                                                     int lane, int n_here) {
    constexpr int WPR = CF::WPR;                              // waves per read: wave (rd, half) takes tiles half + WPR k
    constexpr int NTR = (CF::L1 + 14) / 15;                   // tiles per read
    constexpr int NK = (NTR + WPR - 1) / WPR;                 // tiles of the first wave of a read
    const int j = lane & 15, q = lane >> 4;
    const int rd = wave / WPR, half = wave % WPR;
    f32x4 w[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[blk][c] = *(const f32x4*)(W3 + ((blk * 4 + c) * 64 + lane) * 4);
    f32x4 b4[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) b4[blk] = *(const f32x4*)(W3 + 2 * 4 * 256 + blk * 16 + 4 * q);
    const int lrow = rd * CF::WINDOW + 30 * half + 2 * j;     // row of d0 in the wave's first tile
    const float* pin[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pin[k] = in + lrow * 16 + 4 * (q ^ (2 * (((lrow + k) >> 2) & 1)));
    const int opos = 15 * half + j;                            // pooled position in the wave's first tile
    const bool lane_ok = (j <= 14) && (rd < n_here);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool last_here = half + WPR * (NK - 1) < NTR;        // wave-uniform: the last tile exists for this wave

    f32x4 ring[2][4];
    f32x4 acc[2][2][4];                                       // [tile parity][block][component]
    auto issue = [&](auto tc) {
        constexpr int k = decltype(tc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            constexpr int c0 = 30 * WPR * k;
            ring[k & 1][i] = *(const f32x4*)(pin[(c0 + i) & 7] + (c0 + i) * 16);
        }
    };
    auto epilogue = [&](auto tc) {
        constexpr int k = decltype(tc)::value;
        const int pos = opos + 15 * WPR * k;
        const bool ok = lane_ok && (pos < CF::L1);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4(&a)[4] = acc[k & 1][blk];
            const f32x4 y0 = (a[0] + a[1]) + a[2];             // the bias rides in a[1]
            const f32x4 y1 = (a[1] - a[2]) - a[3];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = CF::act(fmaxf(fmaxf(y0[e], y1[e]), row_shl(y0[e], 1)));
            float* ptr = out + img_off<32, SOUT>(1 + rd * CF::RS1 + pos, 4 * blk + q);
            *(f32x4*)(ok ? ptr : dump) = v;
        }
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NK>([&](auto tc) {
        constexpr int k = decltype(tc)::value;
        constexpr bool tail = (k == NK - 1) && (NTR % WPR != 0);
        if constexpr (k + 1 < NK) issue(std::integral_constant<int, k + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k >= 1) epilogue(std::integral_constant<int, (k >= 1 ? k - 1 : 0)>{});
        if (!tail || last_here) {
            const f32x4 d0 = ring[k & 1][0], d1 = ring[k & 1][1], d2 = ring[k & 1][2], d3 = ring[k & 1][3];
            const f32x4 t0 = pk_sub(d0, d2), t1 = pk_add(d1, d2), t2 = pk_sub(d2, d1), t3 = pk_sub(d1, d3);
            asm volatile("s_nop 1");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    f32x4(&a)[4] = acc[k & 1][blk];
                    a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[blk][0][e], t0[e], e == 0 ? zero4 : a[0], 0, 0, 0);
                    a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[blk][1][e], t1[e], e == 0 ? b4[blk] : a[1], 0, 0, 0);
                    a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[blk][2][e], t2[e], e == 0 ? zero4 : a[2], 0, 0, 0);
                    a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[blk][3][e], t3[e], e == 0 ? zero4 : a[3], 0, 0, 0);
                }
        }
        if constexpr (k == NK - 1) {
            if (!tail || last_here) epilogue(std::integral_constant<int, NK - 1>{});
        }
    });
}

// STAMP (diagnostic instantiation, hello_engine_debug_stamps; never the product launch): every wave records s_memtime at the
// start of each group and on both sides of each of the group's 20 barriers into ReadConvArgs::stamps -- memory nothing else
// reads; no output value is computed from a stamp.  Layout: [workgroup][wave][group < stamp_groups][STAMP_SLOTS] of uint64:
// slot 0 = group start, 1 + 2 i = arrival at barrier i, 2 + 2 i = release from it, 41 = HW_ID | XCC_ID << 32, 42 / 43 =
// s_memrealtime at the group's start / end (100 MHz: the in-kernel clock), 44 = after the workgroup's last flush.
template <class CF, bool STEM, int NB64, bool WINO, bool BF16 = false, bool BF16_32 = false, bool STAMP = false>
__global__ __launch_bounds__(CF::THREADS, 2) void readconv_kernel(ReadConvArgs a) {
    using namespace rc;
    constexpr bool F33 = WINO && CF::F33;                     // 64-channel residual blocks in F(3,3) form
    // BF16 (arithmetic mode "bf16x3", never the default): the seven (eleven) 64 -> 64 convolutions run on the bf16 matrix
    // cores as 3-term splits (bf16x3_layer); everything before them -- stem, 32-channel blocks, the strided convolution
    // and its shortcut -- and the per-allele sums stay exact fp32
    static_assert(!BF16 || (F33 && CF::ACT == ACT_RELU), "bf16x3: the 150 bp ReLU geometry of the Winograd kernel");
    // BF16_32 ("bf16x3+32"): the six 32 -> 32 convolutions of the ResidualBlock(32)s too (otherwise exact fp32 F(3,3))
    static_assert(!BF16_32 || BF16, "the 32-channel split layers extend the bf16x3 mode");
    using O = Offs<WINO, F33>;
    constexpr int L1 = CF::L1, RS1 = CF::RS1, L2 = CF::L2, RS2 = CF::RS2;
    static_assert(CF::COMPACT || WINO, "the zero-row 64-channel geometry is implemented for the Winograd form only");
    constexpr int W3232 = O::W3232, W3264 = O::W3264, W3264S = O::W3264S, W6464 = O::W6464;
    constexpr int OFF_B = O::OFF_B, OFF_C1 = O::OFF_C1, OFF_SC = O::OFF_SC, OFF_C2 = O::OFF_C2;
    constexpr int OFF_S1 = O::OFF_S1, OFF_S2 = O::OFF_S2, OFF_S3 = O::OFF_S3;
    constexpr int SWX = WINO ? SW_W : SW_OLD;                 // swizzle of the images the residual blocks walk
    constexpr int NVA = WINO ? 8 : 6, NVB = WINO ? 16 : 12;   // weight registers of a 32- / 64-channel block conv
    constexpr int G = CF::G, T1 = CF::T1, T2 = CF::T2, BUF_FLOATS = CF::BUF_FLOATS, THREADS = CF::THREADS;
    static_assert(CF::NW == 4, "the layer schedule below maps 4 waves to (2 blocks x 2 groups) / (4 blocks)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;
    float* bufB = smem + BUF_FLOATS;
    float* dump = smem + 2 * BUF_FLOATS + 12;                 // 16 spare bytes of a 64-byte block between the images and the staged bytes
    unsigned char* s_u8 = (unsigned char*)(smem + 2 * BUF_FLOATS + 16);

    const int tid0 = threadIdx.x;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const float* __restrict__ W = a.w;

    // which of a lane's rows are shared zero rows: bit k = the wave's k-th tile, per image geometry
    unsigned pad1 = 0, pad1f = 0;                             // pad1f: the same for a FLIPped Winograd layer
    unsigned pad2 = 0, pad2w = 0;                             // 64 channels: direct 16-row tiles / Winograd tiles
    if (!CF::COMPACT) {                                       // (a compact image holds no zero rows between reads)
        const int j = tid0 & 15;
#pragma unroll
        for (int k = 0; k < T2; ++k) pad2 |= (((k * 16 + j) % RS2) >= L2 ? 1u : 0u) << k;
#pragma unroll
        for (int k = 0; k < (RS2 * G / 2 + 15) / 16; ++k) pad2w |= (((k * 16 + j) % (RS2 / 2)) == RS2 / 2 - 1 ? 1u : 0u) << k;
    }
    {
        const int j = tid0 & 15;
        const int pg1 = wave0 / 2;                             // 32-channel layers: 2 blocks x 2 position groups
        if (WINO) {                                            // tile = 16 pairs of rows; the odd row of pair P = 35 mod 36
#pragma unroll
            for (int k = 0; k < (T1 / 2 + 1) / 2; ++k) {
                pad1 |= ((((pg1 + 2 * k) * 16 + j) % (RS1 / 2)) == RS1 / 2 - 1 ? 1u : 0u) << k;
                pad1f |= ((((1 - pg1 + 2 * k) * 16 + j) % (RS1 / 2)) == RS1 / 2 - 1 ? 1u : 0u) << k;
            }
        } else {
#pragma unroll
            for (int k = 0; k < T1 / 2; ++k) pad1 |= ((((pg1 + 2 * k) * 16 + j) % RS1) >= L1 ? 1u : 0u) << k;
        }
    }

    // X = the trunk's input image [71][32] per read (shared zero rows and row 0 zero), H = the other image
    float* const X = STEM ? bufB : bufA;
    float* const H = STEM ? bufA : bufB;
    f32x4 sreg[CF::NSREG];
    f32x4 wA[NVA], wB[NVB], w2[2];
    f32x4 w3[2][5];                                           // F33: one input group's weights, current / next
    f32x4 (&w6)[6] = reinterpret_cast<f32x4 (&)[6]>(wA);       // the direct-form views of the same registers
    f32x4 (&w12)[12] = reinterpret_cast<f32x4 (&)[12]>(wB);

    // A workgroup walks `groups_per_wg` consecutive groups of G reads and carries the running per-allele sum
    // of its reads in registers across them: a partial slot is written only when the allele changes (or at
    // the workgroup's end), i.e. one slot per (workgroup, allele) incidence instead of per (group, allele).
    const long long wg_read0 = (long long)blockIdx.x * G * a.groups_per_wg;
    const int slot0 = a.slot_of_group[blockIdx.x];
    const int first_allele = a.allele_of_read[wg_read0];
    int cur = first_allele;                                   // allele whose reads `carry` holds (uniform)
    constexpr int NF = (L2 * 16 + THREADS - 1) / THREADS;     // float4 elements of a [36][64] frame per thread
    f32x4 carry[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) carry[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid0 + THREADS * i;
            if (f < L2 * 16)
                *(f32x4*)(a.partial + ((long long)(slot0 + cur - first_allele) * L2 + (f >> 4)) * 64 + 4 * (f & 15)) = carry[i];
            carry[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    for (int grp = 0; grp < a.groups_per_wg; ++grp) {
    // Every per-lane address of the body derives from these three; an opaque zero ties them to the iteration, so
    // the compiler recomputes them per group instead of keeping every layer's pointers alive across the loop
    int opaque_zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(opaque_zero));
    const int tid = tid0 + opaque_zero;
    const int wave = __builtin_amdgcn_readfirstlane(wave0 + opaque_zero);
    const int lane = tid & 63;
    const int cb2 = wave % 2, cb4 = wave;
    const long long read0 = wg_read0 + (long long)grp * G;
    if (read0 >= a.n_reads) break;                            // uniform: the last workgroup may hold fewer groups
    unsigned long long* const stamp_base = STAMP ? a.stamps + (((long long)blockIdx.x * CF::NW + wave) * a.stamp_groups + grp) * STAMP_SLOTS : nullptr;
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (grp < a.stamp_groups && lane == 0) stamp_base[slot] = t;
        }
    };
    auto barrier = [&](int id) {                              // barrier `id` of the group (0 .. 19 in the fp32 Winograd schedule)
        stamp(1 + 2 * id);
        __syncthreads();
        stamp(2 + 2 * id);
    };
    if constexpr (STAMP) {
        stamp(0);
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (grp < a.stamp_groups && lane == 0) {
            stamp_base[41] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
            stamp_base[42] = rt;
        }
    }
    const int n_here = (int)((a.n_reads - read0) < G ? (a.n_reads - read0) : G);
    // first trunk layer: requested before anything else waits
    if constexpr (F33) {
        static_assert(!F33 || CF::F33_32, "the F(3,3) kernel runs its 32-channel blocks in F(3,3) form too");
#pragma unroll
        for (int c = 0; c < 5; ++c) w3[0][c] = *(const f32x4*)(W + OFF_B + cb2 * (10 * 256) + lane * 4 + c * 256);
    } else {
        load_weights<NVA>(wA, W + OFF_B, cb2, lane);
    }
    if (STEM) {
        // the stem, from the uint8 pileups: conv1 bytes -> bufB, conv2 bufB -> bufA, conv3 + max pool
        // bufA -> X (= bufB again, now as the 32-channel image)
        const int ch = a.channels;
        const int n_bytes = n_here * CF::WINDOW * ch;
        const unsigned char* src = a.reads + read0 * CF::WINDOW * ch;
        // The group's bytes in ONE round trip to memory: every thread requests all its dwords first and
        // stores them afterwards (a load-store loop would pay the memory latency once per iteration and
        // leave the workgroup memory-bound for longer than all its MFMAs take).
        constexpr int NDW = (CF::U8_BYTES / 4 + THREADS - 1) / THREADS;
        if ((reinterpret_cast<unsigned long long>(src) & 3ull) == 0) {
            unsigned v[NDW];
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + THREADS * k;                  // dword index inside the staging buffer
                unsigned x = 0;
                if (4 * d + 4 <= n_bytes) {
                    x = ((const unsigned*)src)[d];
                } else if (4 * d < n_bytes) {                     // the last, partial dword (7-channel reads)
                    for (int b = 0; b < n_bytes - 4 * d; ++b) x |= (unsigned)src[4 * d + b] << (8 * b);
                }
                v[k] = x;
            }
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + THREADS * k;
                if (4 * d < CF::U8_BYTES) ((unsigned*)s_u8)[d] = v[k];
            }
        } else {
            for (int i = tid; i < CF::U8_BYTES; i += THREADS) s_u8[i] = (i < n_bytes) ? src[i] : (unsigned char)0;
        }
        f32x4 ws2[3];
        if constexpr (!WINO) load_weights<3>(ws2, W + OFF_S2, 0, lane);
        barrier(0);
        stem_conv1<CF>(s_u8, bufB, W + OFF_S1, ch, dump, wave, lane);
        barrier(1);
        if constexpr (WINO) stem_conv2_wino<CF>(bufB, bufA, W + OFF_S2, dump, wave, lane);
        else conv_layer<CF, 16, 16, 3, 1, 0, 1, 1, 0, CF::ST12, MODE_PLAIN, false, GEOM_STEM, 16, CF::SROWS>(
            bufB, bufA, ws2, nullptr, W + OFF_S2 + 3 * 256, sreg, 0u, dump, wave, lane);
        barrier(2);
        if (tid < 8 * (G + 1)) {                      // the image's shared zero rows: 0, 72, 144, ...
            const int row = (tid >> 3) * RS1;
            *(f32x4*)(X + img_off<32, SWX>(row, tid & 7)) = f32x4{0.f, 0.f, 0.f, 0.f};   // a whole row, chunk by chunk
        }
        if constexpr (WINO) stem_conv3_pool_wino<CF, SWX>(bufA, X, W + OFF_S3, dump, wave, lane, n_here);
        else stem_conv3_pool<CF, SWX>(bufA, X, W + OFF_S3, dump, wave, lane, n_here);
        barrier(3);
        if (tid < 8) ((f32x4*)H)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};        // row 0 of the 32-channel image
    } else {
        for (int i = tid; i < BUF_FLOATS / 4; i += THREADS) ((f32x4*)bufA)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid < 16) ((f32x4*)bufB)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        barrier(0);
        const f32x4* src = (const f32x4*)(a.pooled + read0 * (L1 * 32));
        const int n4 = n_here * L1 * 8;
        for (int f = tid; f < n4; f += THREADS) {
            const int rd = f / (L1 * 8);
            const int rem = f - rd * (L1 * 8);
            const int p = rem >> 3, c = rem & 7;
            const int row = 1 + rd * RS1 + p;
            *(f32x4*)(bufA + img_off<32, SWX>(row, c)) = src[f];
        }
    }

    auto slice = [&](int off, int cb, int nv) { return W + off + cb * nv * 256 + lane * 4; };   // this wave's block, this lane

    // ---- 3 x ResidualBlock(32): x -> relu(conv) -> relu(conv) + x --------------------------------
    // With the in-kernel stem nothing is pending here but wave 0's zeros for row 0 of H, which no wave reads before the next
    // barrier (the first convolution reads X and writes rows >= 1 of H): no barrier.  Without the stem, X was just filled.
    if constexpr (STEM && !BF16_32) {
        stamp(1 + 2 * 4);
        stamp(2 + 2 * 4);
    } else {
        barrier(4);
    }
    // bf16x3: this wave's split weights of the 32-channel layers, [6 layers][2 blocks][3 taps][hi | lo][64 lanes][8] behind
    // the 64-channel layers' block
    const unsigned short* const WS32 = (const unsigned short*)(W + O::off_d(NB64)) + (1 + 2 * NB64) * 24576 + cb2 * 3072 + lane * 8;
    bf16x8 cwh[3], cwl[3];
    if constexpr (BF16_32) {
        // the stem's fp32 output becomes the residual stream (registers) and, in place, the split image the first
        // convolution reads: every lane picks up the float4s it will own in every 32-channel layer
        const int j = lane & 15, q = lane >> 4, pg = wave / 2;
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            cwh[st] = *(const bf16x8*)(WS32 + (st * 2) * 512);
            cwl[st] = *(const bf16x8*)(WS32 + (st * 2 + 1) * 512);
        }
#pragma unroll
        for (int k = 0; k < RS1 * G / 32; ++k)
            sreg[k] = *(const f32x4*)(X + img_off<32, SWX>(16 * (pg + 2 * k) + j + 1, 4 * cb2 + q));
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS1 * G / 32; ++k) store_split32((unsigned char*)X, 16 * (pg + 2 * k) + j + 1, 4 * cb2 + q, sreg[k]);
        __syncthreads();
    }
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        const int off_a = OFF_B + (2 * blk) * (O::W3232D + 32), off_b = off_a + (O::W3232D + 32);
        if constexpr (BF16_32) {
            bf16x3_layer32<CF, BF_PLAIN, false, false>((const unsigned char*)X, (unsigned char*)H, cwh, cwl,
                                                       WS32 + (2 * blk + 1) * 6144, W + off_a + O::W3232D, sreg, wave, lane);
            __syncthreads();
            if (blk < 2)
                bf16x3_layer32<CF, BF_RESID, false, false>((const unsigned char*)H, (unsigned char*)X, cwh, cwl,
                                                           WS32 + (2 * blk + 2) * 6144, W + off_b + O::W3232D, sreg, wave, lane);
            else
                bf16x3_layer32<CF, BF_RESID, true, true>((const unsigned char*)H, (unsigned char*)X, cwh, cwl, nullptr,
                                                         W + off_b + O::W3232D, sreg, wave, lane);
            __syncthreads();
            continue;
        }
        // the block's second conv rolls in the next block's first conv, or the strided conv (4 channel blocks;
        // its 6 registers are the first 6 of the 8 a Winograd layer refills)
        const float* nxt = (blk < 2) ? slice(off_b + (W3232 + 32), cb2, NVA) : slice(OFF_C1, cb4, 6);
        if constexpr (F33) {
            // this wave's block of a 32-channel F(3,3) layer, this lane: [2 input groups][5 components][64 lanes][4]
            auto slice32 = [&](int off) { return W + off + cb2 * (10 * 256) + lane * 4; };
            wino3_layer<CF, 32, MODE_PLAIN, false>(X, H, w3, slice32(off_a), slice32(off_b), W + off_a + O::W3232D, wave, lane);
            barrier(5 + 2 * blk);
            if (blk < 2)
                wino3_layer<CF, 32, MODE_RESID_INPLACE, false>(H, X, w3, slice32(off_b), slice32(off_b + (O::W3232D + 32)),
                                                               W + off_b + O::W3232D, wave, lane);
            else
                wino3_layer<CF, 32, MODE_RESID_INPLACE, true>(H, X, w3, slice32(off_b), nullptr, W + off_b + O::W3232D, wave, lane);
        } else if constexpr (WINO) {
            wino_layer<CF, 32, MODE_PLAIN, true>(X, H, wA, slice(off_b, cb2, NVA), W + off_a + W3232, pad1, dump, wave, lane);
            barrier(5 + 2 * blk);
            wino_layer<CF, 32, MODE_RESID_INPLACE, true, true>(H, X, wA, nxt, W + off_b + W3232, pad1f, dump, wave, lane);
        } else {
            conv_layer<CF, 32, 32, 3, 1, 1, RS1, RS1, L1, T1, MODE_PLAIN, true>(
                X, H, w6, slice(off_b, cb2, 6), W + off_a + W3232, sreg, pad1, dump, wave, lane);
            barrier(5 + 2 * blk);
            conv_layer<CF, 32, 32, 3, 1, 1, RS1, RS1, L1, T1, MODE_RESID_INPLACE, true>(
                H, X, w6, nxt, W + off_b + W3232, sreg, pad1, dump, wave, lane);
        }
        barrier(6 + 2 * blk);
    }

    // ---- strided block 32 -> 64: relu(conv s2) -> relu(conv) + (1x1 s2 shortcut) ----------------
    if constexpr (F33) {                     // (otherwise rolled in by the last 32-channel layer)
        load_weights<6>(w6, W + OFF_C1, cb4, lane);
    } else {
        load_weights<NVB>(wB, W + OFF_C2, cb4, lane);
    }
    load_weights<2>(w2, W + OFF_SC, cb4, lane);
    if (tid < 32) ((f32x4*)H)[(tid & 15) + (tid >> 4) * (RS2 * G + 1) * 16] = f32x4{0.f, 0.f, 0.f, 0.f};   // rows 0 and 36G+1
    conv_layer<CF, 32, 64, 3, 2, 1, RS1, RS2, L2, T2, MODE_PLAIN, false, GEOM_TRUNK, 16, RS2 * CF::G, false, SWX,
               BF16 ? SW_SPLIT : (F33 ? SW_3 : SWX)>(X, H, w6, nullptr, W + OFF_C1 + W3264, sreg, pad2, dump, wave, lane);
    // bf16x3: this wave's split weights of trunk layer l (0 = the strided block's second conv, then the blocks' convs):
    // [layer][4 blocks][6 steps][hi | lo][64 lanes][8 bf16] behind the fp32 blob
    const unsigned short* const WS16 = (const unsigned short*)(W + O::off_d(NB64)) + cb4 * 6144 + lane * 8;
    bf16x8 bwh[6], bwl[6];
    if constexpr (BF16) {
        // the shortcut per 16-row tile, in registers: it becomes the fp32 residual stream of the whole 64-channel trunk
        conv_layer<CF, 32, 64, 1, 2, 0, RS1, RS2, L2, T2, MODE_TO_REGS, false, GEOM_TRUNK, 16, RS2 * CF::G, false, SWX, SWX>(
            X, nullptr, w2, nullptr, W + OFF_SC + W3264S, sreg, pad2, dump, wave, lane);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            bwh[st] = *(const bf16x8*)(WS16 + (st * 2) * 512);
            bwl[st] = *(const bf16x8*)(WS16 + (st * 2 + 1) * 512);
        }
    } else if constexpr (F33) {
        // the shortcut in the row order the F(3,3) epilogue of the block's second conv holds its outputs in
        static_assert(!F33 || CF::NSREG >= 3 * (RS2 * CF::G / 48), "shortcut tiles kept in registers");
        conv_layer<CF, 32, 64, 1, 2, 0, RS1, RS2, L2, 3 * (RS2 * CF::G / 48), MODE_TO_REGS, false, GEOM_WTRIPLE, 16, RS2 * CF::G,
                   false, SWX, SWX>(X, nullptr, w2, nullptr, W + OFF_SC + W3264S, sreg, pad2, dump, wave, lane);
        // the second conv's first input group: requested only now (20 registers less across the two convs above)
#pragma unroll
        for (int c = 0; c < 5; ++c) w3[0][c] = *(const f32x4*)(W + OFF_C2 + cb4 * (20 * 256) + lane * 4 + c * 256);
    } else if constexpr (WINO) {
        // the shortcut in the row order the Winograd epilogue of the block's second conv holds its outputs in
        conv_layer<CF, 32, 64, 1, 2, 0, RS1, RS2, L2, CF::NSREG, MODE_TO_REGS, false, GEOM_WPAIR, 16, RS2 * CF::G, false, SWX,
                   SWX>(X, nullptr, w2, nullptr, W + OFF_SC + W3264S, sreg, pad2, dump, wave, lane);
    } else {
        conv_layer<CF, 32, 64, 1, 2, 0, RS1, RS2, L2, T2, MODE_TO_REGS, false, GEOM_TRUNK, 16, RS2 * CF::G, false, SWX, SWX>(
            X, nullptr, w2, nullptr, W + OFF_SC + W3264S, sreg, pad2, dump, wave, lane);
    }
    barrier(11);
    if (tid < 32) ((f32x4*)X)[(tid & 15) + (tid >> 4) * (RS2 * G + 1) * 16] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's block of an F(3,3) layer, this lane: [4 input groups][5 components][64 lanes][4]
    auto slice3 = [&](int off) { return W + off + cb4 * (20 * 256) + lane * 4; };
    if constexpr (BF16) {
        bf16x3_layer<CF, BF_RESID, false, false>((const unsigned char*)H, (unsigned char*)X, bwh, bwl, WS16 + 24576,
                                                 W + OFF_C2 + O::W6464D, sreg, wave, lane);
    } else if constexpr (F33) {
        // the shortcut moves from registers into the output image (each lane stores what it will read back as the
        // residual of its own outputs): holding it across the layer would not fit beside 15 accumulators
        {
            const int j = lane & 15, q = lane >> 4;
#pragma unroll
            for (int t = 0; t < 3 * (RS2 * G / 48); ++t)
                *(f32x4*)(X + img_off_triple<64, SW_3>(t / 3, j, (t % 3) + 1, 4 * cb4 + q)) = sreg[t];
        }
        wino3_layer<CF, 64, MODE_RESID_INPLACE, false>(H, X, w3, slice3(OFF_C2), slice3(O::off_d(0)), W + OFF_C2 + O::W6464D, wave,
                                                       lane);
    } else if constexpr (WINO) {
        wino_layer<CF, 64, MODE_ADD_REGS, true>(H, X, wB, slice(O::off_d(0), cb4, NVB), W + OFF_C2 + W6464, pad2w, dump, wave,
                                                lane, sreg);
    } else {
        conv_layer<CF, 64, 64, 3, 1, 1, RS2, RS2, L2, T2, MODE_ADD_REGS, true, GEOM_TRUNK, 16, RS2 * CF::G, true, SWX, SWX>(
            H, X, w12, slice(O::off_d(0), cb4, NVB), W + OFF_C2 + W6464, sreg, pad2, dump, wave, lane);
    }
    barrier(12);

    // ---- NB64 x ResidualBlock(64) (3 in the canonical read convolver) ------------------------------
#pragma unroll
    for (int blk = 0; blk < NB64; ++blk) {
        const int off_a = O::off_d(blk), off_b = off_a + (O::W6464D + 64);
        if constexpr (BF16) {
            bf16x3_layer<CF, BF_PLAIN, false, false>((const unsigned char*)X, (unsigned char*)H, bwh, bwl,
                                                     WS16 + (2 + 2 * blk) * 24576, W + off_a + O::W6464D, sreg, wave, lane);
            __syncthreads();
            if (blk < NB64 - 1)
                bf16x3_layer<CF, BF_RESID, false, false>((const unsigned char*)H, (unsigned char*)X, bwh, bwl,
                                                         WS16 + (3 + 2 * blk) * 24576, W + off_b + O::W6464D, sreg, wave, lane);
            else
                bf16x3_layer<CF, BF_RESID, true, true>((const unsigned char*)H, (unsigned char*)X, bwh, bwl, nullptr,
                                                       W + off_b + O::W6464D, sreg, wave, lane);
        } else if constexpr (F33) {
            wino3_layer<CF, 64, MODE_PLAIN, false>(X, H, w3, slice3(off_a), slice3(off_b), W + off_a + O::W6464D, wave, lane);
            barrier(13 + 2 * blk);
            if (blk < NB64 - 1)
                wino3_layer<CF, 64, MODE_RESID_INPLACE, false>(H, X, w3, slice3(off_b), slice3(O::off_d(blk + 1)),
                                                           W + off_b + O::W6464D, wave, lane);
            else
                wino3_layer<CF, 64, MODE_RESID_INPLACE, true>(H, X, w3, slice3(off_b), nullptr, W + off_b + O::W6464D, wave, lane);
        } else if constexpr (WINO) {
            wino_layer<CF, 64, MODE_PLAIN, true>(X, H, wB, slice(off_b, cb4, NVB), W + off_a + W6464, pad2w, dump, wave, lane);
            barrier(13 + 2 * blk);
            if (blk < NB64 - 1)
                wino_layer<CF, 64, MODE_RESID_INPLACE, true>(H, X, wB, slice(O::off_d(blk + 1), cb4, NVB), W + off_b + W6464,
                                                             pad2w, dump, wave, lane);
            else
                wino_layer<CF, 64, MODE_RESID_INPLACE, false>(H, X, wB, nullptr, W + off_b + W6464, pad2w, dump, wave, lane);
        } else {
            conv_layer<CF, 64, 64, 3, 1, 1, RS2, RS2, L2, T2, MODE_PLAIN, true, GEOM_TRUNK, 16, RS2 * CF::G, true>(
                X, H, w12, slice(off_b, cb4, 12), W + off_a + W6464, sreg, pad2, dump, wave, lane);
            barrier(13 + 2 * blk);
            if (blk < NB64 - 1)
                conv_layer<CF, 64, 64, 3, 1, 1, RS2, RS2, L2, T2, MODE_RESID_INPLACE, true, GEOM_TRUNK, 16, RS2 * CF::G, true>(
                    H, X, w12, slice(O::off_d(blk + 1), cb4, 12), W + off_b + W6464, sreg, pad2, dump, wave, lane);
            else
                conv_layer<CF, 64, 64, 3, 1, 1, RS2, RS2, L2, T2, MODE_RESID_INPLACE, false, GEOM_TRUNK, 16, RS2 * CF::G, true>(
                    H, X, w12, nullptr, W + off_b + W6464, sreg, pad2, dump, wave, lane);
        }
        barrier(14 + 2 * blk);
    }

    // ---- the group's reads join the running per-allele sum, in read order ---------------------------
    static_assert(G <= 4, "the allele select below");
    // the alleles of the group's reads, by scalar loads (the index is uniform), all requested at once
    int allele_of[G];
#pragma unroll
    for (int rd = 0; rd < G; ++rd) {
        const long long r = read0 + rd < a.n_reads ? read0 + rd : a.n_reads - 1;
        allele_of[rd] = __builtin_amdgcn_readfirstlane(a.allele_of_read[r]);
    }
#pragma nounroll
    for (int rd = 0; rd < n_here; ++rd) {
        const int al = rd == 0 ? allele_of[0] : rd == 1 ? allele_of[1 % G] : rd == 2 ? allele_of[2 % G] : allele_of[3 % G];
        if (al != cur) {                                      // uniform
            flush();
            cur = al;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + THREADS * i;
            if (f < L2 * 16) {
                const f32x4 v = *(const f32x4*)(X + img_off<64, F33 ? SW_3 : SWX>(1 + rd * RS2 + (f >> 4), f & 15));
#pragma unroll
                for (int e = 0; e < 4; ++e) carry[i][e] += v[e];
            }
        }
    }
    barrier(19);                                          // the next group's stem overwrites the images
    if constexpr (STAMP) {
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        if (grp < a.stamp_groups && lane == 0) stamp_base[43] = rt;
    }
    }   // groups of this workgroup
    flush();
    if constexpr (STAMP) {
        static_assert(!STAMP || (NB64 == 3 && STEM && WINO && !BF16), "stamps: the default fp32 Winograd schedule (20 barriers per group)");
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((tid0 & 63) == 0) a.stamps[(((long long)blockIdx.x * CF::NW + wave0) * a.stamp_groups) * STAMP_SLOTS + 44] = t;
    }
}

template <class CF, int NB64, bool WINO, bool STEM_ONLY = false, bool BF16 = false, bool BF16_32 = false>
static hipError_t launch_cfg(const ReadConvArgs& a, hipStream_t stream) {
    // the LDS opt-in is a per-device attribute of the function: once per device this process launches on
    // (threads that share a device race benignly: the call is idempotent)
    static bool configured_on[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    bool& configured = configured_on[dev];
    if (!configured) {
        hipError_t e = hipSuccess;
        if constexpr (!STEM_ONLY)
            e = hipFuncSetAttribute((const void*)readconv_kernel<CF, false, NB64, WINO, BF16, BF16_32>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)readconv_kernel<CF, true, NB64, WINO, BF16, BF16_32>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured = true;
    }
    if (a.groups_per_wg < 1) return hipErrorInvalidValue;
    const long long per_wg = (long long)CF::G * a.groups_per_wg;
    const unsigned groups = (unsigned)((a.n_reads + per_wg - 1) / per_wg);       // workgroups
    if (a.stamps) {
        // the diagnostic instantiation (hello_engine_debug_stamps): the default schedule only; stamp_mode bit 1 pads the
        // workgroup's LDS past half a CU's, so that ONE workgroup is resident per CU (one wave per SIMD)
        if constexpr (std::is_same<CF, Geometry>::value && NB64 == 3 && WINO && !BF16 && !STEM_ONLY) {
            if (!a.reads || (a.channels != 6 && a.channels != 7) || a.stamp_groups < 1) return hipErrorInvalidValue;
            static bool stamped_on[64] = {};          // (engines on threads that share a device race benignly, like configured_on: the call is idempotent)
            if (!stamped_on[dev]) {
                const hipError_t e = hipFuncSetAttribute((const void*)readconv_kernel<CF, true, NB64, WINO, false, false, true>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
                if (e != hipSuccess) return e;
                stamped_on[dev] = true;
            }
            const int lds = (a.stamp_mode & 2) ? 100 * 1024 : CF::LDS_BYTES;
            hipLaunchKernelGGL((readconv_kernel<CF, true, NB64, WINO, false, false, true>), dim3(groups), dim3(CF::THREADS), lds, stream, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (a.reads) {
        if (a.channels != 6 && a.channels != 7) return hipErrorInvalidValue;
        hipLaunchKernelGGL((readconv_kernel<CF, true, NB64, WINO, BF16, BF16_32>), dim3(groups), dim3(CF::THREADS), CF::LDS_BYTES, stream, a);
    } else if constexpr (!STEM_ONLY) {
        hipLaunchKernelGGL((readconv_kernel<CF, false, NB64, WINO, BF16, BF16_32>), dim3(groups), dim3(CF::THREADS), CF::LDS_BYTES, stream, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_readconv_fused(const ReadConvArgs& a, hipStream_t stream) {
    if (a.n_reads <= 0) return hipSuccess;
    if (a.window == 250) {         // the feature-map variant: whole read convolver from the bytes, Winograd form
        if (!a.reads || !a.winograd || a.extra_blocks != 0) return hipErrorInvalidValue;
        return launch_cfg<Geometry250, 3, true, true>(a, stream);
    }
    if (a.window != 150) return hipErrorInvalidValue;
    if (a.bf16x3) {                // arithmetic mode bf16x3: canonical architecture, whole kernel from the bytes, Winograd form
        if (!a.reads || !a.winograd || a.extra_blocks != 0 || a.softplus) return hipErrorInvalidValue;
        return a.bf16x3 > 1 ? launch_cfg<Geometry, 3, true, true, true, true>(a, stream)
                            : launch_cfg<Geometry, 3, true, true, true, false>(a, stream);
    }
    if (a.softplus) {              // the Softplus configuration: whole kernel, Winograd form
        if (!a.reads || !a.winograd || a.extra_blocks != 0) return hipErrorInvalidValue;
        return launch_cfg<GeometrySoftplus, 3, true, true>(a, stream);
    }
    if (a.extra_blocks == 0) return a.winograd ? launch_cfg<Geometry, 3, true>(a, stream) : launch_cfg<Geometry, 3, false>(a, stream);
    if (a.extra_blocks == 2) return a.winograd ? launch_cfg<Geometry, 5, true>(a, stream) : launch_cfg<Geometry, 5, false>(a, stream);
    return hipErrorInvalidValue;
}

// frames[a] = sum over the allele's partial slots, in slot (= read) order
__global__ void readconv_finalize_kernel(const float* __restrict__ partial, const int32_t* __restrict__ slot_off,
                                         float* __restrict__ frames, int frame_f4) {      // frame_f4 = positions * channels / 4
    const int al = blockIdx.x;
    const int lo = slot_off[al], hi = slot_off[al + 1];
    for (int f = threadIdx.x; f < frame_f4; f += blockDim.x) {
        f32x4 acc = *(const f32x4*)(partial + ((long long)lo * frame_f4 + f) * 4);
        for (int s = lo + 1; s < hi; ++s) {
            const f32x4 v = *(const f32x4*)(partial + ((long long)s * frame_f4 + f) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        *(f32x4*)(frames + ((long long)al * frame_f4 + f) * 4) = acc;
    }
}

hipError_t launch_readconv_finalize(const float* partial, const int32_t* slot_off, float* frames, int n_alleles,
                                    int frame_rows, int channels, hipStream_t stream) {
    if (n_alleles <= 0) return hipSuccess;
    if (channels <= 0 || (channels % 4)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(readconv_finalize_kernel, dim3(n_alleles), dim3(192), 0, stream, partial, slot_off, frames,
                       frame_rows * (channels / 4));
    return hipGetLastError();
}

// =====================================================================================================================
// Fused allele-level compressor (architectures/compressor_conv_small.py:8-55; MixtureOfExpertsAdvanced.py:125):
//     [36][64] per item -> 1x1 64->64 + ReLU -> strided block 64->128 (k3 s2 + ReLU, k3 s1 + ReLU, + 1x1 s2 shortcut)
//     -> NB x ResidualBlock(128) -> [18][128]
// One workgroup of 8 waves carries 8 items (alleles, or site sums) through all 4 + 2 NB convolutions with the
// activations in LDS (two images of 74.75 KB: one workgroup per CU, two waves per SIMD), using the read convolver's
// layer routines at twice the channel counts: the 64-channel image is 8 x 36 = 288 rows = 18 tiles of 16 rows, the
// 128-channel image 8 x 18 = 144 rows = 3 tiles of 16 triples, both stacked without rows between the items (the taps
// that would cross an item boundary read zero); every k3/s1 convolution runs in Winograd F(3,3) form (a wave owns one
// of the 8 channel blocks and walks 8 input groups x 3 tiles).  Weights stream from L2 per layer, 1.5 MB per workgroup.
namespace cc {
struct Cfg {
    static constexpr int ACT = rc::ACT_RELU;
    static __device__ __forceinline__ float act(float x) { return fmaxf(x, 0.f); }
    static constexpr int G = 8;                        // items per workgroup
    static constexpr int NW = 8;                       // waves per workgroup
    static constexpr int THREADS = 64 * NW;
    static constexpr int L0 = 36, L1 = 18;             // rows per item at 64 / 128 channels
    static constexpr int NSREG = 3 * (L1 * G / 48);    // shortcut tiles a wave keeps in registers (triple order)
    static constexpr int BUF_FLOATS = rc::cmax((L0 * G + 2) * 64, (L1 * G + 2) * 128);
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4 + 64;
    // packed weight block (floats): per conv [COUT/16][KT][CIN/16][64 lanes][4] (+ bias[COUT]); the F(3,3) convs
    // [COUT/16][CIN/16][5][64 lanes][4] (+ bias)
    static constexpr int W11 = 4 * 1 * 4 * 256, WS = 8 * 3 * 4 * 256, WSC = 8 * 1 * 4 * 256, WB = 8 * 8 * 5 * 256;
    static constexpr int OFF_11 = 0, OFF_S = OFF_11 + W11 + 64, OFF_SC = OFF_S + WS + 128, OFF_B = OFF_SC + WSC + 128;
    static constexpr int off_conv(int i) { return OFF_B + i * (WB + 128); }    // 0: the strided block's second conv; then the blocks'
    static constexpr int total(int blocks) { return off_conv(1 + 2 * blocks); }
};
}  // namespace cc

int compressor_weight_floats(int blocks) { return cc::Cfg::total(blocks); }
bool compressor_supports_blocks(int blocks) { return blocks >= 2 && blocks <= 4; }

template <int NB>
__global__ __launch_bounds__(cc::Cfg::THREADS, 2) void compressor_kernel(CompressorArgs a) {
    using CF = cc::Cfg;
    constexpr int G = CF::G, L0 = CF::L0, L1 = CF::L1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem;
    float* const bufB = smem + CF::BUF_FLOATS;
    float* const dump = smem + 2 * CF::BUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup carries a.items_per_wg <= G items (fewer in small launches, so that the items spread over the CUs: kernels.h)
    const long long item0 = (long long)blockIdx.x * a.items_per_wg;
    const int n_here = (int)((a.n_items - item0) < a.items_per_wg ? (a.n_items - item0) : a.items_per_wg);
    const float* __restrict__ W = a.w;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 w11[4];
    load_weights<4>(w11, W + CF::OFF_11, wave % 4, lane);
    // the items' frames -> bufA as a 64-channel image walked one row per lane (SW_OLD), row 0 = leading zero row
    {
        // ONE round trip to memory: every thread requests its 9 float4 first and stores them afterwards (a load-store loop
        // would pay the memory latency once per iteration, and no second workgroup of the CU hides it here)
        const f32x4* src = (const f32x4*)(a.frames + item0 * (L0 * 64));
        constexpr int NLD = G * L0 * 16 / CF::THREADS;
        static_assert(NLD * CF::THREADS == G * L0 * 16, "the input image divides evenly over the threads");
        f32x4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + CF::THREADS * k;
            v[k] = (f / (L0 * 16)) < n_here ? src[f] : zero4;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + CF::THREADS * k;
            *(f32x4*)(bufA + img_off<64, SW_OLD>(1 + (f >> 4), f & 15)) = v[k];
        }
        // row 0 of both 64-channel images: the padding row of the first item (the 1x1 convolution writes rows >= 1)
        if (tid < 16) ((f32x4*)bufA)[tid] = zero4;
        else if (tid < 32) ((f32x4*)bufB)[tid - 16] = zero4;
    }
    __syncthreads();
    f32x4 sreg[CF::NSREG];
    // a partly filled workgroup (the last one of a launch; the only one of a one-site call) skips the 16-row tiles that hold
    // no items: tile t of the 64-channel image is tile (wave / 4) + 2 k of its position group, of the 128-channel image tile k
    const bool full = n_here == G;
    const int live0 = (n_here * L0 + 15) / 16, live1 = (n_here * L1 + 15) / 16;
    // 1x1 64 -> 64 + ReLU (4 channel blocks x 2 position groups), written for the stride-2 walk of the next layer (SW_W)
    if (full)
        conv_layer<CF, 64, 64, 1, 1, 0, L0, L0, L0, L0 * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L0 * G, false, SW_OLD, SW_W>(
            bufA, bufB, w11, nullptr, W + CF::OFF_11 + CF::W11, sreg, 0u, dump, wave, lane);
    else
        conv_layer<CF, 64, 64, 1, 1, 0, L0, L0, L0, L0 * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L0 * G, false, SW_OLD, SW_W, true>(
            bufA, bufB, w11, nullptr, W + CF::OFF_11 + CF::W11, sreg, 0u, dump, wave, lane, 0, (live0 - wave / 4 + 1) / 2);
    f32x4 ws[12], wsc[4];
    load_weights<12>(ws, W + CF::OFF_S, wave, lane);
    load_weights<4>(wsc, W + CF::OFF_SC, wave, lane);
    __syncthreads();
    // bufA becomes the 128-channel image: rows 0 and 18 G + 1 are its zero rows
    if (tid < 64) ((f32x4*)bufA)[(tid & 31) + (tid >> 5) * (L1 * G + 1) * 32] = zero4;
    // strided block: k3 s2 64 -> 128 + ReLU (tap 0 of an item's first row reads zero, not the previous item's last row)
    if (full)
        conv_layer<CF, 64, 128, 3, 2, 1, L0, L1, L1, L1 * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L1 * G, true, SW_W, SW_3>(
            bufB, bufA, ws, nullptr, W + CF::OFF_S + CF::WS, sreg, 0u, dump, wave, lane);
    else
        conv_layer<CF, 64, 128, 3, 2, 1, L0, L1, L1, L1 * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L1 * G, true, SW_W, SW_3, true>(
            bufB, bufA, ws, nullptr, W + CF::OFF_S + CF::WS, sreg, 0u, dump, wave, lane, 0, live1);
    // its 1x1 s2 shortcut, in the row order the F(3,3) epilogue of the block's second conv holds its outputs in
    conv_layer<CF, 64, 128, 1, 2, 0, L0, L1, L1, CF::NSREG, MODE_TO_REGS, false, GEOM_WTRIPLE, 16, L1 * G, false, SW_W, SW_3>(
        bufB, nullptr, wsc, nullptr, W + CF::OFF_SC + CF::WSC, sreg, 0u, dump, wave, lane);
    f32x4 w3[2][5];
    auto slice = [&](int off) { return W + off + wave * (8 * 5 * 256) + lane * 4; };   // this wave's block, this lane
#pragma unroll
    for (int c = 0; c < 5; ++c) w3[0][c] = *(const f32x4*)(slice(CF::off_conv(0)) + c * 256);
    __syncthreads();
    if (tid < 64) ((f32x4*)bufB)[(tid & 31) + (tid >> 5) * (L1 * G + 1) * 32] = zero4;
    {
        const int j = lane & 15, q = lane >> 4;                   // the shortcut moves into the output image (see readconv_kernel)
#pragma unroll
        for (int t = 0; t < CF::NSREG; ++t)
            *(f32x4*)(bufB + img_off_triple<128, SW_3>(t / 3, j, (t % 3) + 1, 4 * wave + q)) = sreg[t];
    }
    // a partly filled workgroup (the last one of a launch; the only one of a one-site call) skips the F(3,3) tiles that
    // hold no items: 6 triples per item, 16 per tile
    const int live_tiles = (6 * n_here + 15) / 16;
    auto f33_layers = [&](auto partial) {
        constexpr bool PARTIAL = decltype(partial)::value;
        wino3_layer<CF, 128, MODE_RESID_INPLACE, false, L1, SW_3, true, PARTIAL>(
            bufA, bufB, w3, slice(CF::off_conv(0)), slice(CF::off_conv(1)), W + CF::off_conv(0) + CF::WB, wave, lane, live_tiles);
        __syncthreads();
        static_for<0, NB>([&](auto bc) {
            constexpr int blk = decltype(bc)::value;
            constexpr int off_a = CF::off_conv(1 + 2 * blk), off_b = CF::off_conv(2 + 2 * blk);
            wino3_layer<CF, 128, MODE_PLAIN, false, L1, SW_3, true, PARTIAL>(bufB, bufA, w3, slice(off_a), slice(off_b),
                                                                            W + off_a + CF::WB, wave, lane, live_tiles);
            __syncthreads();
            if constexpr (blk < NB - 1)
                wino3_layer<CF, 128, MODE_RESID_INPLACE, false, L1, SW_3, true, PARTIAL>(
                    bufA, bufB, w3, slice(off_b), slice(CF::off_conv(3 + 2 * blk)), W + off_b + CF::WB, wave, lane, live_tiles);
            else
                wino3_layer<CF, 128, MODE_RESID_INPLACE, true, L1, SW_3, true, PARTIAL>(bufA, bufB, w3, slice(off_b), nullptr,
                                                                                       W + off_b + CF::WB, wave, lane, live_tiles);
            __syncthreads();
        });
    };
    if (live_tiles == L1 * G / 48) f33_layers(std::false_type{});
    else f33_layers(std::true_type{});
    f32x4* dst = (f32x4*)(a.dst + item0 * (L1 * 128));
    for (int f = tid; f < n_here * L1 * 32; f += CF::THREADS)
        dst[f] = *(const f32x4*)(bufB + img_off<128, SW_3>(1 + (f >> 5), f & 31));
}

int small_launch_items_per_wg(long long n_items, int most) {
    const long long cus = device_cus();
    const long long n = (n_items + cus - 1) / cus;
    // up to two items per workgroup while that gives every item-pair its own CU; beyond that whole workgroups: more, emptier
    // workgroups each stream the layer weights again, which costs throughput once launches of several engines share the chip
    // (256-site launches over four engines: 444 k sites/s with whole workgroups, 426 k with three items per workgroup)
    return (int)(n < 1 ? 1 : (n <= 2 ? n : most));
}

hipError_t launch_compressor_fused(const CompressorArgs& args, hipStream_t stream) {
    if (args.n_items <= 0) return hipSuccess;
    CompressorArgs a = args;
    if (a.items_per_wg == 0) a.items_per_wg = small_launch_items_per_wg(a.n_items, cc::Cfg::G);
    if (!compressor_supports_blocks(a.blocks) || !a.frames || !a.dst || !a.w || a.items_per_wg < 1 || a.items_per_wg > cc::Cfg::G) return hipErrorInvalidValue;
    static bool configured_on[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured_on[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)compressor_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cc::Cfg::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)compressor_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, cc::Cfg::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)compressor_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cc::Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured_on[dev] = true;
    }
    const unsigned grid = (unsigned)((a.n_items + a.items_per_wg - 1) / a.items_per_wg);
    if (a.blocks == 2) hipLaunchKernelGGL(compressor_kernel<2>, dim3(grid), dim3(cc::Cfg::THREADS), cc::Cfg::LDS_BYTES, stream, a);
    else if (a.blocks == 3) hipLaunchKernelGGL(compressor_kernel<3>, dim3(grid), dim3(cc::Cfg::THREADS), cc::Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(compressor_kernel<4>, dim3(grid), dim3(cc::Cfg::THREADS), cc::Cfg::LDS_BYTES, stream, a);
    return hipGetLastError();
}


// =====================================================================================================================
// Front of the allele-level expert (architectures/xattn_subtract.py:9-60; MixtureOfExpertsAdvanced.py:142-155), fused:
//     x = a0 a + a1 s[site(a)]   (LinearCombination of the allele's and its site's compressed frames, [18][128])
//     y1 = relu(1x1 128->128 (x))
//     y2 = relu(k3 s2 p1 128->256 (y1))   and   sc = 1x1 s2 128->256 (y1)        -- the strided block's first conv and shortcut
// One workgroup of 8 waves carries 8 items: x and y1 live in LDS (two images of 8 x 18 rows x 128 channels, stacked without
// rows between the items), y2 and sc go straight to HBM ([item][9][256] each), where the block's second convolution (Winograd
// kernel, residual = sc) picks them up.  Replaces four launches (MIX, two 1x1 and one strided CONV1D) and three HBM round
// trips of [A][18][128].  The 256-channel convolutions run as two passes of 8 channel blocks (waves w and w + 8 of a
// 16-wave layout); tap 0 of an item's first output row reads zero instead of the previous item's last row.
namespace xf {
struct Cfg {
    static constexpr int ACT = rc::ACT_RELU;
    static __device__ __forceinline__ float act(float x) { return fmaxf(x, 0.f); }
    static constexpr int G = 8, NW = 8, THREADS = 64 * NW;
    static constexpr int L = 18, LO = 9;                 // rows per item before / after the strided block
    static constexpr int NSREG = 1;
    static constexpr int A_FLOATS = (L * G + 2) * 128;   // x: leading zero row + 144 rows (+1)
    static constexpr int B_ROWS = 2 * (16 * 5) + 2;      // y1: the strided layer's fifth (half-empty) tile reads up to row 160
    static constexpr int LDS_BYTES = (A_FLOATS + B_ROWS * 128) * 4 + 64;
    // packed weights (floats): conv_layer's [COUT/16][KT][CIN/16][64 lanes][4] + bias[COUT], three convolutions
    static constexpr int W11 = 8 * 1 * 8 * 256, WS = 16 * 3 * 8 * 256, WSC = 16 * 1 * 8 * 256;
    static constexpr int OFF_11 = 0, OFF_S = OFF_11 + W11 + 128, OFF_SC = OFF_S + WS + 256, TOTAL = OFF_SC + WSC + 256;
};
struct Cfg16 : Cfg {                                     // the 256-channel layers' wave layout: 16 channel blocks
    static constexpr int NW = 16;
};
struct CfgLinear : Cfg16 {                               // the shortcut carries no activation
    static __device__ __forceinline__ float act(float x) { return x; }
};
}  // namespace xf

int xattn_front_weight_floats() { return xf::Cfg::TOTAL; }

__global__ __launch_bounds__(xf::Cfg::THREADS, 2) void xattn_front_kernel(XattnFrontArgs a) {
    using CF = xf::Cfg;
    using CF16 = xf::Cfg16;
    constexpr int G = CF::G, L = CF::L, LO = CF::LO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const bufA = smem;
    float* const bufB = smem + CF::A_FLOATS;
    float* const dump = smem + CF::A_FLOATS + CF::B_ROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup carries a.items_per_wg <= G items (fewer in small launches, so that the items spread over the CUs: kernels.h)
    const long long item0 = (long long)blockIdx.x * a.items_per_wg;
    const int n_here = (int)((a.n_items - item0) < a.items_per_wg ? (a.n_items - item0) : a.items_per_wg);
    const float* __restrict__ W = a.w;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sreg[CF::NSREG];

    f32x4 w11[8];
    load_weights<8>(w11, W + CF::OFF_11, wave, lane);
    {
        // x = a0 a + a1 s: every thread requests its 9 + 9 float4 first (ONE round trip), then combines and stores
        constexpr int NLD = G * L * 32 / CF::THREADS;
        static_assert(NLD * CF::THREADS == G * L * 32, "the input image divides evenly over the threads");
        const f32x4* src = (const f32x4*)(a.alleles + item0 * (L * 128));
        f32x4 va[NLD], vs[NLD];
        if (a.sites) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int f = tid + CF::THREADS * k;
                const int it = f / (L * 32);
                if (it < n_here) {
                    va[k] = src[f];
                    vs[k] = *((const f32x4*)(a.sites + (long long)a.owner[item0 + it] * (L * 128)) + (f - it * (L * 32)));
                } else {
                    va[k] = zero4;
                    vs[k] = zero4;
                }
            }
        } else {
            // the site sum folded in (reduceSlots over the site's alleles, MixtureOfExpertsAdvanced.py:23-34): rows added in
            // allele order from zero, as segsum_kernel adds them -- the same bits; the r-th rows of all nine are requested together
            const f32x4* first[NLD];
            int count[NLD], most = 0;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int f = tid + CF::THREADS * k;
                const int it = f / (L * 32);
                count[k] = 0;
                first[k] = src;
                va[k] = zero4;
                vs[k] = zero4;
                if (it < n_here) {
                    va[k] = src[f];
                    const int site = a.owner[item0 + it];
                    const int lo = a.site_off[site];
                    count[k] = a.site_off[site + 1] - lo;
                    first[k] = (const f32x4*)(a.alleles + (long long)lo * (L * 128)) + (f - it * (L * 32));
                    most = count[k] > most ? count[k] : most;
                }
            }
            for (int r = 0; r < most; ++r) {
                f32x4 row[NLD];
#pragma unroll
                for (int k = 0; k < NLD; ++k) row[k] = r < count[k] ? first[k][(long long)r * (L * 32)] : zero4;
#pragma unroll
                for (int k = 0; k < NLD; ++k)
                    if (r < count[k]) vs[k] = vs[k] + row[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + CF::THREADS * k;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)                                                  // LinearCombination, NNTools.py:771-777
                v[e] = a.rest ? va[k][e] - (vs[k][e] - va[k][e]) : a.a0 * va[k][e] + a.a1 * vs[k][e];
            *(f32x4*)(bufA + img_off<128, SW_OLD>(1 + (f >> 5), f & 31)) = v;
        }
        if (tid < 32) ((f32x4*)bufA)[tid] = zero4;                         // leading zero rows of both images
        else if (tid < 64) ((f32x4*)bufB)[tid - 32] = zero4;
    }
    __syncthreads();
    // a partly filled workgroup (the last one of a launch; the only one of a one-site call) skips the tile pairs that hold
    // no items
    const int vrows = n_here * LO;
    float* const y2 = a.y2 + item0 * (LO * 256);
    float* const sc = a.sc + item0 * (LO * 256);
    auto layers = [&](auto partial) {
        constexpr bool PARTIAL = decltype(partial)::value;
        const int live1 = (n_here * L + 15) / 16, live2 = (n_here * LO + 15) / 16;
        // 1x1 128 -> 128 + ReLU (8 channel blocks, 9 tiles each), written for the stride-2 walk of the next layers (SW_W)
        conv_layer<CF, 128, 128, 1, 1, 0, L, L, L, L * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L * G, false, SW_OLD, SW_W, PARTIAL>(
            bufA, bufB, w11, nullptr, W + CF::OFF_11 + CF::W11, sreg, 0u, dump, wave, lane, 0, live1);
        __syncthreads();
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int w16 = wave + 8 * half;                               // channel block 0..15
            {
                f32x4 ws[24];
                load_weights<24>(ws, W + CF::OFF_S, w16, lane);
                // k3 s2 p1 128 -> 256 + ReLU; tap 0 of an item's first row reads zero (BMASK), 72 rows = 4.5 tiles
                conv_layer<CF16, 128, 256, 3, 2, 1, L, LO, LO, 5, MODE_PLAIN, false, GEOM_TRUNK, 16, LO * G, true, SW_W, SW_GLOBAL, PARTIAL>(
                    bufB, y2, ws, nullptr, W + CF::OFF_S + CF::WS, sreg, 0u, dump, w16, lane, vrows, live2);
            }
            {
                f32x4 wsc[8];
                load_weights<8>(wsc, W + CF::OFF_SC, w16, lane);
                // its 1x1 s2 shortcut: no activation (a configuration whose act() is the identity)
                conv_layer<xf::CfgLinear, 128, 256, 1, 2, 0, L, LO, LO, 5, MODE_PLAIN, false, GEOM_TRUNK, 16, LO * G, false, SW_W, SW_GLOBAL,
                           PARTIAL>(bufB, sc, wsc, nullptr, W + CF::OFF_SC + CF::WSC, sreg, 0u, dump, w16, lane, vrows, live2);
            }
        }
    };
    if (n_here == G) layers(std::false_type{});
    else layers(std::true_type{});
}

hipError_t launch_xattn_front(const XattnFrontArgs& args, hipStream_t stream) {
    if (args.n_items <= 0) return hipSuccess;
    XattnFrontArgs a = args;
    if (a.items_per_wg == 0) a.items_per_wg = small_launch_items_per_wg(a.n_items, xf::Cfg::G);
    if (!a.alleles || (!a.sites && !a.site_off) || !a.owner || !a.y2 || !a.sc || !a.w || a.items_per_wg < 1 || a.items_per_wg > xf::Cfg::G) return hipErrorInvalidValue;
    static bool configured_on[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured_on[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)xattn_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, xf::Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured_on[dev] = true;
    }
    const unsigned grid = (unsigned)((a.n_items + a.items_per_wg - 1) / a.items_per_wg);
    hipLaunchKernelGGL(xattn_front_kernel, dim3(grid), dim3(xf::Cfg::THREADS), xf::Cfg::LDS_BYTES, stream, a);
    return hipGetLastError();
}


// =====================================================================================================================
// Residual trunk of the 2x-channel ("_wide") read convolver (architectures/read_convolver_wide.py: the layers after the
// max pool) + reads->alleles segment sum:
//     [71][64] per read -> 3 x ResidualBlock(64) -> strided block 64->128 (k3 s2 + ReLU, k3 s1 + ReLU, + 1x1 s2 shortcut)
//     -> 3 x ResidualBlock(128) -> [36][128] per read -> per-allele partial sums
// 18.8 M of the wide read convolver's 20.2 M MAC per read; its stem (6|7 -> 32 -> 32 -> 64 + max pool) runs layer by layer
// ahead of this kernel and hands over the pooled rows.  The images are the compressor kernel's: one workgroup of 8 waves
// carries 4 reads with two 74.75 KB images in LDS (one workgroup per CU, two waves per SIMD).  The 64-channel image is
// the read convolver's 32-channel geometry at twice the channels (reads stacked with ONE shared zero row, row stride 72:
// 288 rows = 6 tiles of 16 triples, 4 channel blocks x 2 position groups of waves), the 128-channel image the compact
// one (4 x 36 = 144 rows = 3 tiles of 16 triples, 8 channel blocks); every k3/s1 convolution runs in Winograd F(3,3)
// form.  The F(3,3) layers walk their images three rows per lane (SW_3) and the stride-2 convolutions two (SW_W), whose
// tiles are 32 rows apart -- not a period of SW_3 -- so the strided block reads a re-swizzled copy of its input.
namespace wt {
struct Cfg {
    static constexpr int ACT = rc::ACT_RELU;
    static __device__ __forceinline__ float act(float x) { return fmaxf(x, 0.f); }
    static constexpr int G = 4;                        // reads per workgroup
    static constexpr int NW = 8;                       // waves per workgroup
    static constexpr int THREADS = 64 * NW;
    static constexpr int L1 = 71, RS1 = 72;            // positions / row stride per read at 64 channels
    static constexpr int L2 = 36;                      // positions per read at 128 channels (compact)
    static constexpr int NSREG = 3 * (L2 * G / 48);    // shortcut tiles a wave keeps in registers (triple order)
    // the last triple of the 64-channel image reads rows up to RS1 G + 1 (its fifth input feeds the zero row's output only);
    // the stem's 32-channel stacks are WINDOW G rows, the second one shifted down a row (+ the row its last pair overhangs)
    static constexpr int WINDOW = 150, SROWS = WINDOW * G;
    static constexpr int BUF_FLOATS = rc::cmax(rc::cmax((RS1 * G + 2) * 64, (L2 * G + 2) * 128), (SROWS + 2) * 32);
    static constexpr int ST1 = ((SROWS + 15) / 16 + NW - 1) / NW * NW;     // conv1 tiles of 16 rows, whole rounds of the waves
    static constexpr int U8_BYTES = ((ST1 * 16 + 8) * 7 + 15) / 16 * 16;   // every conv1 tile reads in bounds
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4 + 64 + U8_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's images + byte staging fit the CU's LDS");
    // packed weight block (floats): the F(3,3) convs [COUT/16][CIN/16][5][64 lanes][4] + bias[COUT]; the strided conv and
    // its shortcut [COUT/16][KT][CIN/16][64 lanes][4] + bias
    static constexpr int WA = 4 * 4 * 5 * 256, WS = 8 * 3 * 4 * 256, WSC = 8 * 1 * 4 * 256, WB = 8 * 8 * 5 * 256;
    static constexpr int off_a(int i) { return i * (WA + 64); }               // 6 convs 64 -> 64
    static constexpr int OFF_S = 6 * (WA + 64), OFF_SC = OFF_S + WS + 128, OFF_B = OFF_SC + WSC + 128;
    static constexpr int off_b(int i) { return OFF_B + i * (WB + 128); }      // 0: the strided block's second conv; then 6 convs 128 -> 128
    static constexpr int W_TRUNK = OFF_B + 7 * (WB + 128);
    // the stem: conv1 [2 blocks][6 steps][64 lanes] + bias[32]; conv2 and conv3 as F(2,3) taps [COUT/16][4][2][64 lanes][4] + bias
    static constexpr int OFF_S1 = W_TRUNK, W_S1 = 2 * rc::S1_STEPS * 64;
    static constexpr int OFF_S2 = OFF_S1 + W_S1 + 32, W_S2 = 2 * 4 * 2 * 256;
    static constexpr int OFF_S3 = OFF_S2 + W_S2 + 32, W_S3 = 4 * 4 * 2 * 256;
    static constexpr int W_TOTAL = OFF_S3 + W_S3 + 64;
};
// what wino_layer needs to walk the stem's stacked rows (150 per read, no rows between reads) as one sequence of pairs
struct StemCfg {
    static constexpr int ACT = rc::ACT_RELU;
    static __device__ __forceinline__ float act(float x) { return fmaxf(x, 0.f); }
    static constexpr int G = Cfg::G, NW = Cfg::NW;
    static constexpr int RS1 = Cfg::WINDOW, RS2 = Cfg::WINDOW;
    static constexpr bool COMPACT = false;
};
}  // namespace wt

// ---- wide stem conv1: pileup bytes -> 32 channels (valid convolution over the stacked reads), as stem_conv1 with both
// 16-channel blocks per wave: tiles wave, wave + 8, ...; position r is stored at row r of a 32-channel SW_W image.
__device__ __forceinline__ void wide_stem_conv1(const unsigned char* __restrict__ s_u8, float* __restrict__ out,
                                                const float* __restrict__ W, int ch, float* __restrict__ dump, int wave,
                                                int lane) {
    using CF = wt::Cfg;
    constexpr int S = rc::S1_STEPS, NK = CF::ST1 / CF::NW;
    const int j = lane & 15, q = lane >> 4;
    float w1[2][S];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int st = 0; st < S; ++st) w1[blk][st] = W[(blk * S + st) * 64 + lane];
    f32x4 b4[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) b4[blk] = *(const f32x4*)(W + 2 * S * 64 + blk * 16 + 4 * q);
    const unsigned char* base = s_u8 + (16 * wave + j) * ch + q;             // tile `wave`, row j, byte q
    const int round_bytes = CF::NW * 16 * ch;
    unsigned char cur[S], nxt[S];
#pragma unroll
    for (int st = 0; st < S; ++st) cur[st] = base[4 * st];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (k + 1 < NK) {
#pragma unroll
            for (int st = 0; st < S; ++st) nxt[st] = base[(k + 1) * round_bytes + 4 * st];
        }
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < S; ++st) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0][st], (float)cur[st], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1][st], (float)cur[st], a1, 0, 0, 0);
        }
        const int r = 16 * (wave + CF::NW * k) + j;
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v0[e] = CF::act(a0[e] + b4[0][e]);
            v1[e] = CF::act(a1[e] + b4[1][e]);
        }
        *(f32x4*)(r < CF::SROWS ? out + img_off<32, SW_W>(r, q) : dump) = v0;
        *(f32x4*)(r < CF::SROWS ? out + img_off<32, SW_W>(r, 4 + q) : dump) = v1;
#pragma unroll
        for (int st = 0; st < S; ++st) cur[st] = nxt[st];
    }
}

// ---- wide stem conv3 (32 -> 64, valid) + ReLU + MaxPool1d(3, 2) in Winograd F(2,3) form, as stem_conv3_pool_wino:
// lane row j of a tile holds the pair of conv3 positions (2P, 2P+1), P = 15 t + j, pooled output P = max(y0, y1, y0 of
// lane j+1); 5 tiles per read.  Wave (read, half) computes channel blocks 2 half, 2 half + 1 of its read's 5 tiles.
// conv2 (wino_layer) stored position p of the stack at row p + 1.  Output: the trunk's 64-channel image (SW_3).
__device__ __forceinline__ void wide_stem_conv3_pool(const float* __restrict__ in, float* __restrict__ out,
                                                     const float* __restrict__ W3, float* __restrict__ dump, int wave,
                                                     int lane, int n_here) {
    using CF = wt::Cfg;
    constexpr int NTR = (CF::L1 + 14) / 15;
    const int j = lane & 15, q = lane >> 4;
    const int rd = wave >> 1, half = wave & 1;
    f32x4 w[2][4][2];                                                        // [block][component][input group]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m) w[b][c][m] = *(const f32x4*)(W3 + ((((2 * half + b) * 4 + c) * 2 + m) * 64 + lane) * 4);
    f32x4 b4[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) b4[b] = *(const f32x4*)(W3 + CF::W_S3 + (2 * half + b) * 16 + 4 * q);
    const int row0 = CF::WINDOW * rd + 1 + 2 * j;                             // row of d0 of this lane's pair in tile 0
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 ring[2][2][4];
    auto issue = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[t & 1][m][i] = *(const f32x4*)(in + img_off<32, SW_W>(row0 + 30 * t + i, 4 * m + q));
    };
    issue(std::integral_constant<int, 0>{});
    static_for<0, NTR>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t + 1 < NTR) issue(std::integral_constant<int, t + 1>{});
        f32x4 a[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4(&d)[4] = ring[t & 1][m];
            const f32x4 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool first = (m == 0) && (e == 0);
                        a[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[b][c][m][e], v[c][e], first ? (c == 1 ? b4[b] : zero4) : a[b][c],
                                                                       0, 0, 0);
                    }
        }
        const int pos = 15 * t + j;
        const bool ok = (j <= 14) && (rd < n_here) && (pos < CF::L1);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 y0 = (a[b][0] + a[b][1]) + a[b][2];                  // the bias rides in a[1]
            const f32x4 y1 = (a[b][1] - a[b][2]) - a[b][3];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = CF::act(fmaxf(fmaxf(y0[e], y1[e]), row_shl(y0[e], 1)));
            float* ptr = out + img_off<64, SW_3>(1 + rd * CF::RS1 + pos, 4 * (2 * half + b) + q);
            *(f32x4*)(ok ? ptr : dump) = v;
        }
    });
}

int readconv_wide_weight_floats() { return wt::Cfg::W_TOTAL; }     // trunk block, then the stem block
int readconv_wide_reads_per_group() { return wt::Cfg::G; }
ReadConvPlan readconv_wide_plan(long long n_reads) {
    return ReadConvPlan{1, (n_reads + wt::Cfg::G - 1) / wt::Cfg::G, 0};
}

template <bool STEM>
__global__ __launch_bounds__(wt::Cfg::THREADS, 2) void readconv_wide_kernel(ReadConvArgs a) {
    using CF = wt::Cfg;
    constexpr int G = CF::G, L1 = CF::L1, RS1 = CF::RS1, L2 = CF::L2, THREADS = CF::THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const X = smem;
    float* const H = smem + CF::BUF_FLOATS;
    int* const s_allele = (int*)(smem + 2 * CF::BUF_FLOATS);                 // G ints
    float* const dump = smem + 2 * CF::BUF_FLOATS + 12;                       // 16 spare bytes of the same 64-byte block
    const int tid0 = threadIdx.x;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const float* __restrict__ W = a.w;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // one group of G reads per workgroup (no sum carried across groups: the registers are the 128-channel layers'),
    // one partial slot per (group, allele) incidence
    const int tid = tid0, wave = wave0, lane = tid0 & 63;
    const int cb4 = wave % 4;
    const long long read0 = (long long)blockIdx.x * G;
    const int n_here = (int)((a.n_reads - read0) < G ? (a.n_reads - read0) : G);
    f32x4 w3[2][5];
    auto slice64 = [&](int off) { return W + off + cb4 * (4 * 5 * 256) + lane * 4; };      // this wave's block, this lane
    auto slice128 = [&](int off) { return W + off + wave * (8 * 5 * 256) + lane * 4; };
#pragma unroll
    for (int c = 0; c < 5; ++c) w3[0][c] = *(const f32x4*)(slice64(CF::off_a(0)) + c * 256);
    if (tid < G) s_allele[tid] = (tid < n_here) ? a.allele_of_read[read0 + tid] : -1;
    if constexpr (STEM) {
        // the stem, from the uint8 pileups: conv1 bytes -> X (32-channel stack), conv2 X -> H, conv3 + max pool H -> X
        // (now the trunk's 64-channel image)
        unsigned char* const s_u8 = (unsigned char*)(smem + 2 * CF::BUF_FLOATS + 16);
        const int ch = a.channels;
        const int n_bytes = n_here * CF::WINDOW * ch;
        const unsigned char* src = a.reads + read0 * CF::WINDOW * ch;
        constexpr int NDW = (CF::U8_BYTES / 4 + THREADS - 1) / THREADS;      // the group's bytes in ONE round trip
        if ((reinterpret_cast<unsigned long long>(src) & 3ull) == 0) {
            unsigned v[NDW];
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + THREADS * k;
                unsigned x = 0;
                if (4 * d + 4 <= n_bytes) {
                    x = ((const unsigned*)src)[d];
                } else if (4 * d < n_bytes) {                                // the last, partial dword (7-channel reads)
                    for (int b = 0; b < n_bytes - 4 * d; ++b) x |= (unsigned)src[4 * d + b] << (8 * b);
                }
                v[k] = x;
            }
#pragma unroll
            for (int k = 0; k < NDW; ++k) {
                const int d = tid + THREADS * k;
                if (4 * d < CF::U8_BYTES) ((unsigned*)s_u8)[d] = v[k];
            }
        } else {
            for (int i = tid; i < CF::U8_BYTES; i += THREADS) s_u8[i] = (i < n_bytes) ? src[i] : (unsigned char)0;
        }
        f32x4 ws2[8];
        load_weights<8>(ws2, W + CF::OFF_S2, wave % 2, lane);
        __syncthreads();
        wide_stem_conv1(s_u8, X, W + CF::OFF_S1, ch, dump, wave, lane);
        __syncthreads();
        wino_layer<wt::StemCfg, 32, MODE_PLAIN, false>(X, H, ws2, nullptr, W + CF::OFF_S2 + CF::W_S2, 0u, dump, wave, lane);
        __syncthreads();
        if (tid < 16 * (G + 1)) *(f32x4*)(X + img_off<64, SW_3>((tid >> 4) * RS1, tid & 15)) = zero4;   // the shared zero rows
        wide_stem_conv3_pool(H, X, W + CF::OFF_S3, dump, wave, lane, n_here);
        __syncthreads();
        if (tid < 16) ((f32x4*)H)[tid] = zero4;                              // row 0 of H as the trunk's 64-channel image
    } else {
        // the group's pooled rows in ONE round trip: every thread requests its float4 first and stores them afterwards
        const f32x4* src = (const f32x4*)(a.pooled + read0 * (long long)(L1 * 64));
        constexpr int NLD = (G * L1 * 16 + THREADS - 1) / THREADS;
        const int n4 = n_here * L1 * 16;
        f32x4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + THREADS * k;
            v[k] = f < n4 ? src[f] : zero4;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + THREADS * k;
            if (f < G * L1 * 16) {
                const int rd = f / (L1 * 16), rem = f - rd * (L1 * 16);
                *(f32x4*)(X + img_off<64, SW_3>(1 + rd * RS1 + (rem >> 4), rem & 15)) = v[k];
            }
        }
        // the shared zero rows 0, 72, ..., 288 of X; row 0 of H (the layers store H's other zero rows themselves)
        if (tid < 16 * (G + 1)) *(f32x4*)(X + img_off<64, SW_3>((tid >> 4) * RS1, tid & 15)) = zero4;
        else if (tid < 16 * (G + 2)) ((f32x4*)H)[tid - 16 * (G + 1)] = zero4;
    }
    __syncthreads();

    // ---- 3 x ResidualBlock(64) ----------------------------------------------------------------------
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        const int off_a = CF::off_a(2 * blk), off_b = CF::off_a(2 * blk + 1);
        wino3_layer<CF, 64, MODE_PLAIN, false, RS1, SW_3, false>(X, H, w3, slice64(off_a), slice64(off_b), W + off_a + CF::WA, wave, lane);
        __syncthreads();
        if (blk < 2)
            wino3_layer<CF, 64, MODE_RESID_INPLACE, false, RS1, SW_3, false>(H, X, w3, slice64(off_b), slice64(CF::off_a(2 * blk + 2)),
                                                                             W + off_b + CF::WA, wave, lane);
        else
            wino3_layer<CF, 64, MODE_RESID_INPLACE, true, RS1, SW_3, false>(H, X, w3, slice64(off_b), nullptr, W + off_b + CF::WA, wave, lane);
        __syncthreads();
    }

    // ---- strided block 64 -> 128 -----------------------------------------------------------------------
    f32x4 ws[12], wsc[4];
    load_weights<12>(ws, W + CF::OFF_S, wave, lane);
    load_weights<4>(wsc, W + CF::OFF_SC, wave, lane);
    // X (walked three rows per lane so far) -> H in the two-rows-per-lane swizzle the stride-2 convolutions walk
    {
        constexpr int NCP = ((RS1 * G + 1) * 16 + THREADS - 1) / THREADS;
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
            const int f = tid + THREADS * k;
            if (f < (RS1 * G + 1) * 16)
                *(f32x4*)(H + img_off<64, SW_W>(f >> 4, f & 15)) = *(const f32x4*)(X + img_off<64, SW_3>(f >> 4, f & 15));
        }
    }
    __syncthreads();
    f32x4 sreg[CF::NSREG];
    // X becomes the 128-channel image: rows 0 and 36 G + 1 are its zero rows
    if (tid < 64) ((f32x4*)X)[(tid & 31) + (tid >> 5) * (L2 * G + 1) * 32] = zero4;
    conv_layer<CF, 64, 128, 3, 2, 1, RS1, L2, L2, L2 * G / 16, MODE_PLAIN, false, GEOM_TRUNK, 16, L2 * G, false, SW_W, SW_3>(
        H, X, ws, nullptr, W + CF::OFF_S + CF::WS, sreg, 0u, dump, wave, lane);
    // the 1x1 s2 shortcut, in the row order the F(3,3) epilogue of the block's second conv holds its outputs in
    conv_layer<CF, 64, 128, 1, 2, 0, RS1, L2, L2, CF::NSREG, MODE_TO_REGS, false, GEOM_WTRIPLE, 16, L2 * G, false, SW_W, SW_3>(
        H, nullptr, wsc, nullptr, W + CF::OFF_SC + CF::WSC, sreg, 0u, dump, wave, lane);
#pragma unroll
    for (int c = 0; c < 5; ++c) w3[0][c] = *(const f32x4*)(slice128(CF::off_b(0)) + c * 256);
    __syncthreads();
    if (tid < 64) ((f32x4*)H)[(tid & 31) + (tid >> 5) * (L2 * G + 1) * 32] = zero4;
    {
        const int j = lane & 15, q = lane >> 4;               // the shortcut moves into the output image (see readconv_kernel)
#pragma unroll
        for (int t = 0; t < CF::NSREG; ++t)
            *(f32x4*)(H + img_off_triple<128, SW_3>(t / 3, j, (t % 3) + 1, 4 * wave + q)) = sreg[t];
    }
    wino3_layer<CF, 128, MODE_RESID_INPLACE, false, L2>(X, H, w3, slice128(CF::off_b(0)), slice128(CF::off_b(1)),
                                                        W + CF::off_b(0) + CF::WB, wave, lane);
    __syncthreads();

    // ---- 3 x ResidualBlock(128) ----------------------------------------------------------------------
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        // an opaque zero ties the per-lane addresses to the block: recomputed per block instead of kept (spilled) across blocks
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        const int lane_b = lane + oz, wave_b = __builtin_amdgcn_readfirstlane(wave + oz);
        auto slice = [&](int off) { return W + off + wave_b * (8 * 5 * 256) + lane_b * 4; };
        const int off_a = CF::off_b(1 + 2 * blk), off_b = CF::off_b(2 + 2 * blk);
        wino3_layer<CF, 128, MODE_PLAIN, false, L2>(H, X, w3, slice(off_a), slice(off_b), W + off_a + CF::WB, wave_b, lane_b);
        __syncthreads();
        if (blk < 2)
            wino3_layer<CF, 128, MODE_RESID_INPLACE, false, L2>(X, H, w3, slice(off_b), slice(CF::off_b(3 + 2 * blk)),
                                                                W + off_b + CF::WB, wave_b, lane_b);
        else
            wino3_layer<CF, 128, MODE_RESID_INPLACE, true, L2>(X, H, w3, slice(off_b), nullptr, W + off_b + CF::WB, wave_b, lane_b);
        __syncthreads();
    }

    // ---- the group's reads, summed per allele in read order; a slot per allele of the group ------------
    const int slot0 = a.slot_of_group[blockIdx.x];
    const int first_allele = s_allele[0];
    int cur = first_allele;
    constexpr int NF = (L2 * 32 + THREADS - 1) / THREADS;     // float4 elements of a [36][128] frame per thread
    f32x4 carry[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) carry[i] = zero4;
    auto flush = [&]() {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + THREADS * i;
            if (f < L2 * 32)
                *(f32x4*)(a.partial + ((long long)(slot0 + cur - first_allele) * L2 + (f >> 5)) * 128 + 4 * (f & 31)) = carry[i];
            carry[i] = zero4;
        }
    };
    for (int rd = 0; rd < n_here; ++rd) {
        const int al = s_allele[rd];
        if (al != cur) {                                      // uniform
            flush();
            cur = al;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int f = tid + THREADS * i;
            if (f < L2 * 32) {
                const f32x4 v = *(const f32x4*)(H + img_off<128, SW_3>(1 + rd * L2 + (f >> 5), f & 31));
#pragma unroll
                for (int e = 0; e < 4; ++e) carry[i][e] += v[e];
            }
        }
    }
    flush();
}

hipError_t launch_readconv_wide(const ReadConvArgs& a, hipStream_t stream) {
    if (a.n_reads <= 0) return hipSuccess;
    if ((!a.pooled == !a.reads) || !a.w || !a.partial || !a.winograd || a.window != 150 || a.extra_blocks != 0 || a.softplus ||
        a.groups_per_wg != 1 || (a.reads && a.channels != 6 && a.channels != 7))
        return hipErrorInvalidValue;
    static bool configured_on[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!configured_on[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)readconv_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           wt::Cfg::LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)readconv_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wt::Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured_on[dev] = true;
    }
    const unsigned grid = (unsigned)((a.n_reads + wt::Cfg::G - 1) / wt::Cfg::G);
    if (a.reads) hipLaunchKernelGGL(readconv_wide_kernel<true>, dim3(grid), dim3(wt::Cfg::THREADS), wt::Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(readconv_wide_kernel<false>, dim3(grid), dim3(wt::Cfg::THREADS), wt::Cfg::LDS_BYTES, stream, a);
    return hipGetLastError();
}

}  // namespace hello
