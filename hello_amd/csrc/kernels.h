// Launchers of the gfx950 kernels behind the C ABI (include/hello_mi355x.h).
// Activations: float32, channels-last [rows][length][channels].  All launchers are asynchronous on
// `stream` and return the hipError_t of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hello {

// Compute units of the CURRENT device (hipGetDevice), cached per device: launch plans and the small- / large-launch kernel
// choice depend on it, and a process may drive several devices (256 when the query fails).
int device_cus();

struct ConvArgs {
    const void* src;      // float or uint8 [rows][lin][cin]
    const float* src2;    // conv_wino.hip, two-source form (a CONCAT folded in): channels [split, cin) come from this tensor
                          // [rows][lin][cin - split] and channels [0, split) from src [rows][lin][split]; nullptr otherwise
    int split;
    float* dst;           // [rows][lout][cout]
    const float* res;     // optional residual, same shape as dst, added after the activation
    const float* w;       // packed [cout_pad][kpad], k index = tap*cin + c, zero padded
    const float* bias;    // [cout_pad]
    long long m_total;    // rows * lout
    int lin, lout, cin, cout, k, stride, pad;     // cin: input channels a filter reads (= row channels / groups)
    int cin_stride;       // channels of an activation row (cin x groups)
    int groups;           // grouped convolution (nn.Conv1d groups): output block g reads input channels [g cin, (g + 1) cin); every
                          // workgroup's channel block lies inside one group (cout / groups a multiple of 128)
    int kpad;             // multiple of 32
    int cout_pad;         // multiple of 32
    int relu;                  // activation: 0 none, 1 ReLU, 2 Softplus (beta 1, threshold 20)
    int src_u8;
    unsigned wino_rows;        // conv_wino.hip: m_total / lin, filled in by its launcher
    int wino;                  // Winograd form (conv_wino.hip): F(3,3) when lin % 3 == 0, else F(2,3); w packed
                               // [cout][cin/8][T components][8], kpad = T*cin, T = outputs per tile + 2
};
hipError_t launch_conv1d(const ConvArgs& a, hipStream_t stream);
bool conv1d_wino_supported(const ConvArgs& a);
int conv1d_wino_outputs_per_tile(int length);      // 3 when the row length is a multiple of 3, else 2
hipError_t launch_conv1d_wino(const ConvArgs& a, hipStream_t stream);

hipError_t launch_maxpool(const float* src, float* dst, long long rows, int lin, int lout, int c,
                          int k, int stride, int pad, hipStream_t stream);

// dst[s][e] = sum_{r in [off[s], off[s+1])} src[r][e], e < row_floats (row_floats % 4 == 0)
hipError_t launch_segsum(const float* src, float* dst, const int32_t* off, int n_seg, int row_floats,
                         hipStream_t stream);

// dst[a][e] = a0*src0[a][e] + a1*src1[owner[a]][e]
hipError_t launch_mix(const float* src0, const float* src1, float* dst, const int32_t* owner,
                      long long rows, int row_floats, float a0, float a1, bool rest, hipStream_t stream);

// out[o*out_stride_o + row*out_stride_row] = b[o] + sum_c W[o][c] * mean_l src[row][l][c]; optional softmax over o
hipError_t launch_head(const float* src, const float* w, const float* b, float* out, long long rows,
                       int len, int c, int cout, long long out_stride_o, long long out_stride_row,
                       int softmax, hipStream_t stream);

hipError_t launch_concat(const float* src0, const float* src1, float* dst, long long rows_x_len, int c0,
                         int c1, hipStream_t stream);

hipError_t launch_add(const float* src0, const float* src1, float* dst, long long n, hipStream_t stream);

// LayerNorm over the c channels of every position, then activation (0 none | 1 ReLU | 2 Softplus), then + res
hipError_t launch_layernorm(const float* src, const float* res, float* dst, const float* gamma, const float* beta,
                            long long positions, int c, float eps, int act, hipStream_t stream);

// uint8 [R][C][L] -> uint8 [R][L][C]
hipError_t launch_rcl_to_rlc(const uint8_t* src, uint8_t* dst, long long rows, int len, int c,
                             hipStream_t stream);

// MoEMergedWrapperAdvanced posterior section; out rows: mix, e0, e1, e2, each [n_pairs_total]
hipError_t launch_posteriors(const float* logits, const float* meta, const int32_t* allele_off,
                             const int64_t* pair_off, int n_sites, long long n_alleles, int n_experts,
                             long long n_pairs_total, float* out, hipStream_t stream);

// ---- fused read convolver (readconv_fused.hip) ------------------------------------------------
struct ReadConvArgs {
    const uint8_t* reads;      // [R][150][channels] pileups: the stem runs inside the kernel; or NULL and
    const float* pooled;       // [R][71][32] output of a layer-by-layer stem (3 valid convs + MaxPool1d(3,2))
    int channels;              // 6 | 7 (only with `reads`)
    const float* w;            // packed block, see hello_amd/readconv_pack.py
    float* partial;            // [n_slots][36][64]: one slot per (workgroup, allele) incidence
    const int32_t* allele_of_read;   // [R]
    const int32_t* slot_of_group;    // [n_workgroups + 1] first partial slot of each workgroup
    int groups_per_wg;         // consecutive groups of readconv_reads_per_group() reads one workgroup walks
    long long n_reads;
    int extra_blocks;          // identity-shortcut 64-channel blocks after the canonical three: 0 | 2
    int winograd;              // k3/s1 convolutions in Winograd form: F(3,3) trunk at 150 bp, else F(2,3) (weights packed accordingly)
    int window;                // pileup window: 150 | 250 (250: `reads` + Winograd form only)
    int softplus;              // Softplus instead of ReLU (`reads` + Winograd form, 150 bp only)
    int bf16x3;                // arithmetic mode bf16x3 (`reads` + Winograd form, 150 bp, ReLU, no extra blocks): 1 = the 64-channel
                               // trunk on the bf16 matrix cores as 3-term splits, 2 = the 32-channel blocks too ("bf16x3+32");
                               // the split weights follow the fp32 blob
    unsigned long long* stamps;   // diagnostic launch (hello_engine_debug_stamps) or NULL: per-wave s_memtime stamps,
                                  // [workgroups][4 waves][stamp_groups][readconv_stamp_slots()] -- see readconv_kernel
    int stamp_groups;          // groups per workgroup that record (>= groups_per_wg records all)
    int stamp_mode;            // bit 1: pad the LDS so that one workgroup is resident per CU
};
int readconv_stamp_slots();
int readconv_stamp_waves();
bool readconv_supports_window(int window);
int readconv_reads_per_group(int window);
int readconv_frame_rows(int window);       // positions per read after the read convolver: 36 | 61
int readconv_groups_per_workgroup(long long n_reads, int window);
// How a launch covers its groups of reads: `bulk_wgs` workgroups of `groups_per_wg` groups (whole rounds of the
// device's resident workgroup slots), then -- in a second launch of the same kernel -- `rest_wgs` workgroups of ONE
// group for what is left, so the partial last round costs one group's time instead of `groups_per_wg`.
struct ReadConvPlan {
    int groups_per_wg;
    long long bulk_wgs;      // each groups_per_wg groups (the last one possibly fewer when rest_wgs == 0)
    long long rest_wgs;      // each one group (the last one possibly fewer reads)
};
ReadConvPlan readconv_plan(long long n_reads, int window);
ReadConvPlan readconv_wide_plan(long long n_reads);   // every workgroup one group, one launch
int readconv_weight_floats(int extra_blocks, bool winograd, int window);
int readconv_bf16x3_extra_floats(int extra_blocks);     // floats the bf16x3 mode's split weights add behind that blob   // 150 bp + Winograd: residual trunk in F(3,3) form
bool readconv_supports_extra_blocks(int extra_blocks);
hipError_t launch_readconv_fused(const ReadConvArgs& a, hipStream_t stream);
// frames[a] = sum of the partial slots of allele a, in slot order
// frames[a] = sum of allele a's partial slots; `channels` per position (64; 128 for the wide trunk)
hipError_t launch_readconv_finalize(const float* partial, const int32_t* slot_off, float* frames, int n_alleles,
                                    int frame_rows, int channels, hipStream_t stream);
// Residual trunk of the 2x-channel read convolver (read_convolver_wide.py after its max pool) + segment sum:
// `pooled` = [R][71][64] rows of a layer-by-layer stem, partial slots [36][128]; one group of
// readconv_wide_reads_per_group() reads per workgroup (groups_per_wg must be 1), Winograd form only.
int readconv_wide_weight_floats();
int readconv_wide_reads_per_group();
hipError_t launch_readconv_wide(const ReadConvArgs& a, hipStream_t stream);

// ---- fused allele-level compressor (readconv_fused.hip) ---------------------------------------------------
// architectures/compressor_conv_small.py (and ExpertAlleleConvolver*.py): 1x1 64->64, strided block 64->128 with
// its 1x1 shortcut, `blocks` identity residual blocks at 128 channels; [items][36][64] -> [items][18][128].
struct CompressorArgs {
    const float* frames;       // [items][36][64]
    float* dst;                // [items][18][128]
    const float* w;            // packed block, hello_amd/readconv_pack.py pack_compressor
    long long n_items;
    int blocks;                // identity residual blocks after the strided one: 2 | 3
    int items_per_wg;          // 1..8 items a workgroup carries; 0: the launcher picks (small_launch_items_per_wg)
};
// Items per workgroup of the LDS-resident allele-stage kernels (compressor_kernel, xattn_front_kernel; one workgroup per CU).  A
// workgroup's time grows with the items it carries (compressor: 84 us with 2, 121 us with 8; front: 29 / 57 us) while a small launch
// leaves most CUs idle: one or two items per workgroup while that gives each its own CU (items <= 2 CUs), else whole workgroups of `most`.
int small_launch_items_per_wg(long long n_items, int most);
int compressor_weight_floats(int blocks);
bool compressor_supports_blocks(int blocks);
hipError_t launch_compressor_fused(const CompressorArgs& a, hipStream_t stream);

// ---- fused front of the allele-level expert (readconv_fused.hip) ----------------------------------------------------------
// architectures/xattn_subtract.py:9-60: x = a0 a + a1 s[owner], 1x1 128->128 + ReLU, then the strided block's first convolution
// (k3 s2 p1 128->256 + ReLU -> y2) and its 1x1 s2 shortcut (-> sc); [items][18][128] -> two [items][9][256] tensors.
struct XattnFrontArgs {
    const float* alleles;      // [items][18][128] compressed allele frames
    const float* sites;        // [sites][18][128] their per-site sums; nullptr: the kernel forms a site's sum from its alleles' rows
    const int32_t* owner;      // [items] site of each allele
    const int32_t* site_off;   // [sites + 1] first allele of each site (read when sites == nullptr)
    float* y2;                 // [items][9][256]
    float* sc;                 // [items][9][256]
    const float* w;            // packed block, hello_amd/readconv_pack.py pack_xattn_front
    long long n_items;
    float a0, a1;              // LinearCombination coefficients (2, -1)
    int rest;                  // x = a - (s - a) in that rounding order (MoEMergedAdvanced, :372-383) instead of a0 a + a1 s
    int items_per_wg;          // 1..8 items a workgroup carries; 0: the launcher picks (small_launch_items_per_wg)
};
int xattn_front_weight_floats();
hipError_t launch_xattn_front(const XattnFrontArgs& a, hipStream_t stream);

// ---- pileup-tensor producer (featurize.hip) ----------------------------------------------------------
struct FeaturizeArgs {
    const uint8_t* bases;            // all reads' bases, concatenated (ASCII)
    const uint8_t* quals;            // all reads' base qualities, same offsets
    const long long* read_off;       // [R+1]
    const uint32_t* cigars;          // all reads' CIGAR operations, BAM packing: length << 4 | operation
    const long long* cigar_off;      // [R+1]
    const long long* ref_start;      // [R] genome position of the first aligned base
    const uint8_t* mapq;             // [R]
    const int8_t* orientation;       // [R] > 0 forward
    const uint8_t* hp;               // [R] haplotag 0 | 1 | 2
    const int32_t* site_of_read;     // [R]
    const uint8_t* ref;              // all sites' reference windows, concatenated (ASCII)
    const long long* ref_off;        // [S+1]
    const long long* window_start;   // [S] genome position of ref window byte 0
    const long long* asm_start;      // [S] allele span (assemblyStart, assemblyStop)
    const long long* asm_stop;       // [S]
    long long n_reads;
    int length, channels;            // feature length L, 6 | 7
    uint8_t* out;                    // [R][L][C]
};
hipError_t launch_featurize(const FeaturizeArgs& a, hipStream_t stream);

}  // namespace hello
