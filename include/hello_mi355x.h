/* hello_mi355x.h -- C ABI of the MI355X (gfx950) engine for HELLO's variant-scoring hot path.
 *
 * The reference has no native interface on this path: the network is a pickled torch module that
 * python/caller_calling.py:863-868 loads and :651-652 calls per site, and that
 * python/MixtureOfExpertsDNNFast.py:120-134 calls per batch.  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md); each entry point names the reference
 * interface it stands in for.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative hello_status; it never throws; the message
 *     for the calling thread's last failure is hello_last_error().
 *   - pileup tensors are uint8, channels-last [reads][window][channels], exactly what the C++
 *     featurizer emits (c++/src/AlleleSearcherLiteFiltered.cpp:1045,1172).  HELLO_LAYOUT_RCL accepts
 *     the training-storage layout [reads][channels][window] (python/MemmapDatasetLoader.py:68-74).
 *   - count arrays (reads per allele, alleles per site) are ALWAYS host memory (they are tiny and the
 *     engine builds its CSR offsets from them on the host); the large tensors and the outputs are
 *     host or device memory according to `flags`.
 *   - an engine instance is not re-entrant: one instance per (process, device), calls issued from one
 *     thread at a time, all on the same stream.  With device outputs the call is asynchronous and
 *     stream-ordered; with host outputs it returns after the results have landed.
 */
#ifndef HELLO_MI355X_H
#define HELLO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HELLO_ABI_VERSION 2      /* 2: CONV1D groups in c1, two-source CONV1D (src1 / seg), XATTN_FRONT, flags 64 / 128 on
                                    READCONV_FUSED only (a CONV1D carrying them is refused), hello_site_records */

typedef enum hello_status {
    HELLO_OK = 0,
    HELLO_ERR_ARG = -1,      /* NULL / negative / inconsistent argument                      */
    HELLO_ERR_SHAPE = -2,    /* counts do not add up (sum reads_per_allele != n_reads, ...)  */
    HELLO_ERR_MODEL = -3,    /* malformed model description                                  */
    HELLO_ERR_HIP = -4,      /* a HIP runtime call failed                                    */
    HELLO_ERR_NOGPU = -5,    /* no gfx950 device visible                                     */
    HELLO_ERR_NOMEM = -6     /* host memory exhausted inside the library (no C++ exception crosses the boundary) */
} hello_status;

/* ---- model description: a flat program over activation buffers ------------------------------
 * The Python host (hello_amd/compiler.py) lowers a MoEAttention model
 * (python/MixtureOfExpertsAdvanced.py:71-252, built from python/NNTools.py:633-657 layer lists) into
 * this program once, with weight-norm / batch-norm already folded into the weights
 * (python/NNTools.py:780-799).  Activations are float32, channels-last [rows][length][channels]. */

typedef enum hello_domain {          /* what a buffer's rows are */
    HELLO_ROWS_READS0 = 0,
    HELLO_ROWS_READS1 = 1,
    HELLO_ROWS_ALLELES = 2,
    HELLO_ROWS_SITES = 3
} hello_domain;

typedef enum hello_segment {         /* which ragged reduction (reduceSlots, MixtureOfExpertsAdvanced.py:23-34) */
    HELLO_SEG_READS0_TO_ALLELES = 0,
    HELLO_SEG_READS1_TO_ALLELES = 1,
    HELLO_SEG_ALLELES_TO_SITES = 2
} hello_segment;

/* reserved buffer ids: the caller's inputs */
#define HELLO_BUF_NONE   (-1)
#define HELLO_BUF_READS0 0           /* uint8 [R0][L][C0] */
#define HELLO_BUF_READS1 1           /* uint8 [R1][L][C1] */
#define HELLO_BUF_REF    2           /* uint8 [S][L][5]  one-hot reference (caller_calling.py:53-97) */
#define HELLO_BUF_FIRST_SCRATCH 3

typedef enum hello_op_kind {
    HELLO_OP_CONV1D = 1,     /* Conv1d + bias (+ReLU) (+ residual add after the activation)          */
    HELLO_OP_MAXPOOL = 2,    /* MaxPool1d(k, stride, pad), floor mode                                 */
    HELLO_OP_SEGSUM = 3,     /* dst[seg] = sum of src rows of the segment (reduceSlots)               */
    HELLO_OP_MIX = 4,        /* dst[a] = a0*src0[a] + a1*src1[site(a)]  (xattn_subtract.py:14-42)      */
    HELLO_OP_HEAD = 5,       /* mean over length, Linear C->cout (terminus, NNTools.py:517-566);      */
                             /* writes output slot `dst`: 0..2 expert logits, 3 meta (+softmax)       */
    HELLO_OP_CONCAT = 6,     /* channel concat of src0 (cin ch) and src1 (c1 ch)                      */
    HELLO_OP_ADD = 7,        /* dst = src0 + src1                                                     */
    HELLO_OP_READCONV_FUSED = 8 /* whole read convolver + reads->alleles segment sum in one kernel:
                                 * canonical architecture (architectures/read_convolver.py) on 150 bp windows
                                 * (+0 | 2 extra blocks in k; from the bytes with FLAG_SRC_U8, else from the
                                 * pooled stem output), or on 250 bp windows (from the bytes, Winograd form);
                                 * dst rows are [36 | 61][64] per allele                                   */
    ,
    HELLO_OP_LAYERNORM = 9   /* LayerNormModule (NNTools.py:802-828) between a convolution and its activation:
                              * dst[r][l][:] = act((src0[r][l][:] - mean) / sqrt(var + a0) * gamma + beta) (+ res), mean and
                              * biased variance over the cin channels of the position; gamma at w_off, beta at b_off,
                              * eps in a0; flags: HELLO_FLAG_RELU | HELLO_FLAG_SOFTPLUS                                */
    ,
    HELLO_OP_COMPRESSOR_FUSED = 10 /* the whole allele-level compressor (architectures/compressor_conv_small.py:8-55) in
                              * one LDS-resident kernel: 1x1 64->64, strided block 64->128 with its 1x1 shortcut, k (2 .. 4)
                              * identity residual blocks; src0 rows [36][64] -> dst rows [18][128]; ReLU; k3/s1 convolutions
                              * in Winograd F(3,3) form, weights packed by hello_amd/readconv_pack.py pack_compressor       */
    ,
    HELLO_OP_XATTN_FRONT = 11 /* the front of the allele-level expert (architectures/xattn_subtract.py:9-60) in one LDS-resident
                              * kernel: x = a0 src0[a] + a1 src1[site(a)] (HELLO_FLAG_MIX_REST: x = src0[a] - (src1[site(a)] - src0[a]), as
                              * MIX) ([18][128] rows; src1 = HELLO_BUF_NONE: the site's row is
                              * the sum of its alleles' src0 rows, formed in the kernel in allele order -- the SEGSUM of src0 over
                              * HELLO_SEG_ALLELES_TO_SITES folded in, same bits), 1x1 128->128 + ReLU, then the strided
                              * block's first convolution (k3 s2 p1 128->256 + ReLU) -> dst and its 1x1 s2 shortcut -> the buffer
                              * named by `res` (an OUTPUT here: the block's second convolution reads it as its residual); both
                              * [9][256] rows; weights packed by hello_amd/readconv_pack.py pack_xattn_front                  */
} hello_op_kind;

#define HELLO_FLAG_RELU     1
#define HELLO_FLAG_SRC_U8   2        /* src0 is one of the uint8 input buffers                       */
#define HELLO_FLAG_SOFTMAX  4        /* HEAD: softmax over cout (meta expert, :229-232)              */
#define HELLO_FLAG_SOFTPLUS 16       /* CONV1D: Softplus(beta 1, threshold 20) instead of ReLU (…_layer_norm.py:16) */
#define HELLO_FLAG_WINOGRAD 32       /* READCONV_FUSED: k3/s1 convolutions in Winograd form (150 bp: residual trunk
                                        F(3,3), stem F(2,3); 250 bp: F(2,3)); weights as hello_amd/readconv_pack.py
                                        packs them for that form and window;
                                        CONV1D (k 3, stride 1, pad 1): weights are the T Winograd taps, packed
                                        [cout][cin/8][T][8]; T = 5 (F(3,3)) when lin % 3 == 0, else 4 (F(2,3)) */
#define HELLO_FLAG_BF16X3   64       /* READCONV_FUSED (whole kernel from the bytes, 150 bp, ReLU, Winograd form, k = 0): arithmetic mode
                                        "bf16x3" -- the seven 64 -> 64 trunk convolutions on the bf16 matrix cores as 3-term
                                        splits x w ~= xh wh + xh wl + xl wh of pre-split operands; their split weights
                                        (hello_amd/readconv_pack.py pack_bf16x3) follow the op's fp32 weight block.  Never
                                        the default: results differ from exact fp32 at the 1e-6 level of the activations */
#define HELLO_FLAG_BF16X3_32 128     /* with HELLO_FLAG_BF16X3 ("bf16x3+32"): the six 32 -> 32 convolutions of the ResidualBlock(32)s too */
#define HELLO_FLAG_MIX_REST 8        /* MIX: dst[a] = src0[a] - (src1[site(a)] - src0[a])  (:372-383)  */
/* Lanes (an addition within ABI version 2): bits 8..10 of `flags` name the stream an op runs on, 0 = the call's own stream.  A model
 * of two read technologies / three experts is several INDEPENDENT chains (MixtureOfExpertsAdvanced.py:161-252: read convolver +
 * compressor + expert per technology, the combined expert, the meta network); a launch of a few sites is latency-bound -- every chain
 * is a handful of workgroups -- so a program for SMALL launches puts the chains on lanes and the engine runs them concurrently,
 * ordering lanes with events wherever an op reads what another lane wrote (derived from the buffer ids).  Such a program must write
 * every scratch buffer from ONE op (no buffer reuse): hello_engine_create refuses it otherwise.  Same kernels, same bits. */
#define HELLO_FLAG_LANE_SHIFT 8
#define HELLO_FLAG_LANE_MASK  (7 << HELLO_FLAG_LANE_SHIFT)
#define HELLO_MAX_LANES 8

typedef struct hello_op {
    int32_t kind;        /* hello_op_kind */
    int32_t domain;      /* hello_domain of dst rows */
    int32_t src0, src1;  /* buffer ids; CONV1D with src1 != HELLO_BUF_NONE: a CONCAT folded in -- input channels [0, seg) are src0's
                          * rows [lin][seg], channels [seg, cin) src1's rows [lin][cin - seg] (dense Winograd form, ReLU, no residual) */
    int32_t dst;         /* buffer id; HEAD: output slot */
    int32_t res;         /* residual buffer id or HELLO_BUF_NONE */
    int32_t cin, cout;   /* channels */
    int32_t k, stride, pad;     /* READCONV_FUSED: k = extra identity 64-channel blocks after the canonical 3 (0 | 2) */
    int32_t lin, lout;   /* positions per row before / after */
    int32_t flags;
    int32_t seg;         /* hello_segment (SEGSUM / MIX / READCONV_FUSED); two-source CONV1D: channels of src0 */
    int32_t c1;          /* CONCAT: channels of src1; CONV1D: groups of a grouped convolution (nn.Conv1d groups; 0 / 1 = dense):
                          * output block g reads input channels [g cin/groups, (g+1) cin/groups), weights packed per output
                          * channel over ITS group's k * cin/groups inputs; cout/groups a multiple of 128, cin/groups of 16 */
    float a0, a1;        /* MIX coefficients */
    int64_t w_off;       /* float offset of the packed weights in the weight blob */
    int64_t b_off;       /* float offset of the bias */
} hello_op;

typedef struct hello_buffer {        /* scratch buffer i (i >= HELLO_BUF_FIRST_SCRATCH) */
    int32_t domain;                  /* hello_domain: rows scale with the batch */
    int32_t floats_per_row;          /* max length*channels over all uses */
} hello_buffer;

typedef struct hello_model_desc {
    int32_t abi_version;             /* HELLO_ABI_VERSION */
    int32_t window;                  /* L, 150 for the shipped models (python/call.py:187) */
    int32_t channels0, channels1;    /* 6 (7 with the haplotag channel); channels1 = 0: single tech */
    int32_t n_experts;               /* rows of `logits`: 1 (single expert) or 3 (ensemble) */
    int32_t has_meta;                /* meta mixing weights are produced */
    int32_t uses_ref;                /* HELLO_BUF_REF is read (meta_convolver_ref models) */
    int32_t n_buffers;               /* including the 3 reserved ids */
    const hello_buffer* buffers;     /* [n_buffers]; entries 0..2 ignored */
    int32_t n_ops;
    const hello_op* ops;
} hello_model_desc;

typedef struct hello_engine hello_engine;

/* ---- flags of hello_engine_forward ---------------------------------------------------------- */
#define HELLO_IN_DEVICE   1          /* reads0 / reads1 / ref_onehot are device pointers             */
#define HELLO_OUT_DEVICE  2          /* logits / meta / posteriors are device pointers               */
#define HELLO_LAYOUT_RCL  4          /* reads are [R][C][L] instead of [R][L][C]                     */

/* Stands in for: torch.load(args.network) + network.eval() (caller_calling.py:863-868); the model is
 * replicated per process like the reference's per-worker copy (call.py:215-221).  `folded_weights`
 * (host, nbytes) is copied to the device. */
int hello_engine_create(const hello_model_desc* desc, const void* folded_weights, size_t nbytes,
                        int hip_device, hello_engine** out);

/* Stands in for: MoEAttention.forward(tensors, numAllelesPerSite, numReadsPerAllele,
 * reference_segments) (MixtureOfExpertsAdvanced.py:161; batched callers
 * MixtureOfExpertsDNNFast.py:128-134).
 *   reads0 [R0][L][C0], reads_per_allele0 [A] (host), reads1/reads_per_allele1 likewise or NULL,
 *   alleles_per_site [S] (host), ref_onehot [S][L][5] or NULL,
 *   logits out [n_experts][A], meta out [S][3] or NULL,
 *   posteriors out [4][sum_s A_s(A_s+1)/2] or NULL: the wrapper's pair posteriors (see
 *   hello_engine_posteriors) computed in the same stream-ordered call.
 * Every allele must own >= 1 read (the featurizer inserts an all-zero dummy read,
 * AlleleSearcherLiteFiltered.cpp:1037-1043); violations are HELLO_ERR_SHAPE. */
int hello_engine_forward(hello_engine* engine,
                         const uint8_t* reads0, const int32_t* reads_per_allele0,
                         const uint8_t* reads1, const int32_t* reads_per_allele1,
                         const int32_t* alleles_per_site, const uint8_t* ref_onehot,
                         int32_t n_sites, int32_t n_alleles, int64_t n_reads0, int64_t n_reads1,
                         float* logits, float* meta, float* posteriors, int32_t flags, void* hip_stream);

/* Stands in for: the posterior section of MoEMergedWrapperAdvanced.forward
 * (MixtureOfExpertsAdvanced.py:530-589): sigmoid, probability of every unordered allele pair in
 * first-seen itertools.product order, mixture over experts.  n_pairs_total = sum_s A_s(A_s+1)/2.
 *   logits [n_experts][A] and meta [S][3] as written by hello_engine_forward (meta NULL => [1,0,0]),
 *   out [4][n_pairs_total]: rows mix, expert0, expert1, expert2.
 * `flags`: HELLO_IN_DEVICE for logits/meta, HELLO_OUT_DEVICE for out. */
int hello_engine_posteriors(hello_engine* engine, const float* logits, const float* meta,
                            const int32_t* alleles_per_site, int32_t n_sites, int32_t n_alleles,
                            int64_t n_pairs_total, float* out, int32_t flags, void* hip_stream);

/* Stands in for: AlleleSearcherLite.computeFeaturesColoredSimple (the C++ featurizer behind
 * python/AlleleSearcherLite.py:232-251, c++/src/AlleleSearcherLiteFiltered.cpp:971-1180), batched: every
 * read of every allele of every site in one launch, written as uint8 [n_reads][feature_length][channels]
 * ready to be passed to hello_engine_forward as `reads0` / `reads1`.
 *   bases / quals          all reads' bases (ASCII) and base qualities, concatenated; read_offsets [R+1]
 *   cigars                 BAM packing (length << 4 | operation), concatenated; cigar_offsets [R+1]
 *   ref_starts, mapq, orientation (> 0 forward), hp (0|1|2), site_of_read      per read
 *   ref_windows            all sites' reference windows (ASCII), concatenated; ref_window_offsets [S+1]
 *   window_starts, assembly_starts, assembly_stops                             per site (genome positions)
 * A read with zero CIGAR operations yields the all-zero dummy row of an unsupported allele (:1037-1043).
 * `flags`: HELLO_IN_DEVICE when ALL input arrays are device pointers, HELLO_OUT_DEVICE for `out`. */
int hello_engine_featurize(hello_engine* engine,
                           const uint8_t* bases, const uint8_t* quals, const int64_t* read_offsets,
                           const uint32_t* cigars, const int64_t* cigar_offsets,
                           const int64_t* ref_starts, const uint8_t* mapq, const int8_t* orientation,
                           const uint8_t* hp, const int32_t* site_of_read,
                           const uint8_t* ref_windows, const int64_t* ref_window_offsets,
                           const int64_t* window_starts, const int64_t* assembly_starts,
                           const int64_t* assembly_stops,
                           int64_t n_reads, int32_t n_sites, int32_t feature_length, int32_t channels,
                           uint8_t* out, int32_t flags, void* hip_stream);

/* The engine's own stream (a hipStream_t): what a call with hip_stream == NULL runs on.  It is created
 * non-blocking, so it is NOT ordered with the legacy default stream: a caller whose other work sits on the default
 * stream (handle 0, indistinguishable from NULL here) orders the two with events on this handle -- the Python
 * binding does (hello_amd/engine.py). */
void* hello_engine_stream(hello_engine* engine);

/* Host memory the GPU reads and writes in place (pinned, mapped, coherent): pointers into such a block may be passed to
 * hello_engine_forward under HELLO_IN_DEVICE / HELLO_OUT_DEVICE -- a small batch then crosses PCIe inside the kernels' own loads and
 * stores, with no staging copy and no copy-engine hop (followed by hello_engine_synchronize before the host reads the outputs).
 * NULL when the allocation fails (no GPU).  An addition within ABI version 2. */
void* hello_pinned_alloc(size_t bytes);
void hello_pinned_free(void* block);

/* Wait for everything the engine has enqueued on its last stream. */
int hello_engine_synchronize(hello_engine* engine);

/* Seconds of device time between the start and end of the most recent forward (HIP events on the
 * call's stream); negative if unavailable.  Used by bench.py's roofline leg. */
int hello_engine_last_forward_ms(hello_engine* engine, float* ms);

/* Per-op device time (HIP events on the call's stream around every op).  set_profiling(n > 0)
 * arms recording for the next n forwards; op_times_ms returns, per op, the SUM over the forwards
 * recorded since then and their number.  set_profiling(0) disarms. */
int hello_engine_set_profiling(hello_engine* engine, int max_forwards);
int hello_engine_op_times_ms(hello_engine* engine, float* ms_sum, int32_t capacity, int32_t* n_ops,
                             int32_t* n_forwards);
/* Restrict the recording to ops of one hello_op_kind (0 = every op): two event records per matching op and
 * forward instead of one per op, cheap enough to stay armed inside a timed region (bench.py times the
 * dominant kernel, HELLO_OP_READCONV_FUSED, this way).  Ops of other kinds then report 0 ms. */
int hello_engine_set_profiling_filter(hello_engine* engine, int32_t op_kind);

/* Debug read-back of one op's output (parity tests of single kernels; the reference's counterpart is a forward
 * hook on the module).  debug_capture(i >= 0) arms: every following forward copies the dst buffer of op i,
 * right after the op has run (scratch buffers are reused by later ops), into an engine-owned device buffer;
 * debug_capture(-1) disarms.  debug_read waits for the last forward and copies that snapshot to the host:
 * float32 [rows of the op's domain][positions][channels], `*n_floats` = its size (also when `out` is NULL or
 * `capacity` too small, which is HELLO_ERR_ARG unless out is NULL). */
int hello_engine_debug_capture(hello_engine* engine, int32_t op_index);
int hello_engine_debug_read(hello_engine* engine, float* out, int64_t capacity, int64_t* n_floats);

/* Diagnostic timeline of the fused read convolver (how its time splits over its barrier-delimited sections; the reference has
 * no counterpart -- its profile is torch's).  debug_stamps(1) arms: following forwards launch a STAMPED instantiation of the kernel
 * (same code, plus per-wave s_memtime records at the start of each group of reads and on both sides of each of its 20 barriers,
 * written to an engine-owned buffer nothing else reads; results are unchanged, the launch is ~10 % slower);  3: the same with ONE
 * workgroup resident per CU (LDS padding); 0 disarms.  Only the canonical fp32 Winograd read convolver from the bytes.
 * debug_read_stamps copies the last stamped forward's records to the host: uint64 [layout[0] workgroups][layout[1] waves]
 * [layout[2] groups per workgroup][layout[3] slots]; layout[4] = how many of the workgroups walk layout[2] groups (the rest, from
 * the second launch, walk one).  Slots: 0 group start, 1 + 2 i / 2 + 2 i arrival at / release from barrier i, 41 HW_ID | XCC_ID << 32,
 * 42 / 43 s_memrealtime (100 MHz) at the group's start / end, 44 (group 0 only) after the workgroup's last flush. */
int hello_engine_debug_stamps(hello_engine* engine, int32_t mode);
int hello_engine_debug_read_stamps(hello_engine* engine, uint64_t* out, int64_t capacity, int64_t* n_words, int32_t* layout);

void hello_engine_destroy(hello_engine* engine);

/* ---- record stage (host only: no engine, no GPU) ---------------------------------------------------------------------
 * Stands in for, per site of a whole launch and on `n_threads` host threads:
 *   - caller_calling.vcfRecords' decision and record (python/caller_calling.py:698-743: best pair of the mixture row,
 *     QUAL = -10 log10(1 - min(p, 1 - 1e-8)), ALT list, genotype, createVcfRecord normalisation
 *     python/vcfFromContigs.py:139-227) with INFO "MixtureOfExpertPrediction" -> the shard's `.vcf` lines;
 *   - its `.features` entry (:743-754) as one pickle stream per shard (the list prepareVcf.py:138 loads);
 *   - prepareVcf.vcfRecords' call on the meta-weighted mean of the experts in float64 (python/prepareVcf.py:154-168) with
 *     INFO "HELLO" -> the lines of the final VCF, with their normalised positions for the final sort;
 *   - the best pair / probability / QUAL of all five rows (mixture, expert 0..2, mean).
 * ALT alleles are written in sorted order; equal pair probabilities are ordered by the pair's allele strings, as Python's
 * sort of (value, key) tuples orders them.  A site whose `keep` byte is 0 is decided but emits nothing. */
typedef struct hello_site_table {
    int32_t n_sites;
    const int32_t* alleles_per_site;         /* [S] */
    const uint8_t* allele_text;              /* every allele string of every site, concatenated (ASCII) */
    const int64_t* allele_text_off;          /* [A + 1] */
    int32_t n_chromosomes;
    const uint8_t* chromosome_text;          /* chromosome names, concatenated */
    const int64_t* chromosome_text_off;      /* [n_chromosomes + 1] */
    const int32_t* chromosome_of_site;       /* [S] index into the names */
    const int64_t* start;                    /* [S] allele span [start, stop) in genome coordinates (0-based) */
    const int64_t* stop;
    const uint8_t* ref_windows;              /* per-site reference windows, concatenated (or NULL with `genome`) */
    const int64_t* ref_window_off;           /* [S + 1] */
    const int64_t* window_start;             /* [S] genome position of each window's first byte */
    const uint8_t* const* genome;            /* [n_chromosomes] whole sequences (entries may be NULL) or NULL */
    const int64_t* genome_len;               /* [n_chromosomes] */
    const uint8_t* keep;                     /* [S] or NULL (= all) */
} hello_site_table;

typedef struct hello_features_format {       /* how `meta` (float32 [3]) is written inside a `.features` entry:     */
    const uint8_t* meta_prefix;              /* pickle opcodes before and after the 12 payload bytes -- the binding  */
    int32_t meta_prefix_len;                 /* takes them from its own NumPy's pickle of such an array, so the file */
    const uint8_t* meta_suffix;              /* loads wherever that NumPy's pickles load                             */
    int32_t meta_suffix_len;
} hello_features_format;

typedef struct hello_records hello_records;

typedef struct hello_records_view {          /* pointers into a hello_records object; valid until it is destroyed */
    int32_t n_sites, n_shards;
    const uint8_t* shard_vcf;                /* record lines ('\n'-terminated) in site order */
    const int64_t* shard_vcf_off;            /* [S + 1]: site s owns bytes [off[s], off[s+1]) (empty: no record) */
    const uint8_t* mean_vcf;                 /* the final VCF's lines for the same sites */
    const int64_t* mean_vcf_off;             /* [S + 1] */
    const int64_t* mean_position;            /* [S] 0-based position of the mean call after normalisation, -1: no record */
    const uint8_t* features;                 /* n_shards pickle streams, concatenated */
    const int64_t* features_off;             /* [n_shards + 1] */
    const int32_t* n_records;                /* [n_shards] */
    const int32_t* best_pair;                /* [5][S] pair index within the site: rows mixture, expert 0..2, mean */
    const double* best_p;                    /* [5][S] */
    const double* qual;                      /* [5][S] */
} hello_records_view;

/*   posteriors [4][n_pairs_total] and meta [S][3] (NULL = [1, 0, 0]) as written by hello_engine_forward, HOST memory;
 *   shard_site_off [n_shards + 1] site offsets of the shards the launch coalesced (NULL = one shard);
 *   fmt NULL = no `.features` streams; n_threads <= 0 = one per hardware thread. */
int hello_site_records(const hello_site_table* sites, const float* posteriors, int64_t n_pairs_total, const float* meta,
                       const int32_t* shard_site_off, int32_t n_shards, const hello_features_format* fmt,
                       int32_t n_threads, hello_records** out);
int hello_records_get(const hello_records* records, hello_records_view* view);
void hello_records_destroy(hello_records* records);

const char* hello_last_error(void);
int hello_abi_version(void);

/* ---- shared scoring server (host only; one per GPU) ------------------------------------------------------------------
 * Stands in for: the model copy every worker process of the reference's pool loads for itself and calls once per site
 * (python/call.py:111,215-221; python/caller_calling.py:863-868,872-891).  The workers keep their loop and their call
 * (hello_amd/shared.py: the client packs a site into its slot of a shared-memory segment, sends one byte on a Unix-domain
 * socket and blocks for the one-byte answer); ONE server process per (model file, GPU) drains all pending slots into one
 * hello_engine_forward launch and scatters logits / meta / pair posteriors back.  An addition within ABI version 2.
 *
 * Slot (offsets: hello_site_slot_layout_of): int32 header [16] = {alleles, reads0, reads1, has_ref, pairs, message length};
 * int32 reads_per_allele0 / 1 [HELLO_SITE_MAX_ALLELES]; uint8 reference one-hot [window][5]; results float32 logits
 * [3][HELLO_SITE_MAX_ALLELES], meta [4], posteriors [4][MAX_ALLELES (MAX_ALLELES + 1) / 2]; a message area (errors, statistics
 * as JSON); then the pileup bytes of technology 0 followed by technology 1.
 * Wire: the client sends <u32 length><JSON with "protocol">, the server answers <u32 length><JSON: slot, slot_bytes, max_clients,
 * shm_path, pid, the model's dimensions, `info_json`>; afterwards one byte each way per request: 'R' (score my slot) or 'S'
 * (statistics into my slot's message area) -> 'K' (done) or 'E' (refused: the reason is in the message area). */
#define HELLO_SITE_PROTOCOL 1
#define HELLO_SITE_MAX_ALLELES 64

typedef struct hello_site_slot_layout {      /* byte offsets inside one slot */
    int64_t header, rpa0, rpa1, ref, logits, meta, post, err, reads, read_capacity;
} hello_site_slot_layout;

typedef struct hello_site_server_config {
    int32_t window, channels0, channels1;    /* the model, as the clients are told (and as every slot is checked against) */
    int32_t n_experts, has_meta, uses_ref;
    int32_t max_clients;                     /* slots of the segment */
    int32_t max_batch_sites;                 /* sites of one launch at most */
    int32_t group_launches;                  /* != 0 with several engines: a launch takes clients / engines sites, the groups run out of phase */
    int32_t reserved;                        /* 0 */
    int64_t slot_bytes;
    double idle_exit_s;                      /* leave after this long without a client; < 0: never */
    double linger_s;                         /* patience for the clients that could still send a site before a launch goes out */
    const char* info_json;                   /* further members of the handshake object ("k": v, ...) or NULL */
} hello_site_server_config;

typedef struct hello_site_server_stats {
    int64_t launches, sites, errors;
    int32_t largest_launch, clients_seen;
} hello_site_server_stats;

/* A scorer other than an engine (tests drive the server without a GPU through this): fill logits [n_experts][A], meta [S][3]
 * (NULL when the model has none) and posteriors [4][sum_s A_s(A_s+1)/2]; return 0, or non-zero with a message in `err`. */
typedef int (*hello_site_scorer)(void* ctx, const uint8_t* reads0, const int32_t* reads_per_allele0, const uint8_t* reads1,
                                 const int32_t* reads_per_allele1, const int32_t* alleles_per_site, const uint8_t* ref_onehot,
                                 int32_t n_sites, int32_t n_alleles, int64_t n_reads0, int64_t n_reads1, float* logits, float* meta,
                                 float* posteriors, char* err, int32_t err_capacity);

typedef struct hello_site_server hello_site_server;

int hello_site_slot_layout_of(int32_t window, int32_t channels0, int32_t channels1, int64_t slot_bytes, hello_site_slot_layout* out);
/* Creates the segment file and the listening socket (a connectable socket = a ready server: add the scorers first or right after). */
int hello_site_server_create(const char* socket_path, const char* shm_path, const hello_site_server_config* config, hello_site_server** out);
/* One scorer thread per engine / scorer added; every engine is used from its own thread only. */
int hello_site_server_add_engine(hello_site_server* server, hello_engine* engine);
int hello_site_server_add_scorer(hello_site_server* server, hello_site_scorer scorer, void* ctx);
/* Blocks: serves until hello_site_server_stop (callable from any thread or a signal handler) or the idle limit. */
int hello_site_server_run(hello_site_server* server);
void hello_site_server_stop(hello_site_server* server);
int hello_site_server_get_stats(hello_site_server* server, hello_site_server_stats* out);
/* Closes the sockets, unmaps and unlinks the segment and the socket file. */
void hello_site_server_destroy(hello_site_server* server);

#ifdef __cplusplus
}
#endif
#endif /* HELLO_MI355X_H */
