// C ABI of the engine (include/hello_mi355x.h): model program interpreter, batch CSR preparation,
// scratch management, stream-ordered execution.  Host code only; kernels live in the other .hip files.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hello_mi355x.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace

namespace hello {
// the same thread-local message, for the other translation units of the library (records.hip)
int set_last_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
// No C++ exception crosses the C ABI: every entry point that allocates is a function-try-block ending here.
int exception_status(const char* where) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try { return set_last_error(HELLO_ERR_NOMEM, "%s: out of host memory", where); } catch (...) { return HELLO_ERR_NOMEM; }
    } catch (const std::exception& ex) {
        try { return set_last_error(HELLO_ERR_ARG, "%s: %s", where, ex.what()); } catch (...) { return HELLO_ERR_ARG; }
    } catch (...) {
        try { return set_last_error(HELLO_ERR_ARG, "%s: unknown C++ exception", where); } catch (...) { return HELLO_ERR_ARG; }
    }
}
}  // namespace hello

namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(HELLO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// grow-only device allocation
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            if (hipFree(p) != hipSuccess) return -1;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;   // slack so slightly larger batches do not reallocate
        if (hipMalloc(&p, want) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        size_t want = bytes + bytes / 8 + 256;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct hello_engine {
    int device = 0;
    hello_model_desc desc{};
    std::vector<hello_op> ops;
    std::vector<hello_buffer> buffers;
    float* d_weights = nullptr;
    size_t n_weight_floats = 0;

    std::vector<DevBuf> scratch;     // indexed by buffer id (0..2 = staged inputs when host pointers)
    DevBuf d_csr;                    // all per-batch index arrays, one allocation
    PinnedBuf h_csr;
    DevBuf d_logits, d_meta, d_post, d_rcl0, d_rcl1;
    DevBuf d_out_small;        // staged host-output calls too large to be written in place: logits | meta | posteriors in ONE block (one copy back)
    PinnedBuf h_io;            // small host calls: inputs and outputs pass through pinned memory (pageable copies stall)
    DevBuf d_partial[2];             // fused read convolver partial sums, per read technology (their ops may run on different lanes)
    // lanes (programs for small launches: HELLO_FLAG_LANE_*): streams 1.., per-op events where another lane reads the op's output
    int n_lanes = 1;
    std::vector<hipStream_t> lane_streams;      // [n_lanes], entry 0 unused (= the call's stream)
    std::vector<hipEvent_t> op_done;            // [n_ops]: recorded after ops whose output another lane reads, and after a lane's last op
    std::vector<std::vector<int>> op_waits;     // [n_ops]: ops on OTHER lanes whose output this op reads
    std::vector<int> lane_tail;                 // [n_lanes]: the last op of each lane
    hipEvent_t ev_lanes_go = nullptr;           // recorded on the call's stream once the inputs are staged: the other lanes start behind it
    DevBuf d_feat_in, d_feat_out;    // featurizer staging (host-pointer callers)
    hipStream_t own_stream = nullptr;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev_staged = nullptr;  // H2D of the pinned CSR block finished
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool staged_pending = false;
    bool timed = false;
    bool profiling = false;
    // profiling ring: events[f][i] brackets op i of the f-th profiled forward (i = n_ops: end marker)
    std::vector<std::vector<hipEvent_t>> prof_events;
    int prof_count = 0;              // forwards recorded since profiling was (re)enabled
    int prof_filter = 0;             // record only around ops of this kind (0: all)
    int debug_op = -1;               // op whose dst is snapshotted after it ran (-1: none)
    DevBuf d_debug;
    size_t debug_floats = 0;         // size of the last snapshot
    int stamp_mode = 0;              // hello_engine_debug_stamps: bit 0 record, bit 1 one workgroup per CU
    DevBuf d_stamps;
    int32_t stamp_layout[5] = {0, 0, 0, 0, 0};   // workgroups, waves, groups recorded per workgroup, slots, bulk workgroups

    // device views into d_csr for the current batch
    int32_t *roff0 = nullptr, *roff1 = nullptr, *aoff = nullptr, *site_of_allele = nullptr;
    int32_t *allele_of_read0 = nullptr, *allele_of_read1 = nullptr;
    int32_t *group_slot0 = nullptr, *group_slot1 = nullptr, *slot_off0 = nullptr, *slot_off1 = nullptr;
    hello::ReadConvPlan plan0{1, 0, 0}, plan1{1, 0, 0};      // how the fused read convolver's launches cover the reads
    bool wide_trunk[2] = {false, false};                      // technology t's fused op is the 2x-channel trunk kernel
    int64_t* pair_off = nullptr;
};

namespace {

long long rows_of(int domain, int32_t S, int32_t A, int64_t R0, int64_t R1) {
    switch (domain) {
        case HELLO_ROWS_READS0: return R0;
        case HELLO_ROWS_READS1: return R1;
        case HELLO_ROWS_ALLELES: return A;
        case HELLO_ROWS_SITES: return S;
    }
    return -1;
}

int validate_model(const hello_model_desc* d) {
    if (!d) return fail(HELLO_ERR_ARG, "model description is NULL");
    if (d->abi_version != HELLO_ABI_VERSION)
        return fail(HELLO_ERR_MODEL, "abi_version %d != %d", d->abi_version, HELLO_ABI_VERSION);
    if (d->window <= 0 || d->channels0 <= 0 || d->channels1 < 0)
        return fail(HELLO_ERR_MODEL, "bad window/channels");
    if (d->n_experts != 1 && d->n_experts != 3) return fail(HELLO_ERR_MODEL, "n_experts must be 1 or 3");
    if (d->n_buffers < HELLO_BUF_FIRST_SCRATCH || !d->buffers) return fail(HELLO_ERR_MODEL, "bad buffer table");
    if (d->n_ops <= 0 || !d->ops) return fail(HELLO_ERR_MODEL, "empty program");
    for (int i = HELLO_BUF_FIRST_SCRATCH; i < d->n_buffers; ++i) {
        if (d->buffers[i].domain < 0 || d->buffers[i].domain > 3 || d->buffers[i].floats_per_row <= 0)
            return fail(HELLO_ERR_MODEL, "buffer %d malformed", i);
    }
    auto buf_ok = [&](int id, bool allow_none) {
        if (id == HELLO_BUF_NONE) return allow_none;
        return id >= 0 && id < d->n_buffers;
    };
    for (int i = 0; i < d->n_ops; ++i) {
        const hello_op& o = d->ops[i];
        if (o.kind < HELLO_OP_CONV1D || o.kind > HELLO_OP_XATTN_FRONT)
            return fail(HELLO_ERR_MODEL, "op %d: unknown kind %d", i, o.kind);
        if (o.domain < 0 || o.domain > 3) return fail(HELLO_ERR_MODEL, "op %d: bad domain", i);
        if (!buf_ok(o.src0, false)) return fail(HELLO_ERR_MODEL, "op %d: bad src0", i);
        if (!buf_ok(o.src1, true) || !buf_ok(o.res, true)) return fail(HELLO_ERR_MODEL, "op %d: bad src1/res", i);
        if (o.kind == HELLO_OP_HEAD) {
            if (o.dst < 0 || o.dst > 3 || o.cout < 1 || o.cout > 4)
                return fail(HELLO_ERR_MODEL, "op %d: bad head slot/cout", i);
        } else if (o.dst < HELLO_BUF_FIRST_SCRATCH || o.dst >= d->n_buffers) {
            return fail(HELLO_ERR_MODEL, "op %d: dst must be a scratch buffer", i);
        }
        if (o.kind == HELLO_OP_CONV1D) {
            if (o.cin <= 0 || o.cout <= 0 || (o.cout % 4) || o.k <= 0 || o.stride <= 0 || o.pad < 0 ||
                o.lin <= 0 || o.lout <= 0 || o.w_off < 0 || o.b_off < 0)
                return fail(HELLO_ERR_MODEL, "op %d: bad conv geometry", i);
            if (o.src1 != HELLO_BUF_NONE &&
                !((o.flags & HELLO_FLAG_WINOGRAD) && (o.flags & HELLO_FLAG_RELU) && !(o.flags & (HELLO_FLAG_BF16X3 | HELLO_FLAG_SOFTPLUS)) &&
                  o.res == HELLO_BUF_NONE && o.c1 <= 1 && o.seg > 0 && o.seg < o.cin && o.seg % 16 == 0 && (o.cin - o.seg) % 16 == 0 &&
                  o.src1 != o.dst && o.src1 >= HELLO_BUF_FIRST_SCRATCH && d->buffers[o.src1].domain == o.domain))
                return fail(HELLO_ERR_MODEL, "op %d: a two-source convolution (src1: a folded CONCAT) is a dense Winograd convolution with "
                                             "ReLU and no residual whose sources hold seg and cin - seg channels, multiples of 16", i);
            if (o.c1 > 1 && !(o.cin % o.c1 == 0 && o.cout % o.c1 == 0 && (o.cout / o.c1) % 128 == 0 && (o.cin / o.c1) % 16 == 0 &&
                              !(o.flags & (HELLO_FLAG_SRC_U8 | HELLO_FLAG_BF16X3))))
                return fail(HELLO_ERR_MODEL, "op %d: a grouped convolution (c1 = %d groups) needs float input, cin / groups a multiple of 16 "
                                             "and cout / groups a multiple of 128 (a workgroup's channel block lies inside one group)", i, o.c1);
            if ((o.lin + 2 * o.pad - o.k) / o.stride + 1 != o.lout)
                return fail(HELLO_ERR_MODEL, "op %d: lout inconsistent", i);
            if (o.flags & (HELLO_FLAG_BF16X3 | HELLO_FLAG_BF16X3_32))
                return fail(HELLO_ERR_MODEL, "op %d: the bf16x3 arithmetic modes exist for the fused read convolver only (ABI 2 "
                                             "dropped the split-operand CONV1D)", i);
        }
        if (o.kind == HELLO_OP_COMPRESSOR_FUSED &&
            !(o.cin == 64 && o.cout == 128 && o.lin == 36 && o.lout == 18 && hello::compressor_supports_blocks(o.k) &&
              (o.flags & HELLO_FLAG_WINOGRAD) && o.w_off >= 0))
            return fail(HELLO_ERR_MODEL, "op %d: the fused compressor maps [36][64] rows to [18][128] with 2 to 4 identity blocks, "
                                         "Winograd form", i);
        if (o.kind == HELLO_OP_XATTN_FRONT &&
            !(o.domain == HELLO_ROWS_ALLELES && o.cin == 128 && o.cout == 256 && o.lin == 18 && o.lout == 9 && o.k == 3 && o.stride == 2 &&
              o.pad == 1 && o.seg == HELLO_SEG_ALLELES_TO_SITES && o.w_off >= 0 &&
              o.res >= HELLO_BUF_FIRST_SCRATCH && o.res != o.dst && o.res != o.src0 && o.res != o.src1 && o.dst != o.src0 && o.dst != o.src1 &&
              d->buffers[o.res].domain == HELLO_ROWS_ALLELES && d->buffers[o.res].floats_per_row >= 9 * 256 &&
              (o.src1 == HELLO_BUF_NONE || d->buffers[o.src1].domain == HELLO_ROWS_SITES)))
            return fail(HELLO_ERR_MODEL, "op %d: the fused expert front maps [18][128] allele rows (src0) and site rows (src1, or none: "
                                         "the sites' sums of src0) to two distinct [9][256] allele buffers, dst and res", i);
        if (o.kind == HELLO_OP_LAYERNORM && (o.cin <= 0 || o.cin > 512 || o.lin <= 0 || o.w_off < 0 || o.b_off < 0 || !(o.a0 > 0.f)))
            return fail(HELLO_ERR_MODEL, "op %d: bad LayerNorm (1..512 channels, eps > 0)", i);
        if ((o.kind == HELLO_OP_SEGSUM || o.kind == HELLO_OP_MIX || o.kind == HELLO_OP_READCONV_FUSED) &&
            (o.seg < 0 || o.seg > 2))
            return fail(HELLO_ERR_MODEL, "op %d: bad segment kind", i);
        if (o.kind == HELLO_OP_CONV1D && (o.flags & HELLO_FLAG_WINOGRAD) &&
            !(o.k == 3 && o.stride == 1 && o.pad == 1 && o.lin == o.lout && !(o.flags & HELLO_FLAG_SRC_U8) &&
              o.cin % 8 == 0 && o.cout % 64 == 0))
            return fail(HELLO_ERR_MODEL, "op %d: this convolution has no Winograd form (needs k 3, stride 1, pad 1, "
                                         "cin %% 8 == 0, cout %% 64 == 0, float input)", i);
        if (o.kind == HELLO_OP_READCONV_FUSED) {
            if (!hello::readconv_supports_window(d->window))
                return fail(HELLO_ERR_MODEL, "op %d: the fused read convolver takes 150 or 250 bp windows, not %d", i, d->window);
            const bool wide = o.cout == 128;       // the 2x-channel trunk: pooled [71][64] rows in, [36][128] frames out
            const bool wide_shape = (o.flags & HELLO_FLAG_SRC_U8) ? ((o.cin == 6 || o.cin == 7) && o.lin == 150) : (o.cin == 64 && o.lin == 71);
            if (wide && !(d->window == 150 && wide_shape && o.lout == 36 && o.k == 0 && (o.flags & HELLO_FLAG_WINOGRAD) &&
                          !(o.flags & HELLO_FLAG_SOFTPLUS)))
                return fail(HELLO_ERR_MODEL, "op %d: the wide read convolver maps pileup bytes [150][6|7] or pooled [71][64] rows to "
                                             "[36][128] frames (150 bp, Winograd form, ReLU, no extra blocks)", i);
            if (!wide && (o.lout != hello::readconv_frame_rows(d->window) || o.cout != 64))
                return fail(HELLO_ERR_MODEL, "op %d: the fused read convolver yields [%d][64] frames", i,
                            hello::readconv_frame_rows(d->window));
            if ((o.flags & HELLO_FLAG_SOFTPLUS) &&
                !(d->window == 150 && (o.flags & HELLO_FLAG_WINOGRAD) && (o.flags & HELLO_FLAG_SRC_U8) && o.k == 0))
                return fail(HELLO_ERR_MODEL, "op %d: the Softplus read convolver runs whole (from the bytes), in Winograd "
                                             "form, on 150 bp windows, without extra blocks", i);
            if ((o.flags & HELLO_FLAG_BF16X3_32) && !(o.flags & HELLO_FLAG_BF16X3))
                return fail(HELLO_ERR_MODEL, "op %d: HELLO_FLAG_BF16X3_32 extends HELLO_FLAG_BF16X3", i);
            if ((o.flags & HELLO_FLAG_BF16X3) &&
                !(d->window == 150 && !wide && (o.flags & HELLO_FLAG_WINOGRAD) && (o.flags & HELLO_FLAG_SRC_U8) && o.k == 0 &&
                  !(o.flags & HELLO_FLAG_SOFTPLUS)))
                return fail(HELLO_ERR_MODEL, "op %d: the bf16x3 arithmetic mode exists for the canonical read convolver run whole "
                                             "(from the bytes) in Winograd form on 150 bp windows, ReLU, without extra blocks", i);
            if (d->window == 250 && !((o.flags & HELLO_FLAG_WINOGRAD) && (o.flags & HELLO_FLAG_SRC_U8) && o.k == 0))
                return fail(HELLO_ERR_MODEL, "op %d: 250 bp windows run whole (from the bytes), in Winograd form, "
                                             "without extra blocks", i);
        }
        if (o.kind == HELLO_OP_READCONV_FUSED && !hello::readconv_supports_extra_blocks(o.k))
            return fail(HELLO_ERR_MODEL, "op %d: the fused read convolver takes 0 or 2 extra blocks (k), got %d", i, o.k);
    }
    return 0;
}

template <typename T>
T* carve(char*& cursor, size_t count) {
    T* p = reinterpret_cast<T*>(cursor);
    cursor += (count * sizeof(T) + 15) & ~size_t(15);
    return p;
}

}  // namespace

extern "C" {

const char* hello_last_error(void) { return g_last_error.c_str(); }
int hello_abi_version(void) { return HELLO_ABI_VERSION; }

int hello_engine_create(const hello_model_desc* desc, const void* folded_weights, size_t nbytes,
                        int hip_device, hello_engine** out) try {
    if (!out) return fail(HELLO_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (int rc = validate_model(desc)) return rc;
    if (!folded_weights || nbytes == 0 || (nbytes % 4)) return fail(HELLO_ERR_ARG, "bad weight blob");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(HELLO_ERR_NOGPU, "no HIP device visible: the HIP engine cannot run");
    if (hip_device < 0 || hip_device >= n_dev) return fail(HELLO_ERR_ARG, "device %d out of range", hip_device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, hip_device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HELLO_ERR_NOGPU, "device %d is %s; this engine is built for gfx950 only", hip_device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(hip_device));

    hello_engine* e = new hello_engine();
    e->device = hip_device;
    e->desc = *desc;
    e->ops.assign(desc->ops, desc->ops + desc->n_ops);
    e->buffers.assign(desc->buffers, desc->buffers + desc->n_buffers);
    e->desc.ops = e->ops.data();
    e->desc.buffers = e->buffers.data();
    e->n_weight_floats = nbytes / 4;
    for (const hello_op& o : e->ops) {
        // every weight block must lie inside the blob, in the layout its op (and arithmetic form) reads
        size_t w_end = 0, b_end = 0;
        if (o.kind == HELLO_OP_CONV1D) {
            const bool wino = (o.flags & HELLO_FLAG_WINOGRAD) != 0;
            const size_t cpad = wino ? (size_t)o.cout : (size_t)((o.cout + 31) / 32) * 32;
            const int cing = o.cin / (o.c1 > 1 ? o.c1 : 1);          // input channels a filter reads (grouped convolutions)
            const size_t kpad = wino ? (size_t)(hello::conv1d_wino_outputs_per_tile(o.lin) + 2) * cing
                                     : (size_t)((o.k * cing + 31) / 32) * 32;
            w_end = (size_t)o.w_off + cpad * kpad;
            b_end = (size_t)o.b_off + cpad;
        } else if (o.kind == HELLO_OP_HEAD) {
            w_end = (size_t)o.w_off + (size_t)o.cout * o.cin;
            b_end = (size_t)o.b_off + o.cout;
        } else if (o.kind == HELLO_OP_LAYERNORM) {
            w_end = (size_t)o.w_off + o.cin;
            b_end = (size_t)o.b_off + o.cin;
        } else if (o.kind == HELLO_OP_COMPRESSOR_FUSED) {
            w_end = (size_t)o.w_off + hello::compressor_weight_floats(o.k);
            b_end = (size_t)o.b_off;
        } else if (o.kind == HELLO_OP_XATTN_FRONT) {
            w_end = (size_t)o.w_off + hello::xattn_front_weight_floats();
            b_end = (size_t)o.b_off;
        } else if (o.kind == HELLO_OP_READCONV_FUSED) {
            w_end = (size_t)o.w_off + (o.cout == 128 ? hello::readconv_wide_weight_floats()
                                                     : hello::readconv_weight_floats(o.k, (o.flags & HELLO_FLAG_WINOGRAD) != 0, desc->window));
            b_end = (size_t)o.b_off;
            if (o.cout == 128) e->wide_trunk[o.seg == HELLO_SEG_READS1_TO_ALLELES ? 1 : 0] = true;
        }
        if (w_end > e->n_weight_floats || b_end > e->n_weight_floats) {
            delete e;
            return fail(HELLO_ERR_MODEL, "a weight block runs past the end of the blob");
        }
    }
    e->scratch.resize(desc->n_buffers);
    hipError_t err = hipMalloc((void**)&e->d_weights, nbytes);
    if (err == hipSuccess) err = hipMemcpy(e->d_weights, folded_weights, nbytes, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_staged, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_start);
    if (err == hipSuccess) err = hipEventCreate(&e->ev_stop);
    if (err != hipSuccess) {
        hello_engine_destroy(e);
        return fail(HELLO_ERR_HIP, "engine setup failed: %s", hipGetErrorString(err));
    }
    // lanes: which op waits for which, from the buffer ids (a laned program writes every scratch buffer from one op)
    const int n_ops = (int)e->ops.size();
    auto lane_of = [&](int i) { return (e->ops[i].flags & HELLO_FLAG_LANE_MASK) >> HELLO_FLAG_LANE_SHIFT; };
    for (int i = 0; i < n_ops; ++i) e->n_lanes = lane_of(i) + 1 > e->n_lanes ? lane_of(i) + 1 : e->n_lanes;
    if (e->n_lanes > 1) {
        std::vector<int> writer(desc->n_buffers, -1);
        e->op_waits.assign(n_ops, {});
        e->op_done.assign(n_ops, nullptr);
        e->lane_tail.assign(e->n_lanes, -1);
        std::vector<char> read_elsewhere(n_ops, 0);
        for (int i = 0; i < n_ops; ++i) {
            const hello_op& o = e->ops[i];
            const bool front = o.kind == HELLO_OP_XATTN_FRONT;          // writes dst AND res; every other op reads res
            const int reads[3] = {o.src0, o.src1, front ? HELLO_BUF_NONE : o.res};
            for (int b : reads) {
                if (b < HELLO_BUF_FIRST_SCRATCH) continue;
                const int w = writer[b];
                if (w < 0) {
                    hello_engine_destroy(e);
                    return fail(HELLO_ERR_MODEL, "laned program: op %d reads buffer %d before any op wrote it", i, b);
                }
                if (lane_of(w) != lane_of(i)) {
                    e->op_waits[i].push_back(w);
                    read_elsewhere[w] = 1;
                }
            }
            const int writes[2] = {o.kind == HELLO_OP_HEAD ? HELLO_BUF_NONE : o.dst, front ? o.res : HELLO_BUF_NONE};
            for (int b : writes) {
                if (b < HELLO_BUF_FIRST_SCRATCH) continue;
                if (writer[b] >= 0) {
                    hello_engine_destroy(e);
                    return fail(HELLO_ERR_MODEL, "laned program: ops %d and %d both write buffer %d (a program with lanes may not reuse buffers)", writer[b], i, b);
                }
                writer[b] = i;
            }
            e->lane_tail[lane_of(i)] = i;
        }
        e->lane_streams.assign(e->n_lanes, nullptr);
        for (int l = 1; l < e->n_lanes && err == hipSuccess; ++l) err = hipStreamCreateWithFlags(&e->lane_streams[l], hipStreamNonBlocking);
        for (int i = 0; i < n_ops && err == hipSuccess; ++i)
            if (read_elsewhere[i] || (lane_of(i) > 0 && e->lane_tail[lane_of(i)] == i)) err = hipEventCreateWithFlags(&e->op_done[i], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_lanes_go, hipEventDisableTiming);
        if (err != hipSuccess) {
            hello_engine_destroy(e);
            return fail(HELLO_ERR_HIP, "lane setup failed: %s", hipGetErrorString(err));
        }
    }
    *out = e;
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_create");
}

void hello_engine_destroy(hello_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (auto& b : e->scratch) b.release();
    e->d_csr.release();
    e->h_csr.release();
    e->d_logits.release();
    e->d_meta.release();
    e->d_post.release();
    e->d_out_small.release();
    e->h_io.release();
    e->d_rcl0.release();
    e->d_rcl1.release();
    e->d_partial[0].release();
    e->d_partial[1].release();
    for (hipStream_t st : e->lane_streams)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : e->op_done)
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_lanes_go) (void)hipEventDestroy(e->ev_lanes_go);
    e->d_feat_in.release();
    e->d_feat_out.release();
    e->d_debug.release();
    e->d_stamps.release();
    if (e->d_weights) (void)hipFree(e->d_weights);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    if (e->ev_staged) (void)hipEventDestroy(e->ev_staged);
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    if (e->ev_stop) (void)hipEventDestroy(e->ev_stop);
    for (auto& ring : e->prof_events)
        for (auto ev : ring)
            if (ev) (void)hipEventDestroy(ev);
    delete e;
}

void* hello_engine_stream(hello_engine* e) { return e ? (void*)e->own_stream : nullptr; }

int hello_engine_synchronize(hello_engine* e) {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->last_stream ? e->last_stream : e->own_stream));
    return HELLO_OK;
}

int hello_engine_last_forward_ms(hello_engine* e, float* ms) {
    if (!e || !ms) return fail(HELLO_ERR_ARG, "NULL argument");
    if (!e->timed) return fail(HELLO_ERR_ARG, "no forward has run yet");
    HIP_TRY(hipEventSynchronize(e->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, e->ev_start, e->ev_stop));
    return HELLO_OK;
}

int hello_engine_set_profiling(hello_engine* e, int max_forwards) {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (max_forwards < 0 || max_forwards > 4096) return fail(HELLO_ERR_ARG, "max_forwards out of range");
    HIP_TRY(hipSetDevice(e->device));
    e->profiling = max_forwards > 0;
    e->prof_count = 0;
    if ((int)e->prof_events.size() < max_forwards) e->prof_events.resize(max_forwards);
    // events are created when a forward first records them (a filtered recording needs two per matching op)
    for (int f = 0; f < max_forwards; ++f) e->prof_events[f].resize(e->ops.size() + 1, nullptr);
    return HELLO_OK;
}

int hello_engine_set_profiling_filter(hello_engine* e, int32_t op_kind) {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (op_kind < 0 || op_kind > HELLO_OP_XATTN_FRONT) return fail(HELLO_ERR_ARG, "unknown op kind %d", op_kind);
    e->prof_filter = op_kind;
    e->prof_count = 0;               // recordings made under another filter do not mix
    return HELLO_OK;
}

int hello_engine_op_times_ms(hello_engine* e, float* ms_sum, int32_t capacity, int32_t* n_ops,
                             int32_t* n_forwards) try {
    if (!e || !ms_sum || !n_ops || !n_forwards) return fail(HELLO_ERR_ARG, "NULL argument");
    if (e->prof_count == 0) return fail(HELLO_ERR_ARG, "no forward was recorded with profiling enabled");
    const int n = (int)e->ops.size() < capacity ? (int)e->ops.size() : capacity;
    for (int i = 0; i < n; ++i) ms_sum[i] = 0.f;
    for (int f = 0; f < e->prof_count; ++f) {
        auto& ring = e->prof_events[f];
        HIP_TRY(hipEventSynchronize(ring[e->ops.size()]));
        for (int i = 0; i < n; ++i) {
            if (e->prof_filter && e->ops[i].kind != e->prof_filter) continue;
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, ring[i], ring[i + 1]));
            ms_sum[i] += t;
        }
    }
    *n_ops = n;
    *n_forwards = e->prof_count;
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_op_times_ms");
}

int hello_engine_debug_capture(hello_engine* e, int32_t op_index) try {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (op_index < -1 || op_index >= (int)e->ops.size()) return fail(HELLO_ERR_ARG, "op index %d out of range", op_index);
    if (op_index >= 0 && e->ops[op_index].kind == HELLO_OP_HEAD)
        return fail(HELLO_ERR_ARG, "op %d is a HEAD: its output is the logits / meta array itself", op_index);
    e->debug_op = op_index;
    e->debug_floats = 0;
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_debug_capture");
}

int hello_engine_debug_stamps(hello_engine* e, int32_t mode) try {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (mode < 0 || mode > 3 || mode == 2) return fail(HELLO_ERR_ARG, "mode %d: 0 = off, 1 = record, 3 = record with one workgroup per CU", mode);
    e->stamp_mode = mode;
    e->stamp_layout[0] = 0;
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_debug_stamps");
}

int hello_engine_debug_read_stamps(hello_engine* e, uint64_t* out, int64_t capacity, int64_t* n_words, int32_t* layout) try {
    if (!e || !n_words || !layout) return fail(HELLO_ERR_ARG, "NULL argument");
    const int32_t* L = e->stamp_layout;
    const int64_t n = (int64_t)L[0] * L[1] * L[2] * L[3];
    *n_words = n;
    for (int i = 0; i < 5; ++i) layout[i] = L[i];
    if (!out) return HELLO_OK;
    if (n == 0) return fail(HELLO_ERR_ARG, "no stamped forward has run since hello_engine_debug_stamps");
    if (capacity < n) return fail(HELLO_ERR_ARG, "capacity %lld < %lld words", (long long)capacity, (long long)n);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->last_stream ? e->last_stream : e->own_stream));
    HIP_TRY(hipMemcpy(out, e->d_stamps.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_debug_read_stamps");
}

int hello_engine_debug_read(hello_engine* e, float* out, int64_t capacity, int64_t* n_floats) try {
    if (!e || !n_floats) return fail(HELLO_ERR_ARG, "NULL argument");
    *n_floats = (int64_t)e->debug_floats;
    if (!out) return HELLO_OK;
    if (e->debug_floats == 0) return fail(HELLO_ERR_ARG, "no forward has run since hello_engine_debug_capture");
    if (capacity < (int64_t)e->debug_floats) return fail(HELLO_ERR_ARG, "capacity %lld < %zu floats", (long long)capacity, e->debug_floats);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->last_stream ? e->last_stream : e->own_stream));
    HIP_TRY(hipMemcpy(out, e->d_debug.p, e->debug_floats * sizeof(float), hipMemcpyDeviceToHost));
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_debug_read");
}

// Build every per-batch index array on the host (pinned), ship them in one copy.
static int stage_batch_indices(hello_engine* e, const int32_t* rpa0, const int32_t* rpa1,
                               const int32_t* aps, int32_t S, int32_t A, int64_t R0, int64_t R1,
                               bool two_tech, hipStream_t stream) {
    // reads one workgroup of the fused read convolver walks (per technology: it depends on the batch size)
    const int win = hello::readconv_supports_window(e->desc.window) ? e->desc.window : 150;
    const int G = hello::readconv_reads_per_group(win);
    if ((e->wide_trunk[0] || e->wide_trunk[1]) && hello::readconv_wide_reads_per_group() != G)
        return fail(HELLO_ERR_ARG, "internal: the wide trunk's group size differs from the read convolver's");
    const hello::ReadConvPlan plan0 = e->wide_trunk[0] ? hello::readconv_wide_plan(R0) : hello::readconv_plan(R0, win);
    const hello::ReadConvPlan plan1 = !two_tech ? hello::ReadConvPlan{1, 0, 0}
                                      : e->wide_trunk[1] ? hello::readconv_wide_plan(R1) : hello::readconv_plan(R1, win);
    const int64_t n_groups0 = plan0.bulk_wgs + plan0.rest_wgs, n_groups1 = plan1.bulk_wgs + plan1.rest_wgs;   // workgroups
    e->plan0 = plan0;
    e->plan1 = plan1;
    size_t bytes = 0;
    auto add = [&](size_t count, size_t elem) { bytes += (count * elem + 15) & ~size_t(15); };
    add(A + 1, 4); add(A + 1, 4); add(S + 1, 4); add(A, 4);          // roff0 roff1 aoff site_of_allele
    add(R0, 4); add(R1, 4);                                          // allele_of_read0/1
    add(n_groups0 + 1, 4); add(n_groups1 + 1, 4);                    // group_slot0/1
    add(A + 1, 4); add(A + 1, 4);                                    // slot_off0/1
    add(S + 1, 8);                                                   // pair_off
    if (e->staged_pending) {
        HIP_TRY(hipEventSynchronize(e->ev_staged));   // previous batch's copy has left the pinned block
        e->staged_pending = false;
    }
    if (e->h_csr.ensure(bytes)) return fail(HELLO_ERR_HIP, "pinned allocation of %zu bytes failed", bytes);
    if (bytes > e->d_csr.cap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (e->d_csr.ensure(bytes)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", bytes);
    }
    char* hc = (char*)e->h_csr.p;
    char* hbase = hc;
    int32_t* h_roff0 = carve<int32_t>(hc, A + 1);
    int32_t* h_roff1 = carve<int32_t>(hc, A + 1);
    int32_t* h_aoff = carve<int32_t>(hc, S + 1);
    int32_t* h_soa = carve<int32_t>(hc, A);
    int32_t* h_aor0 = carve<int32_t>(hc, R0);
    int32_t* h_aor1 = carve<int32_t>(hc, R1);
    int32_t* h_gs0 = carve<int32_t>(hc, n_groups0 + 1);
    int32_t* h_gs1 = carve<int32_t>(hc, n_groups1 + 1);
    int32_t* h_so0 = carve<int32_t>(hc, A + 1);
    int32_t* h_so1 = carve<int32_t>(hc, A + 1);
    int64_t* h_poff = carve<int64_t>(hc, S + 1);

    auto build_reads = [&](const int32_t* rpa, int64_t R, int32_t* roff, int32_t* aor, int32_t* gslot,
                           int32_t* soff, const hello::ReadConvPlan& plan, const char* which) -> int {
        int64_t acc = 0;
        for (int32_t a = 0; a < A; ++a) {
            if (rpa[a] <= 0)
                return fail(HELLO_ERR_SHAPE, "%s[%d] = %d: every allele needs >= 1 read (dummy zero read)",
                            which, a, rpa[a]);
            roff[a] = (int32_t)acc;
            for (int32_t r = 0; r < rpa[a]; ++r) {
                if (acc + r < R) aor[acc + r] = a;
            }
            acc += rpa[a];
            if (acc > R) return fail(HELLO_ERR_SHAPE, "sum(%s) exceeds n_reads = %lld", which, (long long)R);
        }
        roff[A] = (int32_t)acc;
        if (acc != R)
            return fail(HELLO_ERR_SHAPE, "sum(%s) = %lld != n_reads = %lld", which, (long long)acc, (long long)R);
        // partial-sum slots of the fused read convolver: one slot per (workgroup, allele) incidence, numbered in
        // (workgroup, allele) order == (allele, workgroup) order because both are monotone in the read index.
        // Workgroup w of the plan: the bulk workgroups hold groups_per_wg groups of G reads each, the rest (second
        // launch) one group each.  gslot[w] = first slot of workgroup w; soff[a] = first slot of allele a.
        const int64_t n_wgs = plan.bulk_wgs + plan.rest_wgs;
        const int64_t bulk_reads = plan.rest_wgs ? plan.bulk_wgs * plan.groups_per_wg * G : R;
        for (int32_t a = 0; a <= A; ++a) soff[a] = 0;
        int64_t slot = 0;
        for (int64_t w = 0; w < n_wgs; ++w) {
            int64_t r_lo, r_hi;
            if (w < plan.bulk_wgs) {
                r_lo = w * plan.groups_per_wg * G;
                r_hi = r_lo + (int64_t)plan.groups_per_wg * G;
                if (r_hi > bulk_reads) r_hi = bulk_reads;
            } else {
                r_lo = bulk_reads + (w - plan.bulk_wgs) * G;
                r_hi = r_lo + G < R ? r_lo + G : R;
            }
            if (r_lo >= r_hi) return fail(HELLO_ERR_ARG, "internal: empty workgroup in the read-convolver plan");
            gslot[w] = (int32_t)slot;
            const int32_t first = aor[r_lo], last = aor[r_hi - 1];
            for (int32_t a = first; a <= last; ++a) soff[a + 1] += 1;      // counts first, prefix sum below
            slot += (last - first + 1);
        }
        gslot[n_wgs] = (int32_t)slot;
        int64_t s = 0;
        for (int32_t a = 0; a < A; ++a) {
            const int32_t count = soff[a + 1];
            soff[a] = (int32_t)s;
            s += count;
        }
        soff[A] = (int32_t)s;
        if (s != slot) return fail(HELLO_ERR_ARG, "internal: slot accounting mismatch");
        return 0;
    };
    if (int rc = build_reads(rpa0, R0, h_roff0, h_aor0, h_gs0, h_so0, plan0, "reads_per_allele0")) return rc;
    if (two_tech) {
        if (int rc = build_reads(rpa1, R1, h_roff1, h_aor1, h_gs1, h_so1, plan1, "reads_per_allele1")) return rc;
    } else {
        h_gs1[0] = 0;
    }
    int64_t acc = 0, pacc = 0;
    for (int32_t s = 0; s < S; ++s) {
        if (aps[s] <= 0) return fail(HELLO_ERR_SHAPE, "alleles_per_site[%d] = %d", s, aps[s]);
        h_aoff[s] = (int32_t)acc;
        h_poff[s] = pacc;
        for (int32_t k = 0; k < aps[s] && acc + k < A; ++k) h_soa[acc + k] = s;
        acc += aps[s];
        pacc += (int64_t)aps[s] * (aps[s] + 1) / 2;
        if (acc > A) return fail(HELLO_ERR_SHAPE, "sum(alleles_per_site) exceeds n_alleles = %d", A);
    }
    h_aoff[S] = (int32_t)acc;
    h_poff[S] = pacc;
    if (acc != A) return fail(HELLO_ERR_SHAPE, "sum(alleles_per_site) = %lld != n_alleles = %d", (long long)acc, A);

    HIP_TRY(hipMemcpyAsync(e->d_csr.p, hbase, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(e->ev_staged, stream));
    e->staged_pending = true;
    char* dbase = (char*)e->d_csr.p;
    auto dev = [&](void* hp) { return dbase + ((char*)hp - hbase); };
    e->roff0 = (int32_t*)dev(h_roff0);
    e->roff1 = (int32_t*)dev(h_roff1);
    e->aoff = (int32_t*)dev(h_aoff);
    e->site_of_allele = (int32_t*)dev(h_soa);
    e->allele_of_read0 = (int32_t*)dev(h_aor0);
    e->allele_of_read1 = (int32_t*)dev(h_aor1);
    e->group_slot0 = (int32_t*)dev(h_gs0);
    e->group_slot1 = (int32_t*)dev(h_gs1);
    e->slot_off0 = (int32_t*)dev(h_so0);
    e->slot_off1 = (int32_t*)dev(h_so1);
    e->pair_off = (int64_t*)dev(h_poff);
    return 0;
}

int hello_engine_forward(hello_engine* e, const uint8_t* reads0, const int32_t* rpa0,
                         const uint8_t* reads1, const int32_t* rpa1, const int32_t* aps,
                         const uint8_t* ref_onehot, int32_t S, int32_t A, int64_t R0, int64_t R1,
                         float* logits, float* meta, float* posteriors, int32_t flags, void* hip_stream) try {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    const hello_model_desc& d = e->desc;
    const bool two_tech = d.channels1 > 0;
    if (S <= 0 || A <= 0 || R0 <= 0) return fail(HELLO_ERR_ARG, "empty batch (S=%d A=%d R0=%lld)", S, A, (long long)R0);
    if (A < S) return fail(HELLO_ERR_SHAPE, "n_alleles %d < n_sites %d", A, S);
    if (!reads0 || !rpa0 || !aps || !logits) return fail(HELLO_ERR_ARG, "NULL input/output pointer");
    if (two_tech && (!reads1 || !rpa1 || R1 <= 0)) return fail(HELLO_ERR_ARG, "model needs a second read set");
    if (!two_tech) R1 = 0;
    if (d.uses_ref && !ref_onehot) return fail(HELLO_ERR_ARG, "model needs ref_onehot");
    if (d.has_meta && !meta) return fail(HELLO_ERR_ARG, "model produces meta weights: meta output is NULL");
    if (R0 > 0x7fffffffLL || R1 > 0x7fffffffLL) return fail(HELLO_ERR_ARG, "more than 2^31 reads in one batch");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    e->last_stream = stream;
    const bool in_dev = flags & HELLO_IN_DEVICE, out_dev = flags & HELLO_OUT_DEVICE;

    if (int rc = stage_batch_indices(e, rpa0, rpa1, aps, S, A, R0, R1, two_tech, stream)) return rc;

    // ---- scratch sizing (grow-only; growth synchronises the stream first) ----------------------
    bool synced = false;
    auto ensure = [&](DevBuf& b, size_t bytes) -> int {
        if (bytes <= b.cap) return 0;
        if (!synced) {
            if (hipStreamSynchronize(stream) != hipSuccess) return fail(HELLO_ERR_HIP, "stream sync failed");
            synced = true;
        }
        if (b.ensure(bytes)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", bytes);
        return 0;
    };
    for (int i = HELLO_BUF_FIRST_SCRATCH; i < d.n_buffers; ++i) {
        const long long rows = rows_of(e->buffers[i].domain, S, A, R0, R1);
        if (int rc = ensure(e->scratch[i], (size_t)rows * e->buffers[i].floats_per_row * sizeof(float))) return rc;
    }
    const size_t in_bytes0 = (size_t)R0 * d.window * d.channels0;
    const size_t in_bytes1 = (size_t)R1 * d.window * d.channels1;
    const size_t ref_bytes = (size_t)S * d.window * 5;
    const void* in_ptr[3] = {reads0, reads1, ref_onehot};
    const size_t in_bytes[3] = {in_bytes0, two_tech ? in_bytes1 : 0, (d.uses_ref ? ref_bytes : 0)};
    const void* buf_ptr[3] = {nullptr, nullptr, nullptr};
    // Small host calls (one site per call is the reference's deployment form): below SMALL_IO bytes the inputs are copied (memcpy)
    // into the engine's pinned, GPU-mapped block and the kernels read them THERE, across PCIe, and write the outputs into the block's
    // other half the same way -- no copy-engine hop in either direction (a kernel-trace timeline of one-site calls showed ~30 us of a
    // 258 us call in the two copies and the gaps around them; larger calls keep the copies: pageable memory cannot be mapped).
    constexpr size_t SMALL_IO = 256 * 1024;
    auto align16 = [](size_t b) { return (b + 15) & ~size_t(15); };
    const size_t in_total = align16(in_bytes[0]) + align16(in_bytes[1]) + align16(in_bytes[2]);
    const bool stage_in = !in_dev && in_total <= SMALL_IO;
    size_t out_total_small = 0;      // set below when the outputs are staged too
    if (stage_in) {
        if (e->h_io.ensure(2 * SMALL_IO)) return fail(HELLO_ERR_HIP, "pinned allocation failed");
    }
    size_t in_cursor = 0;
    for (int i = 0; i < 3; ++i) {
        if (!in_bytes[i]) continue;
        if (in_dev) {
            buf_ptr[i] = in_ptr[i];
        } else if (stage_in) {
            memcpy((char*)e->h_io.p + in_cursor, in_ptr[i], in_bytes[i]);
            buf_ptr[i] = (char*)e->h_io.p + in_cursor;
            in_cursor += align16(in_bytes[i]);
        } else {
            if (int rc = ensure(e->scratch[i], in_bytes[i])) return rc;
            HIP_TRY(hipMemcpyAsync(e->scratch[i].p, in_ptr[i], in_bytes[i], hipMemcpyHostToDevice, stream));
            buf_ptr[i] = e->scratch[i].p;
        }
    }
    if (flags & HELLO_LAYOUT_RCL) {
        if (int rc = ensure(e->d_rcl0, in_bytes0)) return rc;
        HIP_TRY(hello::launch_rcl_to_rlc((const uint8_t*)buf_ptr[0], (uint8_t*)e->d_rcl0.p, R0, d.window, d.channels0, stream));
        buf_ptr[0] = e->d_rcl0.p;
        if (two_tech) {
            if (int rc = ensure(e->d_rcl1, in_bytes1)) return rc;
            HIP_TRY(hello::launch_rcl_to_rlc((const uint8_t*)buf_ptr[1], (uint8_t*)e->d_rcl1.p, R1, d.window, d.channels1, stream));
            buf_ptr[1] = e->d_rcl1.p;
        }
    }
    float* d_logits = logits;
    float* d_meta = meta;
    float* d_post = posteriors;
    int64_t n_pairs = 0;
    if (posteriors)
        for (int32_t s = 0; s < S; ++s) n_pairs += (int64_t)aps[s] * (aps[s] + 1) / 2;
    const size_t logit_bytes = (size_t)d.n_experts * A * sizeof(float), meta_bytes = (size_t)S * 3 * sizeof(float);
    const size_t post_bytes = (size_t)4 * n_pairs * sizeof(float);
    const size_t out_total = align16(logit_bytes) + (d.has_meta ? align16(meta_bytes) : 0) + (posteriors ? align16(post_bytes) : 0);
    const bool stage_out = !out_dev && out_total <= SMALL_IO;
    // ... written by the kernels IN PLACE only while they are a few KB (the posteriors kernel reads the logits back: thousands of
    // 4-byte reads across PCIe would cost more than the one copy they save); larger staged outputs keep a device block + ONE copy
    constexpr size_t IN_PLACE_OUT = 16 * 1024;
    const bool out_in_place = stage_out && out_total <= IN_PLACE_OUT;
    if (stage_out) {
        if (e->h_io.ensure(2 * SMALL_IO)) return fail(HELLO_ERR_HIP, "pinned allocation failed");
        if (!out_in_place)
            if (int rc = ensure(e->d_out_small, out_total)) return rc;
        char* base = out_in_place ? (char*)e->h_io.p + SMALL_IO : (char*)e->d_out_small.p;   // the output half of the pinned block | device
        d_logits = (float*)base;
        base += align16(logit_bytes);
        if (d.has_meta) {
            d_meta = (float*)base;
            base += align16(meta_bytes);
        }
        if (posteriors) d_post = (float*)base;
        out_total_small = out_total;
    } else if (!out_dev) {
        if (int rc = ensure(e->d_logits, logit_bytes)) return rc;
        d_logits = (float*)e->d_logits.p;
        if (d.has_meta) {
            if (int rc = ensure(e->d_meta, meta_bytes)) return rc;
            d_meta = (float*)e->d_meta.p;
        }
        if (posteriors) {
            if (int rc = ensure(e->d_post, post_bytes)) return rc;
            d_post = (float*)e->d_post.p;
        }
    }
    // fused read convolver partial slots (worst case: one per read + one per group)
    bool has_fused = false;
    for (const hello_op& o : e->ops) has_fused |= (o.kind == HELLO_OP_READCONV_FUSED);
    if (has_fused) {
        const int G = hello::readconv_reads_per_group(d.window);
        const int64_t Rmax = R0 > R1 ? R0 : R1;
        const size_t slots = (size_t)A + (size_t)((Rmax + G - 1) / G) + 1;
        const size_t frame_ch = (e->wide_trunk[0] || e->wide_trunk[1]) ? 128 : 64;
        for (const hello_op& o : e->ops)
            if (o.kind == HELLO_OP_READCONV_FUSED)
                if (int rc = ensure(e->d_partial[o.seg == HELLO_SEG_READS1_TO_ALLELES ? 1 : 0], slots * hello::readconv_frame_rows(d.window) * frame_ch * sizeof(float))) return rc;
    }
    // experts without a head (ensemble of two: third expert is all-zero logits, :244) stay zero
    if (d.n_experts == 3) HIP_TRY(hipMemsetAsync(d_logits, 0, logit_bytes, stream));

    auto ptr = [&](int id) -> void* {
        if (id == HELLO_BUF_NONE) return nullptr;
        if (id < HELLO_BUF_FIRST_SCRATCH) return const_cast<void*>(buf_ptr[id]);
        return e->scratch[id].p;
    };

    HIP_TRY(hipEventRecord(e->ev_start, stream));
    int op_index = 0;
    std::vector<hipEvent_t>* ring = nullptr;
    if (e->profiling && e->prof_count < (int)e->prof_events.size() &&
        e->prof_events[e->prof_count].size() == e->ops.size() + 1)
        ring = &e->prof_events[e->prof_count];
    // an event of the ring is created the first time a forward records it; with a filter only the two events
    // that bracket a matching op are recorded
    auto mark = [&](int slot) -> int {
        hipEvent_t& ev = (*ring)[slot];
        if (!ev) HIP_TRY(hipEventCreate(&ev));
        HIP_TRY(hipEventRecord(ev, stream));
        return 0;
    };
    const int n_ops_total = (int)e->ops.size();
    auto wanted = [&](int i) { return i >= 0 && i < n_ops_total && (!e->prof_filter || e->ops[i].kind == e->prof_filter); };
    // lanes: the independent chains of a multi-technology / multi-expert program run on their own streams (small launches: every
    // chain is a handful of workgroups).  Lane 0 is the call's stream; the others start behind what it has staged so far.
    const bool laned = e->n_lanes > 1;
    if (laned) {
        if (ring || e->debug_op >= 0 || (e->stamp_mode & 1))
            return fail(HELLO_ERR_ARG, "per-op profiling / debug capture / stamps need the sequential program (this engine runs lanes)");
        HIP_TRY(hipEventRecord(e->ev_lanes_go, stream));
        for (int l = 1; l < e->n_lanes; ++l) HIP_TRY(hipStreamWaitEvent(e->lane_streams[l], e->ev_lanes_go, 0));
    }
    hipStream_t const call_stream = stream;
    for (const hello_op& o : e->ops) {
        const int lane = laned ? (o.flags & HELLO_FLAG_LANE_MASK) >> HELLO_FLAG_LANE_SHIFT : 0;
        hipStream_t const stream = lane ? e->lane_streams[lane] : call_stream;      // (shadows the call's stream inside the loop)
        if (laned)
            for (int w : e->op_waits[op_index]) HIP_TRY(hipStreamWaitEvent(stream, e->op_done[w], 0));
        if (ring && (wanted(op_index) || wanted(op_index - 1)))
            if (int rc = mark(op_index)) return rc;
        const long long rows = rows_of(o.domain, S, A, R0, R1);
        switch (o.kind) {
            case HELLO_OP_CONV1D: {
                hello::ConvArgs a{};
                a.src = ptr(o.src0);
                a.dst = (float*)ptr(o.dst);
                a.res = (const float*)ptr(o.res);
                a.w = e->d_weights + o.w_off;
                a.bias = e->d_weights + o.b_off;
                a.m_total = rows * o.lout;
                a.groups = o.c1 > 1 ? o.c1 : 1;
                a.src2 = o.src1 != HELLO_BUF_NONE ? (const float*)ptr(o.src1) : nullptr;
                a.split = o.src1 != HELLO_BUF_NONE ? o.seg : 0;
                a.lin = o.lin; a.lout = o.lout; a.cin = o.cin / a.groups; a.cin_stride = o.cin; a.cout = o.cout;
                a.k = o.k; a.stride = o.stride; a.pad = o.pad;
                a.kpad = ((o.k * a.cin + 31) / 32) * 32;
                a.cout_pad = ((o.cout + 31) / 32) * 32;
                a.relu = (o.flags & HELLO_FLAG_RELU) ? 1 : ((o.flags & HELLO_FLAG_SOFTPLUS) ? 2 : 0);
                a.src_u8 = (o.flags & HELLO_FLAG_SRC_U8) ? 1 : 0;
                a.wino = (o.flags & HELLO_FLAG_WINOGRAD) ? 1 : 0;
                if (!a.src) return fail(HELLO_ERR_ARG, "op %d reads an input the caller did not supply", op_index);
                if (a.wino) {
                    a.kpad = (hello::conv1d_wino_outputs_per_tile(o.lin) + 2) * a.cin;
                    a.cout_pad = o.cout;
                    if (!hello::conv1d_wino_supported(a))
                        return fail(HELLO_ERR_MODEL, "op %d: this convolution has no Winograd form", op_index);
                    HIP_TRY(hello::launch_conv1d_wino(a, stream));
                    break;
                }
                HIP_TRY(hello::launch_conv1d(a, stream));
                break;
            }
            case HELLO_OP_MAXPOOL:
                HIP_TRY(hello::launch_maxpool((const float*)ptr(o.src0), (float*)ptr(o.dst), rows, o.lin, o.lout,
                                              o.cin, o.k, o.stride, o.pad, stream));
                break;
            case HELLO_OP_SEGSUM: {
                const int32_t* off = o.seg == HELLO_SEG_READS0_TO_ALLELES ? e->roff0
                                     : o.seg == HELLO_SEG_READS1_TO_ALLELES ? e->roff1 : e->aoff;
                HIP_TRY(hello::launch_segsum((const float*)ptr(o.src0), (float*)ptr(o.dst), off, (int)rows,
                                             o.lin * o.cin, stream));
                break;
            }
            case HELLO_OP_MIX:
                HIP_TRY(hello::launch_mix((const float*)ptr(o.src0), (const float*)ptr(o.src1), (float*)ptr(o.dst),
                                          e->site_of_allele, rows, o.lin * o.cin, o.a0, o.a1,
                                          (o.flags & HELLO_FLAG_MIX_REST) != 0, stream));
                break;
            case HELLO_OP_HEAD: {
                float* outp;
                long long so, sr;
                if (o.dst == 3) { outp = d_meta; so = 1; sr = 3; }
                else { outp = d_logits + (long long)o.dst * A; so = 0; sr = 1; }
                HIP_TRY(hello::launch_head((const float*)ptr(o.src0), e->d_weights + o.w_off, e->d_weights + o.b_off,
                                           outp, rows, o.lin, o.cin, o.cout, so, sr,
                                           (o.flags & HELLO_FLAG_SOFTMAX) ? 1 : 0, stream));
                break;
            }
            case HELLO_OP_CONCAT:
                HIP_TRY(hello::launch_concat((const float*)ptr(o.src0), (const float*)ptr(o.src1), (float*)ptr(o.dst),
                                             rows * o.lin, o.cin, o.c1, stream));
                break;
            case HELLO_OP_ADD:
                HIP_TRY(hello::launch_add((const float*)ptr(o.src0), (const float*)ptr(o.src1), (float*)ptr(o.dst),
                                          rows * o.lin * o.cin, stream));
                break;
            case HELLO_OP_COMPRESSOR_FUSED: {
                hello::CompressorArgs a{};
                a.frames = (const float*)ptr(o.src0);
                a.dst = (float*)ptr(o.dst);
                a.w = e->d_weights + o.w_off;
                a.n_items = rows;
                a.blocks = o.k;
                HIP_TRY(hello::launch_compressor_fused(a, stream));
                break;
            }
            case HELLO_OP_XATTN_FRONT: {
                hello::XattnFrontArgs a{};
                a.alleles = (const float*)ptr(o.src0);
                a.sites = o.src1 == HELLO_BUF_NONE ? nullptr : (const float*)ptr(o.src1);
                a.owner = e->site_of_allele;
                a.site_off = e->aoff;
                a.y2 = (float*)ptr(o.dst);
                a.sc = (float*)ptr(o.res);
                a.w = e->d_weights + o.w_off;
                a.n_items = rows;
                a.a0 = o.a0;
                a.a1 = o.a1;
                a.rest = (o.flags & HELLO_FLAG_MIX_REST) ? 1 : 0;
                HIP_TRY(hello::launch_xattn_front(a, stream));
                break;
            }
            case HELLO_OP_LAYERNORM:
                HIP_TRY(hello::launch_layernorm((const float*)ptr(o.src0), (const float*)ptr(o.res), (float*)ptr(o.dst),
                                                e->d_weights + o.w_off, e->d_weights + o.b_off, rows * o.lin, o.cin, o.a0,
                                                (o.flags & HELLO_FLAG_RELU) ? 1 : ((o.flags & HELLO_FLAG_SOFTPLUS) ? 2 : 0), stream));
                break;
            case HELLO_OP_READCONV_FUSED: {
                const bool t1 = o.seg == HELLO_SEG_READS1_TO_ALLELES;
                hello::ReadConvArgs a{};
                if (o.flags & HELLO_FLAG_SRC_U8) {
                    a.reads = (const uint8_t*)ptr(o.src0);
                    a.channels = o.cin;
                } else {
                    a.pooled = (const float*)ptr(o.src0);
                }
                if (!ptr(o.src0)) return fail(HELLO_ERR_ARG, "op %d reads an input the caller did not supply", op_index);
                a.w = e->d_weights + o.w_off;
                a.partial = (float*)e->d_partial[t1 ? 1 : 0].p;
                a.allele_of_read = t1 ? e->allele_of_read1 : e->allele_of_read0;
                a.slot_of_group = t1 ? e->group_slot1 : e->group_slot0;
                a.n_reads = t1 ? R1 : R0;
                a.window = d.window;
                a.softplus = (o.flags & HELLO_FLAG_SOFTPLUS) ? 1 : 0;
                a.extra_blocks = o.k;
                a.winograd = (o.flags & HELLO_FLAG_WINOGRAD) ? 1 : 0;
                a.bf16x3 = (o.flags & HELLO_FLAG_BF16X3) ? ((o.flags & HELLO_FLAG_BF16X3_32) ? 2 : 1) : 0;
                const bool wide = o.cout == 128;
                if ((size_t)o.w_off + (wide ? hello::readconv_wide_weight_floats() : hello::readconv_weight_floats(o.k, a.winograd, d.window)) +
                        (a.bf16x3 ? hello::readconv_bf16x3_extra_floats(o.k) : 0) > e->n_weight_floats)
                    return fail(HELLO_ERR_MODEL, "op %d: fused read-convolver weight block truncated", op_index);
                {
                    // the plan of stage_batch_indices: the bulk in whole rounds of n-group workgroups, then the rest as
                    // one-group workgroups in a second launch of the same kernel over the remaining reads
                    const hello::ReadConvPlan& plan = t1 ? e->plan1 : e->plan0;
                    const int G = hello::readconv_reads_per_group(d.window);
                    const long long total = a.n_reads;
                    const long long bulk_reads = plan.rest_wgs ? plan.bulk_wgs * plan.groups_per_wg * G : total;
                    a.groups_per_wg = plan.groups_per_wg;
                    a.n_reads = bulk_reads;
                    int32_t stamp_lay[5] = {0, 0, 0, 0, 0};
                    const int stamp_waves = hello::readconv_stamp_waves();
                    if ((e->stamp_mode & 1) && !t1 && total > 0) {
                        // diagnostic forward: the stamped instantiation of the kernel (refused by the launch for any schedule but
                        // the default fp32 Winograd one from the bytes).  A forward without reads stamps nothing.
                        if (wide || !a.reads) return fail(HELLO_ERR_ARG, "stamps: the canonical fused read convolver from the bytes only");
                        const long long wgs = plan.rest_wgs ? plan.bulk_wgs + plan.rest_wgs : (long long)((total + (long long)G * plan.groups_per_wg - 1) / ((long long)G * plan.groups_per_wg));
                        const int slots = hello::readconv_stamp_slots();
                        const size_t bytes = (size_t)wgs * stamp_waves * plan.groups_per_wg * slots * sizeof(uint64_t);
                        if (bytes > e->d_stamps.cap) {
                            HIP_TRY(hipStreamSynchronize(stream));
                            if (e->d_stamps.ensure(bytes)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", bytes);
                        }
                        HIP_TRY(hipMemsetAsync(e->d_stamps.p, 0, bytes, stream));
                        a.stamps = (unsigned long long*)e->d_stamps.p;
                        a.stamp_groups = plan.groups_per_wg;
                        a.stamp_mode = e->stamp_mode;
                        const int32_t lay[5] = {(int32_t)wgs, stamp_waves, plan.groups_per_wg, slots, (int32_t)(plan.rest_wgs ? plan.bulk_wgs : wgs)};
                        for (int i = 0; i < 5; ++i) stamp_lay[i] = lay[i];
                        e->stamp_layout[0] = 0;               // nothing to read until both launches below have been accepted
                    }
                    HIP_TRY(wide ? hello::launch_readconv_wide(a, stream) : hello::launch_readconv_fused(a, stream));
                    if (plan.rest_wgs) {
                        hello::ReadConvArgs b = a;
                        if (b.stamps) b.stamps += (size_t)plan.bulk_wgs * stamp_waves * plan.groups_per_wg * hello::readconv_stamp_slots();
                        if (b.reads) b.reads += bulk_reads * d.window * o.cin;
                        else b.pooled += bulk_reads * (long long)o.lin * o.cin;
                        b.allele_of_read += bulk_reads;
                        b.slot_of_group += plan.bulk_wgs;
                        b.groups_per_wg = 1;
                        b.n_reads = total - bulk_reads;
                        HIP_TRY(wide ? hello::launch_readconv_wide(b, stream) : hello::launch_readconv_fused(b, stream));
                    }
                    if (a.stamps)                             // both launches went out: the layout debug_read_stamps reports is theirs
                        for (int i = 0; i < 5; ++i) e->stamp_layout[i] = stamp_lay[i];
                }
                HIP_TRY(hello::launch_readconv_finalize((const float*)e->d_partial[t1 ? 1 : 0].p, t1 ? e->slot_off1 : e->slot_off0,
                                                        (float*)ptr(o.dst), A, o.lout, o.cout, stream));
                break;
            }
        }
        if (op_index == e->debug_op) {
            // snapshot of this op's output before a later op reuses the buffer
            const size_t per_row = o.kind == HELLO_OP_CONCAT ? (size_t)o.lout * (o.cin + o.c1)
                                   : (o.kind == HELLO_OP_CONV1D || o.kind == HELLO_OP_READCONV_FUSED || o.kind == HELLO_OP_COMPRESSOR_FUSED || o.kind == HELLO_OP_XATTN_FRONT) ? (size_t)o.lout * o.cout
                                   : (o.kind == HELLO_OP_MAXPOOL ? (size_t)o.lout * o.cin : (size_t)o.lin * o.cin);
            const size_t n = (size_t)rows * per_row;
            if (n * sizeof(float) > e->d_debug.cap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (e->d_debug.ensure(n * sizeof(float))) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", n * sizeof(float));
            }
            HIP_TRY(hipMemcpyAsync(e->d_debug.p, ptr(o.dst), n * sizeof(float), hipMemcpyDeviceToDevice, stream));
            e->debug_floats = n;
        }
        if (laned && e->op_done[op_index]) HIP_TRY(hipEventRecord(e->op_done[op_index], stream));
        ++op_index;
    }
    if (laned)                                       // join: the call's stream goes on (posteriors, outputs) behind every lane's last op
        for (int l = 1; l < e->n_lanes; ++l)
            if (e->lane_tail[l] >= 0) HIP_TRY(hipStreamWaitEvent(stream, e->op_done[e->lane_tail[l]], 0));
    if (ring) {
        if (int rc = mark(op_index)) return rc;     // the end marker (op_times_ms waits for it)
        e->prof_count++;
    }
    if (posteriors)
        HIP_TRY(hello::launch_posteriors(d_logits, d.n_experts == 3 ? d_meta : nullptr, e->aoff, e->pair_off, S, A,
                                         d.n_experts, n_pairs, d_post, stream));
    HIP_TRY(hipEventRecord(e->ev_stop, stream));
    e->timed = true;

    char* const h_out = out_total_small ? (char*)e->h_io.p + SMALL_IO : nullptr;     // the output half of the pinned block
    if (out_total_small) {
        if (!out_in_place) HIP_TRY(hipMemcpyAsync(h_out, e->d_out_small.p, out_total_small, hipMemcpyDeviceToHost, stream));
    } else if (!out_dev) {
        HIP_TRY(hipMemcpyAsync(logits, d_logits, logit_bytes, hipMemcpyDeviceToHost, stream));
        if (d.has_meta) HIP_TRY(hipMemcpyAsync(meta, d_meta, meta_bytes, hipMemcpyDeviceToHost, stream));
        if (posteriors) HIP_TRY(hipMemcpyAsync(posteriors, d_post, post_bytes, hipMemcpyDeviceToHost, stream));
    }
    // host pointers may be pageable and die with the caller's frame: the copies from / to them are complete
    // only after this (device-in / device-out calls stay asynchronous on the stream)
    if (!in_dev || !out_dev) HIP_TRY(hipStreamSynchronize(stream));
    if (out_total_small) {
        const char* src = h_out;
        memcpy(logits, src, logit_bytes);
        src += align16(logit_bytes);
        if (d.has_meta) {
            memcpy(meta, src, meta_bytes);
            src += align16(meta_bytes);
        }
        if (posteriors) memcpy(posteriors, src, post_bytes);
    }
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_forward");
}

// Host memory the GPU reads and writes in place (hipHostMalloc: pinned, mapped, coherent): a caller that builds small batches in such
// a block passes them to hello_engine_forward as DEVICE pointers -- no staging copy, no copy-engine hop either way (the shared scoring
// server does, csrc/site_server.hip).  NULL when the allocation fails (e.g. no GPU).
void* hello_pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void hello_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int hello_engine_posteriors(hello_engine* e, const float* logits, const float* meta, const int32_t* aps,
                            int32_t S, int32_t A, int64_t n_pairs_total, float* out, int32_t flags,
                            void* hip_stream) try {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (!logits || !aps || !out || S <= 0 || A <= 0) return fail(HELLO_ERR_ARG, "bad argument");
    const hello_model_desc& d = e->desc;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    e->last_stream = stream;
    const bool in_dev = flags & HELLO_IN_DEVICE, out_dev = flags & HELLO_OUT_DEVICE;
    // own small CSR block (allele offsets + pair offsets)
    std::vector<int32_t> aoff(S + 1);
    std::vector<int64_t> poff(S + 1);
    int64_t acc = 0, pacc = 0;
    for (int32_t s = 0; s < S; ++s) {
        if (aps[s] <= 0) return fail(HELLO_ERR_SHAPE, "alleles_per_site[%d] = %d", s, aps[s]);
        aoff[s] = (int32_t)acc;
        poff[s] = pacc;
        acc += aps[s];
        pacc += (int64_t)aps[s] * (aps[s] + 1) / 2;
    }
    aoff[S] = (int32_t)acc;
    poff[S] = pacc;
    if (acc != A) return fail(HELLO_ERR_SHAPE, "sum(alleles_per_site) = %lld != n_alleles = %d", (long long)acc, A);
    if (pacc != n_pairs_total)
        return fail(HELLO_ERR_SHAPE, "n_pairs_total = %lld, expected %lld", (long long)n_pairs_total, (long long)pacc);
    const size_t a_bytes = (S + 1) * sizeof(int32_t), p_bytes = (S + 1) * sizeof(int64_t);
    const size_t a_pad = (a_bytes + 15) & ~size_t(15);
    const size_t l_bytes = (size_t)d.n_experts * A * sizeof(float), m_bytes = (size_t)S * 3 * sizeof(float);
    const size_t o_bytes = (size_t)4 * n_pairs_total * sizeof(float);
    const size_t l_pad = (l_bytes + 15) & ~size_t(15), m_pad = (m_bytes + 15) & ~size_t(15);
    const size_t total = a_pad + ((p_bytes + 15) & ~size_t(15)) + l_pad + m_pad + o_bytes;
    HIP_TRY(hipStreamSynchronize(stream));
    if (e->d_post.ensure(total)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", total);
    char* base = (char*)e->d_post.p;
    int32_t* d_aoff = (int32_t*)base;
    int64_t* d_poff = (int64_t*)(base + a_pad);
    float* d_l = (float*)(base + a_pad + ((p_bytes + 15) & ~size_t(15)));
    float* d_m = (float*)((char*)d_l + l_pad);
    float* d_o = (float*)((char*)d_m + m_pad);
    HIP_TRY(hipMemcpyAsync(d_aoff, aoff.data(), a_bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_poff, poff.data(), p_bytes, hipMemcpyHostToDevice, stream));
    const float* lp = logits;
    const float* mp = meta;
    if (!in_dev) {
        HIP_TRY(hipMemcpyAsync(d_l, logits, l_bytes, hipMemcpyHostToDevice, stream));
        lp = d_l;
        if (meta) {
            HIP_TRY(hipMemcpyAsync(d_m, meta, m_bytes, hipMemcpyHostToDevice, stream));
            mp = d_m;
        }
    }
    float* op = out_dev ? out : d_o;
    HIP_TRY(hello::launch_posteriors(lp, (d.n_experts == 3) ? mp : nullptr, d_aoff, d_poff, S, A, d.n_experts,
                                     n_pairs_total, op, stream));
    if (!out_dev) HIP_TRY(hipMemcpyAsync(out, d_o, o_bytes, hipMemcpyDeviceToHost, stream));
    // the host vectors above die with this frame: the copies from them must have completed
    HIP_TRY(hipStreamSynchronize(stream));
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_posteriors");
}

int hello_engine_featurize(hello_engine* e, const uint8_t* bases, const uint8_t* quals, const int64_t* read_offsets,
                           const uint32_t* cigars, const int64_t* cigar_offsets, const int64_t* ref_starts,
                           const uint8_t* mapq, const int8_t* orientation, const uint8_t* hp,
                           const int32_t* site_of_read, const uint8_t* ref_windows,
                           const int64_t* ref_window_offsets, const int64_t* window_starts,
                           const int64_t* assembly_starts, const int64_t* assembly_stops, int64_t n_reads,
                           int32_t n_sites, int32_t feature_length, int32_t channels, uint8_t* out,
                           int32_t flags, void* hip_stream) try {
    if (!e) return fail(HELLO_ERR_ARG, "engine is NULL");
    if (!bases || !quals || !read_offsets || !cigars || !cigar_offsets || !ref_starts || !mapq || !orientation ||
        !hp || !site_of_read || !ref_windows || !ref_window_offsets || !window_starts || !assembly_starts ||
        !assembly_stops || !out)
        return fail(HELLO_ERR_ARG, "NULL pointer");
    if (n_reads <= 0 || n_sites <= 0 || feature_length <= 0) return fail(HELLO_ERR_ARG, "empty batch");
    if (channels != 6 && channels != 7) return fail(HELLO_ERR_ARG, "channels must be 6 or 7");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    e->last_stream = stream;
    const bool in_dev = flags & HELLO_IN_DEVICE, out_dev = flags & HELLO_OUT_DEVICE;
    hello::FeaturizeArgs a{};
    const size_t out_bytes = (size_t)n_reads * feature_length * channels;
    if (in_dev) {
        a.bases = bases; a.quals = quals; a.read_off = (const long long*)read_offsets;
        a.cigars = cigars; a.cigar_off = (const long long*)cigar_offsets;
        a.ref_start = (const long long*)ref_starts; a.mapq = mapq; a.orientation = orientation; a.hp = hp;
        a.site_of_read = site_of_read; a.ref = ref_windows; a.ref_off = (const long long*)ref_window_offsets;
        a.window_start = (const long long*)window_starts; a.asm_start = (const long long*)assembly_starts;
        a.asm_stop = (const long long*)assembly_stops;
    } else {
        // host arrays: validate the offsets, then stage everything in one device block
        if (read_offsets[0] != 0 || cigar_offsets[0] != 0 || ref_window_offsets[0] != 0)
            return fail(HELLO_ERR_SHAPE, "offset arrays must start at 0");
        for (int64_t r = 0; r < n_reads; ++r) {
            if (read_offsets[r + 1] < read_offsets[r] || cigar_offsets[r + 1] < cigar_offsets[r])
                return fail(HELLO_ERR_SHAPE, "offsets of read %lld decrease", (long long)r);
            if (site_of_read[r] < 0 || site_of_read[r] >= n_sites)
                return fail(HELLO_ERR_SHAPE, "site_of_read[%lld] = %d out of range", (long long)r, site_of_read[r]);
        }
        const size_t n_bases = (size_t)read_offsets[n_reads], n_cig = (size_t)cigar_offsets[n_reads];
        const size_t n_ref = (size_t)ref_window_offsets[n_sites];
        struct Part { const void* src; size_t bytes; size_t off; };
        Part parts[15] = {
            {bases, n_bases, 0}, {quals, n_bases, 0}, {read_offsets, (size_t)(n_reads + 1) * 8, 0},
            {cigars, n_cig * 4, 0}, {cigar_offsets, (size_t)(n_reads + 1) * 8, 0}, {ref_starts, (size_t)n_reads * 8, 0},
            {mapq, (size_t)n_reads, 0}, {orientation, (size_t)n_reads, 0}, {hp, (size_t)n_reads, 0},
            {site_of_read, (size_t)n_reads * 4, 0}, {ref_windows, n_ref, 0},
            {ref_window_offsets, (size_t)(n_sites + 1) * 8, 0}, {window_starts, (size_t)n_sites * 8, 0},
            {assembly_starts, (size_t)n_sites * 8, 0}, {assembly_stops, (size_t)n_sites * 8, 0}};
        size_t total = 0;
        for (auto& p : parts) { p.off = total; total += (p.bytes + 15) & ~size_t(15); }
        total += 16;
        HIP_TRY(hipStreamSynchronize(stream));
        if (e->d_feat_in.ensure(total)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", total);
        char* base = (char*)e->d_feat_in.p;
        for (auto& p : parts)
            if (p.bytes) HIP_TRY(hipMemcpyAsync(base + p.off, p.src, p.bytes, hipMemcpyHostToDevice, stream));
        a.bases = (const uint8_t*)(base + parts[0].off); a.quals = (const uint8_t*)(base + parts[1].off);
        a.read_off = (const long long*)(base + parts[2].off); a.cigars = (const uint32_t*)(base + parts[3].off);
        a.cigar_off = (const long long*)(base + parts[4].off); a.ref_start = (const long long*)(base + parts[5].off);
        a.mapq = (const uint8_t*)(base + parts[6].off); a.orientation = (const int8_t*)(base + parts[7].off);
        a.hp = (const uint8_t*)(base + parts[8].off); a.site_of_read = (const int32_t*)(base + parts[9].off);
        a.ref = (const uint8_t*)(base + parts[10].off); a.ref_off = (const long long*)(base + parts[11].off);
        a.window_start = (const long long*)(base + parts[12].off); a.asm_start = (const long long*)(base + parts[13].off);
        a.asm_stop = (const long long*)(base + parts[14].off);
    }
    uint8_t* d_out = out;
    if (!out_dev) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (e->d_feat_out.ensure(out_bytes)) return fail(HELLO_ERR_HIP, "device allocation of %zu bytes failed", out_bytes);
        d_out = (uint8_t*)e->d_feat_out.p;
    }
    a.n_reads = n_reads; a.length = feature_length; a.channels = channels; a.out = d_out;
    HIP_TRY(hello::launch_featurize(a, stream));
    if (!out_dev) HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
    // host inputs were pageable: the copies above are complete only after this
    if (!in_dev || !out_dev) HIP_TRY(hipStreamSynchronize(stream));
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_engine_featurize");
}

}  // extern "C"
