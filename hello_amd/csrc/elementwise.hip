// Bandwidth-side kernels of the scoring path: pooling, ragged segment sums, the 2a-s comparison,
// the expert head, channel concat/add, layout transposition and the genotype-pair posteriors.
// All are simple grid-stride, 16-byte-per-lane kernels; none is on the FLOP critical path.
#include "kernels.h"

namespace hello {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned grid_for(long long n, int block, long long cap = 1 << 20) {
    long long g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---- MaxPool1d (floor mode; padding positions never win) ------------------------------------
__global__ void maxpool_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n4,
                               int lin, int lout, int c4, int k, int stride, int pad) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4);
        const long long rp = i / c4;
        const int p = (int)(rp % lout);
        const long long row = rp / lout;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int tap = 0; tap < k; ++tap) {
            const int pos = p * stride - pad + tap;
            if (pos < 0 || pos >= lin) continue;
            const f32x4 v = *(const f32x4*)(src + ((row * lin + pos) * c4 + cc) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v[e]);
        }
        *(f32x4*)(dst + i * 4) = best;
    }
}

hipError_t launch_maxpool(const float* src, float* dst, long long rows, int lin, int lout, int c, int k,
                          int stride, int pad, hipStream_t stream) {
    const long long n4 = rows * lout * (c / 4);
    if (n4 <= 0) return hipSuccess;
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, stream, src, dst, n4, lin,
                       lout, c / 4, k, stride, pad);
    return hipGetLastError();
}

// ---- LayerNorm over channels (LayerNormModule, NNTools.py:802-828) -----------------------------------
// One wave per position: lane i holds channels i, i + 64, ... (c <= 512); mean and biased variance by two wave
// reductions (the two-pass form: variance of the centred values), then scale, shift, activation, residual.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ src, const float* __restrict__ res,
                                                        float* __restrict__ dst, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, long long positions, int c,
                                                        float eps, int act) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), stride = (long long)gridDim.x * 4;
    for (long long p = wave0; p < positions; p += stride) {
        const float* x = src + p * c;
        float v[8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = (lane + 64 * k < c) ? x[lane + 64 * k] : 0.f;
            sum += v[k];
        }
        const float mean = wave_sum(sum) / (float)c;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = (lane + 64 * k < c) ? v[k] - mean : 0.f;
            sq += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = lane + 64 * k;
            if (ch < c) {
                float y = (v[k] - mean) * rstd * gamma[ch] + beta[ch];
                if (act == 1) y = fmaxf(y, 0.f);
                else if (act == 2) y = y > 20.f ? y : __logf(1.f + __expf(y));
                if (res) y += res[p * c + ch];
                dst[p * c + ch] = y;
            }
        }
    }
}

hipError_t launch_layernorm(const float* src, const float* res, float* dst, const float* gamma, const float* beta,
                            long long positions, int c, float eps, int act, hipStream_t stream) {
    if (positions <= 0) return hipSuccess;
    if (c <= 0 || c > 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel, dim3(grid_for(positions, 4, 1 << 16)), dim3(256), 0, stream, src, res, dst, gamma,
                       beta, positions, c, eps, act);
    return hipGetLastError();
}

// ---- ragged segment sum (reduceSlots) -------------------------------------------------------
// Rows of a segment are contiguous; each thread owns one float4 column of one segment and adds the
// rows in order, so the result is deterministic and independent of the batch composition.
__global__ void segsum_kernel(const float* __restrict__ src, float* __restrict__ dst,
                              const int32_t* __restrict__ off, int row4) {
    const int seg = blockIdx.y;
    const int lo = off[seg], hi = off[seg + 1];
    for (int col = blockIdx.x * blockDim.x + threadIdx.x; col < row4; col += gridDim.x * blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = lo; r < hi; ++r) {
            const f32x4 v = *(const f32x4*)(src + ((long long)r * row4 + col) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        *(f32x4*)(dst + ((long long)seg * row4 + col) * 4) = acc;
    }
}

hipError_t launch_segsum(const float* src, float* dst, const int32_t* off, int n_seg, int row_floats,
                         hipStream_t stream) {
    if (n_seg <= 0) return hipSuccess;
    const int row4 = row_floats / 4;
    // gridDim.y is limited to 65535: walk segments in slabs
    for (int s0 = 0; s0 < n_seg; s0 += 65535) {
        const int ns = (n_seg - s0 < 65535) ? n_seg - s0 : 65535;
        hipLaunchKernelGGL(segsum_kernel, dim3((row4 + 255) / 256, ns), dim3(256), 0, stream, src,
                           dst + (long long)s0 * row_floats, off + s0, row4);
    }
    return hipGetLastError();
}

// ---- dst[a] = a0*src0[a] + a1*src1[owner[a]] -------------------------------------------------
// REST: dst[a] = src0[a] - (src1[owner[a]] - src0[a]), the "allele minus the other alleles of its site"
// expert input of MoEMergedAdvanced (MixtureOfExpertsAdvanced.py:372-383), in the reference's rounding order.
template <bool REST>
__global__ void mix_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                           float* __restrict__ dst, const int32_t* __restrict__ owner, long long n4,
                           int row4, float a0, float a1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / row4;
        const int col = (int)(i - row * row4);
        const f32x4 x = *(const f32x4*)(src0 + i * 4);
        const f32x4 s = *(const f32x4*)(src1 + ((long long)owner[row] * row4 + col) * 4);
        f32x4 v;
        // LinearCombination (NNTools.py:771-777): result = 0; result += c0*x; result += c1*s
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = REST ? x[e] - (s[e] - x[e]) : a0 * x[e] + a1 * s[e];
        *(f32x4*)(dst + i * 4) = v;
    }
}

hipError_t launch_mix(const float* src0, const float* src1, float* dst, const int32_t* owner,
                      long long rows, int row_floats, float a0, float a1, bool rest, hipStream_t stream) {
    const long long n4 = rows * (row_floats / 4);
    if (n4 <= 0) return hipSuccess;
    if (rest)
        hipLaunchKernelGGL(mix_kernel<true>, dim3(grid_for(n4, 256)), dim3(256), 0, stream, src0, src1, dst, owner,
                           n4, row_floats / 4, a0, a1);
    else
        hipLaunchKernelGGL(mix_kernel<false>, dim3(grid_for(n4, 256)), dim3(256), 0, stream, src0, src1, dst, owner,
                           n4, row_floats / 4, a0, a1);
    return hipGetLastError();
}

// ---- expert head: mean over length, Linear, optional softmax ----------------------------------
// one workgroup (256 threads) per row; c <= 1024, cout <= 4
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ src,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out,
                                                   int len, int c, int cout, long long stride_o,
                                                   long long stride_row, int softmax) {
    __shared__ float s_part[4][4];
    const long long row = blockIdx.x;
    const int t = threadIdx.x;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = t; ch < c; ch += 256) {
        float sum = 0.f;
        for (int l = 0; l < len; ++l) sum += src[(row * len + l) * c + ch];
        const float pooled = sum / (float)len;
        for (int o = 0; o < cout; ++o) part[o] += pooled * w[o * c + ch];
    }
    for (int o = 0; o < cout; ++o) {
        float v = part[o];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
        if ((t & 63) == 0) s_part[o][t >> 6] = v;
    }
    __syncthreads();
    if (t == 0) {
        float y[4];
        for (int o = 0; o < cout; ++o) y[o] = ((s_part[o][0] + s_part[o][1]) + (s_part[o][2] + s_part[o][3])) + b[o];
        if (softmax) {
            float mx = y[0];
            for (int o = 1; o < cout; ++o) mx = fmaxf(mx, y[o]);
            float den = 0.f;
            for (int o = 0; o < cout; ++o) { y[o] = expf(y[o] - mx); den += y[o]; }
            for (int o = 0; o < cout; ++o) y[o] /= den;
        }
        for (int o = 0; o < cout; ++o) out[o * stride_o + row * stride_row] = y[o];
    }
}

hipError_t launch_head(const float* src, const float* w, const float* b, float* out, long long rows,
                       int len, int c, int cout, long long out_stride_o, long long out_stride_row,
                       int softmax, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (cout > 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)rows), dim3(256), 0, stream, src, w, b, out, len, c,
                       cout, out_stride_o, out_stride_row, softmax);
    return hipGetLastError();
}

// ---- channel concat / add --------------------------------------------------------------------
__global__ void concat_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                              float* __restrict__ dst, long long n4, int c0q, int c1q) {
    const int cq = c0q + c1q;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long rp = i / cq;
        const int q = (int)(i - rp * cq);
        const f32x4 v = (q < c0q) ? *(const f32x4*)(s0 + (rp * c0q + q) * 4)
                                  : *(const f32x4*)(s1 + (rp * c1q + (q - c0q)) * 4);
        *(f32x4*)(dst + i * 4) = v;
    }
}

hipError_t launch_concat(const float* src0, const float* src1, float* dst, long long rows_x_len, int c0,
                         int c1, hipStream_t stream) {
    const long long n4 = rows_x_len * ((c0 + c1) / 4);
    if (n4 <= 0) return hipSuccess;
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, stream, src0, src1, dst, n4,
                       c0 / 4, c1 / 4);
    return hipGetLastError();
}

__global__ void add_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                           float* __restrict__ dst, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(s0 + i * 4), b = *(const f32x4*)(s1 + i * 4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e] + b[e];
        *(f32x4*)(dst + i * 4) = v;
    }
}

hipError_t launch_add(const float* src0, const float* src1, float* dst, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream, src0, src1, dst, n / 4);
    return hipGetLastError();
}

// ---- [R][C][L] -> [R][L][C] (training-storage layout, MemmapDatasetLoader.py:68-74) ------------
__global__ void rcl_to_rlc_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                  long long n, int len, int c) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const long long rl = i / c;
        const int l = (int)(rl % len);
        const long long r = rl / len;
        dst[i] = src[(r * c + ch) * len + l];
    }
}

hipError_t launch_rcl_to_rlc(const uint8_t* src, uint8_t* dst, long long rows, int len, int c,
                             hipStream_t stream) {
    const long long n = rows * len * c;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rcl_to_rlc_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, src, dst, n, len, c);
    return hipGetLastError();
}

// ---- genotype-pair posteriors (MixtureOfExpertsAdvanced.py:530-589) ---------------------------
// one thread per site.  Single-expert models: experts = [sigmoid(logit), 0, 0], meta = [1, 0, 0]
// (:535-538); ensembles: sigmoid of all three logit rows and the site's meta row (:531-533).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void posteriors_kernel(const float* __restrict__ logits, const float* __restrict__ meta,
                                  const int32_t* __restrict__ allele_off,
                                  const int64_t* __restrict__ pair_off, int n_sites, long long n_alleles,
                                  int n_experts, long long n_pairs, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sites) return;
    const int a0 = allele_off[s], na = allele_off[s + 1] - a0;
    float m[3] = {1.f, 0.f, 0.f};
    if (n_experts == 3 && meta) {
        m[0] = meta[3 * s];
        m[1] = meta[3 * s + 1];
        m[2] = meta[3 * s + 2];
    }
    long long pidx = pair_off[s];
    // log(p tk + (1 - p)(1 - tk) + 1e-10) is log(p + 1e-10) for an allele of the pair and log((1 - p) + 1e-10) otherwise
    // (p * 1 + (1 - p) * 0 == p exactly): both are taken once per (expert, allele) instead of once per pair -- every
    // logit requested before the first is used -- and summed per pair in the same allele order: the same bits as the
    // literal form, which serves sites of more than MAXA alleles
    constexpr int MAXA = 8;
    float in_pair[3][MAXA], off_pair[3][MAXA];
    if (na <= MAXA) {
        float lg[3][MAXA];
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int k = 0; k < MAXA; ++k) lg[e][k] = (e < n_experts && k < na) ? logits[(long long)e * n_alleles + a0 + k] : 0.f;
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int k = 0; k < MAXA; ++k) {
                const float p = (e < n_experts) ? sigmoidf_(lg[e][k]) : 0.f;
                in_pair[e][k] = logf(p + 1e-10f);
                off_pair[e][k] = logf((1.f - p) + 1e-10f);
            }
    }
    for (int i = 0; i < na; ++i) {
        for (int j = i; j < na; ++j, ++pidx) {
            float pe[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                float acc = 0.f;
                if (na <= MAXA) {
#pragma unroll
                    for (int k = 0; k < MAXA; ++k)
                        if (k < na) acc += (k == i || k == j) ? in_pair[e][k] : off_pair[e][k];
                } else {
                    for (int k = 0; k < na; ++k) {
                        const float p = (e < n_experts) ? sigmoidf_(logits[(long long)e * n_alleles + a0 + k]) : 0.f;
                        const float tk = (k == i || k == j) ? 1.f : 0.f;
                        acc += logf(p * tk + (1.f - p) * (1.f - tk) + 1e-10f);
                    }
                }
                pe[e] = expf(acc);
            }
            out[pidx] = m[0] * pe[0] + m[1] * pe[1] + m[2] * pe[2];
            out[n_pairs + pidx] = pe[0];
            out[2 * n_pairs + pidx] = pe[1];
            out[3 * n_pairs + pidx] = pe[2];
        }
    }
}

hipError_t launch_posteriors(const float* logits, const float* meta, const int32_t* allele_off,
                             const int64_t* pair_off, int n_sites, long long n_alleles, int n_experts,
                             long long n_pairs_total, float* out, hipStream_t stream) {
    if (n_sites <= 0) return hipSuccess;
    hipLaunchKernelGGL(posteriors_kernel, dim3((n_sites + 127) / 128), dim3(128), 0, stream, logits, meta,
                       allele_off, pair_off, n_sites, n_alleles, n_experts, n_pairs_total, out);
    return hipGetLastError();
}

int device_cus() {
    static int cached[64] = {};          // benign race: every thread writes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

}  // namespace hello
