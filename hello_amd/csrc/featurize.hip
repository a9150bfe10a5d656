// Pileup-tensor producer: reads + CIGARs + reference window -> uint8 [R][L][C] colour
// tensors, written in exactly the layout the scoring engine consumes.
//
// Reference semantics: AlleleSearcherLiteFiltered::computeFeaturesColoredSimple and its colour helpers
// (c++/src/AlleleSearcherLiteFiltered.cpp:971-1180).  Byte work, HBM-side: 900-1050 output bytes per read.
//
// One wave per read (four reads per workgroup).  The CIGAR is walked operation by operation (wave-uniform
// loop); inside an operation the 64 lanes take the positions.  The row is assembled in LDS -- LDS
// accesses of one wave complete in program order, which is what makes "a later operation overwrites the
// position before it" (deletions and insertions repaint the preceding base as a gap) correct -- and the
// workgroup then writes its four contiguous rows to global memory with 4-byte stores.
#include "kernels.h"

namespace hello {

namespace {
constexpr int BAM_CMATCH = 0, BAM_CINS = 1, BAM_CDEL = 2, BAM_CREF_SKIP = 3, BAM_CSOFT_CLIP = 4, BAM_CEQUAL = 7,
              BAM_CDIFF = 8;

__device__ __forceinline__ int base_color(unsigned char b) {       // :971-985
    switch (b) {
        case 'A': return 40 + 3 * 70;
        case 'G': return 40 + 2 * 70;
        case 'T': return 30 + 1 * 70;
        case 'C': return 30;
        default: return 0;
    }
}
// int(254 * (1.0 * min(q, cap) / cap)) of the reference (:988-999).  In integers: the only q for which
// 254 q / cap is a whole number (q = 0, cap/2, cap) give exact binary fractions, so the double expression
// never lands within rounding distance of an integer from below and (254 q) / cap is bit-identical.
__device__ __forceinline__ int quality_color(int q, int cap) { return (254 * (q < cap ? q : cap)) / cap; }
}  // namespace

__global__ __launch_bounds__(256) void featurize_kernel(FeaturizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row_bytes = a.length * a.channels;
    const long long read0 = (long long)blockIdx.x * 4;
    const int rows_here = (int)((a.n_reads - read0) < 4 ? (a.n_reads - read0) : 4);
    const int total = rows_here * row_bytes;
    for (int i = tid * 4; i < 4 * row_bytes; i += 1024) *(unsigned*)(fz_lds + i) = 0u;   // 4*row_bytes % 4 == 0
    __syncthreads();

    const long long r = read0 + wave;
    if (r < a.n_reads) {
        volatile unsigned char* row = fz_lds + wave * row_bytes;
        const int C = a.channels;
        const int s = a.site_of_read[r];
        const long long wstart = a.window_start[s];
        const unsigned char* ref_text = a.ref + a.ref_off[s];
        const long long ref_len = a.ref_off[s + 1] - a.ref_off[s];
        // device-side guard (host callers are validated, hello_amd/shards.py): a position outside the site's reference
        // window reads as "no base", a CIGAR that runs past its read stops painting
        auto ref_at = [&](long long i) -> unsigned char { return (i >= 0 && i < ref_len) ? ref_text[i] : (unsigned char)0; };
        const long long as0 = a.asm_start[s], as1 = a.asm_stop[s];
        const long long start = (as0 + as1) / 2 - a.length / 2, end = start + a.length;
        const unsigned char* bases = a.bases + a.read_off[r];
        const unsigned char* quals = a.quals + a.read_off[r];
        const long long read_len = a.read_off[r + 1] - a.read_off[r];
        const int mapq_c = quality_color(a.mapq[r], 60);
        const int strand_c = a.orientation[r] > 0 ? 70 : 240;                   // :1002-1005
        const int hp = a.hp[r];
        const int hp_c = hp == 1 ? 120 : (hp == 2 ? 240 : 0);                   // :1019-1028
        auto position_color = [&](long long p) {                                 // :1008-1016, p relative to the window
            return (as0 - wstart <= p && p < as1 - wstart) ? 240 : 70;
        };
        long long rf = a.ref_start[r], rp = 0;
        for (long long ci = a.cigar_off[r]; ci < a.cigar_off[r + 1]; ++ci) {
            const unsigned c = a.cigars[ci];
            const int op = c & 15u;
            const long long len = c >> 4;
            if (op == BAM_CMATCH || op == BAM_CEQUAL || op == BAM_CDIFF) {     // :1074-1096
                for (long long j = lane; j < len; j += 64) {
                    const long long pos = rf + j;
                    if (start <= pos && pos < end && rp + j < read_len) {
                        volatile unsigned char* px = row + (pos - start) * C;
                        px[0] = (unsigned char)base_color(bases[rp + j]);
                        px[1] = (unsigned char)base_color(ref_at(pos - wstart));
                        px[2] = (unsigned char)quality_color(quals[rp + j], 40);
                        px[3] = (unsigned char)mapq_c;
                        px[4] = (unsigned char)strand_c;
                        px[5] = (unsigned char)position_color(pos - wstart);
                        if (C == 7) px[6] = (unsigned char)hp_c;
                    }
                }
                rf += len;
                rp += len;
            } else if (op == BAM_CDEL) {                                        // :1098-1125 (+ fall-through :1126)
                if (start <= rf - 1 && rf - 1 < end) {
                    for (long long i = rf - 1 + lane; i < rf + len; i += 64) {
                        if (start <= i && i < end) {
                            volatile unsigned char* px = row + (i - start) * C;
                            px[1] = (unsigned char)base_color(ref_at(i - wstart));
                            px[3] = (unsigned char)mapq_c;
                            px[4] = (unsigned char)strand_c;
                            px[5] = (unsigned char)position_color(i - wstart);
                            if (C == 7) px[6] = (unsigned char)hp_c;
                        }
                    }
                    if (lane == 0) {
                        volatile unsigned char* px = row + (rf - 1 - start) * C;
                        px[0] = 0;                                               // gap colour
                        px[2] = (unsigned char)((rp > 0 && rp <= read_len) ? quality_color(quals[rp - 1], 40) : 0);
                    }
                }
                rf += len;
            } else if (op == BAM_CREF_SKIP) {                                   // :1126-1129
                rf += len;
            } else if (op == BAM_CINS) {                                        // :1131-1160 (+ fall-through :1161)
                if (start <= rf - 1 && rf - 1 < end && lane == 0) {
                    int qmin = 255;
                    for (long long k = (rp > 0 ? rp - 1 : rp); k < rp + len && k < read_len; ++k) qmin = quals[k] < qmin ? quals[k] : qmin;
                    volatile unsigned char* px = row + (rf - 1 - start) * C;
                    px[0] = 0;
                    px[1] = (unsigned char)base_color(ref_at(rf - 1 - wstart));
                    px[2] = (unsigned char)quality_color(qmin, 40);
                    px[3] = (unsigned char)mapq_c;
                    px[4] = (unsigned char)strand_c;
                    px[5] = (unsigned char)position_color(rf - 1 - wstart);
                    if (C == 7) px[6] = (unsigned char)hp_c;
                }
                rp += len;
            } else if (op == BAM_CSOFT_CLIP) {                                  // :1161-1164
                rp += len;
            }
        }
    }
    __syncthreads();
    unsigned char* dst = a.out + read0 * row_bytes;
    if ((reinterpret_cast<unsigned long long>(dst) & 3ull) == 0) {
        for (int i = tid * 4; i < total; i += 1024) {
            if (i + 4 <= total) {
                *(unsigned*)(dst + i) = *(const unsigned*)(fz_lds + i);
            } else {
                for (int b = i; b < total; ++b) dst[b] = fz_lds[b];
            }
        }
    } else {
        for (int i = tid; i < total; i += 256) dst[i] = fz_lds[i];
    }
}

hipError_t launch_featurize(const FeaturizeArgs& a, hipStream_t stream) {
    if (a.n_reads <= 0) return hipSuccess;
    if (a.channels != 6 && a.channels != 7) return hipErrorInvalidValue;
    const unsigned groups = (unsigned)((a.n_reads + 3) / 4);
    const size_t lds = (size_t)4 * a.length * a.channels;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(featurize_kernel, dim3(groups), dim3(256), lds, stream, a);
    return hipGetLastError();
}

}  // namespace hello
