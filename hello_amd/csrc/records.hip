// Host side of the record stage (include/hello_mi355x.h: hello_site_records): pair posteriors -> genotype call -> normalised
// VCF line, the meta-weighted mean call of the final VCF, and the per-shard ``.features`` pickle streams -- for every site
// of a launch in one multi-threaded call, so the Python driver touches no per-site object.
//
// Reference semantics (string work on the host in the reference too):
//   python/caller_calling.py:698-754   best pair, QUAL, ALT list, genotype, record, ``.features`` entry
//   python/prepareVcf.py:36-105,138-168 callAlleles on the meta-weighted mean of the experts (float64)
//   python/vcfFromContigs.py:139-227   fixEmptyAlleles / createVcfRecord normalisation
// ALT alleles are emitted sorted (hello_amd/vcf.py explains why); ties between pair probabilities are broken the way
// Python's ``sorted([(v, k)], reverse=True)[0]`` breaks them: by the pair of allele strings.
// No device code in this file.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hello_mi355x.h"

namespace hello {
int set_last_error(int code, const char* fmt, ...);      // engine.hip
int exception_status(const char* where) noexcept;         // engine.hip
}

struct hello_records {
    int32_t n_sites = 0, n_shards = 0;
    std::string shard_vcf, mean_vcf, features;
    std::vector<int64_t> shard_vcf_off, mean_vcf_off, features_off, mean_position;
    std::vector<int32_t> n_records;
    std::vector<int32_t> best_pair;      // [5][S]
    std::vector<double> best_p, qual;    // [5][S]
};

namespace {

constexpr double QUAL_CAP = 1 - 1e-8;    // "Quality score restricted to value 80" (prepareVcf.py:61)

struct Str {                             // a view of bytes
    const char* p;
    size_t n;
    bool operator==(const Str& o) const { return n == o.n && (n == 0 || memcmp(p, o.p, n) == 0); }
    std::string str() const { return std::string(p, n); }
};

int cmp(const Str& a, const Str& b) {    // Python's str ordering for ASCII text: by code point, shorter prefix first
    const size_t n = a.n < b.n ? a.n : b.n;
    const int c = n ? memcmp(a.p, b.p, n) : 0;
    if (c) return c;
    return a.n < b.n ? -1 : (a.n > b.n ? 1 : 0);
}

struct Reference {                       // what ``genome[i]`` / ``genome[a:b]`` read: a whole chromosome or the site's window
    const char* text;
    int64_t first, length;               // genome coordinate of text[0], bytes available
    bool covers(int64_t lo, int64_t hi) const { return lo >= first && hi <= first + length && lo <= hi; }
    char at(int64_t i) const { return text[i - first]; }
};

struct Failure {
    bool failed = false;
    char message[256];
    void set(const char* fmt, ...) {
        if (failed) return;
        failed = true;
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(message, sizeof(message), fmt, ap);
        va_end(ap);
    }
};

// ---- pickle stream (protocol 4 opcodes, no framing; the memo holds the five field names, written in the preamble: a
// shard's stream depends on nothing but its own records) ---------------------------------------------------------------
struct Pickle {
    std::string& o;
    explicit Pickle(std::string& out) : o(out) {}
    void op(char c) { o.push_back(c); }
    void u32(uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; o.append(b, 4); }
    void text(const char* p, size_t n) {
        if (n < 256) { op('\x8c'); op((char)n); }            // SHORT_BINUNICODE
        else { op('X'); u32((uint32_t)n); }                  // BINUNICODE
        o.append(p, n);
    }
    void integer(int64_t v) {
        if (v >= 0 && v < 256) { op('K'); op((char)v); }     // BININT1
        else if (v >= INT32_MIN && v <= INT32_MAX) { op('J'); u32((uint32_t)(int32_t)v); }   // BININT
        else { op('\x8a'); op(8); for (int i = 0; i < 8; ++i) op((char)((uint64_t)v >> (8 * i))); }   // LONG1, 8 bytes
    }
    void real(double v) {                                    // BINFLOAT: big-endian IEEE double
        uint64_t bits;
        memcpy(&bits, &v, 8);
        op('G');
        for (int i = 7; i >= 0; --i) op((char)(bits >> (8 * i)));
    }
    void get(uint32_t i) {
        if (i < 256) { op('h'); op((char)i); }               // BINGET
        else { op('j'); u32(i); }                            // LONG_BINGET
    }
    void put(uint32_t i) {
        if (i < 256) { op('q'); op((char)i); }               // BINPUT
        else { op('r'); u32(i); }                            // LONG_BINPUT
    }
};

enum { MEMO_CHROMOSOME = 0, MEMO_POSITION, MEMO_LENGTH, MEMO_META, MEMO_EXPERTS };

// ---- one site --------------------------------------------------------------------------------------------------------
struct SiteCtx {
    int n_alleles;
    const Str* names;                    // the site's allele strings, in site order
    Str chromosome;
    int64_t start, length;
    Reference ref;
};

struct RowCall {
    bool record = false;
    int64_t position = -1;               // 0-based, after normalisation
    int best = -1;                       // pair index within the site
    double p = 0, qual = 0;
};

// best pair of one row: max over (p, (a, b)) tuples
template <class F>
void best_pair(const SiteCtx& s, F value, int& best, double& best_p) {
    int k = 0, bi = 0, bj = 0;
    best = -1;
    for (int i = 0; i < s.n_alleles; ++i)
        for (int j = i; j < s.n_alleles; ++j, ++k) {
            const double p = value(k);
            bool take = best < 0 || p > best_p;
            if (!take && p == best_p) {                       // equal probabilities: the larger pair of strings wins
                const int c = cmp(s.names[i], s.names[bi]);
                take = c > 0 || (c == 0 && cmp(s.names[j], s.names[bj]) > 0);
            }
            if (take) { best = k; best_p = p; bi = i; bj = j; }
        }
}

void pair_of(int n, int k, int& i, int& j) {
    for (i = 0; i < n; ++i) {
        if (k < n - i) { j = i + k; return; }
        k -= n - i;
    }
    i = j = 0;
}

// vcfFromContigs.py:139-160
bool pad_left_if_empty(const SiteCtx& s, int64_t& pos, std::string& ref, std::vector<std::string>& alts, Failure& f) {
    for (auto& a : alts) a.erase(std::remove(a.begin(), a.end(), '-'), a.end());
    bool empty = ref.empty();
    for (auto& a : alts) empty = empty || a.empty();
    if (!empty) return false;
    pos -= 1;
    if (!s.ref.covers(pos, pos + 1)) {
        f.set("site %.*s:%lld: normalisation needs reference base %lld, outside the reference available for the site "
              "[%lld, %lld)", (int)s.chromosome.n, s.chromosome.p, (long long)s.start, (long long)pos,
              (long long)s.ref.first, (long long)(s.ref.first + s.ref.length));
        return false;
    }
    const char anchor = s.ref.at(pos);
    ref.insert(ref.begin(), anchor);
    for (auto& a : alts) a.insert(a.begin(), anchor);
    return true;
}

// callAlleles / caller_calling.vcfRecords for one row; appends the line (with '\n') to `out` when there is a record
template <class F>
RowCall call_row(const SiteCtx& s, F value, const char* info, std::string& out, Failure& f) {
    RowCall c;
    best_pair(s, value, c.best, c.p);
    c.qual = -10.0 * std::log10(1.0 - (c.p < QUAL_CAP ? c.p : QUAL_CAP));
    int bi, bj;
    pair_of(s.n_alleles, c.best, bi, bj);
    const Str ref_allele{s.ref.text + (s.start - s.ref.first), (size_t)s.length};
    std::vector<Str> alt_views;                               // sorted(set(best pair) - {ref allele})
    auto add = [&](const Str& a) {
        if (a == ref_allele) return;
        for (auto& x : alt_views)
            if (x == a) return;
        alt_views.push_back(a);
    };
    add(s.names[bi]);
    add(s.names[bj]);
    int genotype[2] = {0, 0};
    auto by_text = [](const Str& a, const Str& b) { return cmp(a, b) < 0; };
    if (!alt_views.empty()) {
        std::sort(alt_views.begin(), alt_views.end(), by_text);
        const Str top[2] = {s.names[bi], s.names[bj]};
        for (int t = 0; t < 2; ++t) {
            if (top[t] == ref_allele) continue;
            for (size_t x = 0; x < alt_views.size(); ++x)
                if (alt_views[x] == top[t]) genotype[t] = (int)x + 1;
        }
    } else {
        for (int a = 0; a < s.n_alleles; ++a) add(s.names[a]);
        if (alt_views.empty()) return c;                      // no alternative allele at the site: nothing is written
        std::sort(alt_views.begin(), alt_views.end(), by_text);
    }
    // createVcfRecord (vcfFromContigs.py:162-227)
    int64_t pos = s.start;
    std::string ref = ref_allele.str();
    std::vector<std::string> alts;
    for (auto& a : alt_views) alts.push_back(a.str());
    pad_left_if_empty(s, pos, ref, alts, f);
    if (f.failed) return c;
    bool all_ref = true;
    for (auto& a : alts) all_ref = all_ref && a == ref;
    if (alts.empty() || all_ref) return c;
    for (;;) {
        bool trimmed = true;
        const char last = ref.back();
        for (auto& a : alts) trimmed = trimmed && a.back() == last;
        if (trimmed) {
            ref.pop_back();
            for (auto& a : alts) a.pop_back();
        }
        const bool padded = pad_left_if_empty(s, pos, ref, alts, f);
        if (f.failed) return c;
        if (!(trimmed || padded)) break;
    }
    for (;;) {
        bool can = ref.size() > 1;
        for (auto& a : alts) can = can && a.size() > 1 && a[0] == ref[0];
        if (!can) break;
        pos += 1;
        ref.erase(ref.begin());
        for (auto& a : alts) a.erase(a.begin());
    }
    c.record = true;
    c.position = pos;
    char num[64];
    out.append(s.chromosome.p, s.chromosome.n);
    out.append(num, snprintf(num, sizeof(num), "\t%lld\t.\t", (long long)(pos + 1)));
    out.append(ref);
    out.push_back('\t');
    for (size_t x = 0; x < alts.size(); ++x) {
        if (x) out.push_back(',');
        out.append(alts[x]);
    }
    out.append(num, snprintf(num, sizeof(num), "\t%f\tPASS\t", c.qual));
    out.append(info);
    out.append(num, snprintf(num, sizeof(num), "\tGT\t%d/%d\n", genotype[0], genotype[1]));
    return c;
}

// Worker threads that live as long as the library: a call hands out its chunks and waits.  (Threads created per call
// would each attach a fresh malloc arena -- glibc allows 8 per CPU -- and the arenas keep what the short-lived workers
// freed: over 132 launches x 16 threads on a 256-CPU host the resident set grew by 0.5 GB.)  Calls are served one at a
// time; a call is ~1 ms of work.
class WorkerPool {
public:
    void run(int n, const std::function<void(int)>& f) {
        if (n <= 1) {
            if (n == 1) f(0);
            return;
        }
        std::lock_guard<std::mutex> one_call(run_mutex_);
        std::unique_lock<std::mutex> lk(m_);
        while ((int)threads_.size() < n - 1) threads_.emplace_back([this] { loop(); });
        job_ = &f;
        n_jobs_ = n;
        next_ = 0;
        pending_ = n;
        ++generation_;
        cv_work_.notify_all();
        lk.unlock();
        drain();                                  // the calling thread works too
        lk.lock();
        cv_done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void drain() {
        for (;;) {
            int i;
            const std::function<void(int)>* f;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!job_ || next_ >= n_jobs_) return;
                i = next_++;
                f = job_;
            }
            (*f)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            drain();
        }
    }
    std::mutex run_mutex_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* job_ = nullptr;
    int n_jobs_ = 0, next_ = 0, pending_ = 0;
    unsigned long long generation_ = 0;
};

WorkerPool& pool() {
    static WorkerPool* p = new WorkerPool();      // never destroyed: its threads wait on it until the process exits
    return *p;
}

struct Chunk {                           // what one worker produced for its contiguous range of sites
    int32_t lo = 0, hi = 0;
    std::string shard_vcf, mean_vcf;
    std::vector<int32_t> shard_len, mean_len;               // per site of the range
    struct Piece { int32_t shard; int32_t records; std::string bytes; };
    std::vector<Piece> pieces;                               // ``.features`` records, per shard touched, in order
    Failure failure;
};

}  // namespace

extern "C" {

int hello_site_records(const hello_site_table* t, const float* posteriors, int64_t n_pairs_total, const float* meta,
                       const int32_t* shard_site_off, int32_t n_shards, const hello_features_format* fmt,
                       int32_t n_threads, hello_records** out) try {
    using hello::set_last_error;
    if (!t || !posteriors || !out) return set_last_error(HELLO_ERR_ARG, "NULL pointer");
    const int32_t S = t->n_sites;
    if (S < 0 || !t->alleles_per_site || !t->allele_text_off || !t->chromosome_of_site || !t->chromosome_text_off ||
        !t->start || !t->stop || (S > 0 && !t->allele_text))
        return set_last_error(HELLO_ERR_ARG, "incomplete site table");
    if (!t->genome && (!t->ref_windows || !t->ref_window_off || !t->window_start))
        return set_last_error(HELLO_ERR_ARG, "the site table carries neither chromosome sequences nor per-site reference windows");
    if (fmt && (!fmt->meta_prefix || !fmt->meta_suffix || fmt->meta_prefix_len < 0 || fmt->meta_suffix_len < 0))
        return set_last_error(HELLO_ERR_ARG, "incomplete features format");
    const int32_t one_shard[2] = {0, S};
    if (!shard_site_off) { shard_site_off = one_shard; n_shards = 1; }
    if (n_shards < 1 || shard_site_off[0] != 0 || shard_site_off[n_shards] != S)
        return set_last_error(HELLO_ERR_SHAPE, "shard_site_off must run from 0 to n_sites");
    for (int32_t k = 0; k < n_shards; ++k)
        if (shard_site_off[k + 1] < shard_site_off[k]) return set_last_error(HELLO_ERR_SHAPE, "shard_site_off decreases at %d", k);

    // offsets of every site's alleles and pairs
    std::vector<int64_t> aoff(S + 1, 0), poff(S + 1, 0);
    for (int32_t s = 0; s < S; ++s) {
        const int32_t n = t->alleles_per_site[s];
        if (n <= 0) return set_last_error(HELLO_ERR_SHAPE, "alleles_per_site[%d] = %d", s, n);
        aoff[s + 1] = aoff[s] + n;
        poff[s + 1] = poff[s] + (int64_t)n * (n + 1) / 2;
        const int32_t c = t->chromosome_of_site[s];
        if (c < 0 || c >= t->n_chromosomes) return set_last_error(HELLO_ERR_SHAPE, "chromosome_of_site[%d] = %d", s, c);
        if (t->stop[s] < t->start[s]) return set_last_error(HELLO_ERR_SHAPE, "site %d: stop < start", s);
    }
    if (poff[S] != n_pairs_total)
        return set_last_error(HELLO_ERR_SHAPE, "n_pairs_total = %lld, expected %lld", (long long)n_pairs_total, (long long)poff[S]);
    const int64_t A = aoff[S];
    for (int64_t a = 0; a < A; ++a)
        if (t->allele_text_off[a + 1] < t->allele_text_off[a]) return set_last_error(HELLO_ERR_SHAPE, "allele_text_off decreases at %lld", (long long)a);
    std::vector<Str> names(A);
    for (int64_t a = 0; a < A; ++a)
        names[a] = Str{(const char*)t->allele_text + t->allele_text_off[a], (size_t)(t->allele_text_off[a + 1] - t->allele_text_off[a])};
    std::vector<int32_t> shard_of(S);
    for (int32_t k = 0; k < n_shards; ++k)
        for (int32_t s = shard_site_off[k]; s < shard_site_off[k + 1]; ++s) shard_of[s] = k;

    auto* rec = new hello_records();
    rec->n_sites = S;
    rec->n_shards = n_shards;
    rec->best_pair.assign((size_t)5 * S, -1);
    rec->best_p.assign((size_t)5 * S, 0.0);
    rec->qual.assign((size_t)5 * S, 0.0);
    rec->mean_position.assign(S, -1);
    rec->n_records.assign(n_shards, 0);

    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(std::min(T, 64), std::max(1, S / 256)));
    std::vector<Chunk> chunks(T);
    const int64_t P = n_pairs_total;

    auto work = [&](int w) {
        Chunk& ch = chunks[w];
        try {
        ch.lo = (int32_t)((int64_t)S * w / T);
        ch.hi = (int32_t)((int64_t)S * (w + 1) / T);
        ch.shard_len.assign(ch.hi - ch.lo, 0);
        ch.mean_len.assign(ch.hi - ch.lo, 0);
        std::string scratch;
        for (int32_t s = ch.lo; s < ch.hi && !ch.failure.failed; ++s) {
            SiteCtx cx;
            cx.n_alleles = t->alleles_per_site[s];
            cx.names = names.data() + aoff[s];
            const int32_t c = t->chromosome_of_site[s];
            cx.chromosome = Str{(const char*)t->chromosome_text + t->chromosome_text_off[c],
                                (size_t)(t->chromosome_text_off[c + 1] - t->chromosome_text_off[c])};
            cx.start = t->start[s];
            cx.length = t->stop[s] - t->start[s];
            if (t->genome && t->genome[c])
                cx.ref = Reference{(const char*)t->genome[c], 0, t->genome_len[c]};
            else if (t->ref_windows)
                cx.ref = Reference{(const char*)t->ref_windows + t->ref_window_off[s], t->window_start[s],
                                   t->ref_window_off[s + 1] - t->ref_window_off[s]};
            else {
                ch.failure.set("site %d: no sequence for chromosome %.*s", s, (int)cx.chromosome.n, cx.chromosome.p);
                break;
            }
            if (!cx.ref.covers(cx.start, cx.start + cx.length)) {
                ch.failure.set("site %.*s:%lld: the allele span [%lld, %lld) leaves the reference available for the site [%lld, %lld)",
                               (int)cx.chromosome.n, cx.chromosome.p, (long long)cx.start, (long long)cx.start,
                               (long long)(cx.start + cx.length), (long long)cx.ref.first, (long long)(cx.ref.first + cx.ref.length));
                break;
            }
            const float* row[4] = {posteriors + poff[s], posteriors + P + poff[s], posteriors + 2 * P + poff[s],
                                   posteriors + 3 * P + poff[s]};
            const double m[3] = {meta ? (double)meta[3 * (int64_t)s] : 1.0, meta ? (double)meta[3 * (int64_t)s + 1] : 0.0,
                                 meta ? (double)meta[3 * (int64_t)s + 2] : 0.0};
            // prepareVcf.py:154-163: sum(float(e_i) * float(meta_i)), Python's sum from the int 0, in float64
            auto mean = [&](int k) { return (((double)row[1][k] * m[0]) + (double)row[2][k] * m[1]) + (double)row[3][k] * m[2]; };
            const bool keep = !t->keep || t->keep[s];
            RowCall calls[5];
            const size_t before = ch.shard_vcf.size();
            if (keep) {
                calls[0] = call_row(cx, [&](int k) { return (double)row[0][k]; }, "MixtureOfExpertPrediction", ch.shard_vcf, ch.failure);
            } else {
                best_pair(cx, [&](int k) { return (double)row[0][k]; }, calls[0].best, calls[0].p);
                calls[0].qual = -10.0 * std::log10(1.0 - (calls[0].p < QUAL_CAP ? calls[0].p : QUAL_CAP));
            }
            ch.shard_len[s - ch.lo] = (int32_t)(ch.shard_vcf.size() - before);
            for (int e = 0; e < 3; ++e) {                     // the experts' rows: decision only (prepareVcf's expert<N>.vcf)
                best_pair(cx, [&](int k) { return (double)row[1 + e][k]; }, calls[1 + e].best, calls[1 + e].p);
                calls[1 + e].qual = -10.0 * std::log10(1.0 - (calls[1 + e].p < QUAL_CAP ? calls[1 + e].p : QUAL_CAP));
            }
            if (calls[0].record) {                            // a site enters the final VCF through its .features entry
                const size_t mb = ch.mean_vcf.size();
                calls[4] = call_row(cx, mean, "HELLO", ch.mean_vcf, ch.failure);
                ch.mean_len[s - ch.lo] = (int32_t)(ch.mean_vcf.size() - mb);
                if (calls[4].record) rec->mean_position[s] = calls[4].position;
            } else {
                best_pair(cx, mean, calls[4].best, calls[4].p);
                calls[4].qual = -10.0 * std::log10(1.0 - (calls[4].p < QUAL_CAP ? calls[4].p : QUAL_CAP));
            }
            for (int r = 0; r < 5; ++r) {
                rec->best_pair[(size_t)r * S + s] = calls[r].best;
                rec->best_p[(size_t)r * S + s] = calls[r].p;
                rec->qual[(size_t)r * S + s] = calls[r].qual;
            }
            if (fmt && calls[0].record) {                     // caller_calling.py:743-754
                const int32_t shard = shard_of[s];
                if (ch.pieces.empty() || ch.pieces.back().shard != shard) ch.pieces.push_back({shard, 0, std::string()});
                auto& piece = ch.pieces.back();
                piece.records += 1;
                Pickle pk(piece.bytes);
                pk.op('}'); pk.op('(');                        // EMPTY_DICT MARK
                pk.get(MEMO_CHROMOSOME); pk.text(cx.chromosome.p, cx.chromosome.n);
                pk.get(MEMO_POSITION); pk.integer(cx.start);
                pk.get(MEMO_LENGTH); pk.integer(cx.length);
                pk.get(MEMO_META);
                piece.bytes.append((const char*)fmt->meta_prefix, fmt->meta_prefix_len);
                const float m32[3] = {meta ? meta[3 * (int64_t)s] : 1.f, meta ? meta[3 * (int64_t)s + 1] : 0.f,
                                      meta ? meta[3 * (int64_t)s + 2] : 0.f};
                piece.bytes.append((const char*)m32, 12);
                piece.bytes.append((const char*)fmt->meta_suffix, fmt->meta_suffix_len);
                pk.get(MEMO_EXPERTS);
                for (int e = 0; e < 3; ++e) {
                    pk.op('}'); pk.op('(');
                    int k = 0;
                    for (int i = 0; i < cx.n_alleles; ++i)
                        for (int j = i; j < cx.n_alleles; ++j, ++k) {
                            pk.text(cx.names[i].p, cx.names[i].n);
                            pk.text(cx.names[j].p, cx.names[j].n);
                            pk.op('\x86');                     // TUPLE2
                            pk.real((double)row[1 + e][k]);
                        }
                    pk.op('u');                                // SETITEMS
                }
                pk.op('\x87');                                 // TUPLE3
                pk.op('u');
            }
        }
        } catch (const std::exception& ex) {      // a worker thread must not unwind out of its job (std::bad_alloc on a huge launch)
            ch.failure.set("record stage, sites %d..%d: %s", ch.lo, ch.hi, ex.what());
        } catch (...) {
            ch.failure.set("record stage, sites %d..%d: unknown C++ exception", ch.lo, ch.hi);
        }
    };
    pool().run(T, work);
    for (auto& ch : chunks)
        if (ch.failure.failed) {
            const int rc = set_last_error(HELLO_ERR_ARG, "%s", ch.failure.message);
            delete rec;
            return rc;
        }

    // assemble: texts in site order, per-site offsets, one pickle stream per shard
    rec->shard_vcf_off.assign(S + 1, 0);
    rec->mean_vcf_off.assign(S + 1, 0);
    size_t n_shard_text = 0, n_mean_text = 0;
    for (auto& ch : chunks) { n_shard_text += ch.shard_vcf.size(); n_mean_text += ch.mean_vcf.size(); }
    rec->shard_vcf.reserve(n_shard_text);
    rec->mean_vcf.reserve(n_mean_text);
    for (auto& ch : chunks) {
        rec->shard_vcf += ch.shard_vcf;
        rec->mean_vcf += ch.mean_vcf;
        for (int32_t s = ch.lo; s < ch.hi; ++s) {
            rec->shard_vcf_off[s + 1] = rec->shard_vcf_off[s] + ch.shard_len[s - ch.lo];
            rec->mean_vcf_off[s + 1] = rec->mean_vcf_off[s] + ch.mean_len[s - ch.lo];
        }
    }
    rec->features_off.assign(n_shards + 1, 0);
    if (fmt) {
        std::string preamble;
        Pickle pk(preamble);
        pk.op('\x80'); pk.op(4);                               // PROTO 4 (no FRAME: framing is optional for readers)
        const char* fields[5] = {"chromosome", "position", "length", "meta", "expertPredictions"};
        for (uint32_t i = 0; i < 5; ++i) { pk.text(fields[i], strlen(fields[i])); pk.put(i); pk.op('0'); }   // memoise, POP
        pk.op(']');                                            // EMPTY_LIST
        std::vector<std::vector<const Chunk::Piece*>> per_shard(n_shards);
        for (auto& ch : chunks)
            for (auto& piece : ch.pieces) per_shard[piece.shard].push_back(&piece);
        size_t total = 0;
        for (int32_t k = 0; k < n_shards; ++k) {
            total += preamble.size() + 1;
            for (auto* piece : per_shard[k]) total += piece->bytes.size();
            total += 2;
        }
        rec->features.reserve(total);
        for (int32_t k = 0; k < n_shards; ++k) {
            rec->features += preamble;
            // ONE MARK ... APPENDS group per shard around all of its records, whichever worker chunks they came from: the
            // stream's bytes do not depend on the thread count or on where the chunk boundaries fell
            if (!per_shard[k].empty()) rec->features.push_back('(');       // MARK
            for (auto* piece : per_shard[k]) {
                rec->features += piece->bytes;
                rec->n_records[k] += piece->records;
            }
            if (!per_shard[k].empty()) rec->features.push_back('e');       // APPENDS
            rec->features.push_back('.');                      // STOP
            rec->features_off[k + 1] = (int64_t)rec->features.size();
        }
    } else {
        for (int32_t s = 0; s < S; ++s)
            if (rec->shard_vcf_off[s + 1] > rec->shard_vcf_off[s]) rec->n_records[shard_of[s]] += 1;
    }
    *out = rec;
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_site_records");
}

int hello_records_get(const hello_records* r, hello_records_view* v) try {
    if (!r || !v) return hello::set_last_error(HELLO_ERR_ARG, "NULL pointer");
    v->n_sites = r->n_sites;
    v->n_shards = r->n_shards;
    v->shard_vcf = (const uint8_t*)r->shard_vcf.data();
    v->shard_vcf_off = r->shard_vcf_off.data();
    v->mean_vcf = (const uint8_t*)r->mean_vcf.data();
    v->mean_vcf_off = r->mean_vcf_off.data();
    v->mean_position = r->mean_position.data();
    v->features = (const uint8_t*)r->features.data();
    v->features_off = r->features_off.data();
    v->n_records = r->n_records.data();
    v->best_pair = r->best_pair.data();
    v->best_p = r->best_p.data();
    v->qual = r->qual.data();
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_records_get");
}

void hello_records_destroy(hello_records* r) { delete r; }

}  // extern "C"
