// Shared scoring server (host C++; no kernels): ONE process per GPU scores the sites of many worker processes.
//
// Reference deployment form: a pool of single-threaded worker processes, each loading the model and calling
// network(featureDict, ref_segment) once per site (python/call.py:111,215-221; python/caller_calling.py:863-868,872-891).
// The workers keep that loop (hello_amd/shared.py: SharedScoringNetwork packs a site into its slot of a shared-memory segment,
// sends one byte on a Unix-domain socket and blocks for the one-byte answer); this file is the other end.
//
//   threads   one per scorer (= engine), leader / follower: the idle thread that holds the poll mutex waits on the sockets itself
//             (accepting clients, dropping dead ones, reading request bytes), takes EVERY pending slot as its launch, lingers a few
//             tens of microseconds for the clients that could still send one (a launch costs the same for 1 or 16 sites), releases
//             the mutex to the next idle thread and scores: no hand-over between a poller and a scorer on a site's way in, no
//             interpreter lock anywhere on the path.
//   a launch  the slots' headers are checked against the slot's capacity and their own tables (a client is another process: nothing
//             it writes is followed unchecked), pileups / counts / reference segments gathered into the thread's host buffers,
//             scored with ONE hello_engine_forward over host pointers (logits, meta and pair posteriors back on the host), scattered
//             into the slots' result areas, and every client of the launch is answered with one byte.
//   liveness  the socket is the liveness signal both ways: a server that dies closes every client's socket (their blocked recv
//             returns), a client that dies frees its slot (after the launch that may still be reading it).
#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include "../../include/hello_mi355x.h"

namespace hello {
int set_last_error(int code, const char* fmt, ...);
int exception_status(const char* where) noexcept;
}  // namespace hello

namespace {

constexpr int MAX_ALLELES = HELLO_SITE_MAX_ALLELES;
constexpr int MAX_PAIRS = MAX_ALLELES * (MAX_ALLELES + 1) / 2;
constexpr int HEADER_INTS = 16, ERR_BYTES = 1024;
enum { H_ALLELES = 0, H_READS0, H_READS1, H_HAS_REF, H_PAIRS, H_ERRLEN };

double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Scorer {
    hello_site_scorer fn = nullptr;
    void* ctx = nullptr;
    hello_engine* engine = nullptr;
};

}  // namespace

struct hello_site_server {
    hello_site_server_config cfg{};
    std::string info_json, socket_path, shm_path;
    hello_site_slot_layout lay{};
    unsigned char* map = nullptr;
    size_t map_bytes = 0;
    int listener = -1;
    std::vector<Scorer> scorers;
    // the leader's (under poll_mu): sockets and what came off them
    std::mutex poll_mu;
    std::unique_ptr<std::atomic<int>[]> fd_of_slot;   // -1: free or zombie; written by the leader, read by the scorer threads' replies
    std::vector<int> pending;
    double idle_since = 0.0;
    // shared small state (under state_mu)
    std::mutex state_mu;
    std::deque<int> free_slots;
    std::vector<char> inflight, zombie;
    int n_inflight = 0, n_clients = 0;
    hello_site_server_stats stats{};
    std::atomic<int> stop{0};

    unsigned char* slot(int i) const { return map + (size_t)i * (size_t)cfg.slot_bytes; }
    int32_t* header(int i) const { return (int32_t*)(slot(i) + lay.header); }
};

namespace {

using Server = hello_site_server;

bool send_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += k;
        n -= (size_t)k;
    }
    return true;
}

bool recv_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) {
        const ssize_t k = recv(fd, c, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        c += k;
        n -= (size_t)k;
    }
    return true;
}

bool send_msg(int fd, const std::string& json) {
    const uint32_t n = (uint32_t)json.size();
    return send_all(fd, &n, 4) && send_all(fd, json.data(), n);
}

void write_error(Server* s, int index, const char* fmt, ...) {
    char buf[ERR_BYTES];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (n < 0) n = 0;
    if (n >= ERR_BYTES) n = ERR_BYTES - 1;
    memcpy(s->slot(index) + s->lay.err, buf, (size_t)n);
    s->header(index)[H_ERRLEN] = n;
}

void reply(Server* s, int index, char byte) {
    const int fd = s->fd_of_slot[index].load(std::memory_order_relaxed);   // without the poll mutex: a stale fd at worst answers a socket that is closing
    if (fd >= 0) (void)send(fd, &byte, 1, MSG_NOSIGNAL);
}

// ---- the leader: sockets ------------------------------------------------------------------------------------------------
void accept_client(Server* s) {
    const int fd = accept(s->listener, nullptr, nullptr);
    if (fd < 0) return;
    timeval tv{5, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    uint32_t n = 0;
    std::string hello_json;
    if (!recv_all(fd, &n, 4) || n > (1u << 20)) {
        close(fd);
        return;
    }
    hello_json.resize(n);
    if (n && !recv_all(fd, &hello_json[0], n)) {
        close(fd);
        return;
    }
    long protocol = -1;
    const size_t at = hello_json.find("\"protocol\"");
    if (at != std::string::npos) {
        const size_t colon = hello_json.find(':', at);
        if (colon != std::string::npos) protocol = strtol(hello_json.c_str() + colon + 1, nullptr, 10);
    }
    char buf[256];
    if (protocol != HELLO_SITE_PROTOCOL) {
        snprintf(buf, sizeof(buf), "{\"error\": \"protocol %ld != %d\"}", protocol, HELLO_SITE_PROTOCOL);
        send_msg(fd, buf);
        close(fd);
        return;
    }
    int index = -1;
    {
        std::lock_guard<std::mutex> g(s->state_mu);
        if (!s->free_slots.empty()) {
            index = s->free_slots.front();
            s->free_slots.pop_front();
        }
    }
    if (index < 0) {
        snprintf(buf, sizeof(buf), "{\"error\": \"all %d slots are taken\"}", s->cfg.max_clients);
        send_msg(fd, buf);
        close(fd);
        return;
    }
    std::string msg = "{\"protocol\": " + std::to_string(HELLO_SITE_PROTOCOL) + ", \"slot\": " + std::to_string(index) +
                      ", \"slot_bytes\": " + std::to_string((long long)s->cfg.slot_bytes) + ", \"max_clients\": " + std::to_string(s->cfg.max_clients) +
                      ", \"pid\": " + std::to_string((long long)getpid()) + ", \"server\": \"native\", \"shm_path\": \"" + s->shm_path + "\"" +
                      ", \"window\": " + std::to_string(s->cfg.window) + ", \"channels0\": " + std::to_string(s->cfg.channels0) +
                      ", \"channels1\": " + std::to_string(s->cfg.channels1) + ", \"n_experts\": " + std::to_string(s->cfg.n_experts) +
                      ", \"has_meta\": " + (s->cfg.has_meta ? "true" : "false") + ", \"uses_ref\": " + (s->cfg.uses_ref ? "true" : "false");
    if (!s->info_json.empty()) msg += ", " + s->info_json;
    msg += "}";
    if (!send_msg(fd, msg)) {
        close(fd);
        std::lock_guard<std::mutex> g(s->state_mu);
        s->free_slots.push_back(index);
        return;
    }
    tv = timeval{0, 0};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    s->fd_of_slot[index].store(fd);
    std::lock_guard<std::mutex> g(s->state_mu);
    s->n_clients++;
    s->stats.clients_seen++;
}

void drop_client(Server* s, int index) {
    const int fd = s->fd_of_slot[index].load();
    if (fd < 0) return;
    s->fd_of_slot[index].store(-1);
    close(fd);
    for (size_t k = 0; k < s->pending.size(); ++k)          // a dead client's queued site is not scored
        if (s->pending[k] == index) {
            s->pending.erase(s->pending.begin() + (long)k);
            break;
        }
    std::lock_guard<std::mutex> g(s->state_mu);
    s->n_clients--;
    if (s->inflight[index]) s->zombie[index] = 1;            // its slot is reusable -- but not while a launch still reads it
    else s->free_slots.push_back(index);
}

void write_stats(Server* s, int index) {
    hello_site_server_stats st;
    int clients;
    {
        std::lock_guard<std::mutex> g(s->state_mu);
        st = s->stats;
        clients = s->n_clients;
    }
    write_error(s, index, "{\"launches\": %lld, \"sites\": %lld, \"largest_launch\": %d, \"clients_seen\": %d, \"errors\": %lld, \"clients\": %d, "
                          "\"engines\": %d, \"server\": \"native\"}",
                (long long)st.launches, (long long)st.sites, st.largest_launch, st.clients_seen, (long long)st.errors, clients, (int)s->scorers.size());
    reply(s, index, 'K');
}

void poll_once(Server* s, double timeout_s) {
    std::vector<pollfd> fds;
    std::vector<int> who;
    fds.push_back(pollfd{s->listener, POLLIN, 0});
    who.push_back(-1);
    for (int i = 0; i < s->cfg.max_clients; ++i)
        if (s->fd_of_slot[i].load(std::memory_order_relaxed) >= 0) {
            fds.push_back(pollfd{s->fd_of_slot[i].load(std::memory_order_relaxed), POLLIN, 0});
            who.push_back(i);
        }
    timespec ts;
    ts.tv_sec = (time_t)timeout_s;
    ts.tv_nsec = (long)((timeout_s - (double)ts.tv_sec) * 1e9);
    const int n = ppoll(fds.data(), fds.size(), &ts, nullptr);
    if (n <= 0) return;
    for (size_t k = 0; k < fds.size(); ++k) {
        if (!fds[k].revents) continue;
        if (who[k] < 0) {
            accept_client(s);
            continue;
        }
        const int index = who[k];
        char data[64];
        const ssize_t got = recv(fds[k].fd, data, sizeof(data), MSG_DONTWAIT);
        if (got == 0 || (got < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) {
            drop_client(s, index);
        } else if (got > 0) {
            if (memchr(data, 'R', (size_t)got)) s->pending.push_back(index);      // one outstanding request per client
            else if (memchr(data, 'S', (size_t)got)) write_stats(s, index);
        }
    }
}

// Sites one launch takes at most.  With `group_launches` and several engines a launch takes its SHARE of the clients (clients /
// engines), so that the groups run out of phase -- one group's launch is on the GPU while the other group's workers build their
// results and pack their next site -- instead of all workers in lockstep with the GPU idle in between (measured on one MI355X,
// 2 engines: 16 workers 27.1 k -> 31.2 k sites/s, 32 workers 51.2 k -> 57.6 k; profiles/r06_per_site_shared_native.txt).
int launch_cap(Server* s) {
    if (!s->cfg.group_launches || s->scorers.size() < 2) return s->cfg.max_batch_sites;
    int clients;
    {
        std::lock_guard<std::mutex> g(s->state_mu);
        clients = s->n_clients;
    }
    const int share = (clients + (int)s->scorers.size() - 1) / (int)s->scorers.size();
    return share < 1 ? 1 : (share < s->cfg.max_batch_sites ? share : s->cfg.max_batch_sites);
}

// Called with the poll mutex held: wait for requests, linger briefly for the clients that could still send one, return the launch.
std::vector<int> collect(Server* s) {
    double first = -1.0;
    while (!s->stop.load(std::memory_order_relaxed)) {
        if (s->pending.empty()) {
            first = -1.0;
            poll_once(s, 0.25);
            int clients;
            {
                std::lock_guard<std::mutex> g(s->state_mu);
                clients = s->n_clients;
            }
            const double t = now_s();
            if (clients > 0 || !s->pending.empty()) s->idle_since = t;
            else if (s->cfg.idle_exit_s >= 0 && t - s->idle_since > s->cfg.idle_exit_s) s->stop.store(1);
            continue;
        }
        const double t = now_s();
        if (first < 0) first = t;
        int could_still_come;
        {
            std::lock_guard<std::mutex> g(s->state_mu);
            could_still_come = s->n_clients - s->n_inflight - (int)s->pending.size();
        }
        const double left = s->cfg.linger_s - (t - first);
        if (could_still_come <= 0 || left <= 0 || (int)s->pending.size() >= launch_cap(s)) break;
        poll_once(s, left < 50e-6 ? left : 50e-6);
    }
    std::vector<int> take;
    if (s->stop.load()) return take;
    const size_t cap = (size_t)launch_cap(s);
    const size_t n = s->pending.size() < cap ? s->pending.size() : cap;
    take.assign(s->pending.begin(), s->pending.begin() + (long)n);
    s->pending.erase(s->pending.begin(), s->pending.begin() + (long)n);
    std::lock_guard<std::mutex> g(s->state_mu);
    for (int i : take) s->inflight[i] = 1;
    s->n_inflight += (int)take.size();
    return take;
}

// ---- a launch -----------------------------------------------------------------------------------------------------------
// A scorer thread's batch buffers.  With an engine they are pinned, GPU-mapped host memory (hello_pinned_alloc) handed to
// hello_engine_forward as DEVICE pointers: the kernels read the pileups and write logits / meta / posteriors in place across PCIe --
// a launch of a few sites then pays no staging copy and no copy-engine hop in either direction (A/B on one box, 8 / 16 / 32 workers:
// 18.2 -> 18.5, 33.0 -> 33.8, 60.3 -> 61.6 k sites/s; profiles/r06_per_site_shared_native.txt).
struct Block {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool ensure(size_t bytes, bool want_pinned) {
        if (bytes <= cap) return true;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        p = want_pinned ? hello_pinned_alloc(want) : malloc(want);
        pinned = want_pinned;
        cap = p ? want : 0;
        return p != nullptr;
    }
    void release() {
        if (p) {
            if (pinned) hello_pinned_free(p);
            else free(p);
        }
        p = nullptr;
        cap = 0;
    }
    ~Block() { release(); }
};

struct Buffers {
    Block reads0, reads1, ref, logits, meta, post;
    std::vector<int32_t> rpa0, rpa1, aps;
    std::vector<int> kept;
};

void score_batch(Server* s, const Scorer& sc, const std::vector<int>& take, Buffers& b) {
    const auto& lay = s->lay;
    const int64_t rb0 = (int64_t)s->cfg.window * s->cfg.channels0, rb1 = (int64_t)s->cfg.window * s->cfg.channels1;
    const int64_t capacity = s->cfg.slot_bytes - lay.reads;
    b.kept.clear();
    b.rpa0.clear();
    b.rpa1.clear();
    b.aps.clear();
    int second = -1, with_ref = -1;
    int64_t R0 = 0, R1 = 0, A = 0, P = 0;
    for (int index : take) {
        const int32_t* h = s->header(index);
        const int64_t a = h[H_ALLELES], r0 = h[H_READS0], r1 = h[H_READS1];
        const int32_t* t0 = (const int32_t*)(s->slot(index) + lay.rpa0);
        const int32_t* t1 = (const int32_t*)(s->slot(index) + lay.rpa1);
        bool ok = a >= 1 && a <= MAX_ALLELES && r0 >= a && r1 >= 0 && (r1 == 0 || s->cfg.channels1 > 0) && r0 * rb0 + r1 * rb1 <= capacity;
        if (ok) {
            int64_t s0 = 0, s1 = 0;
            for (int k = 0; k < a; ++k) {
                ok = ok && t0[k] >= 1 && (r1 == 0 || t1[k] >= 1);
                s0 += t0[k];
                s1 += t1[k];
            }
            ok = ok && s0 == r0 && (r1 == 0 || s1 == r1);
        }
        // a launch holds sites of one shape of call (every client of a server speaks for the same model): a stray one is refused
        if (ok && second < 0) {
            second = r1 > 0;
            with_ref = h[H_HAS_REF] != 0;
        }
        ok = ok && (r1 > 0) == (second > 0) && (h[H_HAS_REF] != 0) == (with_ref > 0);
        if (!ok) {
            write_error(s, index, "slot %d: the header describes no site that fits the slot and the launch (alleles %lld, reads %lld / %lld; reads per "
                                  "allele must add up, every allele needs a read, optional inputs must match the launch's)",
                        index, (long long)a, (long long)r0, (long long)r1);
            reply(s, index, 'E');
            continue;
        }
        b.kept.push_back(index);
        b.aps.push_back((int32_t)a);
        b.rpa0.insert(b.rpa0.end(), t0, t0 + a);
        if (r1) b.rpa1.insert(b.rpa1.end(), t1, t1 + a);
        R0 += r0;
        R1 += r1;
        A += a;
        P += a * (a + 1) / 2;
    }
    const int S = (int)b.kept.size();
    if (!S) return;
    const int E = s->cfg.n_experts;
    const bool pin = sc.engine != nullptr;
    if (!b.reads0.ensure((size_t)(R0 * rb0), pin) || !b.reads1.ensure((size_t)(R1 * rb1) + 16, pin) || !b.ref.ensure((size_t)S * (size_t)s->cfg.window * 5 + 16, pin) ||
        !b.logits.ensure(sizeof(float) * (size_t)E * (size_t)A, pin) || !b.meta.ensure(sizeof(float) * (size_t)S * 3, pin) ||
        !b.post.ensure(sizeof(float) * 4 * (size_t)P, pin))
        throw std::bad_alloc();
    uint8_t* const reads0 = (uint8_t*)b.reads0.p;
    uint8_t* const reads1 = (uint8_t*)b.reads1.p;
    uint8_t* const ref = (uint8_t*)b.ref.p;
    float* const logits = (float*)b.logits.p;
    float* const meta = (float*)b.meta.p;
    float* const post = (float*)b.post.p;
    {
        size_t at0 = 0, at1 = 0;
        for (int k = 0; k < S; ++k) {
            const int index = b.kept[k];
            const int32_t* h = s->header(index);
            const size_t n0 = (size_t)(h[H_READS0] * rb0), n1 = (size_t)(h[H_READS1] * rb1);
            const unsigned char* src = s->slot(index) + lay.reads;
            memcpy(reads0 + at0, src, n0);
            if (n1) memcpy(reads1 + at1, src + n0, n1);
            at0 += n0;
            at1 += n1;
            if (with_ref > 0) memcpy(ref + (size_t)k * s->cfg.window * 5, s->slot(index) + lay.ref, (size_t)s->cfg.window * 5);
        }
    }
    char err[512] = "";
    int rc;
    if (sc.engine) {
        rc = hello_engine_forward(sc.engine, reads0, b.rpa0.data(), second > 0 ? reads1 : nullptr, second > 0 ? b.rpa1.data() : nullptr, b.aps.data(),
                                  with_ref > 0 ? ref : nullptr, S, (int32_t)A, R0, R1, logits, s->cfg.has_meta ? meta : nullptr, post,
                                  pin ? (HELLO_IN_DEVICE | HELLO_OUT_DEVICE) : 0, nullptr);
        if (!rc && pin) rc = hello_engine_synchronize(sc.engine);     // device-path calls are asynchronous: the outputs are read below
        if (rc) snprintf(err, sizeof(err), "hello_engine_forward: %s (status %d)", hello_last_error(), rc);
    } else {
        rc = sc.fn(sc.ctx, reads0, b.rpa0.data(), second > 0 ? reads1 : nullptr, second > 0 ? b.rpa1.data() : nullptr, b.aps.data(),
                   with_ref > 0 ? ref : nullptr, S, (int32_t)A, R0, R1, logits, s->cfg.has_meta ? meta : nullptr, post, err, (int32_t)sizeof(err));
        err[sizeof(err) - 1] = 0;
        if (rc && !err[0]) snprintf(err, sizeof(err), "the scorer failed with status %d", rc);
    }
    if (rc) {                                    // the whole launch failed: every site of it is answered with the reason
        for (int index : b.kept) {
            write_error(s, index, "%s", err);
            reply(s, index, 'E');
        }
        std::lock_guard<std::mutex> g(s->state_mu);
        s->stats.errors++;
        return;
    }
    int64_t a_off = 0, p_off = 0;
    for (int k = 0; k < S; ++k) {
        const int index = b.kept[k];
        const int a = b.aps[(size_t)k], p = a * (a + 1) / 2;
        float* lg = (float*)(s->slot(index) + lay.logits);
        for (int e = 0; e < E; ++e) memcpy(lg + (size_t)e * MAX_ALLELES, logits + (size_t)e * (size_t)A + (size_t)a_off, sizeof(float) * (size_t)a);
        if (s->cfg.has_meta) memcpy(s->slot(index) + lay.meta, meta + (size_t)k * 3, 12);
        float* po = (float*)(s->slot(index) + lay.post);
        for (int r = 0; r < 4; ++r) memcpy(po + (size_t)r * MAX_PAIRS, post + (size_t)r * (size_t)P + (size_t)p_off, sizeof(float) * (size_t)p);
        a_off += a;
        p_off += p;
    }
    for (int index : b.kept) reply(s, index, 'K');
    std::lock_guard<std::mutex> g(s->state_mu);
    s->stats.launches++;
    s->stats.sites += S;
    if (S > s->stats.largest_launch) s->stats.largest_launch = S;
}

void scorer_loop(Server* s, const Scorer sc) {
    Buffers b;
    while (!s->stop.load(std::memory_order_relaxed)) {
        std::vector<int> take;
        {
            std::lock_guard<std::mutex> g(s->poll_mu);
            if (!s->stop.load()) take = collect(s);
        }
        if (take.empty()) continue;
        try {
            score_batch(s, sc, take, b);
        } catch (...) {                          // host memory exhausted while gathering: the launch's clients hear about it
            for (int index : take) {
                write_error(s, index, "the server ran out of host memory while gathering the launch");
                reply(s, index, 'E');
            }
        }
        std::lock_guard<std::mutex> g(s->state_mu);
        for (int index : take) {
            s->inflight[index] = 0;
            if (s->zombie[index]) {
                s->zombie[index] = 0;
                s->free_slots.push_back(index);
            }
        }
        s->n_inflight -= (int)take.size();
    }
}

int64_t align64(int64_t x) { return (x + 63) & ~(int64_t)63; }

}  // namespace

extern "C" {

int hello_site_slot_layout_of(int32_t window, int32_t channels0, int32_t channels1, int64_t slot_bytes, hello_site_slot_layout* out) {
    if (!out || window <= 0 || channels0 <= 0 || channels1 < 0) return hello::set_last_error(HELLO_ERR_ARG, "slot layout: bad model dimensions");
    int64_t at = 0;
    auto take = [&](int64_t n) {
        const int64_t here = at;
        at = align64(at + n);
        return here;
    };
    out->header = take(4 * HEADER_INTS);
    out->rpa0 = take(4 * MAX_ALLELES);
    out->rpa1 = take(4 * MAX_ALLELES);
    out->ref = take((int64_t)window * 5);
    out->logits = take(4 * 3 * MAX_ALLELES);
    out->meta = take(16);
    out->post = take(16 * MAX_PAIRS);
    out->err = take(ERR_BYTES);
    out->reads = at;
    out->read_capacity = slot_bytes - at;
    if (out->read_capacity < (int64_t)window * channels0)
        return hello::set_last_error(HELLO_ERR_ARG, "slots of %lld bytes cannot hold one read of this model", (long long)slot_bytes);
    return HELLO_OK;
}

int hello_site_server_create(const char* socket_path, const char* shm_path, const hello_site_server_config* cfg, hello_site_server** out) try {
    if (!socket_path || !shm_path || !cfg || !out) return hello::set_last_error(HELLO_ERR_ARG, "site server: NULL argument");
    if (cfg->max_clients < 1 || cfg->max_clients > 1024 || cfg->max_batch_sites < 1 || cfg->n_experts < 1 || cfg->n_experts > 3)
        return hello::set_last_error(HELLO_ERR_ARG, "site server: max_clients in 1..1024, max_batch_sites >= 1, n_experts in 1..3");
    std::unique_ptr<hello_site_server> s(new hello_site_server());
    s->cfg = *cfg;
    s->cfg.info_json = nullptr;
    if (cfg->info_json) s->info_json = cfg->info_json;
    s->socket_path = socket_path;
    s->shm_path = shm_path;
    if (int rc = hello_site_slot_layout_of(cfg->window, cfg->channels0, cfg->channels1, cfg->slot_bytes, &s->lay)) return rc;
    sockaddr_un addr{};
    if (s->socket_path.size() >= sizeof(addr.sun_path)) return hello::set_last_error(HELLO_ERR_ARG, "site server: socket path too long");
    const int fd = open(shm_path, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0) return hello::set_last_error(HELLO_ERR_ARG, "site server: cannot create %s: %s", shm_path, strerror(errno));
    s->map_bytes = (size_t)cfg->max_clients * (size_t)cfg->slot_bytes;
    if (ftruncate(fd, (off_t)s->map_bytes) != 0) {
        close(fd);
        return hello::set_last_error(HELLO_ERR_NOMEM, "site server: cannot size %s to %zu bytes: %s", shm_path, s->map_bytes, strerror(errno));
    }
    void* m = mmap(nullptr, s->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return hello::set_last_error(HELLO_ERR_NOMEM, "site server: cannot map %s: %s", shm_path, strerror(errno));
    s->map = (unsigned char*)m;
    unlink(socket_path);
    s->listener = socket(AF_UNIX, SOCK_STREAM, 0);
    addr.sun_family = AF_UNIX;
    strncpy(addr.sun_path, socket_path, sizeof(addr.sun_path) - 1);
    if (s->listener < 0 || bind(s->listener, (sockaddr*)&addr, sizeof(addr)) != 0 || chmod(socket_path, 0600) != 0 ||
        listen(s->listener, cfg->max_clients) != 0) {
        const int e = errno;
        if (s->listener >= 0) close(s->listener);
        munmap(s->map, s->map_bytes);
        unlink(shm_path);
        return hello::set_last_error(HELLO_ERR_ARG, "site server: cannot listen on %s: %s", socket_path, strerror(e));
    }
    s->fd_of_slot.reset(new std::atomic<int>[(size_t)cfg->max_clients]);
    for (int i = 0; i < cfg->max_clients; ++i) s->fd_of_slot[i].store(-1);
    s->inflight.assign((size_t)cfg->max_clients, 0);
    s->zombie.assign((size_t)cfg->max_clients, 0);
    for (int i = 0; i < cfg->max_clients; ++i) s->free_slots.push_back(i);
    *out = s.release();
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_site_server_create");
}

int hello_site_server_add_engine(hello_site_server* s, hello_engine* engine) try {
    if (!s || !engine) return hello::set_last_error(HELLO_ERR_ARG, "site server: NULL argument");
    Scorer sc;
    sc.engine = engine;
    s->scorers.push_back(sc);
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_site_server_add_engine");
}

int hello_site_server_add_scorer(hello_site_server* s, hello_site_scorer fn, void* ctx) try {
    if (!s || !fn) return hello::set_last_error(HELLO_ERR_ARG, "site server: NULL argument");
    Scorer sc;
    sc.fn = fn;
    sc.ctx = ctx;
    s->scorers.push_back(sc);
    return HELLO_OK;
} catch (...) {
    return hello::exception_status("hello_site_server_add_scorer");
}

int hello_site_server_run(hello_site_server* s) try {
    if (!s) return hello::set_last_error(HELLO_ERR_ARG, "site server: NULL argument");
    if (s->scorers.empty()) return hello::set_last_error(HELLO_ERR_ARG, "site server: no scorer (add an engine first)");
    s->idle_since = now_s();
    std::vector<std::thread> threads;
    for (const Scorer& sc : s->scorers) threads.emplace_back(scorer_loop, s, sc);
    for (auto& t : threads) t.join();
    {
        std::lock_guard<std::mutex> g(s->poll_mu);
        for (int i = 0; i < s->cfg.max_clients; ++i) drop_client(s, i);
    }
    return HELLO_OK;
} catch (...) {
    if (s) s->stop.store(1);
    return hello::exception_status("hello_site_server_run");
}

void hello_site_server_stop(hello_site_server* s) {
    if (s) s->stop.store(1);
}

int hello_site_server_get_stats(hello_site_server* s, hello_site_server_stats* out) {
    if (!s || !out) return hello::set_last_error(HELLO_ERR_ARG, "site server: NULL argument");
    std::lock_guard<std::mutex> g(s->state_mu);
    *out = s->stats;
    return HELLO_OK;
}

void hello_site_server_destroy(hello_site_server* s) {
    if (!s) return;
    s->stop.store(1);
    if (s->listener >= 0) close(s->listener);
    for (int i = 0; i < s->cfg.max_clients && s->fd_of_slot; ++i)
        if (s->fd_of_slot[i].load() >= 0) close(s->fd_of_slot[i].load());
    if (s->map) munmap(s->map, s->map_bytes);
    unlink(s->socket_path.c_str());
    unlink(s->shm_path.c_str());
    delete s;
}

}  // extern "C"
