// ThreadSanitizer harness of the shared scoring server (hello_amd/csrc/site_server.hip), without Python and without a GPU.
//
// tests/test_shared_server.py compiles the server's source as plain host C++ together with this file under -fsanitize=thread
// (the server's only dependencies inside the library -- the thread-local error message and hello_engine_forward -- are given
// trivially here; the harness scores through the ABI's scorer callback, never through an engine), then runs it: N client THREADS
// speak the wire protocol of hello_amd/shared.py (handshake, slot of the shared segment, one byte each way per site) against a
// server with several scorer threads, some clients disconnecting in the middle of their run and reconnecting.  Every answer is
// checked against the scorer's definition; the process must exit 0 with no ThreadSanitizer report.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include "hello_mi355x.h"

// ---- what site_server.hip takes from the rest of the library -----------------------------------------------------------------
namespace hello {
static thread_local std::string g_error;
int set_last_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
int exception_status(const char* where) noexcept {
    g_error = where;
    return HELLO_ERR_NOMEM;
}
}  // namespace hello
extern "C" const char* hello_last_error(void) { return hello::g_error.c_str(); }
extern "C" int hello_engine_forward(hello_engine*, const uint8_t*, const int32_t*, const uint8_t*, const int32_t*, const int32_t*, const uint8_t*, int32_t,
                                    int32_t, int64_t, int64_t, float*, float*, float*, int32_t, void*) {
    return HELLO_ERR_NOGPU;                       // the harness never adds an engine
}
extern "C" int hello_engine_synchronize(hello_engine*) { return HELLO_ERR_NOGPU; }
extern "C" void* hello_pinned_alloc(size_t) { return nullptr; }      // (only engine scorers ask for pinned blocks)
extern "C" void hello_pinned_free(void*) {}

namespace {

constexpr int WINDOW = 150, C0 = 6, ROW = WINDOW * C0, MAXA = HELLO_SITE_MAX_ALLELES, MAXP = MAXA * (MAXA + 1) / 2;

// the scorer: every output of a site is a function of that site's bytes alone
float allele_value(const uint8_t* reads, int n_reads) {
    long long sum = 0;
    for (long long i = 0; i < (long long)n_reads * ROW; ++i) sum += reads[i];
    return (float)(sum % 9973) / 1000.f - 5.f;
}

int scorer(void*, const uint8_t* reads0, const int32_t* rpa0, const uint8_t*, const int32_t*, const int32_t* aps, const uint8_t*, int32_t S, int32_t A,
           int64_t, int64_t, float* logits, float*, float* post, char*, int32_t) {
    std::vector<float> v((size_t)A);
    long long r = 0;
    for (int a = 0; a < A; ++a) {
        v[(size_t)a] = allele_value(reads0 + r * ROW, rpa0[a]);
        r += rpa0[a];
        logits[a] = v[(size_t)a];
    }
    long long P = 0;
    for (int s = 0; s < S; ++s) P += (long long)aps[s] * (aps[s] + 1) / 2;
    long long a0 = 0, p0 = 0;
    for (int s = 0; s < S; ++s) {
        int k = 0;
        for (int i = 0; i < aps[s]; ++i)
            for (int j = i; j < aps[s]; ++j, ++k)
                for (int row = 0; row < 4; ++row) post[row * P + p0 + k] = (float)row + v[(size_t)(a0 + i)] + 2.f * v[(size_t)(a0 + j)];
        a0 += aps[s];
        p0 += k;
    }
    usleep(200);                                  // a launch takes a while: requests pile up behind it
    return 0;
}

bool xfer(int fd, void* p, size_t n, bool out) {
    char* c = (char*)p;
    while (n) {
        const ssize_t k = out ? send(fd, c, n, MSG_NOSIGNAL) : recv(fd, c, n, 0);
        if (k <= 0) return false;
        c += k;
        n -= (size_t)k;
    }
    return true;
}

long long json_int(const std::string& j, const char* key) {
    const size_t at = j.find(std::string("\"") + key + "\"");
    if (at == std::string::npos) return -1;
    return strtoll(j.c_str() + j.find(':', at) + 1, nullptr, 10);
}

std::atomic<int> failures{0};

void client(const char* sock_path, const char* shm_path, int64_t slot_bytes, int max_clients, hello_site_slot_layout lay, int id, int n_calls) {
    unsigned seed = 1234u + (unsigned)id;
    int done = 0;
    while (done < n_calls) {
        const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
        sockaddr_un addr{};
        addr.sun_family = AF_UNIX;
        strncpy(addr.sun_path, sock_path, sizeof(addr.sun_path) - 1);
        if (connect(fd, (sockaddr*)&addr, sizeof(addr)) != 0) {
            close(fd);
            usleep(1000);
            continue;
        }
        const std::string hello_msg = "{\"protocol\": 1, \"pid\": 0}";
        uint32_t n = (uint32_t)hello_msg.size();
        std::string reply;
        if (!xfer(fd, &n, 4, true) || !xfer(fd, (void*)hello_msg.data(), n, true) || !xfer(fd, &n, 4, false)) {
            close(fd);
            continue;
        }
        reply.resize(n);
        if (!xfer(fd, &reply[0], n, false)) {
            close(fd);
            continue;
        }
        const long long slot = json_int(reply, "slot");
        if (slot < 0) {                            // all slots taken for a moment (a reconnecting neighbour's slot is not free yet)
            close(fd);
            usleep(500);
            continue;
        }
        const int shm = open(shm_path, O_RDWR);
        unsigned char* map = (unsigned char*)mmap(nullptr, (size_t)max_clients * (size_t)slot_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, shm, 0);
        close(shm);
        unsigned char* base = map + (size_t)slot * (size_t)slot_bytes;
        const int burst = 20 + (int)(rand_r(&seed) % 40);        // calls before this client hangs up and comes back
        for (int c = 0; c < burst && done < n_calls; ++c, ++done) {
            const int A = 1 + (int)(rand_r(&seed) % 4);
            int32_t* h = (int32_t*)(base + lay.header);
            int32_t* rpa = (int32_t*)(base + lay.rpa0);
            int R = 0;
            for (int a = 0; a < A; ++a) R += (rpa[a] = 1 + (int)(rand_r(&seed) % 12));
            unsigned char* reads = base + lay.reads;
            for (long long i = 0; i < (long long)R * ROW; ++i) reads[i] = (unsigned char)(rand_r(&seed) & 255);
            h[0] = A, h[1] = R, h[2] = 0, h[3] = 0, h[4] = A * (A + 1) / 2;
            char byte = 'R';
            if (!xfer(fd, &byte, 1, true) || !xfer(fd, &byte, 1, false) || byte != 'K') {
                fprintf(stderr, "client %d: call %d got no answer / was refused\n", id, done);
                failures++;
                break;
            }
            // check against the scorer's definition, computed here from what this client wrote
            std::vector<float> v((size_t)A);
            long long r = 0;
            for (int a = 0; a < A; ++a) {
                v[(size_t)a] = allele_value(reads + r * ROW, rpa[a]);
                r += rpa[a];
            }
            const float* lg = (const float*)(base + lay.logits);
            const float* po = (const float*)(base + lay.post);
            bool ok = true;
            for (int a = 0; a < A; ++a) ok = ok && lg[a] == v[(size_t)a];
            int k = 0;
            for (int i = 0; i < A; ++i)
                for (int j = i; j < A; ++j, ++k)
                    for (int row = 0; row < 4; ++row) ok = ok && po[row * MAXP + k] == (float)row + v[(size_t)i] + 2.f * v[(size_t)j];
            if (!ok) {
                fprintf(stderr, "client %d: call %d received another site's answer\n", id, done);
                failures++;
            }
        }
        munmap(map, (size_t)max_clients * (size_t)slot_bytes);
        close(fd);                                 // hang up (sometimes with nothing in flight, sometimes right after an answer)
    }
}

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const std::string sock = dir + "/tsan.sock", shm = dir + "/tsan.slots";
    hello_site_server_config cfg{};
    cfg.window = WINDOW, cfg.channels0 = C0, cfg.channels1 = 0, cfg.n_experts = 1, cfg.has_meta = 0, cfg.uses_ref = 0;
    cfg.max_clients = 12, cfg.max_batch_sites = 64, cfg.group_launches = 1, cfg.slot_bytes = 1 << 18, cfg.idle_exit_s = -1.0, cfg.linger_s = 100e-6;
    hello_site_server* server = nullptr;
    if (hello_site_server_create(sock.c_str(), shm.c_str(), &cfg, &server) != 0) {
        fprintf(stderr, "create: %s\n", hello_last_error());
        return 2;
    }
    for (int k = 0; k < 3; ++k) hello_site_server_add_scorer(server, scorer, nullptr);
    hello_site_slot_layout lay;
    hello_site_slot_layout_of(WINDOW, C0, 0, cfg.slot_bytes, &lay);
    std::thread serving([&] { hello_site_server_run(server); });
    std::vector<std::thread> clients;
    const int n_clients = 10, n_calls = 300;
    for (int i = 0; i < n_clients; ++i) clients.emplace_back(client, sock.c_str(), shm.c_str(), cfg.slot_bytes, cfg.max_clients, lay, i, n_calls);
    for (auto& t : clients) t.join();
    hello_site_server_stats st;
    hello_site_server_get_stats(server, &st);
    hello_site_server_stop(server);
    serving.join();
    hello_site_server_destroy(server);
    printf("sites %lld launches %lld largest %d clients_seen %d errors %lld failures %d\n", (long long)st.sites, (long long)st.launches, st.largest_launch,
           st.clients_seen, (long long)st.errors, failures.load());
    return (failures.load() == 0 && st.sites == (long long)n_clients * n_calls && st.errors == 0 && st.launches < st.sites) ? 0 : 1;
}
