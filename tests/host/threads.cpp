// The reference's concurrency model through the C ABI from NATIVE threads (no interpreter in the way): T threads share one
// index, each owns a context (`&mut Search`, core/lib.rs:352-356) and issues scalar idist_search_batch(nq = 1) calls.
// Every call must return exactly what one wide batch call returns for that query (ids, distance bits, count); prints the
// aggregate calls/s.  usage: threads <n> <dim> <threads> <calls_per_thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/idist.h"

#define CK(x)                                                                      \
    do {                                                                           \
        idist_status _s = (x);                                                     \
        if (_s != IDIST_OK) { fprintf(stderr, "%s: status %d: %s\n", #x, _s, idist_last_error()); exit(2); } \
    } while (0)

int main(int argc, char** argv) {
    // one hardware queue per searching thread's stream: the host's setting, made before the first HIP call
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000, dim = argc > 2 ? (uint32_t)atoi(argv[2]) : 96;
    const int T = argc > 3 ? atoi(argv[3]) : 16, calls = argc > 4 ? atoi(argv[4]) : 100;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    std::vector<float> pts((size_t)n * dim), q((size_t)T * calls * dim);
    for (auto& x : pts) x = u(rng);
    for (auto& x : q) x = u(rng);
    idist_config cfg;
    CK(idist_default_config(&cfg));
    idist_index* idx = nullptr;
    CK(idist_index_build(pts.data(), n, dim, &cfg, 0, &idx));
    const uint32_t ef = cfg.ef_search, nq = (uint32_t)(T * calls);
    std::vector<uint32_t> want_pid((size_t)nq * ef), want_cnt(nq);
    std::vector<float> want_dist((size_t)nq * ef);
    {
        idist_search_ctx* c = nullptr;
        CK(idist_search_ctx_new(idx, 0, &c));
        CK(idist_search_batch(idx, c, q.data(), nq, want_pid.data(), want_dist.data(), want_cnt.data(), nullptr));
        idist_search_ctx_free(c);
    }
    std::vector<int> bad(T, 0);
    std::vector<std::thread> ts;
    std::atomic<int> ready{0}, ready2{0};
    std::atomic<bool> warm{false}, go{false};
    for (int t = 0; t < T; t++)
        ts.emplace_back([&, t] {
            idist_search_ctx* c = nullptr;
            if (idist_search_ctx_new(idx, 1, &c) != IDIST_OK) { bad[t] = 1 << 20; ready++; return; }
            std::vector<uint32_t> pid(ef);
            std::vector<float> dist(ef);
            uint32_t cnt = 0;
            // Search::default() and the first call (the context's buffers come into being) are not what is timed
            if (idist_search_batch(idx, c, q.data(), 1, pid.data(), dist.data(), &cnt, nullptr) != IDIST_OK) bad[t]++;
            ready++;
            while (!warm.load()) std::this_thread::yield();
            for (int i = 0; i < 20; i++)                                   // untimed: the contexts behind combined launches too
                if (idist_search_batch(idx, c, q.data() + ((size_t)t * calls + i % calls) * dim, 1, pid.data(), dist.data(), &cnt, nullptr) != IDIST_OK) bad[t]++;
            ready2++;
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < calls; i++) {
                const size_t j = (size_t)t * calls + i;
                if (idist_search_batch(idx, c, q.data() + j * dim, 1, pid.data(), dist.data(), &cnt, nullptr) != IDIST_OK) { bad[t]++; continue; }
                if (cnt != want_cnt[j] || memcmp(pid.data(), want_pid.data() + j * ef, (size_t)ef * 4) ||
                    memcmp(dist.data(), want_dist.data() + j * ef, (size_t)ef * 4))
                    bad[t]++;
            }
            idist_search_ctx_free(c);
        });
    while (ready.load() < T) std::this_thread::yield();
    warm = true;
    while (ready2.load() < T) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go = true;
    for (auto& th : ts) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int nbad = 0;
    for (int b : bad) nbad += b;
    printf("threads %d calls %u calls_per_s %.0f mismatches %d\n", T, nq, nq / dt, nbad);
    idist_index_free(idx);
    if (nbad) return 1;
    printf("threads ok\n");
    return 0;
}
