// CPU test of idist::Combiner (instant-distance_amd/csrc/idist_combine.hpp): T threads hammer scalar requests through it,
// a fake "launch" (a sleep, like a kernel) serves each batch.  Checks: every request served exactly once with its own
// answer, a leader's batch starts with its own request, never more than max_leaders launches at a time, batches form
// once the threads outnumber the leaders, nothing is left waiting (no deadlock: the program ends).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

#include "../../instant-distance_amd/csrc/idist_combine.hpp"

struct Req {
    int q = 0, out = -1, served = 0, slot = -1;
    bool done = false, lead = false, failed = false;
    std::condition_variable cv;
    void fail() { failed = true; }
};

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 32, calls = argc > 2 ? atoi(argv[2]) : 200;
    const unsigned max_leaders = argc > 3 ? (unsigned)atoi(argv[3]) : 4;
    const size_t max_batch = argc > 4 ? (size_t)atoi(argv[4]) : 6;
    idist::Combiner<Req> comb(max_leaders, max_batch);
    // 6th argument: every k-th launch throws (std::bad_alloc in the real run()): its whole batch must come back failed, nobody
    // may be left waiting, the slot must move on
    const int throw_every = argc > 5 ? atoi(argv[5]) : 0;
    std::atomic<int> in_flight{0}, max_in_flight{0}, launches{0}, bad{0};
    std::atomic<long> failed{0};
    std::atomic<long> served{0}, widest{0};
    std::vector<std::atomic<int>> slot_busy(max_leaders);
    for (auto& x : slot_busy) x = 0;
    auto worker = [&](int t) {
        for (int i = 0; i < calls; i++) {
            Req r;
            r.q = t * 100000 + i;
            comb.submit(r, [&](std::vector<Req*>& b, int slot) {
                if (b[0] != &r) bad++;                                    // a leader's own request leads its batch
                if (slot < 0 || slot >= (int)max_leaders || slot_busy[slot].exchange(1)) bad++;   // a slot serves one leader at a time
                if (b.size() > max_batch) bad++;
                const int now = ++in_flight;
                int m = max_in_flight.load();
                while (now > m && !max_in_flight.compare_exchange_weak(m, now)) {}
                const int nth = ++launches;
                if (throw_every && nth % throw_every == 0) {
                    slot_busy[slot] = 0;
                    --in_flight;
                    throw std::bad_alloc();
                }
                long w = widest.load();
                while ((long)b.size() > w && !widest.compare_exchange_weak(w, (long)b.size())) {}
                std::this_thread::sleep_for(std::chrono::microseconds(200 + (r.q % 7) * 20));
                for (Req* x : b) { x->out = x->q * 2 + 1; x->served++; }
                served += (long)b.size();
                slot_busy[slot] = 0;
                --in_flight;
            });
            if (r.failed) { failed++; if (!r.done || r.served != 0) bad++; }
            else if (!r.done || r.served != 1 || r.out != r.q * 2 + 1) bad++;
        }
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < T; t++) ts.emplace_back(worker, t);
    for (auto& t : ts) t.join();
    const long total = (long)T * calls;
    printf("threads %d calls %ld served %ld launches %d max_in_flight %d widest_batch %ld bad %d\n", T, total, served.load(), launches.load(),
           max_in_flight.load(), widest.load(), bad.load());
    printf("failed (thrown launches) %ld\n", failed.load());
    if (bad || served + failed != total || (throw_every == 0 && failed) || (throw_every && !failed)) return 1;
    if (max_in_flight > (int)max_leaders) return 2;
    if (T > 2 * (int)max_leaders && launches >= total) return 3;        // with more threads than leaders, calls must combine
    if (T <= (int)max_leaders && launches != total) return 4;           // with no more threads than leaders, every call launches itself
    printf("combiner ok\n");
    return 0;
}
