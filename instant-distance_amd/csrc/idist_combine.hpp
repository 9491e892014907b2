// idist_combine.hpp — scalar calls from many threads, combined into few launches (host only, no HIP).
//
// The reference's concurrency model is one `Search` per thread issuing `Hnsw::search` calls on a shared index
// (core/lib.rs:352-356).  On the GPU every such call is a launch, and launches are what saturates first: the kernels of 16
// threads overlap (0.54 ms each, one workgroup each) but ~65 us per launch are serialised inside the HIP runtime — 15k
// calls/s however many threads there are.  Queries, though, batch for free: 64 one-query workgroups in ONE launch take
// what one takes.  So calls that arrive while `max_leaders` launches are already in flight do not launch at all: they queue
// up, and the next thread that may launch — a *leader* — takes everything that is waiting (up to `max_batch`) with it and
// hands each caller its own results.  Every leader holds one of `max_leaders` numbered slots while it runs (the caller keeps
// per-slot resources there: a context wide enough for a batch, whatever the leader's own `Search` was sized for).  Up to
// `max_leaders` threads the behaviour is what it was (every call its own launch on its own stream); beyond, throughput
// grows with the batch width instead of stalling.
//
//   Req        what one call brings (query pointer, result pointers) + done / lead flags + slot; lives on the caller's stack.
//   run(b, s)  executes a batch b (b[0] is the leader's own request) on slot s and fills every request's status; unlocked.
//
// Progress: a leader serves exactly one batch, its own request included, so no thread works for others longer than one
// launch; when it is done and calls are still waiting, it promotes the oldest of them to lead next (so a queue can never
// be left without a leader).  Every request is served exactly once (tests/host/combine_test.cpp hammers this on the CPU).
#pragma once
#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <vector>

namespace idist {

template <class Req>
class Combiner {
public:
    explicit Combiner(unsigned max_leaders = 8, size_t max_batch = 96) : max_leaders_(max_leaders), max_batch_(max_batch) {
        for (unsigned i = max_leaders; i > 0; i--) free_slots_.push_back((int)i - 1);
    }
    unsigned max_leaders() const { return max_leaders_; }

    // Req needs: bool done, lead (both false on entry), int slot, std::condition_variable cv, void fail() (called instead of a
    // result when run() threw: nothing propagates out of submit, which sits under a C ABI).  Returns after r.done.
    template <class Run>
    void submit(Req& r, Run&& run) {
        std::unique_lock<std::mutex> lk(mu_);
        int slot;
        if (active_ >= max_leaders_) {
            pending_.push_back(&r);
            r.cv.wait(lk, [&] { return r.done || r.lead; });
            if (r.done) return;
            slot = r.slot;          // promoted: the slot of the leader that promoted us is ours (active_ was not decremented)
        } else {
            active_++;
            slot = free_slots_.back();
            free_slots_.pop_back();
        }
        std::vector<Req*> batch;
        batch.push_back(&r);
        take_pending(batch);
        lk.unlock();
        // run() may throw (it allocates staging buffers): the riders of this batch are parked on their condition variables
        // and the slot is ours — whatever happens they are woken with a status and the slot moves on
        bool threw = false;
        try {
            run(batch, slot);
        } catch (...) {
            threw = true;
        }
        lk.lock();
        for (Req* q : batch) {
            if (threw) q->fail();
            q->done = true;
            if (q != &r) q->cv.notify_one();
        }
        // hand the slot on, or give it back
        Req* next = nullptr;
        for (size_t i = 0; i < pending_.size(); i++)
            if (!pending_[i]->lead) { next = pending_[i]; pending_.erase(pending_.begin() + (std::ptrdiff_t)i); break; }
        if (next) {
            next->slot = slot;
            next->lead = true;
            next->cv.notify_one();
        } else {
            active_--;
            free_slots_.push_back(slot);
        }
    }

private:
    void take_pending(std::vector<Req*>& batch) {
        size_t k = 0;
        while (k < pending_.size() && batch.size() < max_batch_) {
            if (pending_[k]->lead) { k++; continue; }              // (cannot happen: promoted requests leave the queue)
            batch.push_back(pending_[k]);
            pending_.erase(pending_.begin() + (std::ptrdiff_t)k);
        }
    }
    std::mutex mu_;
    std::vector<Req*> pending_;
    std::vector<int> free_slots_;
    unsigned active_ = 0;
    const unsigned max_leaders_;
    const size_t max_batch_;
};

}  // namespace idist
