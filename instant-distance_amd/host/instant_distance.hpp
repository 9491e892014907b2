// instant_distance.hpp — C++17 host mirror of instant-distance's public Rust API over the
// C ABI of libidist.so (include/idist.h).  Header only; link with -lidist.
//
//   Builder   core/lib.rs:23-113      Hnsw<P>     core/lib.rs:194-397
//   Heuristic core/lib.rs:115-128     HnswMap<P,V> core/lib.rs:131-173
//   Search    core/lib.rs:560-574     Item / MapItem core/lib.rs:399-413 / 175-191
//   PointId   core/types.rs:241-253   Point       core/lib.rs:780-782
//
// (paths relative to the reference checkout, core/ = instant-distance/src/).
//
// `trait Point { fn distance(&self, &Self) -> f32 }` is arbitrary user code in the reference and
// cannot be shipped to a GPU.  Here a Point type exposes its coordinates as f32 and names one of
// the two distances the reference itself ships:
//     static constexpr int METRIC = IDIST_METRIC_L2SQ;   // FloatArray, py/lib.rs:378-421
//     static constexpr int METRIC = IDIST_METRIC_L2;     // tests/all.rs:93-97, examples/colors.rs
//     size_t dim() const;  void write_f32(float* out) const;
// A Point without that interface is a compile-time error — there is no CPU fallback.
// Errors: the reference API is infallible (it panics at core/lib.rs:256 and :148); every non-OK
// idist_status is thrown as instant_distance::Error (the shim's `expect()`).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <exception>
#include <functional>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/idist.h"

namespace instant_distance {

struct Error : std::runtime_error {
    idist_status status;
    Error(idist_status s, const std::string& m) : std::runtime_error("idist status " + std::to_string(s) + ": " + m), status(s) {}
};
inline void check(idist_status s) {
    if (s != IDIST_OK) throw Error(s, idist_last_error());
}

// core/types.rs:241-253
struct PointId {
    uint32_t v = IDIST_INVALID;
    bool is_valid() const { return v != IDIST_INVALID; }
    uint32_t into_inner() const { return v; }
    bool operator==(const PointId& o) const { return v == o.v; }
    bool operator<(const PointId& o) const { return v < o.v; }
};

// core/lib.rs:115-128
struct Heuristic {
    bool extend_candidates = false;
    bool keep_pruned = true;
};

template <class P> class Hnsw;
template <class P, class V> class HnswMap;

// core/lib.rs:560-574, 767-778: reusable scratch; holds the results of the last search
class Search {
public:
    Search() = default;
    Search(const Search&) = delete;
    Search& operator=(const Search&) = delete;
    ~Search() { release(); }
    size_t len() const { return pid_.size(); }

private:
    template <class P> friend class Hnsw;
    template <class P, class V> friend class HnswMap;
    // The context belongs to ONE Hnsw object, recognised by its uid — an address alone would accept a new index that
    // happens to live where a freed one did.  One query slot: Search::default() per thread must cost kilobytes.
    idist_search_ctx* bind(const idist_index* idx, uint64_t uid) {
        if (owner_uid_ != uid || !ctx_) {
            release();
            check(idist_search_ctx_new(idx, 1, &ctx_));
            owner_uid_ = uid;
        }
        return ctx_;
    }
    void release() {
        if (ctx_) idist_search_ctx_free(ctx_);
        ctx_ = nullptr;
        owner_uid_ = 0;
    }
    idist_search_ctx* ctx_ = nullptr;
    uint64_t owner_uid_ = 0;
    std::vector<uint32_t> pid_;
    std::vector<float> dist_;
};

// core/lib.rs:399-413
template <class P> struct Item {
    float distance;
    PointId pid;
    const P* point;
};
// core/lib.rs:175-191
template <class P, class V> struct MapItem {
    float distance;
    PointId pid;
    const P* point;
    const V* value;
};

// core/lib.rs:23-113
class Builder {
public:
    Builder() {
        check(idist_default_config(&cfg_));
        seed_ = (uint64_t(std::random_device{}()) << 32) ^ std::random_device{}();   // rand::random(), :108
    }
    static Builder default_() { return Builder(); }
    Builder ef_construction(size_t ef) && { cfg_.ef_construction = (uint32_t)ef; return std::move(*this); }   // :35-38
    Builder ef_search(size_t ef) && { cfg_.ef_search = (uint32_t)ef; return std::move(*this); }               // :44-47
    Builder select_heuristic(const Heuristic* h) && {                                                          // :49-52 (nullptr = None)
        cfg_.has_heuristic = h ? 1 : 0;
        if (h) { cfg_.extend_candidates = h->extend_candidates; cfg_.keep_pruned = h->keep_pruned; }
        return std::move(*this);
    }
    Builder ml(float ml) && { cfg_.ml = ml; return std::move(*this); }                                          // :57-60
    Builder seed(uint64_t s) && { seed_ = s; return std::move(*this); }                                         // :65-68
    // :70-75 (`indicatif` feature): bar(done, total, layer) is called from a watcher thread while the build runs
    // and once more when it has finished; layer < 0 outside the per-layer loop
    Builder progress(std::function<void(uint64_t, uint64_t, int)> bar) && { progress_ = std::move(bar); return std::move(*this); }
    // engine knobs (not in the reference)
    Builder max_batch(uint32_t k) && { cfg_.max_batch = k; return std::move(*this); }
    Builder tie_policy(int32_t p) && { cfg_.tie_policy = p; return std::move(*this); }   // IDIST_TIES_STRICT / IDIST_TIES_DROP
    Builder tie_capacity(uint32_t n) && { cfg_.tie_capacity = n; return std::move(*this); }
    Builder device(int d) && { device_ = d; return std::move(*this); }

    template <class P, class V> HnswMap<P, V> build(std::vector<P> points, std::vector<V> values) && {         // :78-80
        return HnswMap<P, V>(std::move(points), std::move(values), *this);
    }
    template <class P> std::pair<Hnsw<P>, std::vector<PointId>> build_hnsw(std::vector<P> points) && {         // :83-85
        std::vector<PointId> ids;
        Hnsw<P> h(std::move(points), *this, &ids);
        return {std::move(h), std::move(ids)};
    }

private:
    template <class P> friend class Hnsw;
    idist_config cfg_{};
    uint64_t seed_ = 0;
    int device_ = 0;
    std::function<void(uint64_t, uint64_t, int)> progress_;
};

// polls an idist_progress from a second thread while the blocking build call runs on the caller's
class BuildWatch {
public:
    explicit BuildWatch(const std::function<void(uint64_t, uint64_t, int)>& bar) : bar_(bar) {
        if (!bar_) return;
        check(idist_progress_new(&p_));
        check(idist_progress_watch_next_build(p_));
        thr_ = std::thread([this] {
            uint64_t last = ~0ull;
            while (!stop_.load()) {
                uint64_t done = 0, total = 0;
                int32_t layer = -1;
                idist_progress_get(p_, &done, &total, &layer);
                if (total && done != last) { bar_(done, total, layer); last = done; }
                std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
        });
    }
    ~BuildWatch() {
        if (!p_) return;
        stop_.store(true);
        thr_.join();
        idist_progress_watch_next_build(nullptr);
        uint64_t done = 0, total = 0;
        int32_t layer = -1;
        idist_progress_get(p_, &done, &total, &layer);
        if (!std::uncaught_exceptions()) bar_(done, total, layer);   // bar.finish()
        idist_progress_free(p_);
    }
private:
    std::function<void(uint64_t, uint64_t, int)> bar_;
    idist_progress* p_ = nullptr;
    std::atomic<bool> stop_{false};
    std::thread thr_;
};

// core/lib.rs:194-397
template <class P> class Hnsw {
public:
    Hnsw(std::vector<P> points, const Builder& b, std::vector<PointId>* out_ids) {                              // Hnsw::new, :209-345
        const uint32_t n = (uint32_t)points.size();
        std::vector<uint32_t> out(n ? n : 1), order(n ? n : 1);
        check(idist_permutation(b.seed_, n, out.data(), order.data()));                                         // :257-270
        points_.reserve(n);
        for (uint32_t i = 0; i < n; i++) points_.push_back(points[order[i]]);
        const uint32_t dim = n ? (uint32_t)points_[0].dim() : 1;
        std::vector<float> flat((size_t)n * dim);
        for (uint32_t i = 0; i < n; i++) points_[i].write_f32(flat.data() + (size_t)i * dim);
        idist_config cfg = b.cfg_;
        cfg.metric = P::METRIC;
        ef_search_ = cfg.ef_search;
        {
            BuildWatch watch(b.progress_);
            check(idist_index_build(flat.data(), n, dim, &cfg, b.device_, &idx_));
        }
        if (out_ids) {
            out_ids->resize(n);
            for (uint32_t i = 0; i < n; i++) (*out_ids)[i] = PointId{out[i]};
        }
    }
    Hnsw(Hnsw&& o) noexcept : idx_(o.idx_), uid_(o.uid_), points_(std::move(o.points_)), ef_search_(o.ef_search_) { o.idx_ = nullptr; }
    Hnsw(const Hnsw&) = delete;
    ~Hnsw() { idist_index_free(idx_); }

    static Builder builder() { return Builder(); }                                                              // :205-207

    // Hnsw::search, :352-383: <= ef_search items, nearest first
    std::vector<Item<P>> search(const P& point, Search& search) const {
        run(point, search);
        std::vector<Item<P>> out;
        for (size_t i = 0; i < search.pid_.size(); i++)
            out.push_back(Item<P>{search.dist_[i], PointId{search.pid_[i]}, &points_[search.pid_[i]]});
        return out;
    }
    const P& operator[](PointId pid) const { return points_[pid.v]; }                                           // Index<PointId>
    size_t len() const { return points_.size(); }
    const std::vector<P>& points() const { return points_; }

private:
    template <class Q, class V> friend class HnswMap;
    void run(const P& point, Search& s) const {
        const uint32_t ef = ef_search_;
        std::vector<float> q(points_.empty() ? 1 : points_[0].dim());
        point.write_f32(q.data());
        std::vector<uint32_t> pid(ef ? ef : 1);
        std::vector<float> dist(ef ? ef : 1);
        uint32_t cnt = 0;
        check(idist_search_batch(idx_, s.bind(idx_, uid_), q.data(), 1, pid.data(), dist.data(), &cnt, nullptr));
        s.pid_.assign(pid.begin(), pid.begin() + cnt);
        s.dist_.assign(dist.begin(), dist.begin() + cnt);
    }
    idist_index* idx_ = nullptr;
    static uint64_t next_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
    uint64_t uid_ = next_uid();        // identity of this device index for the Search objects bound to it
    std::vector<P> points_;
    uint32_t ef_search_ = 100;
};

// core/lib.rs:131-173
template <class P, class V> class HnswMap {
public:
    HnswMap(std::vector<P> points, std::vector<V> values, const Builder& b) : hnsw_(std::move(points), b, &ids_) {
        // values re-ordered by PointId, :144-149 (values.at() mirrors the panic at :148)
        this->values.resize(ids_.size());
        for (size_t src = 0; src < ids_.size(); src++) this->values[ids_[src].v] = values.at(src);
    }
    std::vector<MapItem<P, V>> search(const P& point, Search& search) const {                                  // :154-162
        std::vector<MapItem<P, V>> out;
        for (auto& it : hnsw_.search(point, search)) out.push_back(MapItem<P, V>{it.distance, it.pid, it.point, &values[it.pid.v]});
        return out;
    }
    std::vector<V> values;

private:
    std::vector<PointId> ids_;
    Hnsw<P> hnsw_;
};

}  // namespace instant_distance
