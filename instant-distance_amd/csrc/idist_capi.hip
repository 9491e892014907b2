// idist_capi.hip — C ABI (include/idist.h) over the gfx950 kernels.
// Host side only orchestrates: allocation, layout conversion, launch schedule
// of the build (Hnsw::new, core/lib.rs:209-345) and of the batched search.
#include "../../include/idist.h"
#include "idist_kernels.hpp"
#include "idist_mfma.hpp"
#include "idist_combine.hpp"

#ifndef IDIST_EMU
#include <hip/hip_runtime.h>
#define IDIST_LAUNCH(kfn, grid, block, smem, stream, ...) kfn<<<(grid), (block), (smem), (stream)>>>(__VA_ARGS__)
#else
#define IDIST_LAUNCH(kfn, grid, block, smem, stream, ...) \
    ::emu::launch((grid), (block), (smem), [=]() { kfn(__VA_ARGS__); })
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <time.h>
#include <vector>

using namespace idist;

// (The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues — default 4, read once when the
// runtime starts.  One `Search` per host thread = one stream per thread: a host that runs more than four of them sets
// GPU_MAX_HW_QUEUES itself before its first HIP call; this library does not touch the environment.)

namespace {

// Environment knobs.  The product library (libidist.so) reads three — IDIST_COMBINE, IDIST_SYNC, IDIST_KERNEL_EVENTS: host-side
// behaviour an integrator may want to switch.  Every other IDIST_* knob exists for tests and A/B measurements and is compiled
// into the test build (libidist_variants.so, -DIDIST_VARIANTS), the measurement build (-DIDIST_PROBE) and the CPU emulator only:
// in the product test_env() is a constant nullptr, so the branches behind it fold away and no environment variable can change
// which kernels run or which graph is built.
#if defined(IDIST_VARIANTS) || defined(IDIST_PROBE) || defined(IDIST_EMU)
inline const char* test_env(const char* name) { return getenv(name); }
inline void warn_ignored_knobs() {}
#else
constexpr const char* test_env(const char*) { return nullptr; }
// The product ignores every other IDIST_* variable: say so once (a script written against the test build would otherwise measure
// the default behaviour without a hint).  The names are not known here — anything with the prefix that is not one of the three.
extern "C" char** environ;
inline void warn_ignored_knobs() {
    static std::once_flag once;
    std::call_once(once, [] {
        for (char** e = environ; e && *e; e++) {
            if (strncmp(*e, "IDIST_", 6) != 0) continue;
            const char* eq = strchr(*e, '=');
            const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
            if (name == "IDIST_COMBINE" || name == "IDIST_SYNC" || name == "IDIST_KERNEL_EVENTS") continue;
            fprintf(stderr, "libidist: %s is ignored by the product library (test knobs exist in libidist_variants.so only)\n", name.c_str());
        }
    });
}
#endif

char g_err_anchor;                                       // (its address identifies this loaded library, see index_alloc)
thread_local std::string g_err;
thread_local struct idist_progress* g_watch = nullptr;   // armed by idist_progress_watch_next_build

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

idist_status fail(idist_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(IDIST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                     \
    } while (0)

#define CHK(expr)                        \
    do {                                 \
        idist_status _s = (expr);        \
        if (_s != IDIST_OK) return _s;   \
    } while (0)

struct Layout {
    uint32_t stride, nb, rs, tail;
};
Layout make_layout(uint32_t dim) {
    const uint32_t dp = (dim + 3u) & ~3u;      // zero padding is a bitwise no-op (fma(0,0,acc) == acc)
    const uint32_t steps = dp / 8u;            // chunks_exact(8), instant-distance-py/src/lib.rs:391
    Layout L;
    L.tail = (dp % 8u) == 4u ? 1u : 0u;        // :402-405
    L.nb = steps / 4u;
    L.rs = steps % 4u;
    const uint32_t used = 32u * L.nb + 8u * L.rs + 4u * L.tail;
    L.stride = std::max(16u, (used + 15u) & ~15u);
    return L;
}

// Layer sizing of Hnsw::new, core/lib.rs:238-250 (f32 multiply + truncation, `as usize` saturates)
uint32_t layer_sizes(uint32_t n, float ml, uint32_t* cum, uint32_t cap) {
    uint32_t cnt = 0;
    size_t num = n;
    for (;;) {
        volatile float prod = (float)num * ml;
        size_t next;
        if (!(prod > 0.0f)) next = 0;
        else if (prod >= 18446744073709551616.0f) next = (size_t)-1;
        else next = (size_t)prod;
        if (next < IDIST_M) break;
        if (cnt + 1 >= cap) return 0;  // too many layers
        cum[cnt++] = (uint32_t)num;
        num = next;
    }
    cum[cnt++] = (uint32_t)num;
    return cnt;
}

idist_status check_device(int32_t device) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(IDIST_ERR_NO_DEVICE, "no HIP device visible (%s); libidist has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= cnt) return fail(IDIST_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, cnt);
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(IDIST_ERR_NO_DEVICE, "device %d is %s; libidist is built for gfx950 (MI355X) only", device,
                    p.gcnArchName);
    HIPCHK(hipSetDevice(device));
    return IDIST_OK;
}

idist_status validate_config(const idist_config* cfg, bool for_build) {
    if (!cfg) return fail(IDIST_ERR_INVALID_ARG, "config is null");
    if (cfg->ef_search > IDIST_MAX_EF) return fail(IDIST_ERR_INVALID_ARG, "ef_search %u > %u", cfg->ef_search, IDIST_MAX_EF);
    if (cfg->metric != IDIST_METRIC_L2SQ && cfg->metric != IDIST_METRIC_L2)
        return fail(IDIST_ERR_INVALID_ARG, "unknown metric %d", cfg->metric);
    if (cfg->tie_policy != IDIST_TIES_STRICT && cfg->tie_policy != IDIST_TIES_DROP)
        return fail(IDIST_ERR_INVALID_ARG, "unknown tie_policy %d", cfg->tie_policy);
    if (cfg->tie_capacity > 4096) return fail(IDIST_ERR_INVALID_ARG, "tie_capacity %u > 4096", cfg->tie_capacity);
    if (for_build) {
        if (cfg->ef_construction == 0 || cfg->ef_construction > IDIST_MAX_EF)
            return fail(IDIST_ERR_INVALID_ARG, "ef_construction %u out of [1,%u]", cfg->ef_construction, IDIST_MAX_EF);
    }
    return IDIST_OK;
}

}  // namespace

constexpr uint32_t kCombineLeaders = 8, kCombineBatch = 64;   // launches in flight per index before calls ride along; widest combined launch

// one scalar Hnsw::search call waiting to be served, alone or as part of another thread's launch (idist_combine.hpp)
struct ScalarReq {
    const float* q;
    uint32_t* pid;
    float* dist;
    uint32_t* cnt;
    uint32_t* ctr;
    idist_status st = IDIST_OK;
    std::string err;
    float kernel_ms = -1.0f;         // HIP-event duration of the launch that served this call (< 0: events are off)
    bool done = false, lead = false;
    int slot = -1;
    std::condition_variable cv;
    void fail() { st = IDIST_ERR_INTERNAL; err = "the combined launch serving this call threw (out of host memory?)"; }
};

struct idist_index {
    uint64_t uid = 0;            // never reused: a context is bound to (pointer, uid), not to the pointer alone
    mutable idist::Combiner<ScalarReq> comb{kCombineLeaders, kCombineBatch};   // scalar calls of many threads -> few launches (the index is shared, `&self`)
    // one context per leader slot for the batches a leader takes along: sized by the batches (a caller's own `Search` may be
    // backed for one query only), created on a slot's first combined launch, at most comb.max_leaders() of them
    mutable idist_search_ctx* comb_ctx[kCombineLeaders] = {nullptr};
    int32_t device = 0;
    idist_config cfg{};
    uint32_t n = 0, dim = 0;
    Layout L{};
    uint32_t n_upper = 0;
    uint32_t layer_len[IDIST_MAX_LAYERS] = {0};
    uint64_t layer_off[IDIST_MAX_LAYERS] = {0};
    size_t upper_rows = 0;
    float* d_points = nullptr;
    uint32_t* d_zero = nullptr;
    uint32_t* d_upper = nullptr;
    uint64_t* d_layer_off = nullptr;
    int n_cu = 256;
    idist_build_stats stats{};
    // the walk's reject filter (FilterView, idist_device.hpp): a compact copy of the rows, made once — after a build / an import, or
    // on the first search of an index whose buffers were filled from outside (idist_index_alloc + a broadcast) — see filter_ensure
    mutable std::mutex filt_mu;
    mutable std::atomic<int> filt_state{0};    // 0 not made yet, 1 ready, 2 not available (no memory: the walks run without it)
    mutable idist::FilterView filt{};

    IndexView view() const {
        IndexView v;
        v.points = d_points;
        v.zero = d_zero;
        v.upper = d_upper;
        v.layer_off = d_layer_off;
        v.n = n;
        v.dim = dim;
        v.stride = L.stride;
        v.nb = L.nb;
        v.rs = L.rs;
        v.tail = L.tail;
        v.n_upper = n_upper;
        v.metric = (uint32_t)cfg.metric;
        if (filt_state.load(std::memory_order_acquire) == 1) v.f = filt;
        return v;
    }
};

struct idist_progress {
    volatile unsigned long long* slot = nullptr;   // pinned host memory the device writes: [0] done, [1] layer + 1
    unsigned long long total = 0;
};

// Knobs read from the environment (see test_env above: three in the product, the rest in the test / measurement builds),
// sampled once per context or build — not per launch.
struct Knobs {
    uint32_t latency_nq = 1024;   // IDIST_LATENCY_NQ: batches up to this many queries run the latency walk (0 = never)
    bool classic = false;         // IDIST_WALK=classic
    bool bloom = true;            // IDIST_BLOOM=0 disables the LDS Bloom filter in front of the visited bitmap
    uint32_t tab_log2 = 0;        // IDIST_TAB_LOG2=<5..13>: size of the on-chip visited set (test knob: small sets exercise the overflow path)
    bool vis_bitmap = false;      // IDIST_VISITED=bitmap: search with bitmap + Bloom filter (16 waves per CU) instead of the on-chip set
    bool vis_onchip = false;      // IDIST_VISITED=onchip: the on-chip set whatever the policy says (test / A-B knob)
    bool combine = true;          // IDIST_COMBINE=0: every scalar host-pointer call makes its own launch, however many are in flight
    bool sync_flag = true;        // IDIST_SYNC=stream: narrow host-pointer calls wait with hipStreamSynchronize instead of for the completion
                                  // word the kernel writes to the context's pinned buffer (A/B knob)
    bool tie_spill_first = false; // IDIST_TIE_SPILL=1: strict ties go to the HBM bags at the first overflow instead of growing the LDS region first (test knob)
    bool events = true;           // IDIST_KERNEL_EVENTS=0: no HIP events around the search kernels (idist_search_ctx_kernel_times then has nothing)
    bool filter = true;           // IDIST_FILTER=0: wide on-chip walks without the reject filter (test / A-B knob)
    bool tab_ids = false;         // IDIST_TAB_FORMAT=ids: the on-chip set always keeps full ids (4 per bucket, frozen at 7/8), never
                                  // 16-bit quotients (8 per bucket, single ids overflow) (test / A-B knob)
    bool tab_q16 = false;         // IDIST_TAB_FORMAT=q16: quotients wherever they apply, also where the policy would keep ids
    bool no_zero_copy = false;    // IDIST_NO_ZERO_COPY=1: narrow host-pointer batches take the general (staged) path too (test / A-B knob)
    int ea = 0;                   // IDIST_EA=<k> (measurement builds only, -DIDIST_EA_PROBE): early abandon after k blocks of a 300-d row
    uint32_t w2_ef = 0xFFFFFFFFu; // IDIST_W2_EF=<ef>: from this ef_search on, wide on-chip batches run two 256-register waves per SIMD (A/B knob; default: policy)
    uint32_t quad_nq = 0xFFFFFFFFu;   // IDIST_QUAD_NQ: batches up to this many queries run four waves per query (default: two
                                      // workgroups per CU, one for 768-d rows; 0 = never)
    static Knobs from_env() {
        Knobs k;
        warn_ignored_knobs();
        if (const char* e = test_env("IDIST_EA")) k.ea = atoi(e);
        if (const char* e = test_env("IDIST_W2_EF")) k.w2_ef = (uint32_t)strtoul(e, nullptr, 10);
        if (const char* e = test_env("IDIST_LATENCY_NQ")) k.latency_nq = (uint32_t)strtoul(e, nullptr, 10);
        if (const char* e = test_env("IDIST_QUAD_NQ")) k.quad_nq = (uint32_t)std::min<unsigned long>(strtoul(e, nullptr, 10), 0xFFFFFFFEul);
        if (const char* e = test_env("IDIST_WALK")) k.classic = e[0] == 'c';      // (honoured by the test build only, see variants_check)
        if (const char* e = test_env("IDIST_BLOOM")) k.bloom = e[0] != '0';
        if (const char* e = test_env("IDIST_FILTER")) k.filter = e[0] != '0';
        if (const char* e = test_env("IDIST_VISITED")) { k.vis_bitmap = e[0] == 'b'; k.vis_onchip = e[0] == 'o'; }
        if (const char* e = test_env("IDIST_NO_ZERO_COPY")) k.no_zero_copy = e[0] != '0';
        if (const char* e = test_env("IDIST_TAB_FORMAT")) { k.tab_ids = e[0] == 'i'; k.tab_q16 = e[0] == 'q'; }
        if (const char* e = getenv("IDIST_KERNEL_EVENTS")) k.events = e[0] != '0';
        if (const char* e = test_env("IDIST_TIE_SPILL")) k.tie_spill_first = e[0] == '1';
        if (const char* e = getenv("IDIST_SYNC")) k.sync_flag = e[0] != 's';
        if (const char* e = getenv("IDIST_COMBINE")) k.combine = e[0] != '0';
        if (const char* e = test_env("IDIST_TAB_LOG2")) k.tab_log2 = std::min(13u, std::max(5u, (uint32_t)atoi(e)));
        return k;
    }
};

struct idist_search_ctx {
    const idist_index* idx = nullptr;
    uint64_t idx_uid = 0;          // uid of the index this context was made for
    uint32_t slots_req = 0;        // what the caller asked for (0 = as many as the batches need, up to a full chip)
    uint32_t slots = 0;            // query slots currently backed by a visited bitmap
    VisGeom vis{};
    uint32_t* d_visited = nullptr; // [slots][vis.slot_words], all-zero between launches
    uint32_t* d_next = nullptr;    // [0] queue head, [1] status, [2] completion count, [16..19] the reject filter's two 64-bit counters
    uint32_t queue_base = 0;       // value of the queue head before the next launch (it is never reset: each launch of nq queries on
                                   // g workgroups moves it by nq + g, unsigned wrap-around included)
    // narrow host-pointer batches (the reference's one query per call): query and results cross PCIe through one pinned,
    // device-mapped buffer the kernel reads and writes directly — no memcpy / memset calls around the launch
    uint8_t* h_io = nullptr;       // host address
    uint8_t* d_io = nullptr;       // the same memory as the device sees it
    size_t io_cap = 0;             // bytes of that buffer: 64 KB on first use, grown on demand up to kIoMaxBytes
    uint32_t done_seq = 0;         // completion word of the last narrow host-pointer launch (h_io + kIoStatusSlots * 4)
    double call_ns_ema = 0.0;      // how long such a call has taken lately: the host sleeps through the first half of it
    uint64_t n_flag_calls = 0;
    static constexpr size_t kIoMinBytes = 64 * 1024, kIoMaxBytes = 256 * 1024, kIoStatusSlots = 256, kIoHeadBytes = kIoStatusSlots * 4 + 64;   // ~100 queries: beyond that the staged copies are as fast (profiles/r02/probe_r02_quad_single_query_phases.jsonl)
    hipStream_t stream = nullptr;
    hipEvent_t ev0[IDIST_EVENT_RING] = {nullptr}, ev1[IDIST_EVENT_RING] = {nullptr};
    uint64_t n_launch = 0;
    // Kernel times this context reports (idist_search_ctx_kernel_times): one record per call served, in call order — its own
    // launches (slot of the event ring; resolved when asked for) and the calls that rode along in another thread's launch
    // (the leader hands the measured duration back, ScalarReq::kernel_ms)
    struct TimeRec { int32_t slot; float ms; };    // slot >= 0: own launch, event pair `slot`; slot < 0: `ms` measured elsewhere
    TimeRec recs[IDIST_EVENT_RING] = {};
    uint64_t n_rec = 0;
    int32_t device = 0;            // copies of what growing the context needs: nothing is read through `idx` after creation
    int n_cu = 256;
    Knobs knobs;
    // staging for the host-pointer API
    float* d_q = nullptr;
    uint32_t* d_pid = nullptr;
    float* d_dist = nullptr;
    uint32_t* d_cnt = nullptr;
    uint32_t* d_ctr = nullptr;
    size_t cap_q = 0, cap_out = 0, cap_nq = 0;
    bool tie_overflowed = false;
    uint32_t tie_cap = 0;          // tie capacity this context escalated to (0 = the index's)
    // strict ties, last resort: one bag of n keys per slot in HBM (the reference's candidate heap is unbounded, core/lib.rs:564)
    bool tie_spill = false;        // later launches attach the bags
    bool tie_escalation_exhausted = false;
    uint64_t* d_tie_spill = nullptr;
    uint32_t spill_slots = 0, n_points = 0;
    // copies of what idist_search_ctx_status needs, so that it never reads through `idx`
    int32_t tie_policy = IDIST_TIES_STRICT;
    uint32_t base_tie_cap = kTieCap, stride = 0, last_ef = 0;
};

namespace {

// The row geometries the kernels are instantiated for at compile time — ONE list: IDIST_DISPATCH instantiates them, and every
// policy that asks "is this a runtime-geometry index?" (launch_search, run_build, filter_applies) asks has_template_geometry().
#define IDIST_GEO_LIST(X, L, CALL) X(L, CALL, 4, 0, 0) /* dim 128 */ X(L, CALL, 9, 1, 1) /* dim 300 */ X(L, CALL, 24, 0, 0) /* dim 768 */
#define IDIST_GEO_TEST(L, CALL, NB_, RS_, TAIL_) if ((L).nb == (NB_) && (L).rs == (RS_) && (L).tail == (TAIL_)) return true;
inline bool has_template_geometry(const Layout& L) {
    IDIST_GEO_LIST(IDIST_GEO_TEST, L, _)
    return false;
}
#define IDIST_GEO_CASE(L, CALL, NB_, RS_, TAIL_) \
    if (!idist_geo_hit_ && (L).nb == (NB_) && (L).rs == (RS_) && (L).tail == (TAIL_)) { idist_geo_hit_ = true; CALL(NB_, RS_, TAIL_); }
#define IDIST_DISPATCH(L, CALL)                                         \
    do {                                                                \
        bool idist_geo_hit_ = false;                                    \
        IDIST_GEO_LIST(IDIST_GEO_CASE, L, CALL)                         \
        if (!idist_geo_hit_) { CALL(-1, -1, -1); }                      \
    } while (0)

// Which filtered kernels a row geometry can ever be given (launch_search / run_build): instantiating only those keeps the build of
// this translation unit at three minutes.  The thin filtered walk serves compact rows of up to five chunks (128-d, 300-d, runtime
// geometries), one fat filtered wave per SIMD the longer ones (768-d, runtime geometries); fat filtered DESCENTS exist for runtime
// geometries only.  A launch the policy asks for and the geometry has no kernel for is an internal error, never a silent no-op.
template <int NB> struct GeoKernels {
    static constexpr bool thin_search = NB != 24;
    static constexpr bool fat_filtered_search = NB == 24 || NB < 0;
    static constexpr bool fat_filtered_search_ids = NB == 24;      // (the id form of the set, cheaper to probe while a walk cannot fill it: C4)
    // thin filtered descents: rows of at least 192 floats by policy — 128-d rows only under the test knob IDIST_BUILD_FILTER=1
#if defined(IDIST_VARIANTS) || defined(IDIST_EMU) || defined(IDIST_PROBE)
    static constexpr bool thin_descent = true;
#else
    static constexpr bool thin_descent = NB != 4;
#endif
};
template <int NB, int RS, int TAIL, int WALK, bool ON> struct SearchLaunch {
    static bool go(uint32_t grid, size_t smem, hipStream_t stream, const IndexView& view, const SearchArgs& a) {
        auto kS = search_kernel<NB, RS, TAIL, WALK>;
        IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);
        return true;
    }
};
template <int NB, int RS, int TAIL, int WALK> struct SearchLaunch<NB, RS, TAIL, WALK, false> {
    static bool go(uint32_t, size_t, hipStream_t, const IndexView&, const SearchArgs&) { return false; }
};
template <int NB, int RS, int TAIL, int WALK, bool ON> struct DescentLaunch {
    static bool go(uint32_t grid, size_t smem, hipStream_t stream, const IndexView& view, const BuildArgs& a) {
        auto kA = build_insert_kernel<NB, RS, TAIL, WALK>;
        IDIST_LAUNCH(kA, grid, 64, smem, stream, view, a);
        return true;
    }
};
template <int NB, int RS, int TAIL, int WALK> struct DescentLaunch<NB, RS, TAIL, WALK, false> {
    static bool go(uint32_t, size_t, hipStream_t, const IndexView&, const BuildArgs&) { return false; }
};

idist_status index_alloc(uint32_t n, uint32_t dim, const idist_config* cfg, const uint32_t* layer_len,
                         uint32_t n_upper, int32_t device, idist_index** out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (dim == 0 || dim > 65536) return fail(IDIST_ERR_INVALID_ARG, "dim %u out of [1,65536]", dim);
    if (n == 0xFFFFFFFFu) return fail(IDIST_ERR_INVALID_ARG, "n must be < u32::MAX (core/lib.rs:256)");
    if (n_upper >= IDIST_MAX_LAYERS) return fail(IDIST_ERR_INVALID_ARG, "more than %u layers", IDIST_MAX_LAYERS);
    CHK(check_device(device));
    // uids are unique across the libraries of one process too (libidist.so and the test build libidist_variants.so share handles in
    // tests/): the counter starts at a value derived from where THIS library was mapped
    static std::atomic<uint64_t> next_uid{(((uint64_t)(uintptr_t)&g_err_anchor >> 12) << 28) | 1u};
    idist_index* ix = new idist_index();
    ix->uid = next_uid.fetch_add(1);
    ix->device = device;
    ix->cfg = *cfg;
    ix->n = n;
    ix->dim = dim;
    ix->L = make_layout(dim);
    ix->n_upper = n_upper;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) ix->n_cu = p.multiProcessorCount;
    size_t rows = 0;
    for (uint32_t l = 0; l < n_upper; l++) {
        if (layer_len[l] > n) { delete ix; return fail(IDIST_ERR_INVALID_ARG, "layer_len[%u] = %u > n", l, layer_len[l]); }
        ix->layer_len[l] = layer_len[l];
        ix->layer_off[l] = rows;
        rows += layer_len[l];
    }
    ix->upper_rows = rows;
    auto cleanup = [&](idist_status s) { idist_index_free(ix); return s; };
    auto alloc = [&](void** p, size_t bytes) -> idist_status {
        HIPCHK(hipMalloc(p, std::max<size_t>(bytes, 256)));
        return IDIST_OK;
    };
    idist_status s;
    if ((s = alloc((void**)&ix->d_points, (size_t)n * ix->L.stride * 4)) != IDIST_OK) return cleanup(s);
    if ((s = alloc((void**)&ix->d_zero, (size_t)n * IDIST_M2 * 4)) != IDIST_OK) return cleanup(s);
    if ((s = alloc((void**)&ix->d_upper, rows * IDIST_M * 4)) != IDIST_OK) return cleanup(s);
    if ((s = alloc((void**)&ix->d_layer_off, IDIST_MAX_LAYERS * 8)) != IDIST_OK) return cleanup(s);
    if (hipMemcpy(ix->d_layer_off, ix->layer_off, IDIST_MAX_LAYERS * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(ix->d_zero, 0xFF, std::max<size_t>((size_t)n * IDIST_M2 * 4, 4)) != hipSuccess ||
        hipMemset(ix->d_upper, 0xFF, std::max<size_t>(rows * IDIST_M * 4, 4)) != hipSuccess)
        return cleanup(fail(IDIST_ERR_HIP, "index initialisation failed: %s", hipGetErrorString(hipGetLastError())));
    // callers fill these buffers from their own streams (RCCL broadcast, imports): let the fills above land first
    if (hipDeviceSynchronize() != hipSuccess)
        return cleanup(fail(IDIST_ERR_HIP, "index initialisation failed: %s", hipGetErrorString(hipGetLastError())));
    *out = ix;
    return IDIST_OK;
}

// natural row-major device points -> blocked rows of the index
idist_status load_points_device(idist_index* ix, const float* d_nat) {
    if (ix->n == 0) return IDIST_OK;
    const size_t total = (size_t)ix->n * ix->L.stride;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
    IDIST_LAUNCH(permute_rows_kernel, grid, 256, 0, (hipStream_t) nullptr, d_nat, ix->d_points, ix->n, ix->dim, ix->L.stride, ix->L.nb);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    return IDIST_OK;
}
idist_status load_points_host(idist_index* ix, const float* h_nat) {
    if (ix->n == 0) return IDIST_OK;
    float* d_nat = nullptr;
    const size_t bytes = (size_t)ix->n * ix->dim * 4;
    HIPCHK(hipMalloc((void**)&d_nat, bytes));
    hipError_t e = hipMemcpy(d_nat, h_nat, bytes, hipMemcpyHostToDevice);
    idist_status s = e == hipSuccess ? load_points_device(ix, d_nat)
                                     : fail(IDIST_ERR_HIP, "hipMemcpy(points): %s", hipGetErrorString(e));
    hipFree(d_nat);
    return s;
}

uint32_t default_slots(uint32_t n_points, int n_cu) {
    // fill the chip (16 single-wave workgroups per CU) within a memory budget for the visited bitmaps: one bit per
    // point and slot (core/types.rs:13-59), i.e. 512 MB for 4096 slots at 1M points, 5 GB at 10M
    size_t freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) freeb = (size_t)8 << 30;
    const size_t budget = std::min<size_t>(freeb / 3, (size_t)64 << 30);
    const size_t vis = (size_t)vis_geometry(n_points).slot_words * 4;
    size_t s = (size_t)n_cu * 16;
    if (vis) s = std::min(s, std::max<size_t>(budget / vis, 64));
    return (uint32_t)std::max<size_t>(s, 1);
}

// LDS one wave of the on-chip walk may use: one wave per SIMD, four per CU, out of 160 KiB
constexpr size_t kOnChipLdsPerWave = 40 * 1024;

thread_local uint32_t g_tie_cap_msg = kTieCap;
uint32_t tie_capacity(const idist_config& cfg) { return g_tie_cap_msg = cfg.tie_capacity ? cfg.tie_capacity : (uint32_t)kTieCap; }

// IDIST_WALK=classic names walks that only the test build holds (libidist_variants.so): say so instead of running
// something else under that name
idist_status variants_check(bool classic) {
#if defined(IDIST_PROBE) && !defined(IDIST_VARIANTS)   // (the product cannot be asked: it reads no IDIST_WALK)
    if (classic) return fail(IDIST_ERR_UNSUPPORTED, "IDIST_WALK=classic selects a test-only variant of the walk: load libidist_variants.so (make -C instant-distance_amd/csrc variants)");
#endif
    (void)classic;
    return IDIST_OK;
}

idist_status device_status_to_code(uint32_t st, int32_t tie_policy) {
    if (st & kStBadRow) return fail(IDIST_ERR_BAD_GRAPH, "device met an adjacency id >= n");
    if ((st & kStTieOverflow) && tie_policy == IDIST_TIES_STRICT)
        return fail(IDIST_ERR_TIE_OVERFLOW, "more than %d live equidistant candidates beyond ef "
                    "(raise idist_config.tie_capacity, or tie_policy = IDIST_TIES_DROP keeps the nearest ones and goes on)", (int)g_tie_cap_msg);
    if (st & kStQueue)
        return fail(IDIST_ERR_INTERNAL, "a search launch found its context's work queue where no launch of this context left it: the "
                    "launches of one context must run once each, in the order they were enqueued (no graph replay, one stream)");
    if (st & kStGuard) return fail(IDIST_ERR_INTERNAL, "device-side loop guard tripped");
    return IDIST_OK;
}

// ---- build driver: the per-layer insertion schedule of Hnsw::new, core/lib.rs:304-329 ----
bool filter_applies(const idist_index* ix);
idist_status filter_ensure(const idist_index* ix);

idist_status run_build(idist_index* ix, idist_progress* prog, bool tie_spill) {
    const uint32_t n = ix->n;
    if (prog) { prog->total = n; prog->slot[0] = n ? 1 : 0; prog->slot[1] = 0; }   // pid 0 is in from the start
    if (n <= 1) return IDIST_OK;   // pid 0 is never inserted (core/lib.rs:279-280)
    const idist_config& cfg = ix->cfg;
    const uint32_t top = ix->n_upper;
    // a step never holds more than 1/32 of the points already inserted, so scratch is sized by that
    // Heuristic::extend_candidates: defined by the oracle's lock-free restatement (it deadlocks in the reference,
    // core/lib.rs:649 vs :438), one whole insertion per launch in program order — always the sequential schedule
    const bool ext = cfg.has_heuristic && cfg.extend_candidates;
    const uint32_t cap = ext ? 1u : std::min<uint32_t>(cfg.max_batch == 0 ? 8192u : cfg.max_batch, std::max<uint32_t>(1u, n / 32u));
    // the descents keep their visited set on chip, one fat wave per SIMD (up to 8 per CU with the 16-KB quotient set);
    // with the HBM tie bags (one of n keys per slot) as many slots as fit 1 GiB
    uint32_t slots = std::min(cap, (uint32_t)ix->n_cu * 8u);
    if (tie_spill) slots = std::min<uint32_t>(slots, (uint32_t)std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)n * 8)));
    const VisGeom vg = vis_geometry(n);
    const Knobs knobs = Knobs::from_env();
    CHK(variants_check(knobs.classic));
    // no compile-time instantiation of the row geometry (IDIST_DISPATCH): the runtime-geometry kernels
    const bool rt_geometry = !has_template_geometry(ix->L);

    const uint32_t tie_cap = tie_capacity(cfg);
    const uint32_t wcap = cfg.ef_construction + 64 + tie_cap + 64;
    // The descents' on-chip visited set.  Quotient form (16-bit entries, idist_device.hpp q16_*) where n allows it: 16 KB hold
    // 8192 ids — an ef_construction = 100 descent visits ~6k — so a descent wave takes ~22 KB of LDS instead of ~39 KB
    // and four or five of them leave half the CU's LDS to the update stream.  Otherwise (and IDIST_TAB_FORMAT=ids, ext):
    // full ids, the largest set that leaves room for four waves per CU, or the largest that fits at all.
    uint32_t tab_log2 = 0;
    bool tab16 = false;
    if (!ext && !knobs.tab_ids) {
        const uint32_t want = knobs.tab_log2 ? std::max(8u, std::min(13u, knobs.tab_log2))
                                             : (cfg.ef_construction <= 128 ? 12u : 13u);
        for (uint32_t l = want; l <= 13 && !tab16; l++)
            if (q16_applies(l, q16_universe_bits(n, l)) && smem_bytes(ix->L.stride, wcap, true, 1u << l, vg.dirty_words) <= kOnChipLdsPerWave) {
                tab_log2 = l;
                tab16 = true;
            }
    }
    if (!tab16) {
        for (uint32_t l = 13; l >= 8 && !tab_log2; l--)
            if (smem_bytes(ix->L.stride, wcap, true, 1u << l, vg.dirty_words) <= kOnChipLdsPerWave) tab_log2 = l;
        for (uint32_t l = 13; l >= 8 && !tab_log2; l--)
            if (smem_bytes(ix->L.stride, wcap, true, 1u << l, vg.dirty_words) <= 64 * 1024) tab_log2 = l;
        if (!tab_log2) return fail(IDIST_ERR_INVALID_ARG, "dim/ef_construction need more than 64 KiB of LDS per wave");
        if (knobs.tab_log2) tab_log2 = std::max(8u, std::min(tab_log2, knobs.tab_log2));
    }
    const uint32_t dl_shift = tab_log2 + (tab16 ? 1u : 0u);
    const size_t smem = smem_bytes(ix->L.stride, wcap, true, 1u << tab_log2, vg.dirty_words);

    // Tiles of steps B2 / A2 (tile kernel): selected rows kept on chip next to the staging slots.  A tile wave has to find LDS
    // BESIDE the descents of the next step (four waves per CU of `smem` bytes each, resident for a whole launch): a tile that
    // does not fit waits for that launch to drain and the pipeline runs serially — measured at 1024-d, where a 63-KB tile
    // missed the 62 KB left by 1 KB and the full re-selections took 7.4 s of a 9.5-s build.  So: at most what is left of the
    // CU's 160 KiB, at most 64 KiB; long runtime-geometry rows stage four candidates per round instead of eight.
    const bool generic_geo = rt_geometry;
    const size_t lds_beside = (size_t)160 * 1024 > 4 * smem + 8 * 1024 ? (size_t)160 * 1024 - 4 * smem : 8 * 1024;
    const size_t tile_budget = cap > 1 ? std::min<size_t>(64 * 1024, std::max<size_t>(lds_beside, 24 * 1024)) : 64 * 1024;
    uint32_t fc = 8;
    if (generic_geo && smem_bytes_update(ix->L.nb, 4, 8) > tile_budget) fc = 4;
    uint32_t rt = 8;
    if (const char* e = test_env("IDIST_BUILD_RT")) rt = (uint32_t)atoi(e);
    while (rt > 0 && smem_bytes_update(ix->L.nb, rt, fc) > tile_budget) rt--;
    size_t smemB = smem_bytes_update(ix->L.nb, rt, fc);
    if (smemB > tile_budget) {                                   // does not fit beside the descents: take what a CU alone allows
        rt = 8;
        while (rt > 0 && smem_bytes_update(ix->L.nb, rt, fc) > 64 * 1024) rt--;
        smemB = smem_bytes_update(ix->L.nb, rt, fc);
    }
    if (smemB > 64 * 1024) return fail(IDIST_ERR_INVALID_ARG, "dim %u needs %zu B of LDS per wave in the build (> 64 KiB)", ix->dim, smemB);

    // step A2 tile: the new point's selected set (up to 64 rows) — it is not memory bound, so favour rows on chip
    uint32_t rt2 = 16;
    if (const char* e = test_env("IDIST_BUILD_RT2")) rt2 = (uint32_t)atoi(e);
    while (rt2 > 0 && smem_bytes_select(ix->L.nb, rt2, cfg.ef_construction, fc) > tile_budget) rt2--;
    size_t smemA2 = smem_bytes_select(ix->L.nb, rt2, cfg.ef_construction, fc);
    if (smemA2 > tile_budget) {
        rt2 = 16;
        while (rt2 > 0 && smem_bytes_select(ix->L.nb, rt2, cfg.ef_construction, fc) > 64 * 1024) rt2--;
        smemA2 = smem_bytes_select(ix->L.nb, rt2, cfg.ef_construction, fc);
    }
    if (smemA2 > 64 * 1024) return fail(IDIST_ERR_INVALID_ARG, "dim %u / ef_construction %u need %zu B of LDS per wave in the build (> 64 KiB)", ix->dim, cfg.ef_construction, smemA2);

    uint64_t* d_ext_work = nullptr;
    uint32_t ext_cap = 1;
    // a selection's working set, padded for the sort: nearest (<= ef_construction for the new point, <= 65 = new + a full
    // row for a neighbour's re-selection, whatever ef_construction is) plus up to 64 extensions per member
    while (ext_cap < (std::max(cfg.ef_construction, 65u) + 1u) * 65u + 64u) ext_cap <<= 1;
    const size_t smemX = smem_bytes_extend(ix->L.stride, wcap, 1u << tab_log2, vg.dirty_words);
    if (ext && smemX > 64 * 1024) return fail(IDIST_ERR_INVALID_ARG, "dim/ef_construction need %zu B of LDS per wave with extend_candidates (> 64 KiB)", smemX);
    // step A2 on the matrix cores (Gram matrix of the candidates as a filter, idist_mfma.hpp) where it applies
    const bool a2_mfma = cfg.has_heuristic && cfg.metric == IDIST_METRIC_L2SQ && cfg.ef_construction <= 128 &&
                         !(test_env("IDIST_BUILD_A2") && test_env("IDIST_BUILD_A2")[0] == 't');
    const size_t smemA2m = smem_bytes_select_mfma(ix->L.stride);
    uint32_t* d_vis = nullptr;
    uint64_t* d_tie_spill = nullptr;
    uint32_t* d_nbr_dist = nullptr;
    uint32_t *d_edge_pid = nullptr, *d_edge_dist = nullptr, *d_head = nullptr, *d_next = nullptr, *d_touched = nullptr;
    uint32_t* d_small = nullptr;           // [0] n_touched, [1..5] queue (A, B, n_slow, B2, A2), [8] status
    uint64_t *d_wbuf = nullptr, *d_dlog_log = nullptr;
    uint32_t* d_dlog_pd = nullptr;
    uint32_t* d_wcount = nullptr;
    uint32_t *d_row_nsel = nullptr, *d_slow = nullptr, *d_nbr_aux = nullptr;
    unsigned long long* d_stats = nullptr; // [32]
    const size_t n_edges = (size_t)cap * IDIST_M2;
    const size_t n_touch = std::min<size_t>(n_edges, n);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // Pipelined schedule (default for concurrent builds with the heuristic): the descents of step k+1 (HBM-bound)
    // run on one stream while the selections / neighbour updates of step k (LDS- and latency-bound) run on another.
    // Step k's descents read the graph as of step k-2, so nothing they read is being written: the zero layer is
    // kept in two copies, copy k&1 receiving the state after step k.  Deterministic like the sequential schedule;
    // a new point then misses the last two steps' points (<= 1/16 of the graph) instead of the last one's.
    bool pipe = cap > 1 && cfg.has_heuristic;
    if (const char* e = test_env("IDIST_BUILD_PIPELINE")) pipe = pipe && e[0] != '0';
    // the descents (8-12 waves per CU saturate their HBM stream) leave wave slots and LDS to the other stream
    uint32_t a_waves = tab16 ? 4u : 3u;
    if (const char* e = test_env("IDIST_BUILD_A_WAVES")) a_waves = (uint32_t)std::min(8, std::max(1, atoi(e)));
    // the descent's register budget (quotient set only): 256 registers per wave (two waves fit a SIMD: the update stream's
    // waves find room on every SIMD) or 512 (IDIST_BUILD_A_REGS=512, more rounds in flight per wave).  C3: 1.26 vs 1.29-1.32 s
    // Runtime-geometry rows (any dimension without a compile-time instantiation): the register tile, not the row, bounds what a
    // wave keeps on the wire, and the 256-register tile (2 x 3 rounds x 4 blocks = 24 KB) is too little at four waves per
    // CU: one 512-register wave per SIMD (2 x 4 x 8 = 64 KB each) builds 1M x 1024-d in 4.35 s instead of 9.6 s and 1M x 384-d
    // in 1.98 s instead of 2.66 s (profiles/r04/probe_r04d_build_dim*.jsonl) although the update stream then only runs where a
    // descent wave has retired.
    // Which register budget the descents take (same graphs either way; round 5):
    //   * the 128-d instantiation (4 blocks): one fat wave per SIMD at EVERY size — C2 100k x 128: 0.110 against 0.121 s (five to
    //     eight thin waves per CU: 0.117-0.122 s); 300k / 600k / 1M x 128: 0.245 / 0.449 / 0.713 s against 0.286 / 0.550 / 0.915 s
    //     (profiles/r05/probe_r05c_build_descent_waves_c2.jsonl, probe_r05p_build_regs_mid_size_short_rows.jsonl);
    //   * 300-d and 768-d rows: two 256-register waves per SIMD, also where the index sits in the Infinity Cache (C3 1.26 against
    //     1.29-1.32 s, C4 2.78-2.83 against 3.00-3.06 s; 40k x 768: 0.139 against 0.153 s; 60k-200k x 300: within 2 %,
    //     probe_r05q_build_regs_cache_resident_long_rows.jsonl);
    //   * runtime-geometry rows of at most 12 blocks (<= 384-d) fit the 256-register tile whole (<6 blocks, 3 rounds>, two groups in
    //     flight) and build faster on it — 1M x 64 / 100 / 200 / 384-d: 0.91 / 1.11 / 1.30 / 1.75 s against 1.18 / 1.38 / 1.55 / 1.96 s;
    //     longer rows keep the fat waves (512-d: 2.35 against 3.12 s, 500k x 1024-d: 2.26 against 4.19 s; probe_r05g_*, probe_r05p_*).
    const bool rows128 = ix->L.nb == 4 && ix->L.rs == 0 && ix->L.tail == 0;
    bool a_regs256 = tab16 && !rows128 && (!rt_geometry || ix->L.nb <= 12u);
    if (const char* e = test_env("IDIST_BUILD_A_REGS")) a_regs256 = tab16 && atoi(e) == 256;
    // (Fat descent waves take a SIMD's whole register file: where four of them sit on a CU, nothing of the update stream runs until
    //  one retires — at 1024-d the selection's launches stretch to the descents' 13 ms and a step's period is 16.5 ms for 12.9 ms of
    //  descents, profiles/r05/trace_chain_r05i_build_500k_dim1024.json.  Measured and not kept: fewer descent waves per step, so that some
    //  SIMDs stay free — 960 / 896 / 768 waves build 500k x 1024-d in 2.25 / 2.31 / 2.51 s against 2.20 s, probe_r05j; the update
    //  streams at the highest priority — no difference at 1024 / 512 / 768 / 300-d, probe_r05k.)
    // steps of at most two insertions per CU run four waves per insertion (IDIST_BUILD_QUAD=0: never)
    const bool a_quad = !(test_env("IDIST_BUILD_QUAD") && test_env("IDIST_BUILD_QUAD")[0] == '0');
    const uint32_t quad_B = (uint32_t)ix->n_cu * 2u;
    // Narrow steps (four waves per insertion: their time does not depend on their width — one descent ≈ 0.45 ms) hold g / 8
    // insertions until they stop being narrow, wide ones g / 32: the first 16k points of a build take ≈ 60 steps instead of
    // ≈ 180 (100k x 128: 0.156 -> 0.122 s, 20k x 128: 0.078 -> 0.043 s, C3 -2.6 %; recall@10 unchanged to the fourth digit at
    // 5k / 20k / 100k / 1M points, profiles/r04/probe_r04_build_growth_*.jsonl).  IDIST_BUILD_GROWTH=<d> (A/B knob, 8..32): g / d.
    uint32_t growth_div = 8u;
    if (const char* e = test_env("IDIST_BUILD_GROWTH")) growth_div = (uint32_t)std::min(32, std::max(8, atoi(e)));
    // what one CU's LDS holds of them (the sequential schedule runs nothing beside the descents)
    const uint32_t a_waves_max = std::max<uint32_t>(1u, std::min<uint32_t>(8u, (uint32_t)((160u * 1024u) / smem)));
    a_waves = std::min(a_waves, a_waves_max);
    uint32_t* d_zero2 = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr, s3 = nullptr, s4 = nullptr;
    hipEvent_t evA[2] = {nullptr, nullptr}, evS[2] = {nullptr, nullptr}, evA2[2] = {nullptr, nullptr}, evC = nullptr;
    // Narrow steps (the growth phase of every layer: fewer insertions than the chip has room for) are bound by the latency of
    // one descent plus five dependent launches, not by throughput, so there the two parity chains of the pipeline
    //     descents(k) -> selection(k) -> updates(k) -> descents(k + 2)
    // get streams of their own: the descents of odd steps (s3) run beside those of even steps (s1), and the new points'
    // selection (s4: matrix cores and LDS) beside the previous step's neighbour updates (s2: dependent gathers).  Everything
    // a launch owns is kept per parity for that — visited bitmaps, work-queue heads, step-A outputs, the inboxes the selection
    // hands to the updates.  Wide steps saturate the memory system whatever the layout (profiles/r04/probe_r04_build_schedule:
    // the extra overlap costs 2 %), so they keep one descent stream and one update stream.
    // IDIST_BUILD_STREAMS=off | narrow (default) | all.
    int stream_mode = 1;
    if (const char* e = test_env("IDIST_BUILD_STREAMS")) stream_mode = e[0] == 'o' ? 0 : (e[0] == 'a' ? 2 : 1);
    bool two_a = !tie_spill && stream_mode != 0;
    bool own_a2 = stream_mode != 0;
    auto release = [&]() {
        hipFree(d_zero2);
        hipFree(d_ext_work);
        if (s1) hipStreamDestroy(s1);
        if (s2) hipStreamDestroy(s2);
        if (s3) hipStreamDestroy(s3);
        if (s4) hipStreamDestroy(s4);
        for (int i = 0; i < 2; i++) { if (evA[i]) hipEventDestroy(evA[i]); if (evS[i]) hipEventDestroy(evS[i]); if (evA2[i]) hipEventDestroy(evA2[i]); }
        if (evC) hipEventDestroy(evC);
        hipFree(d_nbr_dist); hipFree(d_row_nsel); hipFree(d_slow); hipFree(d_nbr_aux); hipFree(d_wbuf); hipFree(d_wcount); hipFree(d_dlog_log); hipFree(d_dlog_pd);
        hipFree(d_vis); hipFree(d_tie_spill); hipFree(d_edge_pid); hipFree(d_edge_dist); hipFree(d_head);
        hipFree(d_next); hipFree(d_touched); hipFree(d_small); hipFree(d_stats);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
    };
#define BCHK(expr)                                                                                  \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            release();                                                                              \
            return fail(IDIST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                           \
    } while (0)
    if (ext) BCHK(hipMalloc((void**)&d_ext_work, (size_t)ext_cap * 8));
    two_a = two_a && pipe;
    const size_t vis_words = (size_t)slots * vg.slot_words;            // one set of visited bitmaps per descent stream
    BCHK(hipMalloc((void**)&d_vis, vis_words * 4 * (two_a ? 2 : 1)));
    BCHK(hipMemset(d_vis, 0, vis_words * 4 * (two_a ? 2 : 1)));
    if (tie_spill) BCHK(hipMalloc((void**)&d_tie_spill, (size_t)slots * n * 8));
    BCHK(hipMalloc((void**)&d_nbr_dist, (size_t)n * IDIST_M2 * 4));
    BCHK(hipMemset(d_nbr_dist, 0, (size_t)n * IDIST_M2 * 4));
    BCHK(hipMalloc((void**)&d_nbr_aux, (size_t)n * IDIST_M2 * 4));
    BCHK(hipMemset(d_nbr_aux, 0, (size_t)n * IDIST_M2 * 4));
    BCHK(hipMalloc((void**)&d_row_nsel, (size_t)n * 4));
    BCHK(hipMemset(d_row_nsel, 0, (size_t)n * 4));
    own_a2 = own_a2 && pipe;
    const size_t nib = own_a2 ? 2 : 1;                                   // inbox sets
    BCHK(hipMalloc((void**)&d_slow, nib * n_touch * 4));
    const size_t np = pipe ? 2 : 1;       // step-A outputs are double-buffered in the pipelined schedule
    BCHK(hipMalloc((void**)&d_wbuf, np * cap * cfg.ef_construction * 8));
    BCHK(hipMalloc((void**)&d_wcount, np * cap * 4));
    const size_t dl_n = cfg.has_heuristic ? (np * cap) << dl_shift : 64;   // the simple splice looks nothing up
    BCHK(hipMalloc((void**)&d_dlog_log, dl_n * 8));
    BCHK(hipMalloc((void**)&d_dlog_pd, dl_n * 8));
    if (pipe) {
        BCHK(hipMalloc((void**)&d_zero2, (size_t)n * IDIST_M2 * 4));
        BCHK(hipMemset(d_zero2, 0xFF, (size_t)n * IDIST_M2 * 4));
        BCHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        BCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        if (two_a) BCHK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
        if (own_a2) {
            BCHK(hipStreamCreateWithFlags(&s4, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) BCHK(hipEventCreate(&evA2[i]));
            BCHK(hipEventCreate(&evC));
        }
        for (int i = 0; i < 2; i++) { BCHK(hipEventCreate(&evA[i])); BCHK(hipEventCreate(&evS[i])); }
    }
    BCHK(hipMalloc((void**)&d_edge_pid, nib * n_edges * 4));
    BCHK(hipMalloc((void**)&d_edge_dist, nib * n_edges * 4));
    BCHK(hipMalloc((void**)&d_next, nib * n_edges * 4));
    BCHK(hipMalloc((void**)&d_head, nib * (size_t)n * 4));
    BCHK(hipMemset(d_head, 0xFF, nib * (size_t)n * 4));
    BCHK(hipMalloc((void**)&d_touched, nib * n_touch * 4));
    BCHK(hipMalloc((void**)&d_small, 256));
    BCHK(hipMemset(d_small, 0, 256));
    BCHK(hipMalloc((void**)&d_stats, 256));
    BCHK(hipMemset(d_stats, 0, 256));
    BCHK(hipEventCreate(&e0));
    BCHK(hipEventCreate(&e1));

    // Round 6: the descents of concurrent steps run with the reject filter in front of their distance passes (§4.5) — the compact
    // copy of the rows is made now (every row is in place before the first insertion), the log takes bound-form entries for the
    // candidates the filter turned down, step B resolves them (dlog_resolve).  IDIST_BUILD_FILTER=0 (test knob): without.
    // Rows of at least 192 floats: C3 1.20 -> 1.03 s, 1M x 768 2.82 -> 2.01 s, 1M x 384 1.71 -> 1.29 s, 1M x 200 1.22 -> 1.12 s
    // (probe_r06v_build_filter_128_recheck.jsonl); 128-d rows lose (a 192-B compact
    // row saves little of a 512-B one and their fat unfiltered descents are faster: 1M x 128 0.70 -> 0.85 s, C2 0.107 -> 0.118 s) and
    // keep the unfiltered descents (profiles/probe_r06g_build_filter_c3.jsonl, probe_r06h_build_filter_dims.jsonl; same graphs).
    // IDIST_BUILD_FILTER=1 (test knob) forces it for every geometry the filter applies to.
    const char* bf_env = test_env("IDIST_BUILD_FILTER");
    bool build_filter = cfg.has_heuristic && !ext && tab16 && knobs.filter && filter_applies(ix) &&
                        (has_template_geometry(ix->L) || filt_stride(ix->L.stride) <= 128u * (uint32_t)kFiltRtChunks) &&   // (the thin tile)
                        (bf_env ? bf_env[0] != '0' : ix->L.stride >= 192u);
    if (build_filter) {
        CHK(filter_ensure(ix));
        build_filter = ix->filt_state.load(std::memory_order_acquire) == 1;
    }
    // (fat filtered descents for runtime-geometry rows beyond the thin tile: 500k x 1024 / 1536-d 2.04 / 2.78 against 2.10 / 3.13 s —
    //  profiles/probe_r06r_build_fat_filtered_1024.jsonl — not worth their compile time: those descents stay unfiltered)
    IndexView view = ix->view();
    if (!build_filter) view.f = FilterView{};
    // (one FAT filtered descent wave per SIMD, the search's layout for 768-d rows, builds 1M x 768 in 2.56 s against 2.37 s on the thin
    //  ones: the update stream needs the registers — profiles/probe_r06n_build_fat_768.jsonl)
    BuildArgs a{};
    a.top = top;
    a.efc = cfg.ef_construction;
    a.wcap = wcap;
    a.keep_pruned = cfg.keep_pruned ? 1u : 0u;
    a.has_heuristic = cfg.has_heuristic ? 1u : 0u;
    a.use_bloom = knobs.bloom ? 1u : 0u;
    a.visited = d_vis;
    a.vis = vg;
    a.edge_pid = d_edge_pid;
    a.edge_dist = d_edge_dist;
    a.head = d_head;
    a.next = d_next;
    a.touched = d_touched;
    a.nbr_dist = d_nbr_dist;
    a.row_nsel = d_row_nsel;
    a.nbr_aux = d_nbr_aux;
    a.slow = d_slow;
    a.rt = rt;
    a.n_touched = d_small;
    a.queue = d_small + 1;
    a.n_slow = d_small + 3;      // == &queue[2]
    a.status = d_small + 8;
    a.dlog_log = d_dlog_log;
    a.dlog_pd = d_dlog_pd;
    a.tab_log2 = tab_log2;
    a.tab16 = tab16 ? 1u : 0u;
    a.ubits = tab16 ? q16_universe_bits(n, tab_log2) : 0u;
    a.dl_shift = dl_shift;
    a.use_dlog = test_env("IDIST_BUILD_NO_DLOG") ? 0u : 1u;
    a.wbuf = d_wbuf;
    a.wcount = d_wcount;
    a.rt2 = rt2;
    a.fc = fc;
    a.tie_cap = tie_cap;
    a.tie_spill = d_tie_spill;
    a.tie_spill_cap = tie_spill ? n : 0u;
    if (const char* e = test_env("IDIST_BUILD_CHUNK")) a.chunk = (uint32_t)atoi(e);
    const size_t smemF = smem_bytes_update_fast(ix->L.stride);
    [[maybe_unused]] const bool classic = knobs.classic;   // (test build: IDIST_VARIANT_BUILD)
    const bool no_fast = test_env("IDIST_BUILD_NO_FAST") != nullptr;   // test knob: route every update through B2
    a.stats = d_stats;

    uint32_t cum[IDIST_MAX_LAYERS + 1];
    cum[0] = n;
    for (uint32_t l = 1; l <= top; l++) cum[l] = ix->layer_len[l - 1];

    // the set-up above ran on the null stream, which the pipeline's non-blocking streams do not wait for
    if (pipe) BCHK(hipDeviceSynchronize());
    hipStream_t stream = pipe ? s1 : nullptr;
    uint32_t* zbuf[2] = {ix->d_zero, pipe ? d_zero2 : ix->d_zero};     // copy k&1 holds the state after step k
    uint32_t* const smallS = d_small + (pipe ? 16 : 0);                // step A2/B/B2 counters (own stream)
    uint32_t* const d_status = d_small + (pipe ? 32 : 8);
    a.n_touched = smallS;
    a.status = d_status;
    uint64_t n_batches = 0;
    uint32_t prev_start = 0, prev_count = 0;
    BCHK(hipEventRecord(e0, stream));
    for (int layer = (int)top; layer >= 0; layer--) {                    // core/lib.rs:304
        const uint32_t end = cum[layer];
        uint32_t g = (uint32_t)layer == top ? 1u : std::max(cum[layer + 1], 1u);   // ranges, :275-281
        a.layer = (uint32_t)layer;
        bool first_of_layer = true;
        while (g < end) {
            // top layer is sequential in the reference (:313-314); below, at most `cap` inserts run
            // concurrently (:316-318) and never more than 1/32 of the graph they search (1/8 while the step is narrow).
            uint32_t B = 1;
            if (cap > 1 && (uint32_t)layer != top && g >= 64) B = std::min(cap, std::max(std::min(g / growth_div, quad_B), g / 32u));
            B = std::min(B, end - g);
            a.start = g;
            a.count = B;
            const uint64_t k = n_batches + 1;                              // step number, 1-based
            const int par = (int)(k & 1u);
            const bool alt = stream_mode == 2 || (stream_mode == 1 && B <= quad_B);   // this step on the extra streams
            // a sequential step (B = 1: the top layer, the first 63 points) depends on the whole previous step anyway: all of its
            // launches go to the update stream — same-stream boundaries (~2 us) instead of three cross-stream event hops (~27 us each
            // in the trace of profiles/r05/trace_chain_r05b_build_c2.json: 74 of the 179 us such a step took)
            // Only where every descent stream has its own queue head, visited bitmaps and tie bags (two_a): with ONE set of them (a build
            // on the HBM tie bags, IDIST_BUILD_STREAMS=off) the last sequential step's descent on s2 and the next, concurrent step's
            // descent on s1 — which waits for the step BEFORE it only — would share them.
            const bool seq1 = pipe && two_a && B == 1;
            hipStream_t sA = pipe ? (seq1 ? s2 : (two_a && alt && par ? s3 : s1)) : stream, sS = pipe ? s2 : stream;
            IndexView viewA = view, viewS = view;
            BuildArgs aA = a;
            if (pipe) {
                // a sequential (B = 1) step reads the previous step's state, a concurrent one the state before that.
                // The first step of a layer reads the previous step's state too: the snapshot of the layer above was
                // taken from it (s1 has waited for it already), and a descent that enters the zero layer through a
                // node of that last step must find its row, not the all-INVALID one of the older copy.
                const int lag = B > 1 && !first_of_layer ? 2 : 1;
                if (k > (uint64_t)lag) BCHK(hipStreamWaitEvent(sA, evS[(k - lag) & 1u], 0));
                viewA.zero = zbuf[(k - lag) & 1u];
                viewS.zero = zbuf[par];
                aA.queue = d_small + (two_a && par ? 48 : 1);             // step A queue head, one per descent stream
                if (two_a && par) aA.visited = d_vis + vis_words;
                aA.dlog_log = d_dlog_log + (((size_t)par * cap) << dl_shift);
                aA.dlog_pd = d_dlog_pd + (((size_t)par * cap) << (dl_shift + 1u));
                aA.wbuf = d_wbuf + (size_t)par * cap * cfg.ef_construction;
                aA.wcount = d_wcount + (size_t)par * cap;
                BCHK(hipMemsetAsync(aA.queue, 0, 4, sA));
            } else {
                BCHK(hipMemsetAsync(d_small, 0, 32, sA));                 // n_touched, queue heads, n_slow
            }
            const uint32_t gridA = std::min(B, std::min<uint32_t>(slots, (uint32_t)ix->n_cu * (pipe ? a_waves : std::min(a_waves_max, a_regs256 ? 8u : 4u))));
            const uint32_t gridB = (uint32_t)std::min<size_t>(std::min<size_t>((size_t)B * IDIST_M2, g), (size_t)ix->n_cu * 16);
            const uint32_t gridS = std::min<uint32_t>(gridB, (uint32_t)ix->n_cu * 6);
            const uint32_t gridA2 = std::min<uint32_t>(B, (uint32_t)ix->n_cu * 4);
            const uint32_t gridA2m = std::min<uint32_t>(B, (uint32_t)ix->n_cu * 2);
            BuildArgs aS = aA;                                             // same step-A outputs, own counters
            uint32_t* const cnt = smallS + (own_a2 && par ? 8 : 0);        // this step's n_touched / queue heads / n_slow
            aS.n_touched = cnt;
            aS.queue = cnt + 1;
            aS.n_slow = cnt + 3;
            aS.visited = d_vis;
            if (own_a2 && par) {                                           // this step's inboxes
                aS.edge_pid = d_edge_pid + n_edges; aS.edge_dist = d_edge_dist + n_edges; aS.next = d_next + n_edges;
                aS.head = d_head + n; aS.touched = d_touched + n_touch; aS.slow = d_slow + n_touch;
            }
            // (carry-over of the previous step's rows: its touched list)
            const uint32_t* const prev_touched = own_a2 ? d_touched + (par ? 0 : n_touch) : a.touched;
            const uint32_t* const prev_cnt = own_a2 ? smallS + (par ? 0 : 8) : smallS;
            hipStream_t sA2 = own_a2 && alt && !seq1 ? s4 : sS;
            BuildArgs af = aS;
            af.efc = no_fast ? 0u : cfg.ef_construction;                  // efc = 0 makes the fast kernel defer everything
#ifdef IDIST_VARIANTS
#define IDIST_VARIANT_BUILD(NB_, RS_, TAIL_)                                                                          \
    else if (classic && tab16) {                                                                                      \
        auto kA16 = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkClassic, 0, false, 1, true, false, true)>;    \
        IDIST_LAUNCH(kA16, gridA, 64, smem, sA, viewA, aA);                                                           \
    } else if (classic) {                                                                                             \
        auto kA = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkClassic, 0, false, 1, true)>;                   \
        IDIST_LAUNCH(kA, gridA, 64, smem, sA, viewA, aA);                                                             \
    }
#else
#define IDIST_VARIANT_BUILD(NB_, RS_, TAIL_)
#endif
#define LAUNCH_BUILD(NB_, RS_, TAIL_)                                                              \
    {                                                                                              \
        auto kAo = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true)>; \
        auto kAo16 = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true, false, true)>; \
        /* two descent waves per SIMD: 256 registers each, fewer rounds in flight per wave, more waves */ \
        /* narrow steps (the growth phase of a layer, max_batch = 1): four waves per insertion, like narrow search batches */ \
        auto kAq16 = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true, true, true)>; \
        auto kAo16w2 = build_insert_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, (NB_) == 24 ? 1 : ((NB_) == 4 ? 6 : 3), false, 2, true, false, true)>; \
        /* ... and with the reject filter in front of the distance passes (thin: one f32 round in flight, query fragment from LDS) */ \
        auto kF = build_update_fast_kernel<NB_, RS_, TAIL_>;                                       \
        auto kB = build_update_kernel<NB_, RS_, TAIL_>;                                            \
        auto kP = build_update_simple_kernel<NB_, RS_, TAIL_>;                                     \
        auto kA2 = build_select_kernel<NB_, RS_, TAIL_>;                                           \
        auto kX = build_extend_kernel<NB_, RS_, TAIL_>;                                            \
        auto kA2m = build_select_mfma_kernel<NB_, RS_, TAIL_>;                                     \
        if (ext) { IDIST_LAUNCH(kX, 1, 64, smemX, sA, viewA, aA, d_ext_work, ext_cap); }           \
        IDIST_VARIANT_BUILD(NB_, RS_, TAIL_)                                                       \
        else if (tab16 && a_quad && B <= quad_B) { IDIST_LAUNCH(kAq16, std::min(B, slots), 256, smem, sA, viewA, aA); } \
        else if (tab16 && build_filter) { launched_a = DescentLaunch<NB_, RS_, TAIL_, walk_thin_filter(2), GeoKernels<NB_>::thin_descent>::go(gridA, smem, sA, viewA, aA); } \
        else if (tab16 && a_regs256) { IDIST_LAUNCH(kAo16w2, gridA, 64, smem, sA, viewA, aA); }    \
        else if (tab16) { IDIST_LAUNCH(kAo16, gridA, 64, smem, sA, viewA, aA); }                   \
        else { IDIST_LAUNCH(kAo, gridA, 64, smem, sA, viewA, aA); }                                \
        if (pipe) {                                                                                \
            BCHK(hipEventRecord(evA[par], sA));                                                    \
            BCHK(hipStreamWaitEvent(sA2, evA[par], 0));                                            \
            /* this step's inbox set is the one step k - 1's carry-over reads its touched list from (evC, recorded below) */ \
            if (sA2 != s2 && k > 1) BCHK(hipStreamWaitEvent(sA2, evC, 0));                         \
            if (sA2 == s2) {                                                                       \
                IDIST_LAUNCH(copy_rows_kernel, 1024, 64, 0, s2, zbuf[par ^ 1], zbuf[par], prev_touched, prev_cnt, prev_start, prev_count); \
                if (own_a2) BCHK(hipEventRecord(evC, s2));                                         \
            }                                                                                      \
            BCHK(hipMemsetAsync(cnt, 0, 32, sA2));                                                 \
        }                                                                                          \
        if (ext) {                                                                                 \
        } else if (cfg.has_heuristic) {                                                            \
            if (a2_mfma) { IDIST_LAUNCH(kA2m, gridA2m, 256, smemA2m, sA2, viewS, aS); }            \
            else { IDIST_LAUNCH(kA2, gridA2, 64, smemA2, sA2, viewS, aS); }                        \
            if (pipe && sA2 != s2) {                                                               \
                BCHK(hipEventRecord(evA2[par], s4));                                               \
                BCHK(hipStreamWaitEvent(s2, evA2[par], 0));                                        \
                IDIST_LAUNCH(copy_rows_kernel, 1024, 64, 0, s2, zbuf[par ^ 1], zbuf[par], prev_touched, prev_cnt, prev_start, prev_count); \
                BCHK(hipEventRecord(evC, s2));                                                     \
            }                                                                                      \
            IDIST_LAUNCH(kF, gridB, 64, smemF, sS, viewS, af);                                     \
            IDIST_LAUNCH(kB, gridS, 64, smemB, sS, viewS, aS);                                     \
        } else {                                                                                   \
            IDIST_LAUNCH(kP, gridB, 64, (size_t)(72 * 8 + 64 * 4), sS, viewS, aS);                 \
        }                                                                                          \
    }
            bool launched_a = true;
            IDIST_DISPATCH(ix->L, LAUNCH_BUILD);
            if (!launched_a) { release(); return fail(IDIST_ERR_INTERNAL, "no filtered descent kernel for this row geometry"); }
#undef LAUNCH_BUILD
            prev_start = g;
            prev_count = B;
            g += B;
            n_batches++;
            first_of_layer = false;
            if (prog) IDIST_LAUNCH(progress_kernel, 1, 1, 0, sS, prog->slot, (unsigned long long)g, (unsigned long long)layer + 1ull);
            if (pipe) BCHK(hipEventRecord(evS[par], s2));
            if ((n_batches & 1023u) == 0) BCHK(hipGetLastError());
        }
        if (layer > 0) {                                                 // UpperNode::from_zero, :323-328
            const size_t total = (size_t)end * IDIST_M;
            const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
            // pipelined: behind the layer's last step on the update stream; the event the next steps' descents wait for
            // (the last step's) is recorded again behind it, so whichever stream they run on finds the snapshot taken
            IDIST_LAUNCH(snapshot_kernel, grid, 256, 0, pipe ? s2 : stream, zbuf[n_batches & 1u], ix->d_upper + ix->layer_off[layer - 1] * IDIST_M, end);
            if (pipe) BCHK(hipEventRecord(evS[n_batches & 1u], s2));       // (both the first and the second step of the next layer wait for this one)
        }
    }
    if (pipe) {
        if (n_batches) BCHK(hipStreamWaitEvent(s1, evS[n_batches & 1u], 0));
        if (test_env("IDIST_BUILD_CHECK")) {
            // self-check: carry the last step over as well; now the two copies must agree on every row, or some
            // step's carry-over missed a row
            const int par = (int)(n_batches & 1u);
            IDIST_LAUNCH(copy_rows_kernel, 1024, 64, 0, s1, zbuf[par], zbuf[par ^ 1], own_a2 && par ? d_touched + n_touch : a.touched,
                         smallS + (own_a2 && par ? 8 : 0), prev_start, prev_count);
            BCHK(hipMemsetAsync(d_small + 40, 0, 4, s1));
            IDIST_LAUNCH(count_row_mismatch_kernel, 1024, 256, 0, s1, zbuf[0], zbuf[1], n, d_small + 40);
            uint32_t bad = 0;
            BCHK(hipStreamSynchronize(s1));
            BCHK(hipMemcpy(&bad, d_small + 40, 4, hipMemcpyDeviceToHost));
            if (bad) {
                release();
                return fail(IDIST_ERR_INTERNAL, "pipelined build: the two copies of the zero layer differ on %u rows", bad);
            }
        }
        if (n_batches & 1u) std::swap(ix->d_zero, d_zero2);              // the final state lives in copy (last step)&1
    }
    BCHK(hipEventRecord(e1, stream));
    BCHK(hipGetLastError());
    BCHK(hipEventSynchronize(e1));
    float ms = 0;
    BCHK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t small[8] = {0};
    unsigned long long stats[32] = {0};
    BCHK(hipMemcpy(&small[6], d_status, 4, hipMemcpyDeviceToHost));
    BCHK(hipMemcpy(stats, d_stats, 256, hipMemcpyDeviceToHost));
#undef BCHK
    release();
    ix->stats.n_dist = stats[0];
    ix->stats.n_exp0 = stats[1];
    ix->stats.n_expU = stats[2];
    ix->stats.n_sel_pairs = stats[3];
    ix->stats.n_heur_rows = stats[4];
    ix->stats.n_updates = stats[5];
    ix->stats.n_batches = n_batches;
    ix->stats.seconds = ms * 1e-3;
    ix->stats.n_updates_fast = stats[6];
    ix->stats.n_updates_full = stats[7];
    ix->stats.n_filter_examined = stats[16];
    ix->stats.n_filter_rejected = stats[17];
    ix->stats.filter_row_bytes = build_filter ? filt_stride(ix->L.stride) : 0u;
    // the reference's own count exists only where every selection ran in the reference's order
    ix->stats.n_heur_ref = (ext || (cfg.has_heuristic && no_fast && !a2_mfma && cap == 1)) ? stats[8] : 0;
    if (prog) { prog->slot[0] = n; prog->slot[1] = 0; }
#ifdef IDIST_PROBE
    fprintf(stderr, "{\"probe\": \"step_B_lookups\", \"n\": %u, \"updates_single_new\": %llu, \"updates_general\": %llu, \"lookups\": %llu, "
            "\"misses\": %llu, \"misses_member_of_previous_step\": %llu, \"misses_member_of_this_step\": %llu, \"new_vs_new_rows\": %llu, "
            "\"n_updates\": %llu, \"n_updates_full\": %llu, \"n_sel_pairs\": %llu, \"n_heur_rows\": %llu}\n",
            n, stats[9], stats[10], stats[11], stats[12], stats[13], stats[14], stats[15], stats[5], stats[7], stats[3], stats[4]);
#endif
    ix->stats.tie_overflow = (small[6] & kStTieOverflow) ? 1 : 0;
    return device_status_to_code(small[6], cfg.tie_policy);
}

#ifdef IDIST_PROBE
// measurement build: the four-wave walk's segment ticks (g_quad_probe, idist_device.hpp); reset != 0 clears them after the read
extern "C" int idist_probe_quad(unsigned long long* out24, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_quad_probe), 24 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[24] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_quad_probe), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif

idist_status build_common(const void* points, bool on_device, uint32_t n, uint32_t dim, const idist_config* cfg,
                          int32_t device, idist_index** out) {
    CHK(validate_config(cfg, true));
    if (!points && n) return fail(IDIST_ERR_INVALID_ARG, "points is null");
    uint32_t cum[IDIST_MAX_LAYERS + 1];
    uint32_t nl = 1;
    cum[0] = n;
    if (n) {
        nl = layer_sizes(n, cfg->ml, cum, IDIST_MAX_LAYERS);
        if (nl == 0) return fail(IDIST_ERR_INVALID_ARG, "ml = %g yields more than %u layers", (double)cfg->ml, IDIST_MAX_LAYERS);
    }
    idist_progress* prog = g_watch;
    g_watch = nullptr;
    idist_config c = *cfg;
    const bool spill_first = test_env("IDIST_TIE_SPILL") && test_env("IDIST_TIE_SPILL")[0] == '1';   // test knob, see Knobs
    bool tie_spill = false;
    for (;;) {
        idist_index* ix = nullptr;
        CHK(index_alloc(n, dim, &c, cum + 1, n ? nl - 1 : 0, device, &ix));
        idist_status s = on_device ? load_points_device(ix, (const float*)points) : load_points_host(ix, (const float*)points);
        if (s == IDIST_OK) s = run_build(ix, prog, tie_spill);
        if (s == IDIST_OK) { *out = ix; return IDIST_OK; }
        idist_index_free(ix);
        // strict ties: the descent's tie region was too small -> build again with a larger one (x8, at most 4096 entries of
        // LDS), and when that is not enough either, with HBM bags behind it (unbounded, like the reference's heap)
        const uint32_t cap = tie_capacity(c);
        if (s != IDIST_ERR_TIE_OVERFLOW || tie_spill) return s;
        if (cap >= 4096u || spill_first) tie_spill = true;
        else c.tie_capacity = std::min<uint32_t>(4096u, cap * 8u);
    }
}

// A context belongs to ONE index: `ctx->idx == idx` alone would accept a new index that happens to live at the
// address of a freed one (its bitmaps are sized for the old n).
idist_status check_ctx(const idist_index* idx, const idist_search_ctx* ctx) {
    if (!idx || !ctx || ctx->idx != idx || ctx->idx_uid != idx->uid)
        return fail(IDIST_ERR_INVALID_ARG, "ctx does not belong to idx");
    return IDIST_OK;
}

// Back `want` query slots with visited bitmaps (Search::default() grows its scratch on first use too,
// core/lib.rs:363).  Existing slots are all-zero between launches, so growing is allocate + clear.
idist_status ensure_slots(idist_search_ctx* ctx, uint32_t want, hipStream_t stream) {
    if (want <= ctx->slots) return IDIST_OK;
    uint32_t s = 1;
    while (s < want) s <<= 1;                                    // few distinct sizes: powers of two up to the cap ...
    if (ctx->slots) s = std::max(s, ctx->slots * 8u);            // ... and at least 8x per step: growing synchronises the device
    const uint32_t cap = ctx->slots_req ? ctx->slots_req : default_slots(ctx->n_points, ctx->n_cu);
    s = std::max(std::min(s, cap), 1u);
    if (s <= ctx->slots) return IDIST_OK;
    const size_t bytes = std::max<size_t>((size_t)s * ctx->vis.slot_words * 4, 256);
    uint32_t* fresh = nullptr;
    HIPCHK(hipDeviceSynchronize());                              // nothing may still be walking on the old bitmaps
    HIPCHK(hipMalloc((void**)&fresh, bytes));
    hipError_t e = hipMemsetAsync(fresh, 0, bytes, stream);
    if (e != hipSuccess) { hipFree(fresh); return fail(IDIST_ERR_HIP, "visited bitmaps: %s", hipGetErrorString(e)); }
    hipFree(ctx->d_visited);
    ctx->d_visited = fresh;
    ctx->slots = s;
    return IDIST_OK;
}

// Which walk serves a wide batch (profiles/r02/probe_r02_ef_paths_onchip_vs_bitmap.jsonl, probe_r02_configs_c2_c4_c5.jsonl):
//   * a search visits ~53 * ef_search + 600 nodes; the 8192-id on-chip set is frozen at 7168 and the rest of the walk
//     test-and-sets the bitmap.  The fat on-chip waves still win while a row fetch outweighs that extra round trip:
//     up to ef_search ~ 180 at 128-d, ~ 550 at 300-d (id set; ~ 600 with the quotient set), beyond 200 at 768-d;
//   * an index that sits in the Infinity Cache (100k x 128: 51 MB) is served faster by 16 small waves per CU
//     (4.2 vs 5.0 ms per 10k queries): the on-chip walk is for HBM-resident indexes.
//   * round 4 (profiles/r04/probe_r04f_ef_crossover_c3.jsonl, probe_r04i_ef_paths_wide_merge_c3.jsonl): with the quotient set the
//     crossover at 300-d moved from ~1.5 to ~2 x row floats, and once the fat waves merged a `nearest` of up to 1024 entries in
//     one pass (w_push_merge<16>) the on-chip walk stayed ahead of the bitmap walk up to ef_search 1000 (0.651 / 0.637 / 0.625
//     vs 0.628 / 0.620 / 0.611 of spec at 650 / 800 / 1000; beyond the merge's reach too: 0.628 vs 0.565 at ef 1000 in
//     probe_r04k).  Rows of >= 256 floats: on chip as far as LDS allows; shorter rows: the line through the 128-d (~180) and
//     300-d (~600) crossovers.
//   * round 5 (profiles/r05/probe_r05m_walk_policy_cache_resident.jsonl, probe_r05n_walk_policy_by_size_and_row.jsonl: on-chip walk vs
//     bitmap walk at 9 more shapes): which walk wins near the Infinity Cache depends on the ROW LENGTH, not on residency alone.
//     Rows of >= 256 floats: the on-chip walk wins by 5-8 % also where the index sits in the cache (100k x 300: 7.97 vs 8.51 ms,
//     40k x 768: 17.2 vs 18.1 ms per 10k queries at ef 100; the same at ef 150-800).  Shorter rows: the bitmap walk wins well BEYOND
//     the 128-MB mark — 300k x 128 (154 MB): 4.39 vs 4.69 ms at ef 100, 6.03 vs 7.50 at ef 150; 400k x 200-d (320 MB): 7.49 vs 8.76 and
//     13.2 vs 17.8; 1M x 64-d (256 MB): 4.96 vs 6.84 — and loses from ~512 MB on (1M x 128: 6.73 vs 5.29 ms); and once such an index
//     is far beyond the cache the on-chip walk stays ahead to higher ef_search (1M / 2M x 128 at ef 200: 10.6 / 10.9 against 11.8 /
//     15.0 ms; 2M x 128 at ef 400: 23.0 against 27.5; 1M x 128 at ef 400: a tie).
//     Runtime-geometry rows shorter than 128 floats are different again: the fat waves' register tile (<8 blocks, 4 rounds>) keeps only
//     4 x 8 rows x 256 B = 8 KB on the wire per wave at 64-d, and the bitmap walk wins at EVERY size — 4M x 64 (1 GB): 5.93 / 10.7 /
//     20.1 ms against 7.13 / 14.9 / 31.0 ms at ef 100 / 200 / 400; at 100-d (3 blocks) it wins from ef ~150 on (3M x 100: 14.9 / 27.3
//     against 16.3 / 34.0 ms at ef 200 / 400, a tie at ef 100); 200-d rows behave like the 128-d instantiation (2M x 200: on chip 9.83 /
//     19.3 against 11.5 / 21.1 ms) — profiles/r05/probe_r05u_walk_policy_short_rt_rows_large.jsonl.
inline uint32_t on_chip_max_ef(uint32_t stride_floats, size_t index_bytes, bool runtime_geometry) {
    if (stride_floats >= 256u) return 1536u;       // (as far as W and the set fit a wave's LDS: tab_fit in launch_search)
    if (runtime_geometry && stride_floats <= 64u) return 0u;         // never: see above
    if (runtime_geometry && stride_floats < 128u) return 160u;
    const int ef = (int)stride_floats * 61 / 25 - 132;
    return (uint32_t)std::min(1536, std::max(index_bytes >= ((size_t)512 << 20) ? 400 : 160, ef));
}
// short rows (< 256 floats) are served by the bitmap walk up to this many bytes of point rows (see above)
constexpr size_t kShortRowBitmapBytes = (size_t)320 << 20;
// ... and long rows by the on-chip walk from this size on (below it nothing was measured: the round-2 rule stands)
constexpr size_t kLongRowOnChipBytes = (size_t)96 << 20;
constexpr uint32_t kFilterWaves = 2u;    // waves per SIMD of a filtered wide walk (launch_search)
constexpr uint32_t kLongWalkEf = 512u;   // ef_search from which wide on-chip batches run two thinner waves per SIMD (see launch_search)

// Long walks and where the visited bitmaps land (round 5, measured, nothing kept): ef_search 800 at 1M points test-and-sets the HBM
// bitmaps ~36k times per query, and five fresh contexts on ONE index in one process answer the same 10k queries in 72.5 / 72.9 / 79.1 /
// 82.3 / 82.4 ms while fresh replicas of the index behind fresh contexts change nothing — the spread between fresh processes is the
// CONTEXT's allocation, not the index's (profiles/r05/probe_r05b_placement_ef800_contexts_vs_replicas_c3.jsonl).  The pattern alone
// (scripts/micro/bitmap_placement.hip) is placement-independent; beside random row gathers it shows the same discrete levels (3 %): the
// 192 MB of hot bitmap lines and the row gathers compete for the 256-MiB Infinity Cache, and how well the bitmaps stay in it depends on
// physical placement.  Choosing among four candidate allocations by timing the real kernel (round 1's cure for the byte array) found a
// fast one on one box (four of five processes at 74-76 ms) and none on the next (five of five at 82.7 ms) for ~0.2 s per context: not
// kept.  Gathering the rows with the non-temporal hint (leaving the cache to the bitmaps) costs 11-15 % at every ef_search and 30 % of
// the build (`make nt`, profiles/r05/probe_r05e_rows_nontemporal_ab_c3.jsonl): the row gathers live on Infinity-Cache hits too.
// The compact copy of the rows behind the walk's reject filter (FilterView): made once per index, here.  The lattice [lo, lo + 255
// step] comes from a sample of the rows (mean +- 5 sigma of the coordinates, clipped to the sample's range): its choice decides how
// many candidates the filter can reject, never a result — a coordinate outside it is clamped and its error is part of the row's
// recorded |p - p^|.  Runs on the null stream and waits for it (an index is immutable once it is searched; its rows are in place).
// Geometries the search kernels have no filter tile for (compact rows beyond four 128-B chunks without a compile-time instantiation)
// and indexes the copy finds no memory for simply run without it.
bool filter_applies(const idist_index* ix) {
    return ix->n > 0 && (has_template_geometry(ix->L) || filt_stride(ix->L.stride) <= 128u * (uint32_t)kFiltRtChunksFat);
}
idist_status filter_ensure(const idist_index* ix) {
    if (ix->filt_state.load(std::memory_order_acquire) != 0) return IDIST_OK;
    std::lock_guard<std::mutex> lk(ix->filt_mu);
    if (ix->filt_state.load(std::memory_order_acquire) != 0) return IDIST_OK;
    if (!filter_applies(ix)) { ix->filt_state.store(2, std::memory_order_release); return IDIST_OK; }
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t stride = ix->L.stride, fs = filt_stride(stride);
    // sample: up to 2048 rows, evenly spaced
    const uint32_t ns = std::min<uint32_t>(ix->n, 2048u), every = ix->n / ns;
    std::vector<float> smp((size_t)ns * stride);
    HIPCHK(hipMemcpy2D(smp.data(), (size_t)stride * 4, ix->d_points, (size_t)every * stride * 4, (size_t)stride * 4, ns, hipMemcpyDeviceToHost));
    // robust range: mean and sigma of the central 98 % of the sampled coordinates (a few wild values — or heavy tails — must not
    // stretch the lattice: they are clamped, and only their own rows pay for it), sigma rescaled to the whole of a Gaussian
    std::vector<float> vals;
    vals.reserve((size_t)ns * ix->dim);
    for (uint32_t r = 0; r < ns; r++)
        for (uint32_t pos = 0; pos < stride; pos++) {
            if (natural_pos(pos, ix->L.nb) >= ix->dim) continue;
            const float v = smp[(size_t)r * stride + pos];
            if (std::fabs(v) <= 3.0e38f) vals.push_back(v);
        }
    double lo = 0.0, hi = 1.0;
    if (!vals.empty()) {
        const size_t cnt = vals.size(), k0 = cnt / 100, k1 = cnt - 1 - cnt / 100;
        std::nth_element(vals.begin(), vals.begin() + k0, vals.end());
        const float q01 = vals[k0];
        std::nth_element(vals.begin() + k0, vals.begin() + k1, vals.end());
        const float q99 = vals[k1];
        const auto mm = std::minmax_element(vals.begin(), vals.end());
        const double mn = *mm.first, mx = *mm.second;
        double sum = 0.0, sum2 = 0.0;
        size_t m = 0;
        for (const float v : vals)
            if (v >= q01 && v <= q99) { sum += v; sum2 += (double)v * v; m++; }
        const double mean = sum / (double)std::max<size_t>(m, 1);
        const double sd = std::sqrt(std::max(0.0, sum2 / (double)std::max<size_t>(m, 1) - mean * mean)) / 0.93;
        lo = std::max(mn - 0.25 * sd, mean - 5.0 * sd);
        hi = std::min(mx + 0.25 * sd, mean + 5.0 * sd);
    }
    if (!(hi > lo) || !((hi - lo) / 65280.0 > 1e-30)) hi = lo + 1.0;      // degenerate data: any lattice is as good
    FilterView f{};
    f.fstride = fs;
    f.lo = (float)lo;
    f.step256 = (float)((hi - lo) / 65280.0);                            // 255 steps of 256 sub-steps
    f.qscale = 1.0f / f.step256;
    f.dscale = f.step256 * f.step256;
    // every float step of the test (the conversion of I, dscale, the sums, the canonical chain of dim / 8 + 7 roundings) is covered
    f.up = 1.0f + 4.8828125e-4f + 6.0e-8f * (float)(ix->L.stride / 8u + 8u);
    f.slack = 4.8e-7f * std::sqrt((float)ix->dim);
    uint8_t* rows = nullptr;
    if (hipMalloc((void**)&rows, (size_t)ix->n * fs) != hipSuccess) {
        (void)hipGetLastError();
        ix->filt_state.store(2, std::memory_order_release);
        return IDIST_OK;
    }
    const uint32_t grid = std::min<uint32_t>(ix->n, (uint32_t)ix->n_cu * 32u);
    IDIST_LAUNCH(filter_rows_kernel, grid, 64, 0, nullptr, ix->d_points, ix->n, ix->dim, stride, ix->L.nb, rows, fs, f.lo, f.step256);
    if (hipDeviceSynchronize() != hipSuccess) {
        const hipError_t e = hipGetLastError();
        hipFree(rows);
        return fail(IDIST_ERR_HIP, "filter_rows_kernel failed: %s", hipGetErrorString(e));
    }
    f.rows = rows;
    ix->filt = f;
    ix->filt_state.store(1, std::memory_order_release);
    return IDIST_OK;
}

idist_status launch_search(const idist_index* ix, idist_search_ctx* ctx, const float* d_q, uint32_t nq,
                           uint32_t* d_pid, float* d_dist, uint32_t* d_cnt, uint32_t* d_ctr, hipStream_t stream,
                           uint32_t* status_host = nullptr, uint32_t* grid_out = nullptr, uint32_t* done_host = nullptr, uint32_t done_seq = 0) {
    const uint32_t ef = ix->cfg.ef_search;
    CHK(variants_check(ctx->knobs.classic));
    SearchArgs a{};
    a.queries = d_q;
    a.nq = nq;
    a.ef = ef;
    a.tie_cap = std::max(tie_capacity(ix->cfg), ctx->tie_cap);
    a.wcap = ef + 64 + a.tie_cap + 8;
    a.vis = ctx->vis;
    // Default walk: the visited set lives in LDS (an exact hash set of 2^tab_log2 ids per query, the HBM bitmap only
    // takes what does not fit), one wave per SIMD with up to 512 registers for rows in flight.  The largest set that
    // fits a quarter of the CU's LDS next to the query tile and W is used; none fits (huge ef_search) -> bitmap walk.
    uint32_t tab_fit = 0;                                               // largest set that fits at all
    for (uint32_t l = 13; l >= 10 && !tab_fit; l--)
        if (smem_bytes(ix->L.stride, a.wcap, false, 1u << l, a.vis.dirty_words) <= kOnChipLdsPerWave) tab_fit = l;
    if (tab_fit && ctx->knobs.tab_log2) tab_fit = std::min(tab_fit, ctx->knobs.tab_log2);
    if (ctx->knobs.vis_bitmap) tab_fit = 0;
    // Narrow batches (the reference's call pattern is ONE query per Hnsw::search): a four-wave workgroup per query,
    // one workgroup per CU — the rows of an expansion are fetched by all four SIMDs in one round trip.
    // Two such workgroups share a CU where the kernel's registers allow it (all but the 768-d instantiation): up to two
    // queries per CU the four-wave walk beats the single-wave one (512 queries: 0.65 vs 0.81 ms at C3), beyond it loses.
    const uint32_t quad_per_cu = (ix->L.nb == 24 && ix->L.rs == 0 && ix->L.tail == 0) ? 1u : 2u;
    const uint32_t quad_nq = ctx->knobs.quad_nq == 0xFFFFFFFFu ? (uint32_t)ix->n_cu * quad_per_cu : ctx->knobs.quad_nq;
    const bool quad = tab_fit && !ctx->knobs.classic && nq <= quad_nq;
    const size_t row_bytes = (size_t)ix->n * ix->L.stride * 4;
    const bool long_rows = ix->L.stride >= 256u;
    const bool rt_rows = !has_template_geometry(ix->L);   // no compile-time instantiation of the row geometry
    const bool cache_resident = long_rows ? row_bytes < kLongRowOnChipBytes : row_bytes <= kShortRowBitmapBytes;   // "served best by many small waves"
    // With the reject filter in front (round 6) the on-chip walk stays ahead at every ef_search the set fits — 1M x 128 at ef 200 / 400:
    // 8.5 / 15.4 ms against 10.6 / 20.8 ms on the bitmap walk, 4M x 64-d at ef 100 / 200: 5.1 / 10.3 against 5.9 / 11.8 — and the ef
    // caps of on_chip_max_ef only bind for unfiltered indexes (compact rows beyond the filter's tiles, IDIST_FILTER=0).  The
    // cache-residency rule shrinks too: filtered long rows take it at every size (20k x 300: 3.46 against 5.87 ms; 40k x 768 did before),
    // filtered short rows from the L2's reach on (8 x 4 MB: 100k x 128 = C2 3.44 against 3.80 ms at ef 100, 6.56 against 6.64 at ef 200,
    // a tie at 400; 300k x 128 3.65 against 4.45; 400k x 200-d 4.71 against 7.48) — below it the bitmap walk's sixteen small waves per
    // CU win (10k x 128: 1.98 against 2.83 ms, 50k x 64-d: 2.85 against 3.75) — profiles/probe_r06f_filter_policy.jsonl,
    // probe_r06l_c2_policy.jsonl.
    const bool filter_ok = ctx->knobs.filter && filter_applies(ix);
    const bool small_for_filter = !long_rows && row_bytes < ((size_t)32 << 20);
    const bool wide_on_chip = tab_fit && (ctx->knobs.vis_onchip || ctx->knobs.tab_log2 ||
                                          (filter_ok ? !small_for_filter : (ef <= on_chip_max_ef(ix->L.stride, row_bytes, rt_rows) && !cache_resident)));
    const bool on_chip = quad || wide_on_chip;
    uint32_t tab_log2 = on_chip ? tab_fit : 0u;
    // The reject filter in front of the wide on-chip walks' distance passes (its compact rows are made on first use).  A filtered
    // walk is bound by its dependent round trips, so it runs as `fw` thin waves per SIMD (walk_thin_filter, idist_device.hpp) with
    // the largest quotient set that lets 4 * fw of them share a CU's LDS — where the quotient form applies (n <= 33M) and the
    // caller did not pin the set's form or size.
    const bool use_filter = wide_on_chip && !quad && filter_ok;
    if (use_filter) CHK(filter_ensure(ix));
    const bool filtered = use_filter && ix->filt_state.load(std::memory_order_acquire) == 1;
    // (every other wide walk — unfiltered indexes, IDIST_TAB_FORMAT=ids, the classic test walks, n beyond the quotient form's reach —
    //  runs the round-5 kernels, compiled WITHOUT the filter: carrying its registers cost the 1024-d search 15 %)
    // Long rows (compact rows beyond four chunks: 768-d) keep ONE fat wave per SIMD with the filter: a thin wave's registers hold 16
    // of their compact rows and a quarter of an f32 row at a time — six or seven dependent round trips per expansion where the fat
    // wave makes two (C5, ef 200: 162 ms fat against 213 ms thin per 65,536 queries; C4 a tie at 78-83 ms).
    // (always on the quotient form of the set — the id form's cheaper probe was worth 1-2 % and another two large kernels to compile)
    const bool fat_filtered = filtered && !ctx->knobs.classic && !ctx->knobs.tab_ids && filt_stride(ix->L.stride) > 128u * (uint32_t)kFiltRtChunks &&
                              q16_applies(tab_fit, q16_universe_bits(ix->n, tab_fit));
    // ... except the 768-d instantiation while the id form of the set cannot fill up (ef_search <= ~120: C4)
    const bool fat_ids = fat_filtered && ix->L.nb == 24 && ix->L.rs == 0 && ix->L.tail == 0;
    uint32_t fw = 1;
    if (filtered && !fat_filtered && !ctx->knobs.classic && !ctx->knobs.tab_ids) fw = kFilterWaves;
    if (fw > 1) {
        uint32_t l = ctx->knobs.tab_log2 ? std::min(tab_fit, ctx->knobs.tab_log2) : tab_fit;
        // (a smaller set addresses fewer ids in its 16-bit quotients: 10M points need the 16-KB set — C5 runs six or seven thin
        //  waves per CU on it rather than eight on a set it cannot use)
        while (l > 10u && smem_bytes(ix->L.stride, a.wcap, false, 1u << l, a.vis.dirty_words, false) * 4u * fw > (size_t)160 * 1024 &&
               q16_applies(l - 1u, q16_universe_bits(ix->n, l - 1u)))
            l--;
        if (q16_applies(l, q16_universe_bits(ix->n, l))) tab_log2 = l;
        else fw = 1;
    }
    const bool thin = fw > 1;
    a.tab_log2 = tab_log2;
    // the set stores 16-bit quotients (twice the ids in the same LDS) whenever n allows it: up to 33M points with 32 KB
    a.ubits = on_chip ? q16_universe_bits(ix->n, tab_log2) : 0u;
    // ... and the walk can outgrow the id form: a search visits ~53 * ef_search + 600 nodes (C3 / C4 / C5 data alike); while that
    // stays below the 7/8 * 2^tab_log2 ids the plain set takes, the plain set never spills and its cheaper probe wins by
    // 1-2 % (ef_search = 100: 10.25 vs 10.42 ms per 10k queries at C3, profiles/r03/probe_r03a_ef_paths_*)
    const bool ids_suffice = 53u * ef + 600u <= (7u << tab_log2) / 8u;
    const bool q16 = on_chip && !ctx->knobs.tab_ids && q16_applies(tab_log2, a.ubits) &&
                     (thin || (fat_filtered && !(fat_ids && ids_suffice && !ctx->knobs.tab_q16 && !ctx->knobs.tab_log2)) || ctx->knobs.tab_q16 || ctx->knobs.tab_log2 || !ids_suffice);
    // Long walks (ef_search in the hundreds): an expansion costs a wave 9-10 us whatever it fetches, and it fetches fewer new rows
    // the longer the walk runs — more, thinner waves (two 256-register waves per SIMD, as many as the CU's LDS holds) keep more
    // expansions in flight than one fat wave per SIMD.
    // Measured at 300-d (profiles/r04/probe_r04k_ef_paths_two_waves_per_simd_c3.jsonl: ef 650 / 800 / 1000 at 0.704 / 0.690 / 0.651 of
    // spec against 0.653 / 0.639 / 0.628 for the fat waves, 0.787 against 0.795 at ef 400): from ef_search 512 on, for that row
    // geometry; other geometries keep the fat waves until they are measured (IDIST_W2_EF forces either way).
    // Round 5 (profiles/r05/probe_r05r_long_walks_two_waves_other_geometries.jsonl): the same holds for runtime-geometry rows the
    // 256-register tile keeps whole in flight (1M x 384-d: ef 600 / 800 at 79.1 / 105.1 ms against 88.5 / 114.6 ms, ef 400 within
    // 2 %); not for 768-d (fat waves 1 % ahead at ef 400-800) and not for 128-d (thin waves 43 % slower at ef 400).
    const bool w2_geometry = (ix->L.nb == 9 && ix->L.rs == 1 && ix->L.tail == 1) || (rt_rows && ix->L.stride >= 256u && ix->L.nb <= 12u);
    const uint32_t w2_from = ctx->knobs.w2_ef != 0xFFFFFFFFu ? ctx->knobs.w2_ef : (w2_geometry ? kLongWalkEf : 0xFFFFFFFFu);
    const bool w2 = on_chip && q16 && !quad && !thin && !ctx->knobs.classic && ef >= w2_from;
    const uint32_t w2_per_cu = (uint32_t)std::min<size_t>(8, (size_t)160 * 1024 / smem_bytes(ix->L.stride, a.wcap, false, 1u << std::max(tab_log2, 5u), a.vis.dirty_words));
    const uint32_t thin_per_cu = (uint32_t)std::min<size_t>(4u * fw, (size_t)160 * 1024 / smem_bytes(ix->L.stride, a.wcap, false, 1u << std::max(tab_log2, 5u), a.vis.dirty_words, false));
    uint32_t resident = (uint32_t)ix->n_cu * (quad ? 2u : (thin ? std::max(thin_per_cu, 4u) : (w2 ? std::max(w2_per_cu, 4u) : (on_chip ? 4u : 16u))));   // quad: two workgroups per CU where registers allow
    if (ctx->tie_spill) {
        // one bag of n keys per slot (a walk can hold every point as a tie at most once); the slots that fit 1 GiB
        const uint32_t fit = (uint32_t)std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)std::max(ix->n, 1u) * 8));
        resident = std::min(resident, fit);
        if (!ctx->d_tie_spill || ctx->spill_slots < std::min(nq, resident)) {
            HIPCHK(hipDeviceSynchronize());
            hipFree(ctx->d_tie_spill);
            ctx->d_tie_spill = nullptr;
            ctx->spill_slots = std::min(std::max(nq, 1u), resident);
            HIPCHK(hipMalloc((void**)&ctx->d_tie_spill, (size_t)ctx->spill_slots * std::max(ix->n, 1u) * 8));
        }
        resident = std::min(resident, ctx->spill_slots);
        a.tie_spill = ctx->d_tie_spill;
        a.tie_spill_cap = ix->n;
    }
    CHK(ensure_slots(ctx, std::min(nq, resident), stream));
    ctx->last_ef = ef;
    a.out_pid = d_pid;
    a.out_dist = d_dist;
    a.out_count = d_cnt;
    a.out_counters = d_ctr;
    a.visited = ctx->d_visited;
    a.next = ctx->d_next;
    a.status = ctx->d_next + 1;
    a.use_bloom = ctx->knobs.bloom ? 1u : 0u;
    // bitmap walk only: narrow batches cannot fill the chip and run its latency variant (same results)
    const bool lat = !on_chip && nq <= ctx->knobs.latency_nq &&
                     smem_bytes(ix->L.stride, a.wcap, false, kBloomLatWords, a.vis.dirty_words) <= 64 * 1024;
    const size_t smem = smem_bytes(ix->L.stride, a.wcap, false, on_chip ? (1u << tab_log2) : (lat ? kBloomLatWords : kBloomWords),
                                   a.vis.dirty_words, !thin);
    if (smem > 64 * 1024) return fail(IDIST_ERR_INVALID_ARG, "dim/ef_search need %zu B of LDS per wave (> 64 KiB)", smem);
    const uint32_t grid = std::min(std::min(nq, ctx->slots), resident);
    [[maybe_unused]] const bool classic = ctx->knobs.classic;   // (test build: IDIST_VARIANT_SEARCH_*)
    IndexView view = ix->view();
    if (!thin && !fat_filtered) view.f = FilterView{};
    a.queue_base = ctx->queue_base;
    a.status_host = status_host && grid <= idist_search_ctx::kIoStatusSlots ? status_host : nullptr;
    a.done_host = done_host;
    a.done_count = ctx->d_next + 2;
    a.filt_counts = reinterpret_cast<unsigned long long*>(ctx->d_next + 16);
    a.done_seq = done_seq;
    if (grid_out) *grid_out = grid;
    const uint32_t slot = (uint32_t)(ctx->n_launch % IDIST_EVENT_RING);
    if (ctx->knobs.events) HIPCHK(hipEventRecord(ctx->ev0[slot], stream));
// The `classic` walks (one distance round in flight, no adjacency prefetch) are selected by no policy: they exist as
// independent implementations of the same decisions for the parity tests and are compiled into the TEST build only
// (`make variants` -> libidist_variants.so, -DIDIST_VARIANTS; the CPU emulator build defines it too).
#ifdef IDIST_VARIANTS
#define IDIST_VARIANT_SEARCH_ONCHIP_Q16(NB_, RS_, TAIL_)                                                    \
    else if (on_chip && classic && q16) {                                                                   \
        auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkClassic, 0, false, 1, true, false, true)>;  \
        IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                                  \
    }
#define IDIST_VARIANT_SEARCH_ONCHIP_IDS(NB_, RS_, TAIL_)                                                    \
    else if (on_chip && classic) {                                                                          \
        auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkClassic, 0, false, 1, true)>;               \
        IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                                  \
    }
#define IDIST_VARIANT_SEARCH_BITMAP(NB_, RS_, TAIL_)                                                        \
    else if (classic) {                                                                                     \
        auto kS = search_kernel<NB_, RS_, TAIL_, kWalkClassic>;                                             \
        IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                                  \
    }
#else
#define IDIST_VARIANT_SEARCH_ONCHIP_Q16(NB_, RS_, TAIL_)
#define IDIST_VARIANT_SEARCH_ONCHIP_IDS(NB_, RS_, TAIL_)
#define IDIST_VARIANT_SEARCH_BITMAP(NB_, RS_, TAIL_)
#endif
    bool launched = true;
#define LAUNCH_SEARCH(NB_, RS_, TAIL_)                                                             \
    {                                                                                              \
        if (quad && q16) {                                                                         \
            auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true, true, true)>; \
            IDIST_LAUNCH(kS, grid, 256, smem, stream, view, a);                                    \
        } else if (quad) {                                                                         \
            auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true, true)>; \
            IDIST_LAUNCH(kS, grid, 256, smem, stream, view, a);                                    \
        } else if (thin) {                                                                         \
            launched = SearchLaunch<NB_, RS_, TAIL_, walk_thin_filter(2), GeoKernels<NB_>::thin_search>::go(grid, smem, stream, view, a); \
        } IDIST_VARIANT_SEARCH_ONCHIP_Q16(NB_, RS_, TAIL_) else if (w2) {                          \
            auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, (NB_) == 24 ? 1 : ((NB_) == 4 ? 6 : 3), false, 2, true, false, true)>; \
            IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                     \
        } else if (on_chip && fat_filtered && !q16) {                                              \
            launched = SearchLaunch<NB_, RS_, TAIL_, walk_with_filter(walk_code(kWalkOverlap, 0, false, 1, true)), GeoKernels<NB_>::fat_filtered_search_ids>::go(grid, smem, stream, view, a); \
        } else if (on_chip && fat_filtered) {                                                      \
            launched = SearchLaunch<NB_, RS_, TAIL_, walk_with_filter(walk_code(kWalkOverlap, 0, false, 1, true, false, true)), GeoKernels<NB_>::fat_filtered_search>::go(grid, smem, stream, view, a); \
        } else if (on_chip && q16) {                                                               \
            auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true, false, true)>; \
            IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                     \
        } IDIST_VARIANT_SEARCH_ONCHIP_IDS(NB_, RS_, TAIL_) else if (on_chip) {                     \
            auto kS = search_kernel<NB_, RS_, TAIL_, walk_code(kWalkOverlap, 0, false, 1, true)>;  \
            IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                     \
        } else if (lat) {                                                                          \
            auto kS = search_kernel<NB_, RS_, TAIL_, kWalkLatency>;                                \
            IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                     \
        } IDIST_VARIANT_SEARCH_BITMAP(NB_, RS_, TAIL_) else {                                      \
            auto kS = search_kernel<NB_, RS_, TAIL_, kWalkOverlap>;                                \
            IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                     \
        }                                                                                          \
    }
#ifdef IDIST_EA_PROBE
    // measurement build: wide 300-d batches on the id set with partial-distance early abandon after k blocks (dist_rounds_inflight)
    if (on_chip && !quad && !q16 && !classic && ctx->knobs.ea > 0 && ix->L.nb == 9 && ix->L.rs == 1 && ix->L.tail == 1) {
#define EA_CASE(K_)                                                                                   \
    case K_: {                                                                                        \
        auto kS = search_kernel<9, 1, 1, walk_with_ea(walk_code(kWalkOverlap, 0, false, 1, true), K_)>; \
        IDIST_LAUNCH(kS, grid, 64, smem, stream, view, a);                                            \
        break;                                                                                        \
    }
        switch (ctx->knobs.ea) {
            EA_CASE(4) EA_CASE(5) EA_CASE(6) EA_CASE(7)
            default: return fail(IDIST_ERR_INVALID_ARG, "IDIST_EA=%d: 4..7", ctx->knobs.ea);
        }
#undef EA_CASE
    } else
#endif
    IDIST_DISPATCH(ix->L, LAUNCH_SEARCH);
#undef LAUNCH_SEARCH
    if (!launched) return fail(IDIST_ERR_INTERNAL, "no filtered search kernel for this row geometry (stride %u)", ix->L.stride);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) {
        // nothing ran: put the queue head back where a fresh context has it
        hipMemsetAsync(ctx->d_next, 0, 4, stream);
        ctx->queue_base = 0;
        return fail(IDIST_ERR_HIP, "search launch failed: %s", hipGetErrorString(le));
    }
    ctx->queue_base += nq + grid;
    if (ctx->knobs.events) {
        HIPCHK(hipEventRecord(ctx->ev1[slot], stream));
        ctx->n_launch++;
        ctx->recs[ctx->n_rec++ % IDIST_EVENT_RING] = {(int32_t)slot, 0.0f};
    }
    return IDIST_OK;
}

}  // namespace

extern "C" {

const char* idist_last_error(void) { return g_err.c_str(); }
const char* idist_version(void) { return "instant-distance_amd 0.1 (gfx950)"; }

idist_status idist_device_count(int32_t* out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) cnt = 0;
    *out = cnt;
    return IDIST_OK;
}

idist_status idist_default_config(idist_config* cfg) {   // core/lib.rs:101-128
    if (!cfg) return fail(IDIST_ERR_INVALID_ARG, "cfg is null");
    cfg->ef_search = 100;
    cfg->ef_construction = 100;
    cfg->ml = 1.0f / logf((float)IDIST_M);
    cfg->has_heuristic = 1;
    cfg->extend_candidates = 0;
    cfg->keep_pruned = 1;
    cfg->metric = IDIST_METRIC_L2SQ;
    cfg->max_batch = 0;
    cfg->tie_policy = IDIST_TIES_STRICT;
    cfg->tie_capacity = 0;
    return IDIST_OK;
}

idist_status idist_layer_sizes(uint32_t n, float ml, uint32_t* cum, uint32_t cap, uint32_t* n_layers) {
    if (!cum || !n_layers) return fail(IDIST_ERR_INVALID_ARG, "null output");
    uint32_t tmp[IDIST_MAX_LAYERS + 1];
    const uint32_t nl = n ? layer_sizes(n, ml, tmp, IDIST_MAX_LAYERS) : 0;
    if (n && nl == 0) return fail(IDIST_ERR_INVALID_ARG, "more than %u layers", IDIST_MAX_LAYERS);
    if (nl > cap) return fail(IDIST_ERR_INVALID_ARG, "cum holds %u entries, %u needed", cap, nl);
    for (uint32_t i = 0; i < nl; i++) cum[i] = tmp[i];
    *n_layers = nl;
    return IDIST_OK;
}

idist_status idist_permutation(uint64_t seed, uint32_t n, uint32_t* out_pid, uint32_t* order) {
    // SmallRng (xoshiro256++) seeded through SplitMix64; PARITY UNPINNED, see idist.h
    uint64_t st = seed, s[4];
    for (int i = 0; i < 4; i++) {
        uint64_t z = (st += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        s[i] = z ^ (z >> 31);
    }
    auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
    auto next_u32 = [&]() {
        const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return (uint32_t)(result >> 32);
    };
    std::vector<std::pair<uint32_t, uint32_t>> sh(n);
    for (uint32_t i = 0; i < n; i++) {                       // core/lib.rs:257-259
        const uint64_t m = (uint64_t)next_u32() * n;
        uint32_t key = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - n)) {
            const uint32_t hi2 = (uint32_t)(((uint64_t)next_u32() * n) >> 32);
            if ((uint64_t)lo + hi2 > 0xFFFFFFFFull) key += 1;
        }
        sh[i] = {key, i};
    }
    std::sort(sh.begin(), sh.end());                         // sort_unstable, :260 (all pairs distinct)
    for (uint32_t i = 0; i < n; i++) {                       // :262-270
        if (out_pid) out_pid[sh[i].second] = i;
        if (order) order[i] = sh[i].second;
    }
    return IDIST_OK;
}

idist_status idist_index_build(const float* points, uint32_t n, uint32_t dim, const idist_config* cfg,
                               int32_t device, idist_index** out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    return build_common(points, false, n, dim, cfg, device, out);
}

idist_status idist_index_build_device(const void* d_points, uint32_t n, uint32_t dim, const idist_config* cfg,
                                      int32_t device, idist_index** out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    return build_common(d_points, true, n, dim, cfg, device, out);
}

idist_status idist_progress_new(idist_progress** out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    idist_progress* p = new idist_progress();
    void* m = nullptr;
    hipError_t e = hipHostMalloc(&m, 64, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { delete p; return fail(IDIST_ERR_HIP, "hipHostMalloc(progress): %s", hipGetErrorString(e)); }
    p->slot = reinterpret_cast<volatile unsigned long long*>(m);
    p->slot[0] = 0;
    p->slot[1] = 0;
    *out = p;
    return IDIST_OK;
}

void idist_progress_free(idist_progress* p) {
    if (!p) return;
    if (g_watch == p) g_watch = nullptr;
    if (p->slot) hipHostFree(const_cast<unsigned long long*>(p->slot));
    delete p;
}

idist_status idist_progress_watch_next_build(idist_progress* p) {
    g_watch = p;
    return IDIST_OK;
}

idist_status idist_progress_get(const idist_progress* p, uint64_t* done, uint64_t* total, int32_t* layer) {
    if (!p) return fail(IDIST_ERR_INVALID_ARG, "progress is null");
    if (done) *done = p->slot[0];
    if (total) *total = p->total;
    if (layer) *layer = (int32_t)p->slot[1] - 1;
    return IDIST_OK;
}

idist_status idist_index_build_stats(const idist_index* idx, idist_build_stats* out) {
    if (!idx || !out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    *out = idx->stats;
    return IDIST_OK;
}

idist_status idist_index_import(const float* points, uint32_t n, uint32_t dim, const idist_config* cfg,
                                const uint32_t* zero, const uint32_t* const* layers, const uint32_t* layer_len,
                                uint32_t n_upper, int32_t device, idist_index** out) {
    if (!out) return fail(IDIST_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    CHK(validate_config(cfg, false));
    if (n && (!points || !zero)) return fail(IDIST_ERR_INVALID_ARG, "points/zero is null");
    if (n_upper && (!layers || !layer_len)) return fail(IDIST_ERR_INVALID_ARG, "layers/layer_len is null");
    for (uint32_t l = 0; l < n_upper; l++)
        if (layer_len[l] == 0 || layer_len[l] > n || (l && layer_len[l] > layer_len[l - 1]))
            return fail(IDIST_ERR_BAD_GRAPH, "layer_len[%u] = %u: layers must nest (core/lib.rs:323-327)", l, layer_len[l]);
    idist_index* ix = nullptr;
    CHK(index_alloc(n, dim, cfg, layer_len, n_upper, device, &ix));
    auto bail = [&](idist_status s) { idist_index_free(ix); return s; };
    idist_status s = load_points_host(ix, points);
    if (s != IDIST_OK) return bail(s);
    if (n) {
        if (hipMemcpy(ix->d_zero, zero, (size_t)n * IDIST_M2 * 4, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(IDIST_ERR_HIP, "hipMemcpy(zero) failed"));
        for (uint32_t l = 0; l < n_upper; l++)
            if (hipMemcpy(ix->d_upper + ix->layer_off[l] * IDIST_M, layers[l], (size_t)layer_len[l] * IDIST_M * 4,
                          hipMemcpyHostToDevice) != hipSuccess)
                return bail(fail(IDIST_ERR_HIP, "hipMemcpy(layer %u) failed", l + 1));
        // validate against the reference's invariants
        uint32_t* d_bad = nullptr;
        if (hipMalloc((void**)&d_bad, 256) != hipSuccess || hipMemset(d_bad, 0, 256) != hipSuccess)
            return bail(fail(IDIST_ERR_HIP, "hipMalloc(validate) failed"));
        IDIST_LAUNCH(validate_rows_kernel, std::min<uint32_t>(n, 8192), 64, 0, (hipStream_t) nullptr, ix->d_zero, n, kM2, n, d_bad);
        for (uint32_t l = 0; l < n_upper; l++)
            IDIST_LAUNCH(validate_rows_kernel, std::min<uint32_t>(layer_len[l], 8192), 64, 0, (hipStream_t) nullptr,
                         ix->d_upper + ix->layer_off[l] * IDIST_M, layer_len[l], kM, layer_len[l], d_bad);
        uint32_t bad = 0;
        hipError_t e = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        hipFree(d_bad);
        if (e != hipSuccess) return bail(fail(IDIST_ERR_HIP, "validate: %s", hipGetErrorString(e)));
        if (bad)
            return bail(fail(IDIST_ERR_BAD_GRAPH,
                             "%u adjacency rows hold an id outside their layer or a duplicate id", bad));
    }
    *out = ix;
    return IDIST_OK;
}

idist_status idist_index_alloc(uint32_t n, uint32_t dim, const idist_config* cfg, const uint32_t* layer_len,
                               uint32_t n_upper, int32_t device, idist_index** out) {
    CHK(validate_config(cfg, false));
    if (n_upper && !layer_len) return fail(IDIST_ERR_INVALID_ARG, "layer_len is null");
    return index_alloc(n, dim, cfg, layer_len, n_upper, device, out);
}

idist_status idist_index_export(const idist_index* idx, uint32_t* zero, uint32_t* const* layers) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    HIPCHK(hipSetDevice(idx->device));
    if (idx->n) {
        if (!zero) return fail(IDIST_ERR_INVALID_ARG, "zero is null");
        HIPCHK(hipMemcpy(zero, idx->d_zero, (size_t)idx->n * IDIST_M2 * 4, hipMemcpyDeviceToHost));
    }
    for (uint32_t l = 0; l < idx->n_upper; l++) {
        if (!layers || !layers[l]) return fail(IDIST_ERR_INVALID_ARG, "layers[%u] is null", l);
        HIPCHK(hipMemcpy(layers[l], idx->d_upper + idx->layer_off[l] * IDIST_M, (size_t)idx->layer_len[l] * IDIST_M * 4,
                         hipMemcpyDeviceToHost));
    }
    return IDIST_OK;
}

idist_status idist_index_get_info(const idist_index* idx, idist_index_info* out) {
    if (!idx || !out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    memset(out, 0, sizeof(*out));
    out->n = idx->n;
    out->dim = idx->dim;
    out->row_stride = idx->L.stride;
    out->n_upper = idx->n_upper;
    out->ef_search = idx->cfg.ef_search;
    out->metric = idx->cfg.metric;
    out->device = idx->device;
    out->tie_capacity = tie_capacity(idx->cfg);
    for (uint32_t l = 0; l < idx->n_upper; l++) out->layer_len[l] = idx->layer_len[l];
    return IDIST_OK;
}

idist_status idist_index_device_buffers(const idist_index* idx, idist_device_buffers* out) {
    if (!idx || !out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    out->points = idx->d_points;
    out->points_bytes = (size_t)idx->n * idx->L.stride * 4;
    out->zero = idx->d_zero;
    out->zero_bytes = (size_t)idx->n * IDIST_M2 * 4;
    out->upper = idx->d_upper;
    out->upper_bytes = idx->upper_rows * IDIST_M * 4;
    return IDIST_OK;
}

idist_status idist_index_set_ef_search(idist_index* idx, uint32_t ef_search) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    if (ef_search > IDIST_MAX_EF) return fail(IDIST_ERR_INVALID_ARG, "ef_search %u > %u", ef_search, IDIST_MAX_EF);
    idx->cfg.ef_search = ef_search;
    return IDIST_OK;
}

void idist_index_free(idist_index* idx) {
    if (!idx) return;
    hipSetDevice(idx->device);
    for (idist_search_ctx* c : idx->comb_ctx) idist_search_ctx_free(c);
    hipFree(idx->d_points);
    hipFree(idx->d_zero);
    hipFree(idx->d_upper);
    hipFree(idx->d_layer_off);
    hipFree(const_cast<uint8_t*>(idx->filt.rows));
    delete idx;
}

idist_status idist_search_ctx_new(const idist_index* idx, uint32_t slots, idist_search_ctx** out) {
    if (!idx || !out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(idx->device));
    idist_search_ctx* c = new idist_search_ctx();
    c->idx = idx;
    c->idx_uid = idx->uid;
    c->slots_req = slots;
    c->vis = vis_geometry(idx->n);
    c->knobs = Knobs::from_env();
    c->tie_policy = idx->cfg.tie_policy;
    c->base_tie_cap = tie_capacity(idx->cfg);
    c->n_points = idx->n;
    c->stride = idx->L.stride;
    c->device = idx->device;
    c->n_cu = idx->n_cu;
    auto bail = [&](hipError_t e) {
        idist_search_ctx_free(c);
        return fail(IDIST_ERR_HIP, "search ctx allocation failed: %s", hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&c->d_next, 256)) != hipSuccess) return bail(e);
    if ((e = hipMemset(c->d_next, 0, 256)) != hipSuccess) return bail(e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail(e);
    for (uint32_t i = 0; i < IDIST_EVENT_RING; i++) {
        if ((e = hipEventCreate(&c->ev0[i])) != hipSuccess) return bail(e);
        if ((e = hipEventCreate(&c->ev1[i])) != hipSuccess) return bail(e);
    }
    // Search::default() is cheap (core/lib.rs:767-778): one slot = n/8 bytes now; the bitmaps of further slots
    // are allocated when a batch first needs them (ensure_slots), or right away if the caller named a count
    if (ensure_slots(c, slots ? slots : 1u, nullptr) != IDIST_OK) { idist_search_ctx_free(c); return IDIST_ERR_HIP; }
    // the scratch above was cleared on the null stream; a caller's non-blocking stream would not wait for it
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return bail(e);
    *out = c;
    return IDIST_OK;
}

void idist_search_ctx_free(idist_search_ctx* c) {
    if (!c) return;
    // (the index may already be gone: nothing of it is touched here)
    hipFree(c->d_visited);
    hipFree(c->d_tie_spill);
    hipFree(c->d_next);
    if (c->h_io) hipHostFree(c->h_io);
    hipFree(c->d_q);
    hipFree(c->d_pid);
    hipFree(c->d_dist);
    hipFree(c->d_cnt);
    hipFree(c->d_ctr);
    for (uint32_t i = 0; i < IDIST_EVENT_RING; i++) {
        if (c->ev0[i]) hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) hipEventDestroy(c->ev1[i]);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

idist_status idist_search_ctx_reserve(idist_search_ctx* ctx, uint32_t slots) {
    if (!ctx) return fail(IDIST_ERR_INVALID_ARG, "ctx is null");
    // the bitmaps belong on the context's own device whatever device the calling thread has current (replicas on several
    // GPUs); device, CU count and point count were copied at creation — the index itself is not touched (it may be gone)
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t cap = ctx->slots_req ? ctx->slots_req : 0xFFFFFFFFu;
    CHK(ensure_slots(ctx, std::min(std::max(slots, 1u), cap), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));                   // the new bitmaps are cleared before any stream may use them
    return IDIST_OK;
}

idist_status idist_search_batch_device(const idist_index* idx, idist_search_ctx* ctx, const void* d_queries,
                                       uint32_t nq, void* d_out_pid, void* d_out_dist, void* d_out_count,
                                       void* d_out_counters, void* hip_stream) {
    CHK(check_ctx(idx, ctx));
    if (nq == 0) return IDIST_OK;
    if (!d_queries || !d_out_count) return fail(IDIST_ERR_INVALID_ARG, "null device pointer");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t stream = (hipStream_t)hip_stream;
    if (idx->n == 0 || idx->cfg.ef_search == 0) {   // core/lib.rs:359-361
        HIPCHK(hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, stream));
        return IDIST_OK;
    }
    if (!d_out_pid || !d_out_dist) return fail(IDIST_ERR_INVALID_ARG, "null device pointer");
    return launch_search(idx, ctx, (const float*)d_queries, nq, (uint32_t*)d_out_pid, (float*)d_out_dist,
                         (uint32_t*)d_out_count, (uint32_t*)d_out_counters, stream);
}

idist_status idist_search_ctx_status(idist_search_ctx* ctx) {
    if (!ctx) return fail(IDIST_ERR_INVALID_ARG, "ctx is null");
    uint32_t st = 0;
    HIPCHK(hipMemcpy(&st, ctx->d_next + 1, 4, hipMemcpyDeviceToHost));
    if (st) HIPCHK(hipMemset(ctx->d_next + 1, 0, 4));
    if (st & kStTieOverflow) {
        ctx->tie_overflowed = true;
        // strict: later launches of this context get a larger tie region (x4, at most 4096 entries, LDS permitting); when that
        // is exhausted, HBM bags that take whatever the region cannot (unbounded, like the reference's heap)
        const uint32_t cap = std::max(ctx->base_tie_cap, ctx->tie_cap), next = std::min<uint32_t>(4096u, cap * 4u);
        if (ctx->tie_policy == IDIST_TIES_STRICT) {
            if (!ctx->knobs.tie_spill_first && next > cap &&
                smem_bytes(ctx->stride, ctx->last_ef + 64 + next + 8, false, kBloomWords, ctx->vis.dirty_words) <= 64 * 1024)
                ctx->tie_cap = next;
            else if (!ctx->tie_spill) ctx->tie_spill = true;
            else ctx->tie_escalation_exhausted = true;
        }
        g_tie_cap_msg = cap;
    }
    return device_status_to_code(st, ctx->tie_policy);
}

idist_status idist_search_ctx_filter_counts(idist_search_ctx* ctx, uint64_t* examined, uint64_t* rejected, int32_t reset) {
    if (!ctx) return fail(IDIST_ERR_INVALID_ARG, "ctx is null");
    unsigned long long c[2] = {0, 0};
    HIPCHK(hipMemcpy(c, ctx->d_next + 16, sizeof(c), hipMemcpyDeviceToHost));   // (waits for the null stream only: synchronise first)
    if (reset) HIPCHK(hipMemset(ctx->d_next + 16, 0, sizeof(c)));
    if (examined) *examined = c[0];
    if (rejected) *rejected = c[1];
    return IDIST_OK;
}

idist_status idist_search_ctx_tie_overflowed(idist_search_ctx* ctx, int32_t* out) {
    if (!ctx || !out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    CHK(idist_search_ctx_status(ctx));
    *out = ctx->tie_overflowed ? 1 : 0;
    ctx->tie_overflowed = false;
    return IDIST_OK;
}

static idist_status resolve_time(idist_search_ctx* ctx, const idist_search_ctx::TimeRec& r, float* ms) {
    if (r.slot < 0) { *ms = r.ms; return IDIST_OK; }
    HIPCHK(hipEventSynchronize(ctx->ev1[r.slot]));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0[r.slot], ctx->ev1[r.slot]));
    return IDIST_OK;
}

idist_status idist_search_ctx_last_kernel_ms(idist_search_ctx* ctx, float* ms) {
    if (!ctx || !ms) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    if (!ctx->n_rec) return fail(IDIST_ERR_INVALID_ARG, "no search kernel has been timed for this ctx (IDIST_KERNEL_EVENTS=0, or no call yet)");
    return resolve_time(ctx, ctx->recs[(ctx->n_rec - 1) % IDIST_EVENT_RING], ms);
}

idist_status idist_search_ctx_kernel_times(idist_search_ctx* ctx, float* ms, uint32_t cap, uint32_t* n_out) {
    if (!ctx || !ms || !n_out) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    const uint64_t have = std::min<uint64_t>(ctx->n_rec, IDIST_EVENT_RING);
    const uint32_t take = (uint32_t)std::min<uint64_t>(have, cap);
    for (uint32_t i = 0; i < take; i++) CHK(resolve_time(ctx, ctx->recs[(ctx->n_rec - take + i) % IDIST_EVENT_RING], &ms[i]));
    *n_out = take;
    return IDIST_OK;
}

static idist_status search_batch_impl(const idist_index* idx, idist_search_ctx* ctx, const float* queries, uint32_t nq,
                                      uint32_t* out_pid, float* out_dist, uint32_t* out_count, uint32_t* out_counters);

// A leader's launch for the scalar calls it took along (idist_combine.hpp).  Alone: on the leader's own context.  With
// company: on its slot's context, which grows with the batches (the leader's own may be a one-slot `Search`).
static void run_combined(const idist_index* idx, idist_search_ctx* ctx, std::vector<ScalarReq*>& b, int slot) {
    const uint32_t k = (uint32_t)b.size(), ef = idx->cfg.ef_search, dim = idx->dim;
    idist_status st = IDIST_OK;
    if (k == 1) {
        st = search_batch_impl(idx, ctx, b[0]->q, 1, b[0]->pid, b[0]->dist, b[0]->cnt, b[0]->ctr);
    } else if (!idx->comb_ctx[slot] && (st = idist_search_ctx_new(idx, kCombineBatch, &idx->comb_ctx[slot])) != IDIST_OK) {   // backed for a full batch at once: growing a context later would synchronise the device under everybody's feet
        // (reported to every caller of the batch below)
    } else {
        ctx = idx->comb_ctx[slot];
        std::vector<float> q((size_t)k * dim), dd((size_t)k * ef);
        std::vector<uint32_t> pid((size_t)k * ef), cnt(k), ctr((size_t)k * 3);
        for (uint32_t i = 0; i < k; i++) memcpy(q.data() + (size_t)i * dim, b[i]->q, (size_t)dim * 4);
        st = search_batch_impl(idx, ctx, q.data(), k, pid.data(), dd.data(), cnt.data(), ctr.data());
        if (st == IDIST_OK)
            for (uint32_t i = 0; i < k; i++) {
                memcpy(b[i]->pid, pid.data() + (size_t)i * ef, (size_t)ef * 4);
                memcpy(b[i]->dist, dd.data() + (size_t)i * ef, (size_t)ef * 4);
                b[i]->cnt[0] = cnt[i];
                if (b[i]->ctr) memcpy(b[i]->ctr, ctr.data() + (size_t)i * 3, 12);
            }
    }
    float kms = -1.0f;
    if (st == IDIST_OK && k > 1 && ctx->n_rec) (void)resolve_time(ctx, ctx->recs[(ctx->n_rec - 1) % IDIST_EVENT_RING], &kms);
    for (ScalarReq* r : b) {
        r->st = st;
        r->kernel_ms = kms;
        if (st != IDIST_OK) r->err = g_err;
    }
}

idist_status idist_search_batch(const idist_index* idx, idist_search_ctx* ctx, const float* queries, uint32_t nq,
                                uint32_t* out_pid, float* out_dist, uint32_t* out_count, uint32_t* out_counters) {
    CHK(check_ctx(idx, ctx));
    if (nq == 0) return IDIST_OK;
    if (!queries || !out_count) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    const uint32_t ef = idx->cfg.ef_search;
    if (idx->n == 0 || ef == 0) {
        memset(out_count, 0, (size_t)nq * 4);
        if (out_counters) memset(out_counters, 0, (size_t)nq * 12);
        return IDIST_OK;
    }
    if (!out_pid || !out_dist) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    // The reference's scalar call, possibly from many threads at once (one Search each, core/lib.rs:352-356): beyond eight
    // launches in flight on this index a call rides along in another thread's launch instead of making its own
    // (IDIST_TIES_DROP: idist_search_ctx_tie_overflowed is a per-context answer about the caller's OWN queries — such calls
    // always launch on their own context.  Strict ties: where the escalated region / the bags live is unobservable.)
    if (nq == 1 && ctx->knobs.combine && idx->cfg.tie_policy != IDIST_TIES_DROP) {
        ScalarReq r{queries, out_pid, out_dist, out_count, out_counters};
        idx->comb.submit(r, [&](std::vector<ScalarReq*>& b, int slot) { run_combined(idx, ctx, b, slot); });
        if (r.st != IDIST_OK) g_err = r.err;
        // a call that rode along (or led company) ran on a slot context: its kernel time is reported through the caller's own
        else if (r.kernel_ms >= 0.0f) ctx->recs[ctx->n_rec++ % IDIST_EVENT_RING] = {-1, r.kernel_ms};
        return r.st;
    }
    return search_batch_impl(idx, ctx, queries, nq, out_pid, out_dist, out_count, out_counters);
}

static idist_status search_batch_impl(const idist_index* idx, idist_search_ctx* ctx, const float* queries, uint32_t nq,
                                      uint32_t* out_pid, float* out_dist, uint32_t* out_count, uint32_t* out_counters) {
    const uint32_t ef = idx->cfg.ef_search;
    HIPCHK(hipSetDevice(idx->device));
    const size_t qb = (size_t)nq * idx->dim * 4, ob = (size_t)nq * ef * 4;
    // Narrow batches — the reference's call is ONE query per Hnsw::search: query and results cross PCIe through one
    // pinned, device-mapped buffer that the kernel reads and writes itself; the call is a host memcpy, one launch, one
    // stream sync, a host memcpy.  (The general path below costs six copy / memset calls of ~10 us each around the kernel.)
    const size_t io_need = qb + 2 * ob + (size_t)nq * 16 + idist_search_ctx::kIoHeadBytes;
    if (io_need <= idist_search_ctx::kIoMaxBytes && !ctx->knobs.no_zero_copy) {
        if (io_need > ctx->io_cap) {
            size_t cap = idist_search_ctx::kIoMinBytes;
            while (cap < io_need) cap <<= 1;
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->h_io) hipHostFree(ctx->h_io);
            ctx->h_io = nullptr;
            ctx->io_cap = 0;
            // coherent (fine-grained) on purpose: the host polls the completion word and reads the results WHILE the stream is
            // still busy — with a non-coherent mapping (HIP_HOST_COHERENT=0) they would only be visible at kernel end
            HIPCHK(hipHostMalloc((void**)&ctx->h_io, cap, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
            if (hipHostGetDevicePointer((void**)&ctx->d_io, ctx->h_io, 0) != hipSuccess) ctx->d_io = ctx->h_io;
            ctx->io_cap = cap;
        }
        uint8_t* hp = ctx->h_io;
        uint32_t* h_status = (uint32_t*)hp;                                   // [256]
        volatile uint32_t* h_done = (volatile uint32_t*)(hp + idist_search_ctx::kIoStatusSlots * 4);   // completion word
        float* h_q = (float*)(hp + idist_search_ctx::kIoHeadBytes);
        uint32_t* h_pid = (uint32_t*)((uint8_t*)h_q + qb);
        float* h_dist = (float*)((uint8_t*)h_pid + ob);
        uint32_t* h_cnt = (uint32_t*)((uint8_t*)h_dist + ob);
        uint32_t* h_ctr = h_cnt + nq;
        const ptrdiff_t dv = ctx->d_io - ctx->h_io;                           // host address -> device address of the same byte
        for (;;) {
            memcpy(h_q, queries, qb);
            memset(h_status, 0, idist_search_ctx::kIoStatusSlots * 4);
            *h_done = 0u;
            uint32_t grid = 0;
            const bool flag = ctx->knobs.sync_flag;
            const uint32_t seq = ++ctx->done_seq ? ctx->done_seq : ++ctx->done_seq;       // never 0: a fresh buffer reads 0
            const auto t0 = std::chrono::steady_clock::now();
            CHK(launch_search(idx, ctx, (const float*)((uint8_t*)h_q + dv), nq, (uint32_t*)((uint8_t*)h_pid + dv),
                              (float*)((uint8_t*)h_dist + dv), (uint32_t*)((uint8_t*)h_cnt + dv),
                              out_counters ? (uint32_t*)((uint8_t*)h_ctr + dv) : nullptr, ctx->stream,
                              (uint32_t*)((uint8_t*)h_status + dv), &grid, flag ? (uint32_t*)((uint8_t*)h_done + dv) : nullptr, seq));
            if (flag) {
                // The kernel's last workgroup writes `seq` into the pinned buffer behind its results.  One Search per thread
                // is the reference's model, so waiting must not burn a core per thread: sleep through the first half of what
                // such a call has been taking, then poll the word; a stream synchronisation only if it stays away for 2 s.
                bool overslept = false;
                if (ctx->call_ns_ema > 200e3) {
                    struct timespec ts = {0, (long)(ctx->call_ns_ema * 0.5)};
                    nanosleep(&ts, nullptr);
                    overslept = *h_done == seq;                       // done before we looked: the estimate is too long
                }
                bool seen = false;
                for (uint64_t spins = 0; !(seen = (*h_done == seq)); spins++) {
                    if (spins < 2048u) cpu_relax();
                    else std::this_thread::yield();               // 64+ waiting threads must not keep the leaders off the cores
                    if ((spins & 0x3FFFu) == 0x3FFFu &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (!seen) HIPCHK(hipStreamSynchronize(ctx->stream));
                const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
                if (overslept) ctx->call_ns_ema *= 0.7;
                else if (ctx->n_flag_calls++ > 0)                     // (the first call also allocated: not a sample)
                    ctx->call_ns_ema = ctx->call_ns_ema == 0.0 ? ns : 0.8 * ctx->call_ns_ema + 0.2 * ns;
            } else {
                HIPCHK(hipStreamSynchronize(ctx->stream));
            }
            uint32_t any = grid <= idist_search_ctx::kIoStatusSlots ? 0u : 1u;  // too many workgroups for the slots: ask the device
            for (uint32_t g = 0; g < grid && g < idist_search_ctx::kIoStatusSlots; g++) any |= h_status[g];
            if (any) {
                const uint32_t cap_before = std::max(tie_capacity(idx->cfg), ctx->tie_cap);
                const bool spill_before = ctx->tie_spill;
                const idist_status s = idist_search_ctx_status(ctx);
                if (s == IDIST_ERR_TIE_OVERFLOW && (std::max(tie_capacity(idx->cfg), ctx->tie_cap) > cap_before || ctx->tie_spill != spill_before)) continue;
                if (s != IDIST_OK) return s;
            }
            memcpy(out_pid, h_pid, ob);
            memcpy(out_dist, h_dist, ob);
            memcpy(out_count, h_cnt, (size_t)nq * 4);
            if (out_counters) memcpy(out_counters, h_ctr, (size_t)nq * 12);
            return IDIST_OK;
        }
    }
    if (qb > ctx->cap_q) { hipFree(ctx->d_q); ctx->d_q = nullptr; ctx->cap_q = 0; HIPCHK(hipMalloc((void**)&ctx->d_q, qb)); ctx->cap_q = qb; }
    if (ob > ctx->cap_out) {
        hipFree(ctx->d_pid); hipFree(ctx->d_dist); ctx->d_pid = nullptr; ctx->d_dist = nullptr; ctx->cap_out = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_pid, ob));
        HIPCHK(hipMalloc((void**)&ctx->d_dist, ob));
        ctx->cap_out = ob;
    }
    if (nq > ctx->cap_nq) {
        hipFree(ctx->d_cnt); hipFree(ctx->d_ctr); ctx->d_cnt = nullptr; ctx->d_ctr = nullptr; ctx->cap_nq = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_cnt, (size_t)nq * 4));
        HIPCHK(hipMalloc((void**)&ctx->d_ctr, (size_t)nq * 12));
        ctx->cap_nq = nq;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_q, queries, qb, hipMemcpyHostToDevice, ctx->stream));
    for (;;) {
        CHK(launch_search(idx, ctx, ctx->d_q, nq, ctx->d_pid, ctx->d_dist, ctx->d_cnt, out_counters ? ctx->d_ctr : nullptr,
                          ctx->stream));
        HIPCHK(hipMemcpyAsync(out_pid, ctx->d_pid, ob, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(out_dist, ctx->d_dist, ob, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(out_count, ctx->d_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (out_counters) HIPCHK(hipMemcpyAsync(out_counters, ctx->d_ctr, (size_t)nq * 12, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        // strict ties: the tie region was too small -> idist_search_ctx_status enlarged it for this context; the
        // batch is simply searched again (queries are independent and the results are overwritten)
        const uint32_t cap_before = std::max(tie_capacity(idx->cfg), ctx->tie_cap);
        const bool spill_before = ctx->tie_spill;
        const idist_status s = idist_search_ctx_status(ctx);
        if (s == IDIST_ERR_TIE_OVERFLOW && (std::max(tie_capacity(idx->cfg), ctx->tie_cap) > cap_before || ctx->tie_spill != spill_before)) continue;
        return s;
    }
}

// ---- several GPUs of one node: replicate once, shard the queries ----
idist_status idist_replicate(const idist_index* root, const int32_t* devices, uint32_t n_devices, idist_index** replicas) {
    if (!root || !replicas || (n_devices && !devices)) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < n_devices; i++) replicas[i] = nullptr;
    auto undo = [&](idist_status s) {
        const std::string keep = g_err;
        for (uint32_t i = 0; i < n_devices; i++) { idist_index_free(replicas[i]); replicas[i] = nullptr; }
        hipSetDevice(root->device);
        g_err = keep;
        return s;
    };
    std::vector<hipStream_t> streams(n_devices, nullptr);
    auto drop_streams = [&]() {
        for (uint32_t i = 0; i < n_devices; i++)
            if (streams[i]) { hipSetDevice(devices[i]); hipStreamDestroy(streams[i]); }
    };
    const size_t pb = (size_t)root->n * root->L.stride * 4, zb = (size_t)root->n * IDIST_M2 * 4, ub = root->upper_rows * IDIST_M * 4;
    // make sure nothing is still writing the root's buffers (a build on another stream)
    if (hipSetDevice(root->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return fail(IDIST_ERR_HIP, "replicate: root device %d: %s", root->device, hipGetErrorString(hipGetLastError()));
    for (uint32_t i = 0; i < n_devices; i++) {
        idist_index* r = nullptr;
        idist_status st = index_alloc(root->n, root->dim, &root->cfg, root->layer_len, root->n_upper, devices[i], &r);   // sets the device
        if (st != IDIST_OK) { drop_streams(); return undo(st); }
        replicas[i] = r;
        r->stats = root->stats;
        hipError_t e = hipSuccess;
        if (devices[i] != root->device) {
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, devices[i], root->device);
            if (e == hipSuccess && can) {
                e = hipDeviceEnablePeerAccess(root->device, 0);       // direct xGMI reads; already-enabled is fine
                if (e != hipSuccess) { (void)hipGetLastError(); e = hipSuccess; }
            }
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
        // all destinations' copies are queued before any is waited for: the root streams to its peers in parallel
        // (xGMI is point to point: one link per destination)
        if (e == hipSuccess && pb) e = hipMemcpyPeerAsync(r->d_points, devices[i], root->d_points, root->device, pb, streams[i]);
        if (e == hipSuccess && zb) e = hipMemcpyPeerAsync(r->d_zero, devices[i], root->d_zero, root->device, zb, streams[i]);
        if (e == hipSuccess && ub) e = hipMemcpyPeerAsync(r->d_upper, devices[i], root->d_upper, root->device, ub, streams[i]);
        if (e != hipSuccess) {
            fail(IDIST_ERR_HIP, "replicate to device %d: %s", devices[i], hipGetErrorString(e));
            drop_streams();
            return undo(IDIST_ERR_HIP);
        }
    }
    for (uint32_t i = 0; i < n_devices; i++) {
        hipSetDevice(devices[i]);
        hipError_t e = hipStreamSynchronize(streams[i]);
        if (e != hipSuccess) {
            fail(IDIST_ERR_HIP, "replicate to device %d: %s", devices[i], hipGetErrorString(e));
            drop_streams();
            return undo(IDIST_ERR_HIP);
        }
    }
    drop_streams();
    hipSetDevice(root->device);
    return IDIST_OK;
}

// ---- the same replication as RCCL broadcasts (one process, one communicator over the devices involved) ----
#ifndef IDIST_EMU
#include <dlfcn.h>
namespace {
// the six RCCL entry points this file needs, resolved on first use: libidist.so carries no link-time dependency on
// librccl (a process that never replicates never loads it; a process that already holds one — torch's — shares it)
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
    Rccl() {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p && why.empty()) why = std::string("librccl: no symbol ") + n; return p; };
        CommInitAll = (int (*)(comm_t*, int, const int*))sym("ncclCommInitAll");
        CommDestroy = (int (*)(comm_t))sym("ncclCommDestroy");
        GroupStart = (int (*)())sym("ncclGroupStart");
        GroupEnd = (int (*)())sym("ncclGroupEnd");
        Broadcast = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))sym("ncclBroadcast");
        GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        ok = why.empty();
    }
    static Rccl& get() { static Rccl r; return r; }
};
constexpr int kNcclUint8 = 1;   // ncclDataType_t::ncclUint8 (rccl.h)
}  // namespace
#endif

idist_status idist_replicate_rccl(const idist_index* root, const int32_t* devices, uint32_t n_devices, idist_index** replicas,
                                  double* seconds) {
    if (!root || !replicas || (n_devices && !devices)) return fail(IDIST_ERR_INVALID_ARG, "null argument");
    if (seconds) *seconds = 0.0;
#ifdef IDIST_EMU
    return idist_replicate(root, devices, n_devices, replicas);       // (the CPU emulator of tests/simt has no RCCL: same result)
#else
    for (uint32_t i = 0; i < n_devices; i++) replicas[i] = nullptr;
    if (n_devices == 0) return IDIST_OK;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(IDIST_ERR_UNSUPPORTED, "idist_replicate_rccl: %s", R.why.c_str());
    // ranks: the root's device first, then every other distinct destination device
    std::vector<int> devs{root->device};
    for (uint32_t i = 0; i < n_devices; i++)
        if (std::find(devs.begin(), devs.end(), (int)devices[i]) == devs.end()) devs.push_back((int)devices[i]);
    const int nr = (int)devs.size();
    std::vector<Rccl::comm_t> comms(nr, nullptr);
    std::vector<hipStream_t> streams(nr, nullptr);
    std::vector<int> first(nr, -1);                                   // the replica each rank receives into
    auto cleanup = [&](idist_status s) {
        const std::string keep = g_err;
        for (int r = 0; r < nr; r++) {
            if (streams[r]) { hipSetDevice(devs[r]); hipStreamDestroy(streams[r]); }
            if (comms[r]) R.CommDestroy(comms[r]);
        }
        if (s != IDIST_OK)
            for (uint32_t i = 0; i < n_devices; i++) { idist_index_free(replicas[i]); replicas[i] = nullptr; }
        hipSetDevice(root->device);
        g_err = keep;
        return s;
    };
    // nothing may still be writing the root's buffers (a build on another stream)
    if (hipSetDevice(root->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return fail(IDIST_ERR_HIP, "replicate: root device %d: %s", root->device, hipGetErrorString(hipGetLastError()));
    for (uint32_t i = 0; i < n_devices; i++) {
        idist_index* r = nullptr;
        const idist_status st = index_alloc(root->n, root->dim, &root->cfg, root->layer_len, root->n_upper, devices[i], &r);
        if (st != IDIST_OK) return cleanup(st);
        replicas[i] = r;
        r->stats = root->stats;
        const int rk = (int)(std::find(devs.begin(), devs.end(), (int)devices[i]) - devs.begin());
        if (first[rk] < 0) first[rk] = (int)i;
    }
    int rc = R.CommInitAll(comms.data(), nr, devs.data());
    if (rc != 0) return cleanup(fail(IDIST_ERR_HIP, "ncclCommInitAll over %d devices: %s", nr, R.GetErrorString(rc)));
    for (int r = 0; r < nr; r++)
        if (hipSetDevice(devs[r]) != hipSuccess || hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking) != hipSuccess)
            return cleanup(fail(IDIST_ERR_HIP, "replicate: stream on device %d: %s", devs[r], hipGetErrorString(hipGetLastError())));
    const size_t pb = (size_t)root->n * root->L.stride * 4, zb = (size_t)root->n * IDIST_M2 * 4, ub = root->upper_rows * IDIST_M * 4;
    struct Buf { const void* src; size_t bytes; int which; };
    const Buf bufs[3] = {{root->d_points, pb, 0}, {root->d_zero, zb, 1}, {root->d_upper, ub, 2}};
    auto dst_of = [&](const idist_index* x, int which) -> void* {
        return which == 0 ? (void*)x->d_points : (which == 1 ? (void*)x->d_zero : (void*)x->d_upper);
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (const Buf& b : bufs) {
        if (!b.bytes) continue;
        if ((rc = R.GroupStart()) != 0) return cleanup(fail(IDIST_ERR_HIP, "ncclGroupStart: %s", R.GetErrorString(rc)));
        for (int r = 0; r < nr && rc == 0; r++) {
            // rank 0 = the root: it sends its own buffer and, if a replica lives on its device, receives into that one
            void* recv = first[r] >= 0 ? dst_of(replicas[first[r]], b.which) : (void*)b.src;
            rc = R.Broadcast(b.src, recv, b.bytes, kNcclUint8, 0, comms[r], streams[r]);   // (sendbuff is read on the root only)
        }
        const int rc2 = R.GroupEnd();
        if (rc != 0 || rc2 != 0) return cleanup(fail(IDIST_ERR_HIP, "ncclBroadcast: %s", R.GetErrorString(rc ? rc : rc2)));
    }
    for (int r = 0; r < nr; r++)
        if (hipSetDevice(devs[r]) != hipSuccess || hipStreamSynchronize(streams[r]) != hipSuccess)
            return cleanup(fail(IDIST_ERR_HIP, "replicate: broadcast to device %d: %s", devs[r], hipGetErrorString(hipGetLastError())));
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // further replicas on a device that already holds one: plain device-to-device copies
    for (uint32_t i = 0; i < n_devices; i++) {
        const int rk = (int)(std::find(devs.begin(), devs.end(), (int)devices[i]) - devs.begin());
        if (first[rk] == (int)i) continue;
        hipSetDevice(devices[i]);
        for (const Buf& b : bufs)
            if (b.bytes && hipMemcpy(dst_of(replicas[i], b.which), dst_of(replicas[first[rk]], b.which), b.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                return cleanup(fail(IDIST_ERR_HIP, "replicate: copy on device %d: %s", devices[i], hipGetErrorString(hipGetLastError())));
    }
    return cleanup(IDIST_OK);
#endif
}

idist_status idist_search_batch_sharded(const idist_index* const* replicas, idist_search_ctx* const* ctxs, uint32_t n_shards,
                                        const float* queries, uint32_t nq, uint32_t* out_pid, float* out_dist,
                                        uint32_t* out_count, uint32_t* out_counters) {
    if (!replicas || !ctxs || n_shards == 0) return fail(IDIST_ERR_INVALID_ARG, "no shards");
    for (uint32_t i = 0; i < n_shards; i++) {
        CHK(check_ctx(replicas[i], ctxs[i]));
        if (replicas[i]->n != replicas[0]->n || replicas[i]->dim != replicas[0]->dim ||
            replicas[i]->cfg.ef_search != replicas[0]->cfg.ef_search)
            return fail(IDIST_ERR_INVALID_ARG, "shard %u is not a replica of shard 0 (n, dim or ef_search differ)", i);
        for (uint32_t j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return fail(IDIST_ERR_INVALID_ARG, "shards %u and %u share a search context", j, i);
    }
    if (nq == 0) return IDIST_OK;
    if (!queries || !out_count) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    const uint32_t ef = replicas[0]->cfg.ef_search, dim = replicas[0]->dim;
    std::vector<idist_status> st(n_shards, IDIST_OK);
    std::vector<std::string> msg(n_shards);
    auto work = [&](uint32_t i) {
        // contiguous block partition, the reference's own concurrency model (callers split queries over threads)
        const uint32_t lo = (uint32_t)((uint64_t)nq * i / n_shards), hi = (uint32_t)((uint64_t)nq * (i + 1) / n_shards);
        if (hi == lo) return;
        st[i] = idist_search_batch(replicas[i], ctxs[i], queries + (size_t)lo * dim, hi - lo,
                                   out_pid ? out_pid + (size_t)lo * ef : nullptr, out_dist ? out_dist + (size_t)lo * ef : nullptr,
                                   out_count + lo, out_counters ? out_counters + (size_t)lo * 3 : nullptr);
        if (st[i] != IDIST_OK) msg[i] = g_err;                 // g_err is thread-local
    };
    std::vector<std::thread> threads;
    for (uint32_t i = 1; i < n_shards; i++) threads.emplace_back(work, i);
    work(0);
    for (auto& t : threads) t.join();
    for (uint32_t i = 0; i < n_shards; i++)
        if (st[i] != IDIST_OK) return fail(st[i], "shard %u (device %d): %s", i, replicas[i]->device, msg[i].c_str());
    return IDIST_OK;
}


idist_status idist_distance_batch(const idist_index* idx, const float* queries, uint32_t nq, const uint32_t* ids,
                                  uint32_t n_ids, float* out_dist) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    if (nq == 0 || n_ids == 0) return IDIST_OK;
    if (!queries || !ids || !out_dist) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    HIPCHK(hipSetDevice(idx->device));
    float *d_q = nullptr, *d_out = nullptr;
    uint32_t* d_ids = nullptr;
    const size_t qb = (size_t)nq * idx->dim * 4, ib = (size_t)nq * n_ids * 4;
    auto release = [&]() { hipFree(d_q); hipFree(d_out); hipFree(d_ids); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d_q, qb)) != hipSuccess || (e = hipMalloc((void**)&d_out, ib)) != hipSuccess ||
        (e = hipMalloc((void**)&d_ids, ib)) != hipSuccess ||
        (e = hipMemcpy(d_q, queries, qb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_ids, ids, ib, hipMemcpyHostToDevice)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "distance_batch staging: %s", hipGetErrorString(e));
    }
    const uint32_t chunks = (n_ids + 63u) / 64u;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)nq * chunks, 1u << 20);
    const size_t smem = smem_bytes(idx->L.stride, 0, false);
    IndexView view = idx->view();
#define LAUNCH_DIST(NB_, RS_, TAIL_)                                                              \
    {                                                                                             \
        auto kD = distance_batch_kernel<NB_, RS_, TAIL_>;                                         \
        IDIST_LAUNCH(kD, grid, 64, smem, (hipStream_t) nullptr, view, d_q, nq, d_ids, n_ids, d_out); \
    }
    IDIST_DISPATCH(idx->L, LAUNCH_DIST);
#undef LAUNCH_DIST
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out_dist, d_out, ib, hipMemcpyDeviceToHost)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "distance_batch: %s", hipGetErrorString(e));
    }
    release();
    return IDIST_OK;
}

idist_status idist_filter_bound_batch(const idist_index* idx, const float* queries, uint32_t nq, const uint32_t* ids,
                                      uint32_t n_ids, float* out_bound) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    if (nq == 0 || n_ids == 0) return IDIST_OK;
    if (!queries || !ids || !out_bound) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    HIPCHK(hipSetDevice(idx->device));
    if (filter_applies(idx)) CHK(filter_ensure(idx));
    const size_t qb = (size_t)nq * idx->dim * 4, ib = (size_t)nq * n_ids * 4;
    if (idx->filt_state.load(std::memory_order_acquire) != 1) {          // no filter for this index: no bound
        memset(out_bound, 0, ib);
        return IDIST_OK;
    }
    float *d_q = nullptr, *d_out = nullptr;
    uint32_t* d_ids = nullptr;
    auto release = [&]() { hipFree(d_q); hipFree(d_out); hipFree(d_ids); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d_q, qb)) != hipSuccess || (e = hipMalloc((void**)&d_out, ib)) != hipSuccess ||
        (e = hipMalloc((void**)&d_ids, ib)) != hipSuccess ||
        (e = hipMemcpy(d_q, queries, qb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_ids, ids, ib, hipMemcpyHostToDevice)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "filter_bound_batch staging: %s", hipGetErrorString(e));
    }
    const uint32_t chunks = (n_ids + 63u) / 64u;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)nq * chunks, 1u << 20);
    const size_t smem = smem_bytes(idx->L.stride, 0, false);
    IndexView view = idx->view();
#define LAUNCH_FB(NB_, RS_, TAIL_)                                                               \
    {                                                                                             \
        auto kD = filter_bound_kernel<NB_, RS_, TAIL_>;                                           \
        IDIST_LAUNCH(kD, grid, 64, smem, (hipStream_t) nullptr, view, d_q, nq, d_ids, n_ids, d_out); \
    }
    IDIST_DISPATCH(idx->L, LAUNCH_FB);
#undef LAUNCH_FB
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out_bound, d_out, ib, hipMemcpyDeviceToHost)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "filter_bound_batch: %s", hipGetErrorString(e));
    }
    release();
    return IDIST_OK;
}

static idist_status bruteforce_scan(const idist_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                    uint32_t* out_pid, float* out_dist) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    if (nq == 0) return IDIST_OK;
    if (k == 0 || k > IDIST_MAX_EF) return fail(IDIST_ERR_INVALID_ARG, "k %u out of [1,%u]", k, IDIST_MAX_EF);
    if (!queries || !out_pid || !out_dist) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    HIPCHK(hipSetDevice(idx->device));
    float *d_q = nullptr, *d_dist = nullptr;
    uint32_t *d_pid = nullptr, *d_next = nullptr;
    const size_t qb = (size_t)nq * idx->dim * 4, ob = (size_t)nq * k * 4;
    auto release = [&]() { hipFree(d_q); hipFree(d_dist); hipFree(d_pid); hipFree(d_next); };
    hipError_t e;
    if ((e = hipMalloc((void**)&d_q, qb)) != hipSuccess || (e = hipMalloc((void**)&d_dist, ob)) != hipSuccess ||
        (e = hipMalloc((void**)&d_pid, ob)) != hipSuccess || (e = hipMalloc((void**)&d_next, 256)) != hipSuccess ||
        (e = hipMemset(d_next, 0, 256)) != hipSuccess ||
        (e = hipMemcpy(d_q, queries, qb, hipMemcpyHostToDevice)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "bruteforce staging: %s", hipGetErrorString(e));
    }
    const uint32_t wcap = k + 64 + 8;
    const size_t smem = smem_bytes(idx->L.stride, wcap, false);
    const uint32_t grid = std::min<uint32_t>(nq, (uint32_t)idx->n_cu * 16);
    IndexView view = idx->view();
#define LAUNCH_BF(NB_, RS_, TAIL_)                                                                          \
    {                                                                                                       \
        auto kF = bruteforce_kernel<NB_, RS_, TAIL_>;                                                       \
        IDIST_LAUNCH(kF, grid, 64, smem, (hipStream_t) nullptr, view, d_q, nq, k, wcap, d_pid, d_dist, d_next); \
    }
    IDIST_DISPATCH(idx->L, LAUNCH_BF);
#undef LAUNCH_BF
    if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out_pid, d_pid, ob, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(out_dist, d_dist, ob, hipMemcpyDeviceToHost)) != hipSuccess) {
        release();
        return fail(IDIST_ERR_HIP, "bruteforce: %s", hipGetErrorString(e));
    }
    release();
    return IDIST_OK;
}


// Wide-batch exact k-NN: f32-MFMA -2QP^T filter + canonical re-rank (idist_mfma.hpp).  Returns
// IDIST_ERR_INTERNAL with *fell_back = 1 if a candidate list overflowed (caller rescans).
static idist_status bruteforce_mfma(const idist_index* idx, const float* queries, uint32_t nq, uint32_t k,
                                    uint32_t* out_pid, float* out_dist, int* fell_back) {
    *fell_back = 0;
    const uint32_t n = idx->n, stride = idx->L.stride;
    uint32_t S = 32768;
    if (const char* e = test_env("IDIST_BF_SAMPLE")) S = (uint32_t)atoi(e);
    S = std::min(std::max(S, k), n);
    const uint32_t S_pad = (S + kTN - 1) / kTN * kTN;
    const uint32_t QC = 8192;                                    // queries per pass (bounds the dense sample matrix)
    const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)8 * k * n / S + 256, 1024), 1u << 16);
    float *d_qnat = nullptr, *d_qb = nullptr, *d_qn = nullptr, *d_pn = nullptr, *d_dense = nullptr, *d_thr = nullptr, *d_dist = nullptr;
    uint32_t *d_cand = nullptr, *d_cnt = nullptr, *d_pid = nullptr, *d_ovf = nullptr;
    auto release = [&]() {
        hipFree(d_qnat); hipFree(d_qb); hipFree(d_qn); hipFree(d_pn); hipFree(d_dense); hipFree(d_thr); hipFree(d_dist);
        hipFree(d_cand); hipFree(d_cnt); hipFree(d_pid); hipFree(d_ovf);
    };
#define MCHK(expr)                                                                                  \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            release();                                                                              \
            return fail(IDIST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                           \
    } while (0)
    const uint32_t qc_max = std::min(nq, QC);
    const uint32_t qc_pad = (qc_max + kTM - 1) / kTM * kTM;
    MCHK(hipMalloc((void**)&d_qnat, (size_t)qc_max * idx->dim * 4));
    MCHK(hipMalloc((void**)&d_qb, (size_t)qc_pad * stride * 4));
    MCHK(hipMalloc((void**)&d_qn, (size_t)qc_pad * 4));
    MCHK(hipMalloc((void**)&d_pn, (size_t)n * 4));
    MCHK(hipMalloc((void**)&d_dense, (size_t)qc_pad * S_pad * 4));
    MCHK(hipMalloc((void**)&d_thr, (size_t)qc_pad * 4));
    MCHK(hipMalloc((void**)&d_cand, (size_t)qc_max * cap * 4));
    MCHK(hipMalloc((void**)&d_cnt, (size_t)qc_pad * 4));
    MCHK(hipMalloc((void**)&d_pid, (size_t)qc_max * k * 4));
    MCHK(hipMalloc((void**)&d_dist, (size_t)qc_max * k * 4));
    MCHK(hipMalloc((void**)&d_ovf, 256));
    MCHK(hipMemset(d_ovf, 0, 256));
    hipStream_t st = nullptr;
    const uint32_t ngrid = std::min<uint32_t>(n, (uint32_t)idx->n_cu * 32);
    IDIST_LAUNCH(row_norms_kernel, ngrid, 64, 0, st, idx->d_points, n, stride, d_pn);
    std::vector<float> pn_h(n);
    MCHK(hipMemcpy(pn_h.data(), d_pn, (size_t)n * 4, hipMemcpyDeviceToHost));
    const float pn_max = *std::max_element(pn_h.begin(), pn_h.end());
    IndexView view = idx->view();
    const size_t smem_g = (size_t)2 * kTM * kLDP * 4;
    for (uint32_t qb = 0; qb < nq; qb += QC) {
        const uint32_t qc = std::min(QC, nq - qb);
        const uint32_t qpad = (qc + kTM - 1) / kTM * kTM;
        MCHK(hipMemcpy(d_qnat, queries + (size_t)qb * idx->dim, (size_t)qc * idx->dim * 4, hipMemcpyHostToDevice));
        MCHK(hipMemset(d_qb, 0, (size_t)qpad * stride * 4));
        {
            const size_t total = (size_t)qc * stride;
            const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
            IDIST_LAUNCH(permute_rows_kernel, grid, 256, 0, st, d_qnat, d_qb, qc, idx->dim, stride, idx->L.nb);
        }
        IDIST_LAUNCH(row_norms_kernel, std::min<uint32_t>(qpad, 8192), 64, 0, st, d_qb, qpad, stride, d_qn);
        MfmaArgs a{};
        a.Q = d_qb; a.P = idx->d_points; a.qn = d_qn; a.pn = d_pn;
        a.nq = qc; a.n = n; a.stride = stride;
        a.dense = d_dense; a.dense_ld = S_pad; a.thr = d_thr; a.cand = d_cand; a.cnt = d_cnt; a.cap = cap;
        const uint32_t nqt = qpad / kTM;
        // pass 1: dense distances to the sample [0, S) -> per-query threshold
        a.mode = 0; a.p_begin = 0; a.p_end = S;
        IDIST_LAUNCH(mfma_dist_kernel, nqt * (S_pad / kTN), 256, smem_g, st, a);
        IDIST_LAUNCH(kth_threshold_kernel, std::min<uint32_t>(qc, 8192), 64, (size_t)(k + 72) * 8, st, d_dense, S_pad, S, qc, k,
                     k + 72, d_qn, pn_max, d_thr);
        // pass 2: all points, keep those under the threshold
        MCHK(hipMemsetAsync(d_cnt, 0, (size_t)qpad * 4, st));
        a.mode = 1; a.p_begin = 0; a.p_end = n;
        IDIST_LAUNCH(mfma_dist_kernel, nqt * ((n + kTN - 1) / kTN), 256, smem_g, st, a);
        // pass 3: canonical re-rank of the candidates
        const uint32_t wcap = k + 64 + 8;
        const size_t smem_r = (size_t)stride * 4 + (size_t)wcap * 8 + 2 * 64 * 4;
        const uint32_t gridr = std::min<uint32_t>(qc, (uint32_t)idx->n_cu * 16);
#define LAUNCH_RR(NB_, RS_, TAIL_)                                                                                   \
    {                                                                                                                \
        auto kR = rerank_kernel<NB_, RS_, TAIL_>;                                                                    \
        IDIST_LAUNCH(kR, gridr, 64, smem_r, st, view, d_qnat, qc, k, wcap, d_cand, d_cnt, cap, d_pid, d_dist, d_ovf); \
    }
        IDIST_DISPATCH(idx->L, LAUNCH_RR);
#undef LAUNCH_RR
        MCHK(hipGetLastError());
        MCHK(hipMemcpy(out_pid + (size_t)qb * k, d_pid, (size_t)qc * k * 4, hipMemcpyDeviceToHost));
        MCHK(hipMemcpy(out_dist + (size_t)qb * k, d_dist, (size_t)qc * k * 4, hipMemcpyDeviceToHost));
    }
    uint32_t ovf = 0;
    MCHK(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost));
#undef MCHK
    release();
    if (ovf) { *fell_back = 1; return fail(IDIST_ERR_INTERNAL, "MFMA filter: %u candidate lists overflowed", ovf); }
    return IDIST_OK;
}

idist_status idist_bruteforce(const idist_index* idx, const float* queries, uint32_t nq, uint32_t k,
                              uint32_t* out_pid, float* out_dist) {
    if (!idx) return fail(IDIST_ERR_INVALID_ARG, "idx is null");
    if (nq == 0) return IDIST_OK;
    if (k == 0 || k > IDIST_MAX_EF) return fail(IDIST_ERR_INVALID_ARG, "k %u out of [1,%u]", k, IDIST_MAX_EF);
    if (!queries || !out_pid || !out_dist) return fail(IDIST_ERR_INVALID_ARG, "null pointer");
    // the -2QP^T contraction only pays when the batch is wide enough to be a dense GEMM
    bool mfma = nq >= 256 && idx->n >= 16384;
    if (const char* e = test_env("IDIST_BRUTEFORCE")) mfma = strcmp(e, "mfma") == 0 ? true : (strcmp(e, "scan") == 0 ? false : mfma);
    if (mfma && idx->n >= k && idx->n >= 128) {
        HIPCHK(hipSetDevice(idx->device));
        int fell_back = 0;
        idist_status s = bruteforce_mfma(idx, queries, nq, k, out_pid, out_dist, &fell_back);
        if (s == IDIST_OK || !fell_back) return s;
    }
    return bruteforce_scan(idx, queries, nq, k, out_pid, out_dist);
}

}  // extern "C"
