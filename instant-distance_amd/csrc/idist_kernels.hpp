// idist_kernels.hpp — gfx950 kernels of the HNSW hot path (search, build, layout).
// Every kernel runs single-wave workgroups (blockDim.x == 64); grids are
// persistent pools of "slots" that pull work items from an atomic queue.
#pragma once
#include "idist_device.hpp"

namespace idist {

// ---------------------------------------------------------------------------
// layout: natural row-major [n][dim] -> blocked [n][stride]  
// ---------------------------------------------------------------------------
__global__ void permute_rows_kernel(const float* __restrict__ in, float* __restrict__ out, uint32_t n,
                                    uint32_t dim, uint32_t stride, uint32_t nb) {
    const size_t total = (size_t)n * stride;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t row = (uint32_t)(idx / stride), o = (uint32_t)(idx % stride);
        uint32_t e = o;
        if (o < 32u * nb) {
            const uint32_t t = o >> 5, r = o & 31u, j = r >> 2, c = r & 3u;
            e = (t << 5) + (c << 3) + j;
        }
        out[idx] = e < dim ? in[(size_t)row * dim + e] : 0.0f;
    }
}

// The compact copy of the point rows behind the walk's reject filter (FilterView, idist_device.hpp): one wave per row.
// u_k = round((p_k - lo) / step) clamped to [0, 255] for the stored positions that hold a coordinate (padding: 0), in the f32 row's
// own element order; the row ends with {|p - p^|_2 rounded UP as f32, sum u_k^2}.  A row with a non-finite coordinate gets +inf there:
// the filter never rejects it.
__global__ __launch_bounds__(64) void filter_rows_kernel(const float* __restrict__ points, uint32_t n, uint32_t dim, uint32_t stride,
                                                        uint32_t nb, uint8_t* __restrict__ rows, uint32_t fstride, float lo, float step256) {
    const int lane = lane_id();
    const float inv = 1.0f / (256.0f * step256);
    const double step = 256.0 * (double)step256;
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float* src = points + (size_t)r * stride;
        uint8_t* dst = rows + (size_t)r * fstride;
        double e2 = 0.0;
        uint32_t su = 0;
        bool bad = false;
        for (uint32_t pos = (uint32_t)lane; pos < fstride - 8u; pos += 64u) {
            uint32_t u = 0;
            if (pos < stride && natural_pos(pos, nb) < dim) {
                const float p = src[pos];
                if (__builtin_fabsf(p) <= 3.0e38f) {
                    u = (uint32_t)__builtin_fminf(__builtin_fmaxf(__builtin_rintf((p - lo) * inv), 0.0f), 255.0f);
                    const double e = (double)p - ((double)lo + (double)u * step);
                    e2 += e * e;
                    su += u * u;
                } else {
                    bad = true;
                }
            }
            dst[pos] = (uint8_t)u;
        }
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, e2);
            const uint32_t blo = (uint32_t)__shfl_xor((int)(uint32_t)b, m, 64), bhi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), m, 64);
            e2 += __builtin_bit_cast(double, ((unsigned long long)bhi << 32) | blo);
            su += (uint32_t)__shfl_xor((int)su, m, 64);
        }
        const bool any_bad = __ballot(bad) != 0ull;
        if (lane == 0) {
            float ep = (float)(__builtin_sqrt(e2) * (1.0 + 1e-6));
            ep = __uint_as_float(__float_as_uint(ep) + 1u);                // one more ulp up: the cast rounded to nearest
            if (any_bad) ep = __uint_as_float(0x7f800000u);
            reinterpret_cast<uint32_t*>(dst + fstride - 8u)[0] = __float_as_uint(ep);
            reinterpret_cast<uint32_t*>(dst + fstride - 8u)[1] = su;
        }
    }
}

// UpperNode::from_zero for a whole layer, core/lib.rs:323-328, core/types.rs:66-70
__global__ void snapshot_kernel(const uint32_t* __restrict__ zero, uint32_t* __restrict__ upper_rows, uint32_t rows) {
    const size_t total = (size_t)rows * kM;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / kM, c = i % kM;
        upper_rows[i] = zero[r * kM2 + c];
    }
}

// Pipelined build: the rows step k-1 rewrote (its new nodes [start, start+count) and the nodes on its touched
// list) are carried over into the other copy of the zero layer before step k works on that copy in place.
__global__ __launch_bounds__(64) void copy_rows_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                      const uint32_t* __restrict__ touched, const uint32_t* n_touched,
                                                      uint32_t start, uint32_t count) {
    const uint32_t total = count + *n_touched;
    for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
        const uint32_t row = i < count ? start + i : touched[i - count];
        dst[(size_t)row * kM2 + threadIdx.x] = src[(size_t)row * kM2 + threadIdx.x];
    }
}

// Self-check of the pipelined build (IDIST_BUILD_CHECK): counts the zero rows on which the two copies disagree
__global__ void count_row_mismatch_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t rows,
                                          uint32_t* n_bad) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        bool bad = false;
        for (int i = 0; i < kM2; i++) bad = bad || a[(size_t)r * kM2 + i] != b[(size_t)r * kM2 + i];
        if (bad) atomicAdd(n_bad, 1u);
    }
}

// Builder::progress: one thread publishes {done, layer} to pinned host memory after a build step
__global__ void progress_kernel(volatile unsigned long long* slot, unsigned long long done, unsigned long long layer) {
    slot[0] = done;
    slot[1] = layer;
    __threadfence_system();
}

// Reference invariants of an adjacency array: ids < limit, no duplicate before the
// first INVALID (Visited makes duplicates impossible in the reference builder).
__global__ __launch_bounds__(64) void validate_rows_kernel(const uint32_t* __restrict__ rows, uint32_t n_rows,
                                                          int row_stride, uint32_t limit, uint32_t* bad) {
    const int lane = lane_id();
    for (uint32_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        uint32_t id = kInvalid;
        if (lane < row_stride) id = rows[(size_t)r * row_stride + lane];
        const uint64_t inval = __ballot(id == kInvalid);
        const int nvalid = inval ? __builtin_ctzll(inval) : 64;
        bool err = lane < nvalid && id >= limit;
        for (int s = 1; s < 64; s++) {
            const int o = (lane + s) & 63;
            const uint32_t other = bcast_u32(id, o);
            if (lane < nvalid && o < nvalid && other == id) err = true;
        }
        if (__ballot(err) && lane == 0) atomicAdd(bad, 1u);
    }
}

// ---------------------------------------------------------------------------
// LDS carve-up shared by the kernels.
// ---------------------------------------------------------------------------
struct Smem {
    float* q;            // stride floats
    uint64_t* W;         // wcap
    uint64_t* aux;       // 3*64+1 u64 (build only): news / sel / disc
    uint32_t* act_pid;   // 64
    uint32_t* act_dist;  // 64
    QuadCtl* ctl;        // hand-over block of the four-wave walk: command words + the speculated row, its new ids, their distances
    uint32_t* dirty;     // dirty-block bitmap of the visited set (graph walks only), dirty_words dwords
    uint32_t* bloom;     // kBloomWords / kBloomLatWords, last in the carve-up (graph walks only)
};
// quad_ctl = false: a single-wave search kernel has no use for the four-wave walk's hand-over block (1 KB: at ef_search 100 it is the
// difference between seven and eight resident walks per CU with a 16-KB visited set)
__host__ __device__ inline size_t smem_bytes(uint32_t stride, uint32_t wcap, bool build, uint32_t bloom_words = kBloomWords,
                                             uint32_t dirty_words = 0, bool quad_ctl = true) {
    wcap = (wcap + 1u) & ~1u;   // keeps everything behind W 16-B aligned
    size_t b = (size_t)stride * 4 + (size_t)wcap * 8 + 2 * 64 * 4 + (quad_ctl ? sizeof(QuadCtl) : 0) + (size_t)(bloom_words + dirty_words) * 4;
    if (build) b += (size_t)(3 * 64 + 8) * 8;
    return b;
}
__device__ __forceinline__ Smem carve(uint8_t* base, uint32_t stride, uint32_t wcap, bool build, uint32_t dirty_words = 0, bool quad_ctl = true) {
    Smem s;
    s.q = reinterpret_cast<float*>(base);
    base += (size_t)stride * 4;
    s.W = reinterpret_cast<uint64_t*>(base);
    base += (size_t)((wcap + 1u) & ~1u) * 8;
    s.aux = reinterpret_cast<uint64_t*>(base);
    if (build) base += (size_t)(3 * 64 + 8) * 8;
    s.act_pid = reinterpret_cast<uint32_t*>(base);
    s.act_dist = s.act_pid + 64;
    s.ctl = reinterpret_cast<QuadCtl*>(s.act_dist + 64);
    s.dirty = s.act_dist + 64 + (quad_ctl ? sizeof(QuadCtl) / 4 : 0);
    s.bloom = s.dirty + dirty_words;
    return s;
}

// ---------------------------------------------------------------------------
// Hnsw::search (core/lib.rs:352-383) for a batch: persistent slots pull queries.
// ---------------------------------------------------------------------------
struct SearchArgs {
    const float* queries;   // [nq][dim] natural order, device
    uint32_t nq, ef, wcap;
    uint32_t* out_pid;      // [nq][ef]
    float* out_dist;        // [nq][ef]
    uint32_t* out_count;    // [nq]
    uint32_t* out_counters; // [nq][3] or null
    uint32_t* visited;      // [slots][vis.slot_words] bitmaps, all-zero between launches
    VisGeom vis;            // vis_geometry(n)
    uint32_t* next;         // work queue head: counts on from launch to launch (no reset between launches), see queue_base
    uint32_t queue_base;    // value of *next when this launch starts: every launch adds nq + gridDim.x (one failed dequeue per workgroup)
    uint32_t* status;
    uint32_t* status_host;  // [gridDim.x] in pinned host memory or null: per-workgroup copy of a non-zero status (host-pointer calls
                            // of narrow batches read it after the stream sync instead of copying the device word back)
    uint32_t use_bloom;     // LDS Bloom filter in front of the visited bitmap (walks without the on-chip set)
    uint32_t tab_log2;      // log2(entries) of the on-chip visited set (walks with it): 4 << tab_log2 bytes of LDS
    uint32_t ubits;         // quotient form of that set: bits of the id universe, ceil(log2 n) (q16_* in idist_device.hpp)
    uint32_t tie_cap;       // capacity of the tie region (idist_config.tie_capacity)
    uint64_t* tie_spill;    // [slots][tie_spill_cap] or null: HBM bags for the ties beyond that capacity (WState::spill)
    uint32_t tie_spill_cap;
    // narrow host-pointer calls: the last workgroup to finish writes done_seq to pinned host memory — the host waits for
    // that word instead of a stream synchronisation (done_count: device word, back to zero when the launch ends)
    uint32_t* done_host;
    uint32_t* done_count;
    uint32_t done_seq;
    unsigned long long* filt_counts;   // [2] device words of the context: candidates the reject filter examined / rejected
};
// the LDS tail region (after the dirty-block bitmap) holds the Bloom filter or the on-chip visited set
__device__ __forceinline__ void visited_attach_tab(Visited& v, uint32_t* mem, uint32_t log2_entries) {
    v.bloom = nullptr;
    v.tab = mem;
    v.tmask = (1u << (log2_entries - 2u)) - 1u;     // buckets of four ids
    v.tshift = 32u - (log2_entries - 2u);
    v.tlimit = (7u << log2_entries) / 8u;
}
// the same LDS as 2^(log2_entries - 2) buckets of eight 16-bit quotients (twice the ids), ids drawn from [0, 2^ubits)
__host__ __device__ inline bool q16_applies(uint32_t log2_entries, uint32_t ubits) {
    const uint32_t bbits = log2_entries - 2u;
    return ubits > bbits && ubits - bbits <= 14u;      // a remainder + the which-hash bit stay below 0xFFFF (= empty)
}
__host__ __device__ inline uint32_t q16_universe_bits(uint32_t n, uint32_t log2_entries) {
    uint32_t u = 1;
    while (u < 32u && (1ull << u) < (uint64_t)n) u++;
    const uint32_t bbits = log2_entries - 2u;
    return u > bbits ? u : bbits + 1u;
}
__device__ __forceinline__ void visited_attach_q16(Visited& v, uint32_t log2_entries, uint32_t ubits) {
    v.q16 = true;
    v.ubits = ubits;
    v.rbits = ubits - (log2_entries - 2u);
    v.tlimit = 0xFFFFFFFFu;                            // never frozen: single ids overflow to the bitmap
}

// LAT: walk mode (kWalkClassic / kWalkLatency / kWalkOverlap, see search_layer).
// waves per SIMD a walk code asks the register allocator for (0 = its own choice)
#ifndef IDIST_WAVES_ATTR
// (the four-wave walk may share a SIMD with a second workgroup's wave: batches between one and two queries per CU)
#define IDIST_WAVES_ATTR(LAT_) \
    __attribute__((amdgpu_waves_per_eu(walk_waves(LAT_) ? walk_waves(LAT_) : 1, walk_quad(LAT_) ? 2 : (walk_waves(LAT_) ? walk_waves(LAT_) : 8))))
#endif
// Walk codes with the quad bit run four-wave workgroups (256 threads): wave 0 is the walk below, waves 1-3 only
// take their share of every distance pass (QuadCtl, idist_device.hpp).
template <int NB, int RS, int TAIL, int LAT = 0>
__global__ __launch_bounds__(walk_quad(LAT) ? 256 : 64) IDIST_WAVES_ATTR(LAT) void search_kernel(IndexView ix, SearchArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, a.wcap, false, a.vis.dirty_words, !walk_thin(LAT));
    const int lane = lane_id();
    const uint32_t slot = blockIdx.x;
    Visited vis{a.visited + (size_t)slot * a.vis.slot_words, ix.n, sm.dirty, a.vis.shift, a.vis.dirty_words,
                a.use_bloom ? sm.bloom : nullptr, walk_mode(LAT) == kWalkLatency ? kBloomLatLog2Words : kBloomLog2Words};
    if constexpr (walk_vis_lds(LAT)) visited_attach_tab(vis, sm.bloom, a.tab_log2);
    if constexpr (walk_vis16(LAT)) visited_attach_q16(vis, a.tab_log2, a.ubits);
    if constexpr (walk_quad(LAT)) {
        const int wv = (int)uniform_u32(threadIdx.x >> 6);
        if (wv != 0) {                                                 // (the helpers only ever READ the on-chip set through `vis`)
            quad_helper_loop<NB, RS, TAIL, LAT>(ix, sm.q, sm.ctl, sm.act_pid, sm.act_dist, wv, vis);
            return;
        }
    }
    uint32_t status = 0;
    QuadLead ql{sm.ctl, 0u};                                           // (four-wave walk: this wave leads)
    const uint32_t nb = NB >= 0 ? (uint32_t)NB : ix.nb;
    for (uint32_t i = lane; i < a.vis.dirty_words; i += 64) sm.dirty[i] = 0u;
    visited_clear(vis);                                                // LDS side; the slot's bitmap is clean between launches
    bool first_pull = true;
    for (;;) {
        uint32_t qi = 0;
        if (lane == 0) qi = atomicAdd(a.next, 1u) - a.queue_base;
        qi = uniform_u32(qi);
        // the head counts on from launch to launch: a launch that does not find it inside its own window [queue_base,
        // queue_base + nq + gridDim.x) was replayed (graph) or interleaved with another launch of the same context —
        // it would write nothing and the caller would read stale results: say so
        if (first_pull && qi >= a.nq + gridDim.x) status |= kStQueue;
        first_pull = false;
        if (qi >= a.nq) break;

        // stage the query tile in LDS in the blocked order of the point rows
        for (uint32_t o = lane; o < ix.stride; o += 64) sm.q[o] = 0.0f;
        wave_sync();
        const float* qsrc = a.queries + (size_t)qi * ix.dim;
        for (uint32_t e = lane; e < ix.dim; e += 64) sm.q[blocked_pos(e, nb)] = qsrc[e];
        wave_sync();

        // the query on the reject filter's lattice (walks compiled with it): registers, for the whole walk
        FilterQFor<NB, RS, TAIL, LAT> fq;
        if constexpr (walk_filter(LAT)) filter_stage_query(ix, sm.q, fq);

        WState st{sm.W, 0, 1, 0, 0u, (int)a.tie_cap};
        if (a.tie_spill) { st.spill = a.tie_spill + (size_t)slot * a.tie_spill_cap; st.spill_cap = a.tie_spill_cap; }
        Counters ctr{0, 0, 0};
        DistLog nolog{nullptr, 0u};
        // search.reset(), :357: the visited set was emptied when the slot's previous search ended
        push_entry<NB, RS, TAIL>(ix, sm.q, st, vis, sm.act_pid, sm.act_dist, ctr, nolog);  // :364
        for (int cur = (int)ix.n_upper;; cur--) {                      // :365
            const bool is_zero = cur == 0;
            st.ef = is_zero ? (int)a.ef : 1;                           // :366-371
            if (is_zero) {
                search_layer<NB, RS, TAIL, LAT>(ix, ix.zero, kM2, kM2, sm.q, st, vis, sm.act_pid, sm.act_dist, ctr, true, nolog, &ql, fq);
                break;
            }
            const uint32_t* rows = ix.upper + (size_t)ix.layer_off[cur - 1] * kM;
            search_layer<NB, RS, TAIL, LAT>(ix, rows, kM, kM, sm.q, st, vis, sm.act_pid, sm.act_dist, ctr, false, nolog, &ql, fq);
            w_cull(st);                                                // :377-379
            visited_clear(vis);
            visited_begin(vis, (uint32_t)st.plen);
            for (int i = lane; i < st.plen; i += 64) visited_mark(vis, (uint32_t)st.W[i]);
            visited_added(vis, (uint32_t)st.plen);
            wave_sync();
        }
        const int cnt = st.plen < st.ef ? st.plen : st.ef;             // search.iter(), :382
        for (uint32_t i = lane; i < a.ef; i += 64) {
            uint32_t pid = kInvalid;
            float d = __uint_as_float(0x7f800000u);
            if ((int)i < cnt) {
                const uint64_t k = st.W[i] & kKeyMask;
                pid = (uint32_t)k;
                d = __uint_as_float((uint32_t)(k >> 32));
            }
            a.out_pid[(size_t)qi * a.ef + i] = pid;
            a.out_dist[(size_t)qi * a.ef + i] = d;
        }
        if (lane == 0) {
            a.out_count[qi] = (uint32_t)cnt;
            if (a.out_counters) {
                a.out_counters[3 * (size_t)qi + 0] = ctr.n_dist;
                a.out_counters[3 * (size_t)qi + 1] = ctr.n_exp0;
                a.out_counters[3 * (size_t)qi + 2] = ctr.n_expU;
            }
        }
        status |= st.status;
        visited_clear(vis);                                            // leave the slot empty for its next search
        if constexpr (walk_filter(LAT)) {                               // what the reject filter did for this query (idist_search_ctx_filter_counts)
            if (lane == 0 && fq.seen) {
                atomicAdd(&a.filt_counts[0], (unsigned long long)fq.seen);
                atomicAdd(&a.filt_counts[1], (unsigned long long)fq.rejected);
            }
        }
    }
    if constexpr (walk_quad(LAT)) quad_release_helpers(ql);
    if (lane == 0 && status) {
        atomicOr(a.status, status);
        if (a.status_host) a.status_host[blockIdx.x] = status;
    }
    if (a.done_host) {
        __threadfence_system();                                        // this workgroup's results and status are out
        if (lane == 0 && atomicAdd(a.done_count, 1u) + 1u == gridDim.x) {
            *a.done_count = 0u;                                        // (the context's next launch is ordered behind this one)
            __threadfence_system();
            *reinterpret_cast<volatile uint32_t*>(a.done_host) = a.done_seq;
        }
    }
}

// ---------------------------------------------------------------------------
// Point::distance over id lists (the gather-L2 kernel on its own).
// grid = nq * ceil(n_ids/64) waves.
// ---------------------------------------------------------------------------
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void distance_batch_kernel(IndexView ix, const float* __restrict__ queries, uint32_t nq,
                                                           const uint32_t* __restrict__ ids, uint32_t n_ids,
                                                           float* __restrict__ out) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, 0, false);
    const int lane = lane_id();
    const uint32_t chunks = (n_ids + 63u) / 64u;
    const uint32_t nb = NB >= 0 ? (uint32_t)NB : ix.nb;
    for (uint32_t w = blockIdx.x; w < nq * chunks; w += gridDim.x) {
        const uint32_t qi = w / chunks, c0 = (w % chunks) * 64u;
        wave_sync();
        for (uint32_t o = lane; o < ix.stride; o += 64) sm.q[o] = 0.0f;
        wave_sync();
        for (uint32_t e = lane; e < ix.dim; e += 64) sm.q[blocked_pos(e, nb)] = queries[(size_t)qi * ix.dim + e];
        const uint32_t i = c0 + lane;
        uint32_t id = kInvalid;
        if (i < n_ids) id = ids[(size_t)qi * n_ids + i];
        const bool ok = id != kInvalid && id < ix.n;
        const uint64_t m = __ballot(ok);
        const int my = __popcll(m & ((1ull << lane) - 1ull));
        if (ok) sm.act_pid[my] = id;
        wave_sync();
        dist_rounds<NB, RS, TAIL>(ix, sm.q, sm.act_pid, sm.act_dist, __popcll(m));
        wave_sync();
        if (i < n_ids) out[(size_t)qi * n_ids + i] = ok ? __uint_as_float(sm.act_dist[my]) : __uint_as_float(0x7f800000u);
    }
}

// The reject filter on its own (idist_filter_bound_batch): for id lists, the lower bound of the canonical distance the compact
// rows give — what the walk's filter compares with nearest[ef-1] — through the walk's own loads and arithmetic (filter_rounds).
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void filter_bound_kernel(IndexView ix, const float* __restrict__ queries, uint32_t nq,
                                                         const uint32_t* __restrict__ ids, uint32_t n_ids, float* __restrict__ out) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, 0, false);
    const int lane = lane_id();
    const uint32_t chunks = (n_ids + 63u) / 64u;
    const uint32_t nb = NB >= 0 ? (uint32_t)NB : ix.nb;
    constexpr int NCH = filt_chunks<NB, RS, TAIL, kWalkFilterBit>();       // (runtime geometries: the larger tile)
    constexpr bool T8 = filt_tail8<NB, RS, TAIL>();
    for (uint32_t w = blockIdx.x; w < nq * chunks; w += gridDim.x) {
        const uint32_t qi = w / chunks, c0 = (w % chunks) * 64u;
        wave_sync();
        for (uint32_t o = lane; o < ix.stride; o += 64) sm.q[o] = 0.0f;
        wave_sync();
        for (uint32_t e = lane; e < ix.dim; e += 64) sm.q[blocked_pos(e, nb)] = queries[(size_t)qi * ix.dim + e];
        wave_sync();
        FilterQ<NCH, T8> fq;
        filter_stage_query(ix, sm.q, fq);
        const uint32_t i = c0 + lane;
        uint32_t id = kInvalid;
        if (i < n_ids) id = ids[(size_t)qi * n_ids + i];
        const bool ok = fq.on && id != kInvalid && id < ix.n;
        const uint64_t m = __ballot(ok);
        const int my = __popcll(m & ((1ull << lane) - 1ull));
        if (ok) sm.act_pid[my] = id;
        wave_sync();
        if (m) filter_rounds<NCH, T8, 2, NoMid, true>(ix, fq, sm.act_pid, sm.act_dist, __popcll(m), 0.0f);
        wave_sync();
        if (i < n_ids) out[(size_t)qi * n_ids + i] = ok ? __uint_as_float(sm.act_dist[my]) : 0.0f;
    }
}

// ---------------------------------------------------------------------------
// Exhaustive exact top-k with the canonical distance (ground truth; the brute
// force of tests/all.rs:60-67).  One wave per query, W machinery with ef = k.
// ---------------------------------------------------------------------------
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void bruteforce_kernel(IndexView ix, const float* __restrict__ queries, uint32_t nq,
                                                       uint32_t k, uint32_t wcap, uint32_t* out_pid, float* out_dist,
                                                       uint32_t* next) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, wcap, false);
    const int lane = lane_id();
    const uint32_t nb = NB >= 0 ? (uint32_t)NB : ix.nb;
    for (;;) {
        uint32_t qi = 0;
        if (lane == 0) qi = atomicAdd(next, 1u);
        qi = uniform_u32(qi);
        if (qi >= nq) break;
        wave_sync();
        for (uint32_t o = lane; o < ix.stride; o += 64) sm.q[o] = 0.0f;
        wave_sync();
        for (uint32_t e = lane; e < ix.dim; e += 64) sm.q[blocked_pos(e, nb)] = queries[(size_t)qi * ix.dim + e];
        wave_sync();
        WState st{sm.W, 0, (int)k, 0, 0u};
        for (uint32_t base = 0; base < ix.n; base += 64) {
            const int na = ix.n - base < 64u ? (int)(ix.n - base) : 64;
            if (lane < na) sm.act_pid[lane] = base + lane;
            wave_sync();
            dist_rounds<NB, RS, TAIL>(ix, sm.q, sm.act_pid, sm.act_dist, na);
            wave_sync();
            uint64_t key = kMaxKey;
            if (lane < na) key = ((uint64_t)sm.act_dist[lane] << 32) | (base + lane);
            const uint64_t thr = st.plen >= st.ef ? (st.W[st.ef - 1] & kKeyMask) : kMaxKey + 1ull;
            uint64_t pm = __ballot(lane < na && key < thr);
            while (pm) {
                const int i = __builtin_ctzll(pm);
                pm &= pm - 1ull;
                const uint64_t kk = bcast_u64(key, i);
                const int idx = w_rank(st, kk);
                if (idx < st.ef) w_insert(st, idx, kk);
            }
            if (st.plen > st.ef) st.plen = st.ef;   // plain truncate (no candidates here)
            wave_sync();
        }
        for (uint32_t i = lane; i < k; i += 64) {
            uint32_t pid = kInvalid;
            float d = __uint_as_float(0x7f800000u);
            if ((int)i < st.plen) {
                pid = (uint32_t)st.W[i];
                d = __uint_as_float((uint32_t)((st.W[i] & kKeyMask) >> 32));
            }
            out_pid[(size_t)qi * k + i] = pid;
            out_dist[(size_t)qi * k + i] = d;
        }
    }
}

// ---------------------------------------------------------------------------
// Build step A: Construction::insert up to and including select_heuristic
// (core/lib.rs:437-473, :516) for the pids [start, start+count) of one layer
// range.  The graph is read-only here; the new node's own row and one edge
// record per selected neighbour are the only writes.
// ---------------------------------------------------------------------------
struct BuildArgs {
    uint32_t start, count;      // batch = pids [start, start+count)
    uint32_t layer, top;        // LayerId of this range / of the top layer
    uint32_t efc, wcap;
    uint32_t keep_pruned;
    uint32_t use_bloom;         // LDS Bloom filter in front of the visited bytes (descent)
    uint32_t has_heuristic;     // 0 = Builder::select_heuristic(None): select_simple + sorted splice
    uint32_t* visited;          // [slots][vis.slot_words] bitmaps, all-zero between launches
    VisGeom vis;                // vis_geometry(n)
    uint32_t* edge_pid;         // [max_batch*64] neighbour selected by item*64+i
    uint32_t* edge_dist;        // its distance bits
    uint32_t* head;             // [n] inbox head per existing node (edge index), kInvalid = empty
    uint32_t* next;             // [max_batch*64] inbox links
    uint32_t* touched;          // distinct nodes with a non-empty inbox
    uint32_t* n_touched;
    uint32_t* nbr_dist;         // [n][64] distance bits of every zero-row entry to its owner (build scratch)
    uint32_t* nbr_aux;          // [n][64] for a back-filled entry: pid of a selected member that pruned it
    uint32_t* row_nsel;         // [n] how many leading entries of a zero row were SELECTED (the rest is back-fill)
    uint32_t* slow;             // nodes whose update needs the full re-selection (step B2)
    uint32_t* n_slow;
    // distance log of every new point's descent (DistLog): append log, then the published set ids / distances
    uint64_t* dlog_log;         // [max_batch][1 << tab_log2]
    uint32_t* dlog_pd;          // [max_batch][2 << tab_log2] published sets: per bucket four ids + their four distances
    uint32_t tab_log2;          // LDS of the descent's on-chip visited set: 4 << tab_log2 bytes
    uint32_t tab16;             // 1: that set stores 16-bit quotients (2 << tab_log2 ids), else full ids (1 << tab_log2)
    uint32_t ubits;             // quotient form: bits of the id universe
    uint32_t dl_shift;          // log2(entries) of one insertion's distance log = tab_log2 + tab16; its published set takes 2 << dl_shift dwords
    uint32_t use_dlog;          // 0: nothing is logged, step B computes every distance it needs (IDIST_BUILD_NO_DLOG, test / A-B knob)
    uint64_t* wbuf;             // [max_batch][efc] Search.nearest of every new point (step A -> step A2)
    uint32_t* wcount;           // [max_batch]
    uint32_t rt;                // step B2: selected rows kept in the LDS tile
    uint32_t rt2;               // step A2: same for the new point's own selection
    uint32_t fc;                // candidates staged per round in those tiles (8; fewer for long runtime-geometry rows, so that a tile
                                // wave still finds LDS beside the descents)
    uint32_t chunk;             // step B: items per dequeue, 0 = by load (IDIST_BUILD_CHUNK, test knob)
    uint32_t tie_cap;           // capacity of the tie region of the descent (idist_config.tie_capacity)
    uint64_t* tie_spill;        // [slots][tie_spill_cap] or null: HBM bags for the ties beyond that capacity (WState::spill)
    uint32_t tie_spill_cap;
    uint32_t* queue;            // work queue heads: [0] step A, [1] step B, [3] step B2, [4] step A2 ([2] = n_slow)
    unsigned long long* stats;  // [32] n_dist n_exp0 n_expU n_sel_pairs n_heur_rows n_updates n_fast n_full n_heur_ref, [9..15] probe build, [16] [17] reject filter: examined, rejected
    uint32_t* status;
};

// node.set(i, pid) for every found neighbour (core/lib.rs:516; the row was all-INVALID) and one inbox
// record per selected neighbour for step B.
__device__ __forceinline__ void emit_new_node(const IndexView& ix, const BuildArgs& a, uint32_t item, uint32_t nw_pid,
                                              const uint64_t* sel, int nsel) {
    const int lane = lane_id();
    ix.zero[(size_t)nw_pid * kM2 + lane] = lane < nsel ? (uint32_t)sel[lane] : kInvalid;
    a.nbr_dist[(size_t)nw_pid * kM2 + lane] = lane < nsel ? (uint32_t)(sel[lane] >> 32) : 0u;
    bool first = false;
    uint32_t my_pid = 0;
    if (lane < nsel) {
        const uint64_t k = sel[lane];
        const uint32_t e = item * kM2 + (uint32_t)lane;
        const uint32_t pid = (uint32_t)k;
        a.edge_pid[e] = pid;
        a.edge_dist[e] = (uint32_t)(k >> 32);
        const uint32_t old = atomicExch(&a.head[pid], e);
        a.next[e] = old;
        first = old == kInvalid;
        my_pid = pid;
    }
    // one atomic per wave, not per edge: a single word saturates at ~88 atomics/us (MI355X_MICROARCH.md, "dequeue")
    const uint64_t fm = __ballot(first);
    if (fm) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.n_touched, (uint32_t)__popcll(fm));
        base = uniform_u32(base);
        if (first) a.touched[base + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = my_pid;
    }
}

// Construction::insert up to the end of the layer search (core/lib.rs:442-461): the new point's row into the LDS query
// tile, the descent from the top layer, ef_construction from the insertion layer down.  Leaves Search.nearest in st.W and
// the insertion layer's visited set in `vis`.
template <int NB, int RS, int TAIL, int LAT>
__device__ __forceinline__ void insert_descent(const IndexView& ix, const BuildArgs& a, const Smem& sm, WState& st, Visited& vis,
                                               uint32_t nw_pid, Counters& tot, DistLog& dl, QuadLead* ql = nullptr) {
    const int lane = lane_id();
    wave_sync();
    // point = &points[new] (:442): rows are stored blocked already
    const float* prow = ix.points + (size_t)nw_pid * ix.stride;
    for (uint32_t o = lane * 4; o < ix.stride; o += 256)
        *reinterpret_cast<float4*>(sm.q + o) = *reinterpret_cast<const float4*>(prow + o);
    wave_sync();
    // the new point on the reject filter's lattice (descents compiled with it, §4.5)
    FilterQFor<NB, RS, TAIL, LAT> fq;
    if constexpr (walk_filter(LAT)) filter_stage_query(ix, sm.q, fq);
    // search.reset(), :443: the visited set was emptied when the slot's previous descent ended
    push_entry<NB, RS, TAIL>(ix, sm.q, st, vis, sm.act_pid, sm.act_dist, tot, dl);  // :444
    const int num = a.layer == 0 ? kM2 : kM;                      // :445
    for (int cur = (int)a.top;; cur--) {                          // :447
        st.ef = cur <= (int)a.layer ? (int)a.efc : 1;             // :448-452
        if (cur > (int)a.layer) {                                 // :453-457
            const uint32_t* rows = ix.upper + (size_t)ix.layer_off[cur - 1] * kM;
            search_layer<NB, RS, TAIL, LAT>(ix, rows, kM, num, sm.q, st, vis, sm.act_pid, sm.act_dist, tot, false, dl, ql, fq);
            w_cull(st);
            visited_clear(vis);
            visited_begin(vis, (uint32_t)st.plen);
            for (int i = lane; i < st.plen; i += 64) visited_mark(vis, (uint32_t)st.W[i]);
            visited_added(vis, (uint32_t)st.plen);
            wave_sync();
            dl.reset();                                           // the set was emptied: its indices start over
            if (dl.log)
                for (int i0 = 0; i0 < st.plen; i0 += 64) {
                    const bool on = i0 + lane < st.plen;
                    const uint64_t k = on ? st.W[i0 + lane] & kKeyMask : 0ull;
                    dlog_append(dl, on ? vis_index(vis, (uint32_t)k) : -1, (uint32_t)(k >> 32));
                }
        } else {                                                  // :458-461
            search_layer<NB, RS, TAIL, LAT>(ix, ix.zero, kM2, num, sm.q, st, vis, sm.act_pid, sm.act_dist, tot, a.layer == 0, dl, ql, fq);
            break;
        }
    }
    if constexpr (walk_filter(LAT)) { tot.f_seen += fq.seen; tot.f_rej += fq.rejected; }
}

// Walk codes with the quad bit (narrow steps: the growth phase of a layer, max_batch = 1) run four-wave workgroups like the
// search kernel: wave 0 is the descent below, waves 1-3 take their share of every distance pass.
template <int NB, int RS, int TAIL, int LAT = 0>
__global__ __launch_bounds__(walk_quad(LAT) ? 256 : 64) IDIST_WAVES_ATTR(LAT) void build_insert_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, a.wcap, true, a.vis.dirty_words);
    uint64_t* sel = sm.aux + 64 + 8;
    const int lane = lane_id();
    const uint32_t slot = blockIdx.x;
    Visited vis{a.visited + (size_t)slot * a.vis.slot_words, ix.n, sm.dirty, a.vis.shift, a.vis.dirty_words, nullptr, 0};
    visited_attach_tab(vis, sm.bloom, a.tab_log2);                    // the descent always keeps its visited set on chip
    if constexpr (walk_vis16(LAT)) visited_attach_q16(vis, a.tab_log2, a.ubits);
    if constexpr (walk_quad(LAT)) {
        const int wv = (int)uniform_u32(threadIdx.x >> 6);
        if (wv != 0) {                                                 // (the helpers only ever READ the on-chip set through `vis`)
            quad_helper_loop<NB, RS, TAIL, LAT>(ix, sm.q, sm.ctl, sm.act_pid, sm.act_dist, wv, vis);
            return;
        }
    }
    uint32_t status = 0;
    QuadLead ql{sm.ctl, 0u};                                          // (four-wave descents: this wave leads)
    Counters tot{0, 0, 0};
    for (uint32_t i = lane; i < a.vis.dirty_words; i += 64) sm.dirty[i] = 0u;
    visited_clear(vis);                                               // LDS side; the slot's bitmap is clean between launches
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(&a.queue[0], 1u);
        item = uniform_u32(item);
        if (item >= a.count) break;
        const uint32_t nw_pid = a.start + item;
        WState st{sm.W, 0, 1, 0, 0u, (int)a.tie_cap};
        if (a.tie_spill) { st.spill = a.tie_spill + (size_t)slot * a.tie_spill_cap; st.spill_cap = a.tie_spill_cap; }
        // only the heuristic's re-selections look distances up
        DistLog dl{a.has_heuristic && a.use_dlog ? a.dlog_log + ((size_t)item << a.dl_shift) : nullptr, 0u};
        if constexpr (walk_vis16(LAT)) dl.half = 4u << (a.tab_log2 - 2u);     // dwords of the set = entries of one half
        insert_descent<NB, RS, TAIL, LAT>(ix, a, sm, st, vis, nw_pid, tot, dl, walk_quad(LAT) ? &ql : nullptr);
        const int nw = st.plen < st.ef ? st.plen : st.ef;             // Search.nearest
        // (the hand-over addresses below depend on nothing but the item: hipcc computed them up here and kept them in
        // scratch across the whole walk — an opaque copy of the item pins their computation to where they are used)
        uint32_t item_l = item;
#ifndef IDIST_EMU
        asm volatile("" : "+s"(item_l));
#endif
        if (a.has_heuristic) {
            // select_heuristic (:470-472) runs in step A2 with the selected rows on chip; hand Search.nearest over
            for (int i = lane; i < nw; i += 64) a.wbuf[(size_t)item_l * a.efc + i] = st.W[i] & kKeyMask;
            if (lane == 0) a.wcount[item_l] = (uint32_t)nw;
            if (dl.log) {
                uint32_t* pd = a.dlog_pd + ((size_t)item_l << (a.dl_shift + 1u));
                if constexpr (walk_vis16(LAT)) dlog_publish_q16(dl, vis, pd);
                else dlog_publish(dl, vis, pd);
            }
        } else {                                                      // select_simple, :466-469, :758-760
            const int nsel = nw < kM2 ? nw : kM2;
            if (lane < nsel) sel[lane] = st.W[lane] & kKeyMask;
            wave_sync();
            if (lane == 0) a.row_nsel[nw_pid] = (uint32_t)nsel;
            a.nbr_aux[(size_t)nw_pid * kM2 + lane] = 0u;
            emit_new_node(ix, a, item, nw_pid, sel, nsel);
        }
        status |= st.status;
        visited_clear(vis);                                           // leave the slot empty for its next descent
    }
    if constexpr (walk_quad(LAT)) quad_release_helpers(ql);
    if (lane == 0) {
        if (status) atomicOr(a.status, status);
        if (tot.n_dist | tot.n_exp0 | tot.n_expU) {
            atomicAdd(&a.stats[0], (unsigned long long)tot.n_dist);
            atomicAdd(&a.stats[1], (unsigned long long)tot.n_exp0);
            atomicAdd(&a.stats[2], (unsigned long long)tot.n_expU);
        }
        if (tot.f_seen) {
            atomicAdd(&a.stats[16], (unsigned long long)tot.f_seen);
            atomicAdd(&a.stats[17], (unsigned long long)tot.f_rej);
        }
    }
}

// ---------------------------------------------------------------------------
// Heuristic { extend_candidates: true } (core/lib.rs:648-664).
//
// In the reference this option cannot run: `select_heuristic` read-locks every candidate's node (:649,
// core/types.rs:145-149) while `add_neighbor_heuristic` has pushed the NEW node (:626), whose write lock the inserting
// thread holds since :438 — parking_lot locks are not re-entrant, so the second insert deadlocks.  What it WOULD
// compute is well defined once the locks are ignored; the oracle restates exactly that (oracle/idist_oracle.c,
// select_heuristic with extend_candidates, construction_insert with node[i] visible at once), and this kernel is the
// device side of the same definition: one wave performs one whole insertion in program order — descent, the new point's
// selection over {nearest} U {their unvisited zero-layer neighbours}, then for each found neighbour the re-selection
// over {new, its row} U {their unvisited neighbours}, ZeroNode::rewrite, node.set.  Sequential by construction
// (idist_config.max_batch is ignored), byte-identical to the oracle; a correctness path, not a fast one: the working
// set of one selection holds up to 65 * ef_construction candidates and is sorted in HBM scratch.
// ---------------------------------------------------------------------------
// ascending bitonic sort of n (a power of two) keys in global memory by one wave; each pass touches every pair once
__device__ __forceinline__ void wave_sort_u64(uint64_t* a, uint32_t n) {
    const uint32_t lane = (uint32_t)lane_id();
    for (uint32_t k = 2; k <= n; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = lane; i < n; i += 64) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t x = a[i], y = a[l];
                    if ((x > y) == ((i & k) == 0u)) { a[i] = y; a[l] = x; }
                }
            }
            visited_drain();                                   // the pass's stores have landed before the next one reads
            wave_sync();
        }
}

// Search::select_heuristic(extend_candidates = true) for the point whose blocked row sits in the LDS tile q.
// nearest[0..nn): Search.nearest (sorted keys, flags cleared); vis: Search.visited as the caller left it (:651).
// Result: sel[0..return) = selected-then-backfilled keys (NOT re-sorted).  q2: LDS tile for one candidate row.
template <int NB, int RS, int TAIL>
__device__ __forceinline__ int select_extend(const IndexView& ix, const float* q, float* q2, const uint64_t* nearest, int nn,
                                             Visited& vis, uint64_t* work, uint32_t work_cap, bool keep_pruned, uint64_t* sel,
                                             uint64_t* disc, uint32_t* act_pid, uint32_t* act_dist, HeurCounters& hc,
                                             uint32_t& status) {
    const int lane = lane_id();
    // working = nearest + every not yet visited zero-layer neighbour of its members, :643-660
    uint32_t m = (uint32_t)nn;
    for (int i = lane; i < nn; i += 64) work[i] = nearest[i];
    for (int i = 0; i < nn; i++) {
        const uint32_t cpid = (uint32_t)nearest[i];
        const uint32_t nb = ix.zero[(size_t)cpid * kM2 + lane];                     // layer.nearest_iter(candidate.pid), :649
        const uint64_t inval = __ballot(nb == kInvalid);
        const int nvalid = inval ? __builtin_ctzll(inval) : 64;
        bool fresh = false;
        int tab_idx = -1;
        visited_begin(vis);
        if (lane < nvalid) {
            if (nb >= ix.n) status |= kStBadRow;
            else fresh = visited_insert(vis, nb, tab_idx);                           // :650
        }
        const uint64_t fm = __ballot(fresh);
        const int na = __popcll(fm);
        visited_added(vis, (uint32_t)na);
        wave_sync();
        if (!na) continue;
        const int my = __popcll(fm & ((1ull << lane) - 1ull));
        if (fresh) act_pid[my] = nb;
        wave_sync();
        dist_rounds<NB, RS, TAIL>(ix, q, act_pid, act_dist, na);                     // :654-657
        wave_sync();
        hc.n_dist += (uint32_t)na;
        hc.n_ref += (uint32_t)na;
        if (m + (uint32_t)na > work_cap) { status |= kStGuard; break; }
        if (fresh) work[m + (uint32_t)my] = ((uint64_t)act_dist[my] << 32) | nb;
        m += (uint32_t)na;
    }
    uint32_t np2 = 1;
    while (np2 < m) np2 <<= 1;
    for (uint32_t i = m + (uint32_t)lane; i < np2; i += 64) work[i] = ~0ull;
    visited_drain();
    wave_sync();
    if (np2 > 1) wave_sort_u64(work, np2);                                           // working.sort_unstable(), :662-664
    // :666-685
    int nsel = 0, ndisc = 0;
    for (uint32_t w = 0; w < m && nsel < kM2; w++) {
        const uint64_t c = work[w];
        const uint32_t cpid = (uint32_t)c, cd = (uint32_t)(c >> 32);
        bool pruned = false;
        if (nsel > 0) {
            wave_sync();
            const float* prow = ix.points + (size_t)cpid * ix.stride;                // candidate_point, :675
            for (uint32_t o = lane * 4; o < ix.stride; o += 256)
                *reinterpret_cast<float4*>(q2 + o) = *reinterpret_cast<const float4*>(prow + o);
            if (lane < nsel) act_pid[lane] = (uint32_t)sel[lane];
            wave_sync();
            dist_rounds<NB, RS, TAIL>(ix, q2, act_pid, act_dist, nsel);              // :676-679
            wave_sync();
            hc.n_dist += (uint32_t)nsel;
            const uint64_t cm = __ballot(lane < nsel && act_dist[lane] < cd);        // OrderedFloat order == order of the canonical bits
            pruned = cm != 0ull;
            hc.n_ref += pruned ? (uint32_t)__builtin_ctzll(cm) + 1u : (uint32_t)nsel;  // `any` stops at the first closer member
        }
        wave_sync();
        if (!pruned) {                                                               // :681-684
            if (lane == 0) sel[nsel] = c;
            nsel++;
        } else if (ndisc < kM2) {                                                    // only the first 64 can ever be back-filled
            if (lane == 0) disc[ndisc] = c;
            ndisc++;
        }
    }
    wave_sync();
    if (keep_pruned)                                                                 // :687-695
        for (int i = 0; i < ndisc && nsel < kM2; i++) {
            if (lane == 0) sel[nsel] = disc[i];
            nsel++;
        }
    wave_sync();
    return nsel;
}

__host__ __device__ inline size_t smem_bytes_extend(uint32_t stride, uint32_t wcap, uint32_t tab_words, uint32_t dirty_words) {
    return smem_bytes(stride, wcap, true, tab_words, dirty_words) + (size_t)stride * 4 + (size_t)((wcap + 1u) & ~1u) * 8 + 64 * 8;
}

template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void build_extend_kernel(IndexView ix, BuildArgs a, uint64_t* work, uint32_t work_cap) {
    IDIST_DYN_SMEM(smem_raw);
    const Smem sm = carve(smem_raw, ix.stride, a.wcap, true, a.vis.dirty_words);
    uint64_t* sel = sm.aux + 64 + 8;
    uint64_t* disc = sel + 64;
    float* q2 = reinterpret_cast<float*>(sm.bloom + (1u << a.tab_log2));             // behind the on-chip visited set
    uint64_t* W2 = reinterpret_cast<uint64_t*>(q2 + ix.stride);                      // the `insertion` Search's nearest
    uint64_t* own = W2 + ((a.wcap + 1u) & ~1u);                                      // `found`, :465-472
    const int lane = lane_id();
    constexpr int LAT = walk_code(kWalkClassic, 0, false, 1, true);
    Visited vis{a.visited, ix.n, sm.dirty, a.vis.shift, a.vis.dirty_words, nullptr, 0};
    visited_attach_tab(vis, sm.bloom, a.tab_log2);
    for (uint32_t i = lane; i < a.vis.dirty_words; i += 64) sm.dirty[i] = 0u;
    visited_clear(vis);
    uint32_t status = 0;
    Counters tot{0, 0, 0};
    HeurCounters hc{0, 0};
    const uint32_t nw_pid = a.start;                                                 // one insertion per launch
    WState st{sm.W, 0, 1, 0, 0u, (int)a.tie_cap};
    if (a.tie_spill) { st.spill = a.tie_spill; st.spill_cap = a.tie_spill_cap; }
    DistLog nolog{nullptr, 0u};
    insert_descent<NB, RS, TAIL, LAT>(ix, a, sm, st, vis, nw_pid, tot, nolog);       // :442-461
    const int nn = st.plen < st.ef ? st.plen : st.ef;
    for (int i = lane; i < nn; i += 64) sm.W[i] &= kKeyMask;
    wave_sync();
    status |= st.status;
    // :470-472 for the new point; search.visited is the insertion layer's
    const int nf = select_extend<NB, RS, TAIL>(ix, sm.q, q2, sm.W, nn, vis, work, work_cap, a.keep_pruned != 0, sel, disc,
                                               sm.act_pid, sm.act_dist, hc, status);
    if (lane < nf) own[lane] = sel[lane];
    wave_sync();
    for (int i = 0; i < nf; i++) {                                                   // :481-516
        const uint32_t pid = (uint32_t)own[i];
        wave_sync();
        const float* prow = ix.points + (size_t)pid * ix.stride;                     // old = &points[pid], :484
        for (uint32_t o = lane * 4; o < ix.stride; o += 256)
            *reinterpret_cast<float4*>(sm.q + o) = *reinterpret_cast<const float4*>(prow + o);
        const uint32_t cur = ix.zero[(size_t)pid * kM2 + lane];                      // zero.nearest_iter(pid), :487
        const uint64_t inval = __ballot(cur == kInvalid);
        const int ncur = inval ? __builtin_ctzll(inval) : 64;
        // add_neighbor_heuristic, :616-631: insertion.reset(); push(new); push(current...); select_heuristic
        visited_clear(vis);                                                          // :625
        WState ns{W2, 0, (int)a.efc, 0, 0u, (int)a.tie_cap};                         // insertion.ef = ef_construction, :440
        visited_begin(vis);
        if (lane == 0) { visited_mark(vis, nw_pid); sm.act_pid[0] = nw_pid; }
        visited_added(vis, 1u);
        wave_sync();
        dist_rounds<NB, RS, TAIL>(ix, sm.q, sm.act_pid, sm.act_dist, 1);
        wave_sync();
        hc.n_dist += 1u;
        hc.n_ref += 1u;
        w_push_keys(ns, lane == 0 ? (((uint64_t)sm.act_dist[0] << 32) | nw_pid) : kMaxKey, lane == 0);   // :626
        bool fresh = false;
        int tab_idx = -1;
        visited_begin(vis);
        if (lane < ncur) {
            if (cur >= ix.n) status |= kStBadRow;
            else fresh = visited_insert(vis, cur, tab_idx);
        }
        const uint64_t fm = __ballot(fresh);
        const int na = __popcll(fm);
        visited_added(vis, (uint32_t)na);
        wave_sync();
        if (na) {                                                                    // :627-629, slot order
            const int my = __popcll(fm & ((1ull << lane) - 1ull));
            if (fresh) sm.act_pid[my] = cur;
            wave_sync();
            dist_rounds<NB, RS, TAIL>(ix, sm.q, sm.act_pid, sm.act_dist, na);
            wave_sync();
            hc.n_dist += (uint32_t)na;
            hc.n_ref += (uint32_t)na;
            w_push_keys(ns, fresh ? (((uint64_t)sm.act_dist[my] << 32) | cur) : kMaxKey, fresh);
        }
        status |= ns.status;                                                         // a tie overflow of the re-selection's own list
        const int nres = select_extend<NB, RS, TAIL>(ix, sm.q, q2, W2, ns.plen, vis, work, work_cap, a.keep_pruned != 0, sel, disc,
                                                     sm.act_pid, sm.act_dist, hc, status);       // :630
        ix.zero[(size_t)pid * kM2 + lane] = lane < nres ? (uint32_t)sel[lane] : kInvalid;        // ZeroNode::rewrite, :495
        if (lane == 0) ix.zero[(size_t)nw_pid * kM2 + i] = pid;                       // node.set(i, pid), :516
        visited_drain();
        wave_sync();
    }
    visited_clear(vis);
    if (lane == 0) {
        if (status) atomicOr(a.status, status);
        atomicAdd(&a.stats[0], (unsigned long long)tot.n_dist);
        atomicAdd(&a.stats[1], (unsigned long long)tot.n_exp0);
        atomicAdd(&a.stats[2], (unsigned long long)tot.n_expU);
        atomicAdd(&a.stats[3], (unsigned long long)hc.n_dist);
        atomicAdd(&a.stats[5], (unsigned long long)nf);
        atomicAdd(&a.stats[7], (unsigned long long)nf);
        atomicAdd(&a.stats[8], (unsigned long long)hc.n_ref);
    }
}

// ---------------------------------------------------------------------------
// Build step A2: Search::select_heuristic for the new points (core/lib.rs:470-472) on the
// Search.nearest that step A left in wbuf, with the selected rows in an LDS tile (inside step A the
// same selection re-gathered ~1.8 TB of selected rows from L2/HBM per 1M points, and a tile there
// would have cost the HBM-bound descent its occupancy); then node.set + the inbox records.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t smem_bytes_select(uint32_t nb, uint32_t rt, uint32_t efc, uint32_t fc = 8) {
    return tile_floats(nb, rt + fc) * 4 + (size_t)(efc + 8 + 64 + 64) * 8 + 4 * 64 * 4;
}

template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void build_select_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    const int nb = NB >= 0 ? NB : (int)ix.nb;
    Tile tile;
    tile.rt = (int)a.rt2;
    tile.fc = NB >= 0 ? 8 : (int)a.fc;
    tile.slots = tile.rt + tile.fc;
    tile.blk = reinterpret_cast<float*>(smem_raw);
    tile.rem = tile.blk + (size_t)nb * tile.slots * 32;
    uint64_t* Wl = reinterpret_cast<uint64_t*>(tile.blk + tile_floats((uint32_t)nb, (uint32_t)tile.slots));   // efc + 8
    uint64_t* sel = Wl + a.efc + 8;
    uint64_t* disc = sel + 64;
    uint32_t* act_pid = reinterpret_cast<uint32_t*>(disc + 64);
    uint32_t* act_dist = act_pid + 64;
    uint32_t* dprn = act_dist + 64;
    uint32_t* out_aux = dprn + 64;
    const int lane = lane_id();
    HeurCounters hc{0, 0};
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(&a.queue[4], 1u);
        item = uniform_u32(item);
        if (item >= a.count) break;
        const uint32_t nw_pid = a.start + item;
        const int nw = (int)a.wcount[item];
        wave_sync();
        for (int i = lane; i < nw; i += 64) Wl[i] = a.wbuf[(size_t)item * a.efc + i];
        wave_sync();
        int n_selected = 0;
        const int nsel = select_heuristic_tiled<NB, RS, TAIL>(ix, Wl, nw, a.keep_pruned != 0, tile, sel, disc, act_pid,
                                                              act_dist, hc, n_selected, dprn, out_aux);
        if (lane == 0) a.row_nsel[nw_pid] = (uint32_t)n_selected;
        a.nbr_aux[(size_t)nw_pid * kM2 + lane] = out_aux[lane];
        emit_new_node(ix, a, item, nw_pid, sel, nsel);
        wave_sync();
    }
    if (lane == 0 && (hc.n_dist | hc.n_rows)) {
        atomicAdd(&a.stats[3], (unsigned long long)hc.n_dist);
        atomicAdd(&a.stats[4], (unsigned long long)hc.n_rows);
        atomicAdd(&a.stats[8], (unsigned long long)hc.n_ref);
    }
}

// ---------------------------------------------------------------------------
// Build step B for Builder::select_heuristic(None) (core/lib.rs:497-515): splice the new
// point into each selected neighbour's row at the index Rust's slice::binary_search_by
// returns for the reference's comparator — which is REVERSED (it returns
// target.cmp(element), :510, and Greater for an empty slot, :505-508), so the result is
// whatever std's probe sequence makes of it.  Restated literally (Rust >= 1.82 branch-free
// form); d(points[pid], points[third]) comes from the stored nbr_dist.  ZeroNode::insert:
// core/types.rs:100-113.  Several new points for one node are applied in PointId order.
// ---------------------------------------------------------------------------
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void build_update_simple_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    uint64_t* news = reinterpret_cast<uint64_t*>(smem_raw);     // 72: key = edge index (ascending = PointId order)
    int* cmps = reinterpret_cast<int*>(news + 72);              // 64
    const int lane = lane_id();
    const uint32_t ntouched = *a.n_touched;
    uint32_t updates = 0, status = 0;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&a.queue[1], 1u);
        t = uniform_u32(t);
        if (t >= ntouched) break;
        const uint32_t pid = a.touched[t];
        wave_sync();
        WState ns{news, 0, kM2, 0, 0u};
        uint32_t e = a.head[pid];
        uint32_t guard = 0;
        while (e != kInvalid) {
            const uint64_t k = (uint64_t)e;
            const int idx = w_rank(ns, k);
            if (idx < ns.ef) w_insert(ns, idx, k);
            if (ns.plen > ns.ef) ns.plen = ns.ef;
            e = a.next[e];
            if (++guard > a.count) { status |= kStGuard; break; }
        }
        if (lane == 0) a.head[pid] = kInvalid;
        uint32_t id = ix.zero[(size_t)pid * kM2 + lane];
        uint32_t dd = a.nbr_dist[(size_t)pid * kM2 + lane];
        for (int i = 0; i < ns.plen; i++) {
            const uint32_t ei = (uint32_t)news[i];
            const uint32_t nw_pid = a.start + ei / kM2;
            const uint32_t target = a.edge_dist[ei];              // `distance` = d(new, pid), :483
            wave_sync();
            // f(third): empty -> Greater (:505-508), else distance.cmp(d(old, third)) (:510)
            cmps[lane] = id == kInvalid ? 1 : (target > dd) - (target < dd);
            wave_sync();
            int size = kM2, base = 0;                              // slice::binary_search_by
            while (size > 1) {
                const int half = size / 2, mid = base + half;
                base = cmps[mid] > 0 ? base : mid;
                size -= half;
            }
            const int c = cmps[base];
            const int idx = c == 0 ? base : base + (c < 0 ? 1 : 0);   // .unwrap_or_else(|e| e), :512
            if (idx < kM2) {                                       // ZeroNode::insert, core/types.rs:100-113
                const bool occupied = bcast_u32(id, idx) != kInvalid;
                const uint32_t pid_l = bcast_u32(id, lane > 0 ? lane - 1 : 0);
                const uint32_t dd_l = bcast_u32(dd, lane > 0 ? lane - 1 : 0);
                if (occupied && lane > idx) { id = pid_l; dd = dd_l; }     // copy_within(idx..63, idx+1)
                if (lane == idx) { id = nw_pid; dd = target; }
            }
        }
        ix.zero[(size_t)pid * kM2 + lane] = id;
        a.nbr_dist[(size_t)pid * kM2 + lane] = dd;
        updates++;
        wave_sync();
    }
    if (lane == 0) {
        if (status) atomicOr(a.status, status);
        if (updates) atomicAdd(&a.stats[5], (unsigned long long)updates);
    }
}

// ---------------------------------------------------------------------------
// Build step B (fast path): memoised re-selection.
//
// select_heuristic (core/lib.rs:636-698) walks its candidates nearest-first and the
// verdict on a candidate depends only on the candidates SELECTED before it.  Every zero
// row this builder writes has the form [selected, ascending | back-filled discarded,
// ascending] and row_nsel keeps the split.  A discarded candidate never influences
// anyone, so re-running the selection on {row} U {new points} gives every old entry in
// front of the first new point its old verdict, and afterwards
//   * an old SELECTED entry can only be newly pruned by members ADDED since (it already
//     passed every older selected member that precedes it),
//   * an old DISCARDED entry stays discarded as long as no older selected member has
//     been removed (its pruner is still there),
//   * a new point is checked against the current selected set.
// All distances that can matter therefore involve a new point: one gather of
// {old selected} U {new} rows per new point (~20 rows) instead of ~800 pairwise
// distances.  If an old selected entry does get pruned and an old discarded entry
// follows (a cascade), or more than kMaxNewFast points arrive, the node is handed to the
// full re-selection (step B2, build_update_kernel).  Results are identical by
// construction; tests/test_parity.py checks byte-identity with the oracle.
// ---------------------------------------------------------------------------
// step B: d(new point, pid) from the set the new point's descent published, kDlogMiss if it is not there
__device__ __forceinline__ uint32_t build_dlog_find(const BuildArgs& a, uint32_t new_pid, uint32_t pid) {
    const uint32_t* TPD = a.dlog_pd + ((size_t)(new_pid - a.start) << (a.dl_shift + 1u));
    if (a.tab16) return dlog_find_q16(TPD, a.ubits, a.ubits - (a.tab_log2 - 2u), pid);
    return dlog_find(TPD, (1u << (a.tab_log2 - 2u)) - 1u, 32u - (a.tab_log2 - 2u), pid);
}

constexpr int kUpdW = 136;   // <= 64 current + 64 new + slack
constexpr int kMaxNewFast = 8;
constexpr int kFastX = 80;   // columns of the new-vs-{old selected, new} distance table
__host__ __device__ inline size_t smem_bytes_update_fast(uint32_t stride) {
    return (size_t)stride * 4 + (size_t)(kUpdW + 72 + 64 + 64 + 64) * 8 +
           (size_t)(kUpdW + kFastX + kMaxNewFast * kFastX + 64 + kMaxNewFast + 64 + 64 + 3 * kFastX) * 4;
}

template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void build_update_fast_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    float* cq = reinterpret_cast<float*>(smem_raw);
    uint64_t* W = reinterpret_cast<uint64_t*>(cq + ix.stride);
    uint64_t* news = W + kUpdW;
    uint64_t* sel = news + 72;
    uint64_t* disc = sel + 64;
    uint64_t* curk = disc + 64;
    uint32_t* kindx = reinterpret_cast<uint32_t*>(curk + 64);   // kUpdW: kind << 8 | x
    uint32_t* X = kindx + kUpdW;                                // kFastX pids: old selected, then new
    uint32_t* Dn = X + kFastX;                                  // [kMaxNewFast][kFastX] distance bits
    uint32_t* Rx = Dn + kMaxNewFast * kFastX;                   // x of every member of R
    uint32_t* addx = Rx + 64;                                   // active points that entered R
    uint32_t* dprn = addx + kMaxNewFast;                        // pruner of disc[j]
    uint32_t* curaux = dprn + 64;                               // stored pruners of the current row
    uint32_t* act_x = curaux + 64;                              // columns whose distance must be gathered
    uint32_t* act_p = act_x + kFastX;                           // ... their pids
    uint32_t* act_d = act_p + kFastX;                           // ... the results
    enum : uint32_t { OLD_SEL = 0, OLD_DISC = 1, NEW = 2 };
    const int lane = lane_id();
    const uint32_t ntouched = *a.n_touched;
    HeurCounters hc{0, 0};
    uint32_t updates = 0, deferred = 0, status = 0;
#ifdef IDIST_PROBE
    // measurement build (make probe): where step B's distance look-ups miss.  [0] single-new-point updates, [1] general-path
    // updates, [2] look-ups, [3] misses, [4] ... of members that joined in the previous step, [5] ... in this step, [6] rows
    // gathered for the new-vs-new columns of the general path
    uint32_t pb[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // items per dequeue: 37 M updates through one counter would cost ~0.4 s; small steps keep 1 item per wave
    uint32_t kChunk = a.chunk ? a.chunk : ntouched / (gridDim.x * 4u);
    kChunk = kChunk < 1u ? 1u : (kChunk > 16u ? 16u : kChunk);
    uint32_t t_next = 0, t_end = 0;
    for (;;) {
        if (t_next == t_end) {
            uint32_t t0 = 0;
            if (lane == 0) t0 = atomicAdd(&a.queue[1], kChunk);
            t0 = uniform_u32(t0);
            if (t0 >= ntouched) break;
            t_next = t0;
            t_end = t0 + kChunk < ntouched ? t0 + kChunk : ntouched;
        }
        const uint32_t t = t_next++;
        const uint32_t pid = a.touched[t];
        wave_sync();
        // the row and its side arrays only depend on pid: issue their loads before the (dependent) inbox walk
        const uint32_t cur = ix.zero[(size_t)pid * kM2 + lane];
        const uint32_t curd = a.nbr_dist[(size_t)pid * kM2 + lane];
        const uint32_t cura = a.nbr_aux[(size_t)pid * kM2 + lane];
        const int ns0 = (int)a.row_nsel[pid];
        // the new points that chose `pid` (inbox), nearest first
        WState ns{news, 0, kM2, 0, 0u};
        uint32_t e = a.head[pid];
        // Nearly every update has ONE new point, and which one is known as soon as the inbox head is: look its distances
        // to the row's selected members up NOW, in flight together with the inbox walk (edge_dist / next), instead of
        // after it.  (Several new points: the value is simply not used.)
        uint32_t ed0 = 0, en0 = kInvalid;
        if (e != kInvalid) { ed0 = a.edge_dist[e]; en0 = a.next[e]; }   // (requested first: the probe below waits for its own loads)
        uint32_t dn_spec = kDlogMiss;
        if (e != kInvalid && a.use_dlog && lane < ns0 && cur != kInvalid)
            dn_spec = build_dlog_find(a, a.start + e / kM2, cur);
        uint32_t guard = 0;
        bool first_edge = true;
        while (e != kInvalid) {
            const uint32_t ed = first_edge ? ed0 : a.edge_dist[e];
            const uint32_t en = first_edge ? en0 : a.next[e];
            first_edge = false;
            const uint64_t k = ((uint64_t)ed << 32) | (a.start + e / kM2);
            const int idx = w_rank(ns, k);
            if (idx < ns.ef) w_insert(ns, idx, k);
            if (ns.plen > ns.ef) ns.plen = ns.ef;
            e = en;
            if (++guard > a.count) { status |= kStGuard; break; }
        }
        const int k_new = ns.plen;
        curaux[lane] = cura;
        const uint64_t inval = __ballot(cur == kInvalid);
        const int ncur = inval ? __builtin_ctzll(inval) : 64;
        const uint64_t key = lane < ncur ? (((uint64_t)curd << 32) | cur) : kMaxKey;
        const int total = ncur + k_new;
        bool defer = k_new > kMaxNewFast || total > (int)a.efc || guard > a.count;
        int nR = 0, nD = 0;
        bool handled = false;
        if (!defer && k_new == 1) {
            // One new point (the common case; the only case with max_batch = 1): the replay below collapses to a
            // few wave-wide tests, because every stored verdict in front of the new point stands, and behind it
            //   * the new point is pruned iff a selected entry in front of it is closer to it than `pid` is;
            //   * if it is selected instead, an old selected entry s behind it is newly pruned iff
            //     d(new, s) < d(pid, s)   (set P), and an old discarded entry keeps its verdict unless its
            //     stored pruner is in P — that cascade is left to the sequential replay.
            const uint64_t knew = news[0] & kKeyMask;
            const uint32_t new_pid = (uint32_t)knew, cd_new = (uint32_t)(knew >> 32);
            const bool selL = lane < ns0, discL = lane >= ns0 && lane < ncur;
            const uint64_t below = (1ull << lane) - 1ull;
            uint32_t dn = selL ? dn_spec : kDlogMiss;        // d(new, selected entry of this lane), requested before the inbox walk
            // (an entry the descent's reject filter turned down is there in bound form: it decides `d(new, s) < x` for the one x this
            //  lane compares with — d(new, pid) in front of the new point, d(pid, s) behind it — or counts as a miss)
            dn = dlog_resolve(dn, (lane < ncur && key < knew) ? cd_new : curd);
            const bool miss = selL && dn == kDlogMiss;
            const uint64_t mm = __ballot(miss);
#ifdef IDIST_PROBE
            pb[0] += 1;
            pb[2] += (uint32_t)__popcll(__ballot(selL));
            pb[3] += (uint32_t)__popcll(mm);
            pb[4] += (uint32_t)__popcll(__ballot(miss && cur < a.start && cur + a.count >= a.start));
            pb[5] += (uint32_t)__popcll(__ballot(miss && cur >= a.start));
#endif
            if (mm) {
                const int n0 = __popcll(mm), at = __popcll(mm & below);
                if (miss) act_p[at] = cur;
                const float* prow = ix.points + (size_t)new_pid * ix.stride;
                for (uint32_t o = lane * 4; o < ix.stride; o += 256)
                    *reinterpret_cast<float4*>(cq + o) = *reinterpret_cast<const float4*>(prow + o);
                wave_sync();
                dist_rounds<NB, RS, TAIL>(ix, cq, act_p, act_d, n0);
                wave_sync();
                if (miss) dn = act_d[at];
                hc.n_dist += (uint32_t)n0;
                hc.n_rows += 1;
            }
            const bool before = lane < ncur && key < knew;
            const int nb = __popcll(__ballot(selL && before));           // |R| when the new point's turn comes
            const uint64_t cm = __ballot(selL && before && dn < cd_new);   // strict <, :678
            const int nd_old = ncur - ns0;
            wave_sync();
            if (nb >= kM2) {
                // R is full before the new point is reached (:669-671): the row stands
                if (selL) sel[lane] = key;
                nR = ns0;
                nD = 0;
                handled = true;
            } else if (cm) {
                // pruned: it joins the discarded list at its place; R is unchanged
                const uint32_t pr = readlane_u32(cur, __builtin_ctzll(cm));
                const int npos = __popcll(__ballot(discL && before));
                if (selL) sel[lane] = key;
                if (discL) {
                    const int dp = (lane - ns0) + (before ? 0 : 1);
                    if (dp < kM2) { disc[dp] = key; dprn[dp] = cura; }
                }
                if (lane == 0 && npos < kM2) { disc[npos] = knew; dprn[npos] = pr; }
                nR = ns0;
                nD = nd_old + 1;
                handled = true;
            } else {
                const bool inP = selL && !before && dn < curd;         // d(new, s) < d(pid, s)
                const uint64_t P = __ballot(inP);
                bool casc = false;
                int dp = lane - ns0;                                   // place in the merged discarded list
                int pi = 0;
                for (uint64_t pm = P; pm; pm &= pm - 1ull, pi++) {
                    const int i = __builtin_ctzll(pm);
                    const uint32_t ppid = readlane_u32(cur, i);
                    const uint64_t kp = bcast_u64(key, i);
                    casc = casc || (discL && cura == ppid);
                    const int less = __popcll(__ballot(discL && key < kp));
                    if (discL && kp < key) dp += 1;
                    if (lane == i) dp = pi + less;
                }
                if (__ballot(casc) == 0ull) {
                    const int nP = __popcll(P);
                    if (selL && !inP) {
                        const int rp = lane - __popcll(P & below) + (before ? 0 : 1);
                        if (rp < kM2) sel[rp] = key;
                    }
                    if (lane == 0) sel[nb] = knew;
                    if ((discL || inP) && dp < kM2) { disc[dp] = key; dprn[dp] = inP ? new_pid : cura; }
                    nR = ns0 - nP + 1;
                    if (nR > kM2) nR = kM2;
                    nD = nd_old + nP;
                    handled = true;
                }
            }
            wave_sync();
        }
        if (!defer && !handled) {
            // candidates = sort(current U new) — nothing can be dropped by `idx < ef` (total <= ef)
            if (lane < ncur) curk[lane] = key;
            if (lane < ns0) X[lane] = cur;
            if (lane < k_new) X[ns0 + lane] = (uint32_t)news[lane];
            wave_sync();
            const uint64_t keyb = lane < k_new ? (news[lane] & kKeyMask) : kMaxKey;
            int ra = 0, rb = lane;
            for (int i = 0; i < ncur; i++) {
                const uint64_t o = curk[i];
                ra += o < key ? 1 : 0;
                rb += o < keyb ? 1 : 0;
            }
            for (int i = 0; i < k_new; i++) ra += (news[i] & kKeyMask) < key ? 1 : 0;
            if (lane < ncur) { W[ra] = key; kindx[ra] = ((lane < ns0 ? OLD_SEL : OLD_DISC) << 8) | (uint32_t)lane; }
            if (lane < k_new) { W[rb] = keyb; kindx[rb] = (NEW << 8) | (uint32_t)(ns0 + lane); }
            wave_sync();
            // distances of every new point to {old selected} U {new}: the new point's own descent (step A)
            // computed almost all of them — every member of its final W was expanded, so every entry of
            // their rows was visited — and logged them; only the misses (rows seen through a 32-link
            // upper-layer expansion, other new points of this step) are gathered again.
            const int nx = ns0 + k_new;
            for (int ai = 0; ai < k_new; ai++) {
                const uint32_t a_pid = (uint32_t)news[ai];
                uint32_t dv = kDlogMiss;
                if (lane < ns0 && a.use_dlog) dv = build_dlog_find(a, a_pid, X[lane]);
                if (dlog_is_bound(dv)) dv = kDlogMiss;      // (bound form, §4.5: this path compares a column with several values — recompute)
                if (lane < ns0) Dn[ai * kFastX + lane] = dv;
                int nmiss = 0;
                // columns [0, ns0) that missed + the other new points: gather those rows
                const uint64_t mm = __ballot(lane < ns0 && dv == kDlogMiss);
                const int n0 = __popcll(mm);
#ifdef IDIST_PROBE
                if (ai == 0) pb[1] += 1;
                pb[2] += (uint32_t)ns0;
                pb[3] += (uint32_t)n0;
                pb[4] += (uint32_t)__popcll(__ballot(lane < ns0 && dv == kDlogMiss && X[lane] < a.start && X[lane] + a.count >= a.start));
                pb[5] += (uint32_t)__popcll(__ballot(lane < ns0 && dv == kDlogMiss && X[lane] >= a.start));
                pb[6] += (uint32_t)ai;
#endif
                if (lane < ns0 && dv == kDlogMiss) {
                    const int at = __popcll(mm & ((1ull << lane) - 1ull));
                    act_x[at] = (uint32_t)lane;
                }
                // ... and the new points nearer to `pid` than this one: the replay reads d(a, b) of two new points from the row
                // of the LATER one (the one further from `pid`) only, and never the self column — so point ai needs the
                // columns of points 0 .. ai - 1, and the nearest new point (ai = 0) none at all
                if (lane < ai) act_x[n0 + lane] = (uint32_t)(ns0 + lane);
                nmiss = n0 + ai;
                wave_sync();
                if (nmiss > 0) {
                    for (int i = lane; i < nmiss; i += 64) act_p[i] = X[act_x[i]];
                    const float* prow = ix.points + (size_t)a_pid * ix.stride;
                    for (uint32_t o = lane * 4; o < ix.stride; o += 256)
                        *reinterpret_cast<float4*>(cq + o) = *reinterpret_cast<const float4*>(prow + o);
                    wave_sync();
                    dist_rounds<NB, RS, TAIL>(ix, cq, act_p, act_d, nmiss);
                    wave_sync();
                    for (int i = lane; i < nmiss; i += 64) Dn[ai * kFastX + act_x[i]] = act_d[i];
                    hc.n_dist += (uint32_t)nmiss;
                    hc.n_rows += 1;
                    wave_sync();
                }
            }
            (void)nx;
            // replay of select_heuristic (core/lib.rs:668-685) on stored verdicts.  Candidate i lives in
            // lane i & 63 (two registers per lane cover the <= 128 candidates); R, the discarded list and the
            // added set are register-resident too (entry j in lane j), so an iteration is readlanes + a ballot.
            wave_sync();
            uint32_t cw_lo[2], cw_hi[2], ckx[2];
            for (int h = 0; h < 2; h++) {
                const int i = h * 64 + lane;
                const uint64_t v = i < total ? W[i] : 0ull;
                cw_lo[h] = (uint32_t)v;
                cw_hi[h] = (uint32_t)(v >> 32);
                ckx[h] = i < total ? kindx[i] : 0u;
            }
            uint32_t r_pid = kInvalid, r_x = 0;        // lane j: R[j]
            uint32_t s_lo = 0, s_hi = 0;               // lane j: sel[j] (key)
            uint32_t d_lo = 0, d_hi = 0, d_pr = 0;     // lane j: disc[j] (key, pruner)
            uint32_t add_a = 0;                        // lane j: index of the j-th active point that entered R
            int nAdd = 0, nAct = k_new;
            for (int i = 0; i < total; i++) {
                if (nR >= kM2) break;                                   // :669-671
                const int h = i >> 6, li = i & 63;
                const uint32_t c_lo = readlane_u32(h ? cw_lo[1] : cw_lo[0], li);
                const uint32_t cd = readlane_u32(h ? cw_hi[1] : cw_hi[0], li);
                const uint32_t kx = readlane_u32(h ? ckx[1] : ckx[0], li);
                const uint32_t kind = kx >> 8, x = kx & 255u;
                bool pruned;
                uint32_t pr_pid = 0, cx = x;
                if (kind == OLD_SEL) {
                    // passed every older selected member already: only members added since can prune it
                    pruned = false;
                    if (nAdd > 0) {
                        bool closer = false;
                        if (lane < nAdd) closer = Dn[add_a * kFastX + x] < cd;           // strict <, :678
                        const uint64_t cm = __ballot(closer);
                        pruned = cm != 0ull;
                        if (pruned) pr_pid = X[ns0 + readlane_u32(add_a, __builtin_ctzll(cm))];
                    }
                } else if (kind == OLD_DISC) {
                    // stays discarded while the member that pruned it is still selected
                    const uint32_t p = curaux[x];
                    const bool still = __ballot(lane < nR && r_pid == p) != 0ull;
                    if (still) {
                        pruned = true;
                        pr_pid = p;
                    } else {
                        // its pruner is gone: evaluate it like a new point against the current selected set
                        if (nAct >= kMaxNewFast) { defer = true; break; }
                        const int ci = nAct++;
                        if (lane == 0) X[ns0 + ci] = c_lo;
                        const float* prow = ix.points + (size_t)c_lo * ix.stride;
                        for (uint32_t o = lane * 4; o < ix.stride; o += 256)
                            *reinterpret_cast<float4*>(cq + o) = *reinterpret_cast<const float4*>(prow + o);
                        wave_sync();
                        dist_rounds<NB, RS, TAIL>(ix, cq, X, Dn + ci * kFastX, ns0 + ci);
                        wave_sync();
                        hc.n_dist += (uint32_t)(ns0 + ci);
                        hc.n_rows += 1;
                        bool closer = false;
                        if (lane < nR) closer = Dn[ci * kFastX + r_x] < cd;             // columns of all earlier actives exist
                        const uint64_t cm = __ballot(closer);
                        pruned = cm != 0ull;
                        if (pruned) pr_pid = readlane_u32(r_pid, __builtin_ctzll(cm));
                        cx = (uint32_t)(ns0 + ci);
                    }
                } else {
                    const uint32_t ai = x - (uint32_t)ns0;
                    bool closer = false;
                    if (lane < nR) {
                        // distance between two active points: the row of the later-activated one has the column
                        const uint32_t dv = (r_x < (uint32_t)ns0 || r_x - (uint32_t)ns0 < ai) ? Dn[ai * kFastX + r_x]
                                                                                            : Dn[(r_x - (uint32_t)ns0) * kFastX + x];
                        closer = dv < cd;
                    }
                    const uint64_t cm = __ballot(closer);
                    pruned = cm != 0ull;
                    if (pruned) pr_pid = readlane_u32(r_pid, __builtin_ctzll(cm));
                }
                if (!pruned) {
                    if (lane == nR) { s_lo = c_lo; s_hi = cd; r_pid = c_lo; r_x = cx; }
                    if (cx >= (uint32_t)ns0) {
                        if (lane == nAdd) add_a = cx - (uint32_t)ns0;
                        nAdd++;
                    }
                    nR++;
                } else {
                    if (lane == nD) { d_lo = c_lo; d_hi = cd; d_pr = pr_pid; }   // nD < 64 whenever it matters (back-fill room)
                    nD++;
                }
            }
            // hand the register-resident lists to the common tail below
            wave_sync();
            if (lane < nR) sel[lane] = ((uint64_t)s_hi << 32) | s_lo;
            if (lane < nD && lane < kM2) { disc[lane] = ((uint64_t)d_hi << 32) | d_lo; dprn[lane] = d_pr; }
            wave_sync();
        }
        if (defer) {
            // leave the inbox in place; step B2 redoes this node from scratch
            if (lane == 0) a.slow[atomicAdd(a.n_slow, 1u)] = pid;
            deferred++;
            wave_sync();
            continue;
        }
        if (lane == 0) { a.head[pid] = kInvalid; a.row_nsel[pid] = (uint32_t)nR; }
        int nsel = nR;
        if (a.keep_pruned) {                                            // :687-695
            if (nD > kM2) nD = kM2;
            int take = kM2 - nsel;
            if (take > nD) take = nD;
            if (lane < take) sel[nsel + lane] = disc[lane];
            if (take > 0) nsel += take;
            wave_sync();
        }
        ix.zero[(size_t)pid * kM2 + lane] = lane < nsel ? (uint32_t)sel[lane] : kInvalid;     // ZeroNode::rewrite
        a.nbr_dist[(size_t)pid * kM2 + lane] = lane < nsel ? (uint32_t)(sel[lane] >> 32) : 0u;
        a.nbr_aux[(size_t)pid * kM2 + lane] = (lane >= nR && lane < nsel) ? dprn[lane - nR] : 0u;
        updates++;
        wave_sync();
    }
    if (lane == 0) {
        if (status) atomicOr(a.status, status);
#ifdef IDIST_PROBE
        for (int i = 0; i < 7; i++) if (pb[i]) atomicAdd(&a.stats[9 + i], (unsigned long long)pb[i]);
#endif
        if (updates | deferred) {
            atomicAdd(&a.stats[3], (unsigned long long)hc.n_dist);
            atomicAdd(&a.stats[4], (unsigned long long)hc.n_rows);
            atomicAdd(&a.stats[5], (unsigned long long)updates);
            atomicAdd(&a.stats[6], (unsigned long long)updates);
        }
    }
}

// ---------------------------------------------------------------------------
// Build step B: for every node that was selected by at least one new point,
// Search::add_neighbor_heuristic + ZeroNode::rewrite (core/lib.rs:485-496,
// :616-631; core/types.rs:88-98).  With one new point per step this is the
// reference's loop body verbatim; with several, all new points that chose the
// node are pushed (nearest first) before its single re-selection.
//
// On-chip working set: the distances of the current neighbours are kept next to
// the ids (nbr_dist, bit-identical to recomputing them: the canonical distance
// is symmetric), candidate rows are fetched 8 at a time into an LDS tile, and
// the selected set stays in that tile, so every row crosses HBM once per update.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t smem_bytes_update(uint32_t nb, uint32_t rt, uint32_t fc = 8) {
    return tile_floats(nb, rt + fc) * 4 + (size_t)(kUpdW + 72 + 64 + 64 + 64) * 8 + 4 * 64 * 4;
}

template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void build_update_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    const int nb = NB >= 0 ? NB : (int)ix.nb;
    Tile tile;
    tile.rt = (int)a.rt;
    tile.fc = NB >= 0 ? 8 : (int)a.fc;
    tile.slots = tile.rt + tile.fc;
    tile.blk = reinterpret_cast<float*>(smem_raw);
    tile.rem = tile.blk + (size_t)nb * tile.slots * 32;
    uint64_t* W = reinterpret_cast<uint64_t*>(tile.blk + tile_floats((uint32_t)nb, (uint32_t)tile.slots));
    uint64_t* news = W + kUpdW;          // 64 + 8, sorted nearest first
    uint64_t* sel = news + 72;
    uint64_t* disc = sel + 64;
    uint64_t* curk = disc + 64;
    uint32_t* act_pid = reinterpret_cast<uint32_t*>(curk + 64);
    uint32_t* act_dist = act_pid + 64;
    uint32_t* dprn = act_dist + 64;
    uint32_t* out_aux = dprn + 64;
    const int lane = lane_id();
    const uint32_t nslow = *a.n_slow;
    HeurCounters hc{0, 0};
    uint32_t updates = 0, status = 0;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&a.queue[3], 1u);
        t = uniform_u32(t);
        if (t >= nslow) break;
        const uint32_t pid = a.slow[t];
        wave_sync();
        // drain the inbox: keep the <= 64 nearest new points (a node holds 64 links at most)
        WState ns{news, 0, kM2, 0, 0u};
        uint32_t e = a.head[pid];
        uint32_t guard = 0;
        while (e != kInvalid) {
            const uint64_t k = ((uint64_t)a.edge_dist[e] << 32) | (a.start + e / kM2);
            const int idx = w_rank(ns, k);
            if (idx < ns.ef) w_insert(ns, idx, k);
            if (ns.plen > ns.ef) ns.plen = ns.ef;
            e = a.next[e];
            if (++guard > a.count) { status |= kStGuard; break; }
        }
        if (lane == 0) a.head[pid] = kInvalid;
        const int k_new = ns.plen;

        // current = zero.nearest_iter(pid) (:487) with the stored distances to points[pid] (:489)
        const uint32_t cur = ix.zero[(size_t)pid * kM2 + lane];
        const uint32_t curd = a.nbr_dist[(size_t)pid * kM2 + lane];
        const uint64_t inval = __ballot(cur == kInvalid);
        const int ncur = inval ? __builtin_ctzll(inval) : 64;
        const uint64_t key = lane < ncur ? (((uint64_t)curd << 32) | cur) : kMaxKey;

        // insertion.reset(); push(new); push(current...) with ef = ef_construction (:440, :625-629)
        WState st{W, 0, (int)a.efc, 0, 0u};
        if (ncur + k_new <= (int)a.efc) {
            // nothing can be dropped by `idx < ef` => W is simply all keys sorted: parallel rank sort
            if (lane < ncur) curk[lane] = key;
            wave_sync();
            const uint64_t keyb = lane < k_new ? (news[lane] & kKeyMask) : kMaxKey;
            int ra = 0, rb = lane;
            for (int i = 0; i < ncur; i++) {
                const uint64_t o = curk[i];
                ra += o < key ? 1 : 0;
                rb += o < keyb ? 1 : 0;
            }
            for (int i = 0; i < k_new; i++) ra += (news[i] & kKeyMask) < key ? 1 : 0;
            if (lane < ncur) W[ra] = key;
            if (lane < k_new) W[rb] = keyb;
            st.plen = ncur + k_new;
            wave_sync();
        } else {
            for (int i = 0; i < k_new; i++) {
                const uint64_t k = news[i] & kKeyMask;
                const int idx = w_rank(st, k);
                if (idx < st.ef) w_insert(st, idx, k);
            }
            for (int i = 0; i < ncur; i++) {
                const uint64_t k = bcast_u64(key, i);
                const int idx = w_rank(st, k);
                if (idx < st.ef) w_insert(st, idx, k);
            }
        }
        hc.n_ref += (uint32_t)(k_new + ncur);                           // the pushes of :626-629: one distance call each
        // select_heuristic over ALL of `nearest` (no truncate in add_neighbor_heuristic, :630)
        int n_selected = 0;
        const int nsel = select_heuristic_tiled<NB, RS, TAIL>(ix, st.W, st.plen, a.keep_pruned != 0, tile, sel, disc,
                                                              act_pid, act_dist, hc, n_selected, dprn, out_aux);
        if (lane == 0) a.row_nsel[pid] = (uint32_t)n_selected;
        a.nbr_aux[(size_t)pid * kM2 + lane] = out_aux[lane];
        // ZeroNode::rewrite (core/types.rs:88-98): rows are prefix-valid, so clearing to the end is identical
        ix.zero[(size_t)pid * kM2 + lane] = lane < nsel ? (uint32_t)sel[lane] : kInvalid;
        a.nbr_dist[(size_t)pid * kM2 + lane] = lane < nsel ? (uint32_t)(sel[lane] >> 32) : 0u;
        updates++;
        wave_sync();
    }
    if (lane == 0) {
        if (status) atomicOr(a.status, status);
        if (updates) {
            atomicAdd(&a.stats[3], (unsigned long long)hc.n_dist);
            atomicAdd(&a.stats[4], (unsigned long long)hc.n_rows);
            atomicAdd(&a.stats[5], (unsigned long long)updates);
            atomicAdd(&a.stats[7], (unsigned long long)updates);
            atomicAdd(&a.stats[8], (unsigned long long)hc.n_ref);
        }
    }
}

}  // namespace idist
