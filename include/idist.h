/*
 * idist.h — C ABI of libidist.so, the MI355X (gfx950) HNSW build+search engine
 * that sits under instant-distance's Builder / Hnsw / HnswMap / Search / Point
 * API (the drop-in boundary).
 *
 * The reference has NO FFI of its own: its boundary is the public Rust API.
 * Each entry point below names the reference item it stands in for
 * (paths relative to /root/reference/, core/ = instant-distance/src/).  The
 * Rust-side binding a maintainer adds is shown in INTEGRATION.md and shipped as
 * source in instant-distance_amd/rust-shim/.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer, the
 *     library owns all device memory behind the opaque handles;
 *   - every call returns an idist_status (0 = ok); idist_last_error() gives a
 *     thread-local message.  The reference API is infallible (it only panics at
 *     core/lib.rs:256 and :148) so the shim `expect()`s these;
 *   - points are handed over ALREADY IN PointId ORDER: the seed -> permutation
 *     step (core/lib.rs:214,257-270, `rand` crate) stays on the caller's side;
 *   - an idist_index is immutable after build/import and may be shared by any
 *     number of threads; every idist_search_ctx owns its own stream + scratch
 *     (the role of `&mut Search`, core/lib.rs:352-356).
 *   - there is no CPU fallback: without a gfx950 device every compute entry
 *     point fails with IDIST_ERR_NO_DEVICE.
 *
 * Environment: libidist.so reads exactly three variables, all host-side behaviour, sampled once per context —
 *   IDIST_COMBINE=0         every scalar host-pointer call makes its own launch (no riding along in another thread's launch)
 *   IDIST_SYNC=stream       narrow host-pointer calls wait with hipStreamSynchronize instead of for the kernel's completion word
 *   IDIST_KERNEL_EVENTS=0   no HIP events around the search kernels (idist_search_ctx_kernel_times then has nothing)
 * — and nothing in the environment can change which kernels it runs or which graph it builds.  The knobs that select other
 * implementations of the same decisions (IDIST_WALK, IDIST_VISITED, IDIST_TAB_*, IDIST_BUILD_*, ...) exist for the parity tests
 * and A/B measurements and are compiled into the TEST build of the same sources only (libidist_variants.so, `make variants`;
 * DESIGN.md's appendix lists them).  The library never edits the environment either: GPU_MAX_HW_QUEUES (one hardware queue per
 * searching thread's stream) is the host's to set before its first HIP call.
 */
#ifndef IDIST_H
#define IDIST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDIST_M 32u                 /* core/lib.rs:787 */
#define IDIST_M2 64u                /* ZeroNode slots, core/types.rs:83-85 */
#define IDIST_INVALID 0xFFFFFFFFu   /* PointId INVALID, core/types.rs:293 */
#define IDIST_MAX_LAYERS 64u
#define IDIST_MAX_EF 4096u

typedef int32_t idist_status;
enum {
    IDIST_OK = 0,
    IDIST_ERR_INVALID_ARG = 1,
    IDIST_ERR_NO_DEVICE = 2,      /* no gfx950 GPU visible: there is no CPU path */
    IDIST_ERR_HIP = 3,            /* HIP runtime error, see idist_last_error() */
    IDIST_ERR_UNSUPPORTED = 4,    /* option the GPU engine does not implement (yet) */
    IDIST_ERR_BAD_GRAPH = 5,      /* imported adjacency violates the reference's invariants */
    IDIST_ERR_TIE_OVERFLOW = 6,   /* device-pointer launches only: the tie region overflowed, enqueue the batch again (idist_config.tie_policy) */
    IDIST_ERR_INTERNAL = 7        /* device-side guard tripped */
};

enum { IDIST_TIES_STRICT = 0, IDIST_TIES_DROP = 1 };

enum {
    IDIST_METRIC_L2SQ = 0, /* FloatArray::distance, instant-distance-py/src/lib.rs:378-421 */
    IDIST_METRIC_L2 = 1    /* sqrt of it: tests/all.rs:93-97, examples/colors.rs:21-25 */
};

/* Builder fields, core/lib.rs:23-31 (defaults :101-128). */
typedef struct idist_config {
    uint32_t ef_search;         /* Builder::ef_search, default 100 */
    uint32_t ef_construction;   /* Builder::ef_construction, default 100 */
    float ml;                   /* Builder::ml, default 1/ln(32) */
    int32_t has_heuristic;      /* Builder::select_heuristic(Some/None), default Some */
    int32_t extend_candidates;  /* Heuristic::extend_candidates, default false.  true: core/lib.rs:648-664 without the
                                   locks that make it deadlock upstream (:649 vs :438) = the oracle's restatement;
                                   the build is then sequential whatever max_batch says (one insertion per launch) */
    int32_t keep_pruned;        /* Heuristic::keep_pruned, default true */
    int32_t metric;             /* IDIST_METRIC_* (the Point::distance the caller would supply) */
    uint32_t max_batch;         /* build scheduling: 1 = strictly sequential insertion (the
                                   deterministic contract = reference with one rayon thread);
                                   0 = library default; k = at most k concurrent inserts per
                                   step (the rayon for_each of core/lib.rs:316-318). */
    int32_t tie_policy;         /* IDIST_TIES_*: what happens when more un-expanded candidates sit exactly at the furthest
                                   distance of a full `nearest` than the engine's tie region holds (mass duplicates, dense
                                   integer grids).  The reference's candidate heap is unbounded (core/lib.rs:564).
                                   STRICT (default): the result is ALWAYS the reference's.  The region grows on demand — a
                                   build that overflows is repeated with 8x the region, a host-pointer search batch is
                                   searched again with 4x — and beyond 4096 entries the ties that do not fit go to a bag
                                   in HBM and come back in (distance, pid) order: unbounded like the reference's heap, slow
                                   only on data that needs it.  IDIST_ERR_TIE_OVERFLOW is returned in ONE place only:
                                   idist_search_ctx_status after a device-pointer launch (idist_search_batch_device cannot
                                   re-run a launch it did not wait for); that context uses the larger region / the bags
                                   from its next launch on, so the caller enqueues the same batch again.
                                   DROP: keep the configured region, never expand the ties that do not fit, go on —
                                   deterministic, flagged in idist_build_stats.tie_overflow /
                                   idist_search_ctx_tie_overflowed, no longer bit-identical on such data. */
    uint32_t tie_capacity;      /* initial size of that tie region, 0 = 64 (the default), at most 4096.  It lives in LDS
                                   next to `nearest` (8 B per entry): a larger one spares such data the repeated launch at
                                   the price of fewer resident waves per CU.  (HBM bags: n keys per query slot, as many
                                   slots as fit 1 GiB.) */
} idist_config;

typedef struct idist_index idist_index;
typedef struct idist_search_ctx idist_search_ctx;

typedef struct idist_index_info {
    uint32_t n, dim;
    uint32_t row_stride;        /* floats per stored row (blocked layout) */
    uint32_t n_upper;           /* Hnsw.layers.len() */
    uint32_t ef_search;
    int32_t metric;
    int32_t device;
    uint32_t layer_len[IDIST_MAX_LAYERS]; /* layer_len[l-1] = rows of layers[l-1], l = 1..n_upper */
    uint32_t tie_capacity;      /* the tie region the build ended up using (see idist_config.tie_capacity) */
} idist_index_info;

/* Raw device views (for RCCL replication by the host: torch.distributed / ncclBroadcast). */
typedef struct idist_device_buffers {
    void* points;  size_t points_bytes;  /* f32 [n][row_stride], blocked layout */
    void* zero;    size_t zero_bytes;    /* u32 [n][64] */
    void* upper;   size_t upper_bytes;   /* u32 [sum layer_len][32], layer 1 first */
} idist_device_buffers;

/* Work counters that define the algorithmic bytes. */
typedef struct idist_build_stats {
    uint64_t n_dist;        /* descent distance evaluations (Search::push past `visited`) */
    uint64_t n_exp0;        /* zero-array expansions */
    uint64_t n_expU;        /* snapshot (UpperNode) expansions */
    uint64_t n_sel_pairs;   /* candidate pairs of select_heuristic DECIDED here (by the Gram-matrix filter, a memoised
                               verdict or the canonical distance) — this engine's work, not the reference's count */
    uint64_t n_heur_rows;   /* candidate rows staged for select_heuristic */
    uint64_t n_updates;     /* neighbour rows rewritten (ZeroNode::rewrite) */
    uint64_t n_updates_fast; /* ... of which through the memoised re-selection */
    uint64_t n_updates_full; /* ... of which through the full re-selection */
    uint64_t n_batches;
    double seconds;         /* device time of the whole build (HIP events) */
    uint64_t tie_overflow;  /* 1 if the tie capacity was exceeded during the build (only with IDIST_TIES_DROP) */
    uint64_t n_heur_ref;    /* distance calls of select_heuristic / add_neighbor_heuristic exactly as the reference makes them
                               (early-exit `any`, core/lib.rs:676-679; the pushes of :626-629) — available (non-zero) only when
                               the whole build ran through the reference-order kernels: max_batch = 1 with
                               IDIST_BUILD_A2=tile IDIST_BUILD_NO_FAST=1, or extend_candidates; tests compare it with the oracle */
    uint64_t n_filter_examined; /* descent pushes (of n_dist) the reject filter looked up in the compact copy of the rows first ... */
    uint64_t n_filter_rejected; /* ... and of those, the candidates whose f32 row was never fetched (0 / 0: unfiltered descents) */
    uint64_t filter_row_bytes;  /* bytes of one compact row (0: unfiltered descents): what an examined candidate costs instead of 4 * dim */
} idist_build_stats;

/* ---- library ------------------------------------------------------------ */
const char* idist_last_error(void);
const char* idist_version(void);
idist_status idist_device_count(int32_t* out);
/* Builder::default(), core/lib.rs:101-113 (seed stays with the caller). */
idist_status idist_default_config(idist_config* cfg);
/* Layer sizing of Hnsw::new, core/lib.rs:238-250 (f32 multiply + truncation).
 * cum[l] = nodes present on layer l, cum[0] = n; returns the layer count in *n_layers. */
idist_status idist_layer_sizes(uint32_t n, float ml, uint32_t* cum, uint32_t cap, uint32_t* n_layers);

/* Host helper (no GPU): the shuffle of Hnsw::new, core/lib.rs:214,257-270 — keys drawn with
 * SmallRng::seed_from_u64(seed).random_range(0..n), sort_unstable by (key, index).
 * out_pid[orig] = PointId, order[pid] = original index (either may be NULL).
 * PARITY UNPINNED: the `rand` crate is not part of /root/reference (Cargo.lock is git-ignored);
 * this restates xoshiro256++ / SplitMix64 seeding / widening-multiply range sampling.  A Rust
 * caller keeps using the real crate and passes points in PointId order. */
idist_status idist_permutation(uint64_t seed, uint32_t n, uint32_t* out_pid, uint32_t* order);

/* ---- index -------------------------------------------------------------- */
/* Builder::build_hnsw / Hnsw::new, core/lib.rs:83-85, 209-345 (after the permutation):
 * inserts points 1..n per layer range (Construction::insert, :437-528, select_heuristic
 * :636-698) on the GPU.  `points` is host memory, row-major n x dim, PointId order. */
idist_status idist_index_build(const float* points, uint32_t n, uint32_t dim,
                               const idist_config* cfg, int32_t device, idist_index** out);
/* Same, points already resident on `device` (row-major n x dim f32). */
idist_status idist_index_build_device(const void* d_points, uint32_t n, uint32_t dim,
                                      const idist_config* cfg, int32_t device, idist_index** out);
idist_status idist_index_build_stats(const idist_index* idx, idist_build_stats* out);

/* Builder::progress (core/lib.rs:70-75 behind the `indicatif` feature; the bar is advanced at :217-221,
 * :306-309, :332-334, :520-526).  No callbacks cross this ABI, so progress is a pollable object: create it,
 * arm it on the thread that is about to call idist_index_build*, and read it from any other thread while
 * that (blocking) call runs.  `total` = points.len() (bar.set_length, :219), `done` = points inserted so
 * far (the bar's position), `layer` = the layer being built (the bar's message, :307), -1 before the first
 * and after the last.  The device advances it after every build step. */
typedef struct idist_progress idist_progress;
idist_status idist_progress_new(idist_progress** out);
void idist_progress_free(idist_progress* p);
/* the next idist_index_build / idist_index_build_device call made by THIS thread reports into `p` */
idist_status idist_progress_watch_next_build(idist_progress* p);
idist_status idist_progress_get(const idist_progress* p, uint64_t* done, uint64_t* total, int32_t* layer);

/* Adopt an existing graph: the fields of `struct Hnsw`, core/lib.rs:194-199
 * (points, zero: Vec<ZeroNode>, layers: Vec<Vec<UpperNode>>).  Rows are validated
 * against the reference's invariants (ids < n, no duplicate before the first INVALID). */
idist_status idist_index_import(const float* points, uint32_t n, uint32_t dim,
                                const idist_config* cfg, const uint32_t* zero,
                                const uint32_t* const* layers, const uint32_t* layer_len,
                                uint32_t n_upper, int32_t device, idist_index** out);
/* Empty index with the layer structure `layer_len` (replication target). */
idist_status idist_index_alloc(uint32_t n, uint32_t dim, const idist_config* cfg,
                               const uint32_t* layer_len, uint32_t n_upper, int32_t device,
                               idist_index** out);
/* Copy the graph back to host: zero must hold n*64, layers[l-1] layer_len[l-1]*32 u32. */
idist_status idist_index_export(const idist_index* idx, uint32_t* zero, uint32_t* const* layers);
idist_status idist_index_get_info(const idist_index* idx, idist_index_info* out);
idist_status idist_index_device_buffers(const idist_index* idx, idist_device_buffers* out);
/* Hnsw.ef_search is a field of the index (core/lib.rs:195); bench sweeps change it. */
idist_status idist_index_set_ef_search(idist_index* idx, uint32_t ef_search);
void idist_index_free(idist_index* idx);

/* ---- search ------------------------------------------------------------- */
/* Search::default(), core/lib.rs:767-778: reusable scratch (visited set, W, candidates) for up to `slots`
 * queries in flight.  The visited set is one BIT per point and slot (core/types.rs:13-59 keeps a byte and a
 * generation; membership is all that is observable).  slots = 0: like the reference's Search, which sizes its
 * scratch on first use (core/lib.rs:363), the context starts with ONE slot (n/8 bytes) and grows to what the
 * batches it is given need, at most a full chip (16 waves per CU = 4096 slots: 512 MB at 1M points; the default walk
 * keeps the visited set of a query in LDS and touches this bitmap only when it overflows).
 * A context is bound to the index it was created for (by identity, not by address). */
idist_status idist_search_ctx_new(const idist_index* idx, uint32_t slots, idist_search_ctx** out);
void idist_search_ctx_free(idist_search_ctx* ctx);
/* Back `slots` query slots now (Vec::reserve on the Search's scratch).  Growing is the one operation of a context that
 * synchronises the whole device and allocates (the old bitmaps may still be in use by launches on any stream): a launch
 * wider than the slots a context has grows it first, so a caller of idist_search_batch_device who needs launches that
 * never synchronise (stream capture, other contexts busy on the device) reserves min(widest batch, a full chip) up front. */
idist_status idist_search_ctx_reserve(idist_search_ctx* ctx, uint32_t slots);

/* Hnsw::search, core/lib.rs:352-383, for nq queries at once (nq == 1 backs the scalar
 * call; scalar calls from many threads — one context each, the reference's `&mut Search` per thread — are combined into
 * few launches once more than eight are in flight on the index: same results, higher aggregate rate).  Results are Search.nearest: <= ef_search (pid, distance) pairs, nearest first
 * (the caller's `.take(k)` is a prefix).  out_pid/out_dist: nq*ef_search, padded with
 * IDIST_INVALID / +inf; out_count: nq; out_counters (optional): nq*3
 * {n_dist, n_exp0, n_expU} per query.  Host pointers; blocks until done. */
idist_status idist_search_batch(const idist_index* idx, idist_search_ctx* ctx,
                                const float* queries, uint32_t nq, uint32_t* out_pid,
                                float* out_dist, uint32_t* out_count, uint32_t* out_counters);
/* Same with every pointer in device memory, enqueued on `hip_stream` (a hipStream_t, may
 * be NULL) without synchronising: inputs/outputs stay resident in HBM.  A context is one
 * `&mut Search`: its launches must be ordered among themselves (one stream, or explicit
 * dependencies) — they share its visited slots and its work queue; use one context per
 * concurrent stream.  Batches of up to ~100 queries through idist_search_batch (host
 * pointers) cross PCIe through a pinned buffer owned by the context, without copy calls. */
idist_status idist_search_batch_device(const idist_index* idx, idist_search_ctx* ctx,
                                       const void* d_queries, uint32_t nq, void* d_out_pid,
                                       void* d_out_dist, void* d_out_count, void* d_out_counters,
                                       void* hip_stream);
/* Device-side status of the last launches on ctx (after the stream is synchronised). */
idist_status idist_search_ctx_status(idist_search_ctx* ctx);
/* IDIST_TIES_DROP only: *out = 1 if a search since the last call exceeded the tie capacity (then reset). */
idist_status idist_search_ctx_tie_overflowed(idist_search_ctx* ctx, int32_t* out);
/* Diagnostics of the walk's reject filter: wide batches on indexes the filter applies to look every new
 * candidate up in a one-byte-per-coordinate copy of the rows first and fetch the f32 row only of those that copy cannot prove to
 * lie beyond `nearest`'s furthest entry — the candidates `Search::push` turns down at core/lib.rs:712-714 without using their
 * distance.  Results and the counters of idist_search_batch are the reference's either way; these two numbers say how many
 * candidates the filter examined and how many f32 rows it spared, summed over the searches of this context since the last
 * reset (bench.py derives the bytes a launch really requests from them).  Synchronise the context's stream first. */
idist_status idist_search_ctx_filter_counts(idist_search_ctx* ctx, uint64_t* examined, uint64_t* rejected, int32_t reset);
/* HIP-event duration of the last search kernel launched through ctx, milliseconds. */
idist_status idist_search_ctx_last_kernel_ms(idist_search_ctx* ctx, float* ms);
/* Durations (ms) of the most recent search-kernel launches through ctx, oldest first, measured
 * with HIP events recorded on the launch stream around each kernel (a ring of IDIST_EVENT_RING
 * pairs; no synchronisation happens until this call).  *n_out = number written (<= cap). */
#define IDIST_EVENT_RING 64u
idist_status idist_search_ctx_kernel_times(idist_search_ctx* ctx, float* ms, uint32_t cap, uint32_t* n_out);

/* ---- several GPUs of one node --------------------------------------------------- */
/* `Hnsw` is Sync — Hnsw::search takes &self and every mutable bit lives in the caller's Search
 * (core/lib.rs:352-356) — so the reference shares ONE index between all its threads.  Across GPUs the index
 * is replicated once and the queries are block-partitioned; the build itself does not shard (every insert
 * reads and mutates one graph): it runs on one GPU and is replicated ("replicas only").
 *
 * idist_replicate: replicas[i] receives a copy of `root` on devices[i], device to device over xGMI
 * (hipMemcpyPeerAsync, peer access enabled where the link allows; all destinations are in flight together,
 * one stream per destination).  devices[i] may be the root's own device (a plain copy).  Each replica is an
 * ordinary index: search it with its own contexts, free it with idist_index_free.  On error nothing is left
 * allocated.  (The multi-PROCESS flavour — one rank per GPU, RCCL broadcast into idist_index_alloc'ed
 * replicas through idist_index_device_buffers — is instant-distance_amd/dist.py.) */
idist_status idist_replicate(const idist_index* root, const int32_t* devices, uint32_t n_devices,
                             idist_index** replicas);
/* The same replication as ONE RCCL broadcast per buffer (points, zero, upper) over a single-process communicator
 * (ncclCommInitAll over the root's device and the distinct destination devices, ncclBroadcast inside one group,
 * one stream per device): RCCL picks the ring / tree over xGMI instead of the root feeding every peer itself.
 * librccl.so is loaded on first use (dlopen; IDIST_ERR_UNSUPPORTED if it is missing).  A destination on the root's
 * own device is served inside the same broadcast (recvbuff != sendbuff on the root rank).  *seconds (optional):
 * wall time of the broadcasts, communicator set-up excluded. */
idist_status idist_replicate_rccl(const idist_index* root, const int32_t* devices, uint32_t n_devices,
                                  idist_index** replicas, double* seconds);
/* Hnsw::search for nq host queries over n_shards (replica, context) pairs: the queries are cut into the
 * contiguous ranges [nq*i/n_shards, nq*(i+1)/n_shards) — what a caller partitioning over threads does —
 * each searched by idist_search_batch on its own GPU from its own host thread, results written to the same
 * ranges of the outputs.  No collective, no device-to-device traffic; the result is identical to one
 * idist_search_batch over the whole batch. */
idist_status idist_search_batch_sharded(const idist_index* const* replicas, idist_search_ctx* const* ctxs,
                                        uint32_t n_shards, const float* queries, uint32_t nq,
                                        uint32_t* out_pid, float* out_dist, uint32_t* out_count,
                                        uint32_t* out_counters);

/* Point::distance for id lists (core/lib.rs:780-782 as used at :709-710): out[q][i] =
 * distance(queries[q], points[ids[q][i]]) for i < n_ids; IDIST_INVALID ids give +inf.
 * Host pointers. The batched gather-L2 kernel on its own. */
idist_status idist_distance_batch(const idist_index* idx, const float* queries, uint32_t nq,
                                  const uint32_t* ids, uint32_t n_ids, float* out_dist);

/* The walk's reject filter for id lists, on its own: out[q][i] = a LOWER BOUND of
 * distance(queries[q], points[ids[q][i]]) in the index's metric, computed from the one-byte-per-coordinate copy of the row
 * alone with the walk's own loads and arithmetic — `Search::push` (core/lib.rs:704-720) is spared the f32 row of a candidate
 * whose bound exceeds the furthest distance of a full `nearest`.  0 where there is no bound (IDIST_INVALID ids, rows with a
 * non-finite coordinate, indexes without the copy).  Host pointers.  tests/ hold the bound against idist_distance_batch;
 * bench.py uses one pass over all rows as the known byte count its traffic counters are calibrated on. */
idist_status idist_filter_bound_batch(const idist_index* idx, const float* queries, uint32_t nq,
                                      const uint32_t* ids, uint32_t n_ids, float* out_bound);

/* Exact k nearest neighbours by exhaustive scan with the canonical distance (the
 * brute-force check of tests/all.rs:60-67); host pointers. */
idist_status idist_bruteforce(const idist_index* idx, const float* queries, uint32_t nq,
                              uint32_t k, uint32_t* out_pid, float* out_dist);

#ifdef __cplusplus
}
#endif
#endif /* IDIST_H */
