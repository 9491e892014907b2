/*
 * idist_oracle.h — CPU ORACLE for the instant-distance HNSW hot path.
 *
 * >>> TEST INFRASTRUCTURE ONLY. <<<
 * This is a plain-C restatement of the reference algorithm
 * (/root/reference/instant-distance/src/{lib.rs,types.rs} and the FloatArray
 * distance of /root/reference/instant-distance-py/src/lib.rs:378-421).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.  The product
 * (instant_distance_amd + libidist.so) never links, imports or calls it.
 *
 * Parity pins:
 *   - the reference crate cannot be compiled here (no rustc/cargo), so the
 *     oracle is pinned against every known answer the reference's own tests
 *     hold for this path (tests/all.rs `map` exact distances, recall floors of
 *     `random_heuristic` / `random_simple`, examples/colors.rs nearest colour,
 *     test.py self-query) — tests/test_oracle_golden.py.
 *   - neighbour-ID golden vectors do not exist in the reference (its tests
 *     use random seeds and never assert IDs); the rand-crate permutation is
 *     therefore "parity unpinned" and is kept on the caller's side.
 */
#ifndef IDIST_ORACLE_H
#define IDIST_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDO_M 32u              /* core/lib.rs:787 */
#define IDO_M2 64u             /* zero-layer slots, core/types.rs:83-85 */
#define IDO_INVALID 0xFFFFFFFFu /* core/types.rs:293 */

enum { IDO_METRIC_L2SQ = 0, /* py/lib.rs:378-421 (no sqrt)                  */
       IDO_METRIC_L2 = 1 }; /* tests/all.rs:93-97, examples/colors.rs:21-25 */

typedef struct {
    uint32_t ef_search;        /* core/lib.rs:104 default 100 */
    uint32_t ef_construction;  /* core/lib.rs:105 default 100 */
    float ml;                  /* core/lib.rs:107 default 1/ln(32) */
    int32_t has_heuristic;     /* core/lib.rs:106 default Some(..) */
    int32_t extend_candidates; /* core/lib.rs:124 default false */
    int32_t keep_pruned;       /* core/lib.rs:125 default true */
    int32_t metric;            /* IDO_METRIC_* */
} ido_config;

typedef struct {
    uint64_t n_dist;  /* push() calls that passed the visited test, core/lib.rs:705-710 */
    uint64_t n_exp0;  /* layer-0 expansions, core/lib.rs:606 */
    uint64_t n_expU;  /* upper-layer expansions */
    uint64_t n_heur;  /* distance calls inside select_heuristic, core/lib.rs:677 */
} ido_counters;

typedef struct ido_index ido_index;
typedef struct ido_search ido_search;

void ido_default_config(ido_config* c);

/* Canonical distance (arithmetic contract, SURVEY §8c choice A): 8 stride-8
 * FMA chains, hi+lo fold, 4-wide FMA tail, (s0+s2)+(s1+s3).  dim is padded
 * with zeros to a multiple of 4 (the binding zero-pads too, py/lib.rs:363-376). */
float ido_distance(const float* a, const float* b, uint32_t dim, int metric);
float ido_distance_scalar(const float* a, const float* b, uint32_t dim, int metric);

/* Layer sizing, core/lib.rs:238-250.  Writes cumulative sizes bottom layer
 * last into cum[0..n_layers) ordered TOP FIRST?  No: cum[l] = number of
 * points on layer l (layer 0 = n).  Returns the number of layers. */
uint32_t ido_layer_sizes(uint32_t n, float ml, uint32_t* cum, uint32_t cap);

/* rand-crate permutation restatement (PARITY UNPINNED — rand 0.10 is not in
 * /root/reference): SmallRng = xoshiro256++ seeded via SplitMix64,
 * random_range(0..n) by widening multiply with rejection.  core/lib.rs:214,257-270.
 * out_pid[orig] = new PointId ; order[pid] = original index. */
void ido_permutation(uint64_t seed, uint32_t n, uint32_t* out_pid, uint32_t* order);

/* Build from points ALREADY in PointId order (row-major n x dim).
 * threads == 1 : pid-ascending sequential insertion — the exact contract
 *                (= reference with RAYON_NUM_THREADS=1).
 * threads  > 1 : per-layer parallel-for with per-node locks, mirrors
 *                core/lib.rs:316-318 (non-deterministic, timing only).
 * threads < -1 : the exact contract again — insertions strictly sequential — with the <= 64 independent neighbour
 *                updates of one insertion (core/lib.rs:481-516, extend_candidates = false) on |threads| threads: the
 *                same bytes as threads == 1 (tests/test_oracle_golden.py pins that), minutes instead of tens of
 *                minutes for the 100k-point exact-build tests. */
ido_index* ido_build(const float* points, uint32_t n, uint32_t dim,
                     const ido_config* cfg, int threads, ido_counters* counters);

/* Import an existing graph.  layers[l-1] has layer_len[l-1] rows of 32. */
ido_index* ido_import(const float* points, uint32_t n, uint32_t dim,
                      const ido_config* cfg, const uint32_t* zero,
                      const uint32_t* const* layers, const uint32_t* layer_len,
                      uint32_t n_upper);
/* the same over the caller's points / zero arrays (no copy; search only; the caller keeps them alive) */
ido_index* ido_import_borrowed(const float* points, uint32_t n, uint32_t dim,
                      const ido_config* cfg, const uint32_t* zero,
                      const uint32_t* const* layers, const uint32_t* layer_len,
                      uint32_t n_upper);
void ido_free(ido_index*);

uint32_t ido_n(const ido_index*);
uint32_t ido_dim(const ido_index*);
uint32_t ido_n_upper(const ido_index*);
uint32_t ido_layer_len(const ido_index*, uint32_t l /*1-based upper layer*/);
const uint32_t* ido_zero(const ido_index*);                 /* n*64 */
const uint32_t* ido_layer(const ido_index*, uint32_t l);    /* layer_len*32 */
const float* ido_points(const ido_index*);
void ido_set_ef_search(ido_index*, uint32_t ef);

ido_search* ido_search_new(void);          /* Search::default(), core/lib.rs:767-778 */
void ido_search_free(ido_search*);

/* Hnsw::search, core/lib.rs:352-383.  Writes <= ef_search results, nearest
 * first; returns the count.  out arrays must hold ef_search entries. */
uint32_t ido_search_one(const ido_index*, ido_search*, const float* query,
                        uint32_t* out_pid, float* out_dist, ido_counters* c);

/* Batch of queries on `threads` threads, one Search per thread, queries
 * block-partitioned (the reference's concurrency model).  out_* are
 * nq*ef_search, out_count nq, counters (optional) nq*3 u32 {n_dist,n_exp0,n_expU}. */
void ido_search_batch(const ido_index*, const float* queries, uint32_t nq,
                      int threads, uint32_t* out_pid, float* out_dist,
                      uint32_t* out_count, uint32_t* out_counters);

/* Exact brute-force top-k with the canonical distance (ground truth). */
void ido_bruteforce(const float* points, uint32_t n, uint32_t dim, int metric,
                    const float* queries, uint32_t nq, uint32_t k, int threads,
                    uint32_t* out_pid, float* out_dist);

#ifdef __cplusplus
}
#endif
#endif
