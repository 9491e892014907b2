/*
 * idist_oracle.c — CPU ORACLE (test infrastructure, never shipped, never
 * called by the product path).  Plain-C restatement of djc/instant-distance.
 * Every function cites the reference file:line it follows; paths are relative
 * to /root/reference/ ("core/" = instant-distance/src/, "py/" =
 * instant-distance-py/src/).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off).
 */
#define _GNU_SOURCE
#include "idist_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif

/* ------------------------------------------------------------------ */
/* Candidate + ordering: core/types.rs:229-234 (derive(Ord) on         */
/* (OrderedFloat<f32>, PointId)); ordered-float: NaN is greatest, all  */
/* NaNs equal, -0 == +0.                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    float distance;
    uint32_t pid;
} cand_t;

static inline int of32_cmp(float a, float b) {
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn; /* NaN > everything, NaN == NaN */
    return (a > b) - (a < b);
}
static inline int cand_cmp(cand_t a, cand_t b) {
    int c = of32_cmp(a.distance, b.distance);
    if (c) return c;
    return (a.pid > b.pid) - (a.pid < b.pid);
}

/* ------------------------------------------------------------------ */
/* Canonical distance: py/lib.rs:378-421                               */
/* ------------------------------------------------------------------ */
float ido_distance_scalar(const float* a, const float* b, uint32_t dim, int metric) {
    /* dim padded with zeros to a multiple of 4 (fma(0,0,acc) == acc exactly) */
    uint32_t dp = (dim + 3u) & ~3u;
    uint32_t steps = dp / 8u;       /* chunks_exact(8), py/lib.rs:391 */
    int tail = (dp % 8u) == 4u;     /* debug_assert len%8==4, py/lib.rs:387 */
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* _mm256_setzero_ps, :390 */
    for (uint32_t s = 0; s < steps; s++) {
        for (uint32_t j = 0; j < 8; j++) {
            uint32_t i = 8u * s + j;
            float x = i < dim ? a[i] : 0.0f, y = i < dim ? b[i] : 0.0f;
            float d = x - y;                 /* _mm256_sub_ps, :394 */
            acc[j] = fmaf(d, d, acc[j]);     /* _mm256_fmadd_ps, :395 */
        }
    }
    float a4[4];
    for (uint32_t j = 0; j < 4; j++) a4[j] = acc[j + 4] + acc[j]; /* :398-400 */
    if (tail) {                              /* :402-405 */
        for (uint32_t j = 0; j < 4; j++) {
            uint32_t i = 8u * steps + j;
            float x = i < dim ? a[i] : 0.0f, y = i < dim ? b[i] : 0.0f;
            float d = x - y;
            a4[j] = fmaf(d, d, a4[j]);
        }
    }
    float s02 = a4[0] + a4[2];               /* movehl + add_ps, :407-408 */
    float s13 = a4[1] + a4[3];
    float r = s02 + s13;                     /* shuffle 0x1 + add_ss, :409-411 */
    if (metric == IDO_METRIC_L2) r = sqrtf(r); /* tests/all.rs:96 */
    return r;
}

float ido_distance(const float* a, const float* b, uint32_t dim, int metric) {
#if defined(__AVX2__) && defined(__FMA__)
    if ((dim & 3u) == 0) {
        uint32_t steps = dim / 8u;
        int tail = (dim % 8u) == 4u;
        __m256 acc8 = _mm256_setzero_ps();
        for (uint32_t s = 0; s < steps; s++) {
            __m256 l = _mm256_loadu_ps(a + 8u * s);
            __m256 r = _mm256_loadu_ps(b + 8u * s);
            __m256 d = _mm256_sub_ps(l, r);
            acc8 = _mm256_fmadd_ps(d, d, acc8);
        }
        __m128 acc4 = _mm256_extractf128_ps(acc8, 1);
        __m128 right = _mm256_castps256_ps128(acc8);
        acc4 = _mm_add_ps(acc4, right);
        if (tail) {
            __m128 l = _mm_loadu_ps(a + dim - 4);
            __m128 r = _mm_loadu_ps(b + dim - 4);
            __m128 d = _mm_sub_ps(l, r);
            acc4 = _mm_fmadd_ps(d, d, acc4);
        }
        __m128 lower = _mm_movehl_ps(acc4, acc4);
        acc4 = _mm_add_ps(acc4, lower);
        __m128 upper = _mm_shuffle_ps(acc4, acc4, 0x1);
        acc4 = _mm_add_ss(acc4, upper);
        float r = _mm_cvtss_f32(acc4);
        if (metric == IDO_METRIC_L2) r = sqrtf(r);
        return r;
    }
#endif
    return ido_distance_scalar(a, b, dim, metric);
}

/* ------------------------------------------------------------------ */
/* Visited: core/types.rs:13-59                                        */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t* store;
    size_t len;
    uint8_t generation;
} visited_t;

static void visited_init(visited_t* v, size_t cap) { /* :19-24 */
    v->store = cap ? (uint8_t*)calloc(cap, 1) : NULL;
    v->len = cap;
    v->generation = 1;
}
static void visited_reserve(visited_t* v, size_t cap) { /* :26-30 */
    if (v->len != cap) {
        uint8_t fill = (uint8_t)(v->generation - 1);
        uint8_t* ns = (uint8_t*)malloc(cap ? cap : 1);
        size_t keep = v->len < cap ? v->len : cap;
        if (keep) memcpy(ns, v->store, keep);
        if (cap > keep) memset(ns + keep, fill, cap - keep);
        free(v->store);
        v->store = ns;
        v->len = cap;
    }
}
static inline int visited_insert(visited_t* v, uint32_t pid) { /* :32-40 */
    uint8_t* slot = &v->store[pid];
    if (*slot != v->generation) {
        *slot = v->generation;
        return 1;
    }
    return 0;
}
static void visited_clear(visited_t* v) { /* :48-58 */
    if (v->generation < 249) {
        v->generation += 1;
        return;
    }
    memset(v->store, 0, v->len);
    v->generation = 1;
}

/* ------------------------------------------------------------------ */
/* BinaryHeap<Reverse<Candidate>> (core/lib.rs:564): min-heap.  All     */
/* keys are distinct (pid unique through `visited`), so pop order is   */
/* independent of the heap implementation.                             */
/* ------------------------------------------------------------------ */
typedef struct {
    cand_t* a;
    size_t len, cap;
} heap_t;
static void heap_push(heap_t* h, cand_t c) {
    if (h->len == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->a = (cand_t*)realloc(h->a, h->cap * sizeof(cand_t));
    }
    size_t i = h->len++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (cand_cmp(h->a[p], c) <= 0) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = c;
}
static int heap_pop(heap_t* h, cand_t* out) {
    if (!h->len) return 0;
    *out = h->a[0];
    cand_t last = h->a[--h->len];
    size_t i = 0, n = h->len;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m;
        if (l >= n) break;
        m = (r < n && cand_cmp(h->a[r], h->a[l]) < 0) ? r : l;
        if (cand_cmp(last, h->a[m]) <= 0) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (n) h->a[i] = last;
    return 1;
}

typedef struct {
    cand_t* a;
    size_t len, cap;
} vec_t;
static void vec_reserve(vec_t* v, size_t n) {
    if (n > v->cap) {
        v->cap = n * 2 + 16;
        v->a = (cand_t*)realloc(v->a, v->cap * sizeof(cand_t));
    }
}
static void vec_push(vec_t* v, cand_t c) {
    vec_reserve(v, v->len + 1);
    v->a[v->len++] = c;
}

/* ------------------------------------------------------------------ */
/* Search: core/lib.rs:560-574                                         */
/* ------------------------------------------------------------------ */
struct ido_search {
    visited_t visited;
    heap_t candidates;
    vec_t nearest;
    vec_t working;
    vec_t discarded;
    size_t ef;
    ido_counters ctr;
};

ido_search* ido_search_new(void) { /* Default, core/lib.rs:767-778 */
    ido_search* s = (ido_search*)calloc(1, sizeof(*s));
    visited_init(&s->visited, 0);
    s->ef = 1;
    return s;
}
static ido_search* search_with_capacity(size_t n) { /* Search::new, :577-582 */
    ido_search* s = (ido_search*)calloc(1, sizeof(*s));
    visited_init(&s->visited, n);
    s->ef = 1;
    return s;
}
void ido_search_free(ido_search* s) {
    if (!s) return;
    free(s->visited.store);
    free(s->candidates.a);
    free(s->nearest.a);
    free(s->working.a);
    free(s->discarded.a);
    free(s);
}

/* Layer abstraction: core/types.rs:134-192.  `stride` is 32 (UpperNode) or
 * 64 (ZeroNode); `locks` non-NULL == &[RwLock<ZeroNode>] (parallel build). */
typedef struct {
    const uint32_t* rows;
    uint32_t stride;
    atomic_flag* locks;
} layer_t;

static inline void node_lock(atomic_flag* l) {
    while (atomic_flag_test_and_set_explicit(l, memory_order_acquire)) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
static inline void node_unlock(atomic_flag* l) {
    atomic_flag_clear_explicit(l, memory_order_release);
}

/* NearestIter: copies the row (a read-lock guard in the reference) and the
 * caller iterates until the first INVALID, core/types.rs:172-192. */
static inline uint32_t layer_row(const layer_t* L, uint32_t pid, uint32_t* out) {
    const uint32_t* r = L->rows + (size_t)pid * L->stride;
    if (L->locks) {
        node_lock(&L->locks[pid]);
        memcpy(out, r, L->stride * sizeof(uint32_t));
        node_unlock(&L->locks[pid]);
    } else {
        memcpy(out, r, L->stride * sizeof(uint32_t));
    }
    uint32_t n = 0;
    while (n < L->stride && out[n] != IDO_INVALID) n++;
    return n;
}

typedef struct {
    const float* points;
    uint32_t dim;
    int metric;
} pts_t;
static inline float pdist(const pts_t* P, const float* q, uint32_t pid) {
    return ido_distance(q, P->points + (size_t)pid * P->dim, P->dim, P->metric);
}

static void search_reset(ido_search* s) { /* core/lib.rs:740-755 */
    visited_clear(&s->visited);
    s->candidates.len = 0;
    s->nearest.len = 0;
    s->working.len = 0;
    s->discarded.len = 0;
}

/* Search::push, core/lib.rs:704-720 */
static void search_push(ido_search* s, uint32_t pid, const float* point, const pts_t* P) {
    if (!visited_insert(&s->visited, pid)) return;       /* :705-707 */
    cand_t nw = {pdist(P, point, pid), pid};              /* :709-711 */
    s->ctr.n_dist++;
    /* nearest.binary_search(&new): keys are distinct => lower bound, :712 */
    size_t lo = 0, hi = s->nearest.len;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (cand_cmp(s->nearest.a[mid], nw) < 0) lo = mid + 1; else hi = mid;
    }
    size_t idx = lo;
    if (!(idx < s->ef)) return;                           /* :713-714 */
    vec_reserve(&s->nearest, s->nearest.len + 1);         /* :718 */
    memmove(s->nearest.a + idx + 1, s->nearest.a + idx,
            (s->nearest.len - idx) * sizeof(cand_t));
    s->nearest.a[idx] = nw;
    s->nearest.len++;
    heap_push(&s->candidates, nw);                        /* :719 */
}

/* Search::search, core/lib.rs:598-614 */
static void search_layer(ido_search* s, const float* point, const layer_t* L,
                         const pts_t* P, uint32_t links, int is_zero) {
    cand_t c;
    uint32_t row[IDO_M2];
    while (heap_pop(&s->candidates, &c)) {                /* :599 */
        if (s->nearest.len) {                             /* :600-604 */
            cand_t furthest = s->nearest.a[s->nearest.len - 1];
            if (of32_cmp(c.distance, furthest.distance) > 0) break;
        }
        if (is_zero) s->ctr.n_exp0++; else s->ctr.n_expU++;
        uint32_t nn = layer_row(L, c.pid, row);           /* :606 */
        if (nn > links) nn = links;                       /* .take(links) */
        for (uint32_t i = 0; i < nn; i++) search_push(s, row[i], point, P); /* :607 */
        if (s->nearest.len > s->ef) s->nearest.len = s->ef; /* truncate, :612 */
    }
}

/* Search::cull, core/lib.rs:729-737 */
static void search_cull(ido_search* s) {
    s->candidates.len = 0;
    for (size_t i = 0; i < s->nearest.len; i++) heap_push(&s->candidates, s->nearest.a[i]);
    visited_clear(&s->visited);
    for (size_t i = 0; i < s->nearest.len; i++) visited_insert(&s->visited, s->nearest.a[i].pid);
}

static int cand_qsort_cmp(const void* x, const void* y) {
    return cand_cmp(*(const cand_t*)x, *(const cand_t*)y);
}

/* Search::select_heuristic, core/lib.rs:636-698.  Result is left in
 * s->nearest (selected-then-backfilled order, NOT re-sorted). */
static void select_heuristic(ido_search* s, const float* point, const layer_t* L,
                             const pts_t* P, int extend_candidates, int keep_pruned) {
    s->working.len = 0;                                   /* :643 */
    uint32_t row[IDO_M2];
    for (size_t i = 0; i < s->nearest.len; i++) {         /* :646 */
        cand_t c = s->nearest.a[i];
        vec_push(&s->working, c);                         /* :647 */
        if (extend_candidates) {                          /* :648-659 */
            /* NOTE: in the reference this path takes a read lock on a node
             * whose write lock the inserting thread already holds
             * (add_neighbor_heuristic's W contains `new`), i.e. it deadlocks;
             * restated lock-free here. */
            uint32_t nn = layer_row(L, c.pid, row);
            for (uint32_t h = 0; h < nn; h++) {
                if (!visited_insert(&s->visited, row[h])) continue;
                cand_t nw = {pdist(P, point, row[h]), row[h]};
                s->ctr.n_heur++;
                vec_push(&s->working, nw);
            }
        }
    }
    if (extend_candidates)                                /* :662-664 */
        qsort(s->working.a, s->working.len, sizeof(cand_t), cand_qsort_cmp);

    s->nearest.len = 0;                                   /* :666-667 */
    s->discarded.len = 0;
    for (size_t w = 0; w < s->working.len; w++) {         /* :668 */
        cand_t c = s->working.a[w];
        if (s->nearest.len >= IDO_M2) break;              /* :669-671 */
        const float* cp = P->points + (size_t)c.pid * P->dim; /* :675 */
        int pruned = 0;
        for (size_t r = 0; r < s->nearest.len; r++) {     /* :676-679 `any` */
            float d = pdist(P, cp, s->nearest.a[r].pid);
            s->ctr.n_heur++;
            if (of32_cmp(d, c.distance) < 0) { pruned = 1; break; }
        }
        if (!pruned) vec_push(&s->nearest, c); else vec_push(&s->discarded, c); /* :681-684 */
    }
    s->working.len = 0; /* drain(..) */
    if (keep_pruned) {                                    /* :687-695 */
        for (size_t i = 0; i < s->discarded.len; i++) {
            if (s->nearest.len >= IDO_M2) break;
            vec_push(&s->nearest, s->discarded.a[i]);
        }
    }
    s->discarded.len = 0;
}

/* Search::add_neighbor_heuristic, core/lib.rs:616-631 */
static void add_neighbor_heuristic(ido_search* s, uint32_t nw, const uint32_t* current,
                                   uint32_t ncur, const layer_t* L, const float* point,
                                   const pts_t* P, int extend, int keep) {
    search_reset(s);                                      /* :625 */
    search_push(s, nw, point, P);                         /* :626 */
    for (uint32_t i = 0; i < ncur; i++) search_push(s, current[i], point, P); /* :627-629 */
    select_heuristic(s, point, L, P, extend, keep);       /* :630 */
}

/* ------------------------------------------------------------------ */
/* Index                                                               */
/* ------------------------------------------------------------------ */
struct ido_index {
    ido_config cfg;
    uint32_t n, dim;
    float* points;        /* n*dim, PointId order */
    uint32_t* zero;       /* n*64 */
    uint32_t n_upper;     /* layers.len() */
    uint32_t** layers;    /* layers[l-1] : layer_len[l-1]*32 */
    uint32_t* layer_len;
    int borrowed;         /* points / zero belong to the caller (ido_import_borrowed) */
};

void ido_default_config(ido_config* c) { /* core/lib.rs:101-128 */
    c->ef_search = 100;
    c->ef_construction = 100;
    c->ml = 1.0f / logf((float)IDO_M);
    c->has_heuristic = 1;
    c->extend_candidates = 0;
    c->keep_pruned = 1;
    c->metric = IDO_METRIC_L2SQ;
}

/* core/lib.rs:238-250.  cum[l] = cumulative number of points on layers >= l
 * i.e. the number of nodes present in layer l; cum[0] = n. */
uint32_t ido_layer_sizes(uint32_t n, float ml, uint32_t* cum, uint32_t cap) {
    /* sizes pushed bottom-up: (num-next, num) ... then (num,num); reversed. */
    uint32_t tmp[64];
    uint32_t cnt = 0;
    size_t num = n;
    for (;;) {
        volatile float prod = (float)num * ml;            /* f32 multiply, :241 */
        size_t next;
        if (!(prod > 0.0f)) next = 0;                     /* `as usize` saturates */
        else if (prod >= 18446744073709551616.0f) next = (size_t)-1;
        else next = (size_t)prod;
        if (next < IDO_M) break;                          /* :242 */
        tmp[cnt++] = (uint32_t)num;
        num = next;
        if (cnt >= 63) break;
    }
    tmp[cnt++] = (uint32_t)num;                           /* :248 */
    /* tmp[0] = n (layer 0), tmp[cnt-1] = top layer size */
    for (uint32_t l = 0; l < cnt && l < cap; l++) cum[l] = tmp[l];
    return cnt;
}

/* ---- rand restatement (PARITY UNPINNED) ---- */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
typedef struct { uint64_t s[4]; } xoshiro_t;
static uint64_t splitmix64(uint64_t* st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t xoshiro_next(xoshiro_t* x) { /* xoshiro256++ */
    uint64_t* s = x->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}
typedef struct { uint32_t key, idx; } shuf_t;
static int shuf_cmp(const void* a, const void* b) {
    const shuf_t* x = (const shuf_t*)a; const shuf_t* y = (const shuf_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}
void ido_permutation(uint64_t seed, uint32_t n, uint32_t* out_pid, uint32_t* order) {
    xoshiro_t rng;
    uint64_t st = seed;
    for (int i = 0; i < 4; i++) rng.s[i] = splitmix64(&st);
    shuf_t* sh = (shuf_t*)malloc((size_t)(n ? n : 1) * sizeof(shuf_t));
    for (uint32_t i = 0; i < n; i++) {                    /* core/lib.rs:257-259 */
        /* random_range(0..n): widening multiply of a u32 draw, one extra draw
         * to resolve the bias when the low word is large (Canon's method as in
         * rand 0.9's UniformInt::sample_single) — unpinned restatement. */
        uint32_t v = (uint32_t)(xoshiro_next(&rng) >> 32);
        uint64_t m = (uint64_t)v * (uint64_t)n;
        uint32_t key = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo > (uint32_t)(0u - n)) {
            uint32_t v2 = (uint32_t)(xoshiro_next(&rng) >> 32);
            uint32_t hi2 = (uint32_t)(((uint64_t)v2 * (uint64_t)n) >> 32);
            if ((uint64_t)lo + (uint64_t)hi2 > 0xFFFFFFFFull) key += 1;
        }
        sh[i].key = key;
        sh[i].idx = i;
    }
    qsort(sh, n, sizeof(shuf_t), shuf_cmp);               /* sort_unstable, :260 */
    for (uint32_t i = 0; i < n; i++) {                    /* :262-270 */
        out_pid[sh[i].idx] = i;
        if (order) order[i] = sh[i].idx;
    }
    free(sh);
}

/* ---- construction ---- */
typedef struct {
    ido_index* ix;
    pts_t P;
    uint32_t top;               /* LayerId of the top layer */
    atomic_flag* locks;         /* per-node, parallel mode only */
    /* SearchPool, core/lib.rs:531-554 */
    pthread_mutex_t pool_mu;
    ido_search** pool;          /* pairs */
    size_t pool_len, pool_cap;
    ido_counters total;
    struct helpers_s* helpers;  /* threads < 0 only: sequential insertions, one insertion's neighbour updates on helper threads */
} construction_t;

/* threads = -k (test infrastructure for LARGE exact builds): the insertions stay strictly sequential in PointId order —
 * the exact contract — but the <= 64 neighbour updates of ONE insertion (core/lib.rs:481-516) run on k threads.  With
 * extend_candidates = false they are independent: iteration i reads points, the row of found[i].pid and `new`, and
 * rewrites only that row, so any interleaving gives the bytes of the serial loop (tests pin threads=-k == threads=1).
 * Each thread owns its `insertion` Search; their work counters are summed. */
typedef struct helpers_s {
    int k;                        /* threads, the caller included */
    pthread_t* th;
    pthread_barrier_t bar;
    ido_search** ins;             /* [k] */
    construction_t* C;
    uint32_t nw;
    const cand_t* found;
    size_t nfound;
    atomic_uint next;
    int stop;
} helpers_t;

static void pool_pop(construction_t* C, ido_search** a, ido_search** b) {
    pthread_mutex_lock(&C->pool_mu);
    if (C->pool_len >= 2) {
        *b = C->pool[--C->pool_len];
        *a = C->pool[--C->pool_len];
        pthread_mutex_unlock(&C->pool_mu);
        return;
    }
    pthread_mutex_unlock(&C->pool_mu);
    *a = search_with_capacity(C->ix->n);
    *b = search_with_capacity(C->ix->n);
}
static void pool_push(construction_t* C, ido_search* a, ido_search* b) {
    pthread_mutex_lock(&C->pool_mu);
    if (C->pool_len + 2 > C->pool_cap) {
        C->pool_cap = C->pool_cap ? C->pool_cap * 2 : 32;
        C->pool = (ido_search**)realloc(C->pool, C->pool_cap * sizeof(*C->pool));
    }
    C->pool[C->pool_len++] = a;
    C->pool[C->pool_len++] = b;
    C->total.n_dist += a->ctr.n_dist;                  /* descent only */
    C->total.n_exp0 += a->ctr.n_exp0;
    C->total.n_expU += a->ctr.n_expU;
    C->total.n_heur += a->ctr.n_heur + b->ctr.n_heur + b->ctr.n_dist; /* selection */
    memset(&a->ctr, 0, sizeof(a->ctr));
    memset(&b->ctr, 0, sizeof(b->ctr));
    pthread_mutex_unlock(&C->pool_mu);
}

/* Rust std slice::binary_search_by (1.82+ branchless form), used ONLY by the
 * heuristic=None splice, core/lib.rs:500-512, whose comparator is reversed:
 * the result depends on this exact probe sequence (best-effort parity). */
static size_t rust_binary_search_by(const int* cmp_of_elem /*-1,0,1 per elem*/, size_t len) {
    size_t size = len;
    if (size == 0) return 0;
    size_t base = 0;
    while (size > 1) {
        size_t half = size / 2, mid = base + half;
        base = (cmp_of_elem[mid] > 0) ? base : mid;
        size -= half;
    }
    int c = cmp_of_elem[base];
    if (c == 0) return base;
    return base + (c < 0 ? 1 : 0);
}

/* ZeroNode::rewrite, core/types.rs:88-98 */
static void zeronode_rewrite(uint32_t* row, const cand_t* found, size_t nf) {
    size_t k = 0;
    for (uint32_t i = 0; i < IDO_M2; i++) {
        if (k < nf) row[i] = found[k++].pid;
        else if (row[i] != IDO_INVALID) row[i] = IDO_INVALID;
        else break;
    }
}
/* ZeroNode::insert, core/types.rs:100-113 */
static void zeronode_insert(uint32_t* row, size_t idx, uint32_t pid) {
    if (idx >= IDO_M2) return;
    if (row[idx] != IDO_INVALID)
        memmove(row + idx + 1, row + idx, (IDO_M2 - 1 - idx) * sizeof(uint32_t));
    row[idx] = pid;
}

static void add_neighbor_heuristic(ido_search* s, uint32_t nw, const uint32_t* current, uint32_t ncur, const layer_t* L,
                                   const float* point, const pts_t* P, int extend, int keep);

/* core/lib.rs:484-496 for found[i] (heuristic, extend_candidates = false), run by thread t of the helper pool */
static void helpers_job(helpers_t* H, int t) {
    construction_t* C = H->C;
    ido_index* ix = C->ix;
    layer_t Lz = {ix->zero, IDO_M2, NULL};
    uint32_t cur_row[IDO_M2];
    for (;;) {
        uint32_t i = atomic_fetch_add(&H->next, 1);
        if (i >= H->nfound) break;
        uint32_t pid = H->found[i].pid;
        uint32_t* prow = ix->zero + (size_t)pid * IDO_M2;
        const float* old = ix->points + (size_t)pid * ix->dim;
        uint32_t ncur = layer_row(&Lz, pid, cur_row);
        add_neighbor_heuristic(H->ins[t], H->nw, cur_row, ncur, &Lz, old, &C->P, 0, ix->cfg.keep_pruned);
        zeronode_rewrite(prow, H->ins[t]->nearest.a, H->ins[t]->nearest.len);
    }
}
typedef struct { helpers_t* H; int t; } helper_arg_t;
static void* helpers_main(void* arg) {
    helper_arg_t* a = (helper_arg_t*)arg;
    for (;;) {
        pthread_barrier_wait(&a->H->bar);                 /* a job is posted (or stop) */
        if (a->H->stop) break;
        helpers_job(a->H, a->t);
        pthread_barrier_wait(&a->H->bar);                 /* every row of this insertion is rewritten */
    }
    return NULL;
}

/* Construction::insert, core/lib.rs:437-528 */
static void construction_insert(construction_t* C, uint32_t nw, uint32_t layer) {
    ido_index* ix = C->ix;
    const ido_config* cfg = &ix->cfg;
    uint32_t* node = ix->zero + (size_t)nw * IDO_M2;      /* :438 (own row) */
    uint32_t own[IDO_M2];
    for (uint32_t i = 0; i < IDO_M2; i++) own[i] = IDO_INVALID;
    ido_search *search, *insertion;
    pool_pop(C, &search, &insertion);                     /* :439 */
    insertion->ef = cfg->ef_construction;                 /* :440 */

    const float* point = ix->points + (size_t)nw * ix->dim; /* :442 */
    search_reset(search);                                 /* :443 */
    search_push(search, 0, point, &C->P);                 /* :444 */
    uint32_t num = layer == 0 ? IDO_M2 : IDO_M;           /* :445 */

    layer_t Lz = {ix->zero, IDO_M2, C->locks};
    for (uint32_t cur = C->top;; cur--) {                 /* :447 descend */
        search->ef = cur <= layer ? cfg->ef_construction : 1; /* :448-452 */
        if (cur > layer) {                                /* :453-457 */
            layer_t Lu = {ix->layers[cur - 1], IDO_M, NULL};
            search_layer(search, point, &Lu, &C->P, num, 0);
            search_cull(search);
        } else {                                          /* :458-461 */
            search_layer(search, point, &Lz, &C->P, num, layer == 0);
            break;
        }
        if (cur == 0) break;
    }

    const cand_t* found;
    size_t nfound;
    if (!cfg->has_heuristic) {                            /* :466-469 */
        found = search->nearest.a;
        nfound = search->nearest.len < IDO_M2 ? search->nearest.len : IDO_M2;
    } else {                                              /* :470-472 */
        select_heuristic(search, point, &Lz, &C->P, cfg->extend_candidates, cfg->keep_pruned);
        found = search->nearest.a;
        nfound = search->nearest.len;
    }

    uint32_t cur_row[IDO_M2];
    if (C->helpers && cfg->has_heuristic && !cfg->extend_candidates) {
        helpers_t* H = C->helpers;                        /* :481-496 on k threads, see helpers_t */
        H->nw = nw; H->found = found; H->nfound = nfound;
        atomic_store(&H->next, 0);
        pthread_barrier_wait(&H->bar);
        helpers_job(H, 0);
        pthread_barrier_wait(&H->bar);
        for (int t = 0; t < H->k; t++) {                  /* their distance calls are this insertion's */
            insertion->ctr.n_heur += H->ins[t]->ctr.n_heur + H->ins[t]->ctr.n_dist;
            memset(&H->ins[t]->ctr, 0, sizeof(H->ins[t]->ctr));
        }
        for (size_t i = 0; i < nfound; i++) node[i] = own[i] = found[i].pid;   /* node.set(i,pid), :516 */
        nfound = 0;                                       /* nothing left for the serial loop */
    }
    for (size_t i = 0; i < nfound; i++) {                 /* :481 */
        float distance = found[i].distance;
        uint32_t pid = found[i].pid;
        uint32_t* prow = ix->zero + (size_t)pid * IDO_M2;
        const float* old = ix->points + (size_t)pid * ix->dim;
        if (cfg->has_heuristic) {                         /* :484-496 */
            uint32_t ncur = layer_row(&Lz, pid, cur_row); /* zero.nearest_iter(pid) */
            add_neighbor_heuristic(insertion, nw, cur_row, ncur, &Lz, old, &C->P,
                                   cfg->extend_candidates, cfg->keep_pruned);
            if (C->locks) node_lock(&C->locks[pid]);
            zeronode_rewrite(prow, insertion->nearest.a, insertion->nearest.len);
            if (C->locks) node_unlock(&C->locks[pid]);
        } else {                                          /* :497-515 */
            int cmps[IDO_M2];
            if (C->locks) node_lock(&C->locks[pid]);
            memcpy(cur_row, prow, sizeof(cur_row));
            if (C->locks) node_unlock(&C->locks[pid]);
            for (uint32_t t = 0; t < IDO_M2; t++) {
                if (cur_row[t] == IDO_INVALID) { cmps[t] = 1; continue; } /* :505-508 Greater */
                float dt = pdist(&C->P, old, cur_row[t]);
                cmps[t] = of32_cmp(distance, dt);         /* :510 distance.cmp(&third) */
            }
            size_t idx = rust_binary_search_by(cmps, IDO_M2);
            if (C->locks) node_lock(&C->locks[pid]);
            zeronode_insert(prow, idx, nw);               /* :514 */
            if (C->locks) node_unlock(&C->locks[pid]);
        }
        own[i] = pid;                                     /* node.set(i,pid), :516 */
        if (!C->locks) node[i] = pid;
    }
    /* the reference writes `node` under a write lock held since :438; other
     * threads only observe it after release => publish the row at the end. */
    if (C->locks) {
        node_lock(&C->locks[nw]);
        for (size_t i = 0; i < nfound; i++) node[i] = own[i];
        node_unlock(&C->locks[nw]);
    }

    pool_push(C, search, insertion);                      /* :527 */
}

typedef struct {
    construction_t* C;
    uint32_t layer, end;
    atomic_uint* next;
} worker_t;
static void* build_worker(void* arg) {
    worker_t* w = (worker_t*)arg;
    for (;;) {
        uint32_t i = atomic_fetch_add(w->next, 1);
        if (i >= w->end) break;
        construction_insert(w->C, i, w->layer);
    }
    return NULL;
}

static ido_index* index_alloc(uint32_t n, uint32_t dim, const ido_config* cfg) {
    ido_index* ix = (ido_index*)calloc(1, sizeof(*ix));
    ix->cfg = *cfg;
    ix->n = n;
    ix->dim = dim;
    ix->points = (float*)malloc(((size_t)n * dim + 1) * sizeof(float));
    ix->zero = (uint32_t*)malloc(((size_t)n * IDO_M2 + 1) * sizeof(uint32_t));
    memset(ix->zero, 0xFF, (size_t)n * IDO_M2 * sizeof(uint32_t)); /* ZeroNode::default */
    return ix;
}

/* Hnsw::new, core/lib.rs:209-345 (permutation done by the caller). */
ido_index* ido_build(const float* points, uint32_t n, uint32_t dim, const ido_config* cfg,
                     int threads, ido_counters* counters) {
    ido_index* ix = index_alloc(n, dim, cfg);
    if (counters) memset(counters, 0, sizeof(*counters));
    if (n == 0) return ix;                                /* :224-234 */
    memcpy(ix->points, points, (size_t)n * dim * sizeof(float));

    uint32_t cum[64];
    uint32_t nl = ido_layer_sizes(n, cfg->ml, cum, 64);   /* :238-250 */
    uint32_t top = nl - 1;
    ix->n_upper = top;                                    /* vec![vec![]; top.0], :285 */
    ix->layers = (uint32_t**)calloc(top ? top : 1, sizeof(uint32_t*));
    ix->layer_len = (uint32_t*)calloc(top ? top : 1, sizeof(uint32_t));

    construction_t C;
    memset(&C, 0, sizeof(C));
    C.ix = ix;
    C.P.points = ix->points; C.P.dim = dim; C.P.metric = cfg->metric;
    C.top = top;
    pthread_mutex_init(&C.pool_mu, NULL);
    if (threads > 1) {
        C.locks = (atomic_flag*)malloc((size_t)n * sizeof(atomic_flag));
        for (uint32_t i = 0; i < n; i++) atomic_flag_clear(&C.locks[i]);
    }
    helpers_t H;
    helper_arg_t* hargs = NULL;
    if (threads < -1) {                                   /* sequential insertions, helper threads inside one insertion */
        memset(&H, 0, sizeof(H));
        H.k = -threads;
        H.C = &C;
        H.ins = (ido_search**)malloc(sizeof(ido_search*) * (size_t)H.k);
        for (int t = 0; t < H.k; t++) { H.ins[t] = search_with_capacity(n); H.ins[t]->ef = cfg->ef_construction; }
        pthread_barrier_init(&H.bar, NULL, (unsigned)H.k);
        H.th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)H.k);
        hargs = (helper_arg_t*)malloc(sizeof(helper_arg_t) * (size_t)H.k);
        for (int t = 1; t < H.k; t++) { hargs[t].H = &H; hargs[t].t = t; pthread_create(&H.th[t], NULL, helpers_main, &hargs[t]); }
        C.helpers = &H;
    }

    /* ranges, :275-281: layer L gets pids [max(start,1), cum[L]) */
    for (int32_t layer = (int32_t)top; layer >= 0; layer--) { /* :304 */
        uint32_t end = cum[layer];
        uint32_t start = (uint32_t)layer == top ? 0 : cum[layer + 1];
        if (start < 1) start = 1;
        if ((uint32_t)layer == top || threads <= 1) {     /* :313-314 */
            for (uint32_t i = start; i < end; i++) construction_insert(&C, i, (uint32_t)layer);
        } else {                                          /* :316-318 */
            atomic_uint next;
            atomic_init(&next, start);
            pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
            worker_t w = {&C, (uint32_t)layer, end, &next};
            for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, build_worker, &w);
            for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
            free(th);
        }
        if (layer > 0) {                                  /* :323-328 UpperNode::from_zero */
            uint32_t* up = (uint32_t*)malloc((size_t)end * IDO_M * sizeof(uint32_t));
            for (uint32_t i = 0; i < end; i++)
                memcpy(up + (size_t)i * IDO_M, ix->zero + (size_t)i * IDO_M2, IDO_M * sizeof(uint32_t));
            ix->layers[layer - 1] = up;
            ix->layer_len[layer - 1] = end;
        }
    }
    if (C.helpers) {
        H.stop = 1;
        pthread_barrier_wait(&H.bar);
        for (int t = 1; t < H.k; t++) pthread_join(H.th[t], NULL);
        for (int t = 0; t < H.k; t++) ido_search_free(H.ins[t]);
        pthread_barrier_destroy(&H.bar);
        free(H.ins); free(H.th); free(hargs);
    }
    for (size_t i = 0; i < C.pool_len; i++) ido_search_free(C.pool[i]);
    free(C.pool);
    free(C.locks);
    pthread_mutex_destroy(&C.pool_mu);
    if (counters) *counters = C.total;
    return ix;
}

ido_index* ido_import(const float* points, uint32_t n, uint32_t dim, const ido_config* cfg,
                      const uint32_t* zero, const uint32_t* const* layers,
                      const uint32_t* layer_len, uint32_t n_upper) {
    ido_index* ix = index_alloc(n, dim, cfg);
    if (n) {
        memcpy(ix->points, points, (size_t)n * dim * sizeof(float));
        memcpy(ix->zero, zero, (size_t)n * IDO_M2 * sizeof(uint32_t));
    }
    ix->n_upper = n_upper;
    ix->layers = (uint32_t**)calloc(n_upper ? n_upper : 1, sizeof(uint32_t*));
    ix->layer_len = (uint32_t*)calloc(n_upper ? n_upper : 1, sizeof(uint32_t));
    for (uint32_t l = 0; l < n_upper; l++) {
        size_t bytes = (size_t)layer_len[l] * IDO_M * sizeof(uint32_t);
        ix->layers[l] = (uint32_t*)malloc(bytes ? bytes : 1);
        memcpy(ix->layers[l], layers[l], bytes);
        ix->layer_len[l] = layer_len[l];
    }
    return ix;
}

/* The same index over the CALLER's points / zero arrays (no copy: a 10M x 768 index is 31 GB of points); the caller
 * keeps them alive and unchanged for the life of the handle.  Search only. */
ido_index* ido_import_borrowed(const float* points, uint32_t n, uint32_t dim, const ido_config* cfg,
                               const uint32_t* zero, const uint32_t* const* layers,
                               const uint32_t* layer_len, uint32_t n_upper) {
    ido_index* ix = (ido_index*)calloc(1, sizeof(*ix));
    ix->cfg = *cfg;
    ix->n = n;
    ix->dim = dim;
    ix->borrowed = 1;
    ix->points = (float*)points;
    ix->zero = (uint32_t*)zero;
    ix->n_upper = n_upper;
    ix->layers = (uint32_t**)calloc(n_upper ? n_upper : 1, sizeof(uint32_t*));
    ix->layer_len = (uint32_t*)calloc(n_upper ? n_upper : 1, sizeof(uint32_t));
    for (uint32_t l = 0; l < n_upper; l++) {
        size_t bytes = (size_t)layer_len[l] * IDO_M * sizeof(uint32_t);
        ix->layers[l] = (uint32_t*)malloc(bytes ? bytes : 1);
        memcpy(ix->layers[l], layers[l], bytes);
        ix->layer_len[l] = layer_len[l];
    }
    return ix;
}

void ido_free(ido_index* ix) {
    if (!ix) return;
    for (uint32_t l = 0; l < ix->n_upper; l++) free(ix->layers[l]);
    free(ix->layers);
    free(ix->layer_len);
    if (!ix->borrowed) {
        free(ix->zero);
        free(ix->points);
    }
    free(ix);
}

uint32_t ido_n(const ido_index* ix) { return ix->n; }
uint32_t ido_dim(const ido_index* ix) { return ix->dim; }
uint32_t ido_n_upper(const ido_index* ix) { return ix->n_upper; }
uint32_t ido_layer_len(const ido_index* ix, uint32_t l) { return ix->layer_len[l - 1]; }
const uint32_t* ido_zero(const ido_index* ix) { return ix->zero; }
const uint32_t* ido_layer(const ido_index* ix, uint32_t l) { return ix->layers[l - 1]; }
const float* ido_points(const ido_index* ix) { return ix->points; }
void ido_set_ef_search(ido_index* ix, uint32_t ef) { ix->cfg.ef_search = ef; }

/* Hnsw::search, core/lib.rs:352-383 */
uint32_t ido_search_one(const ido_index* ix, ido_search* s, const float* query,
                        uint32_t* out_pid, float* out_dist, ido_counters* c) {
    memset(&s->ctr, 0, sizeof(s->ctr));
    search_reset(s);                                      /* :357 */
    if (ix->n == 0) return 0;                             /* :359-361 */
    pts_t P = {ix->points, ix->dim, ix->cfg.metric};
    visited_reserve(&s->visited, ix->n);                  /* :363 */
    search_push(s, 0, query, &P);                         /* :364 */
    for (uint32_t cur = ix->n_upper;; cur--) {            /* :365 */
        int is_zero = cur == 0;
        s->ef = is_zero ? ix->cfg.ef_search : 1;          /* :366-371 */
        uint32_t num = is_zero ? IDO_M2 : IDO_M;
        if (is_zero) {                                    /* :372-375 */
            layer_t L = {ix->zero, IDO_M2, NULL};
            search_layer(s, query, &L, &P, num, 1);
        } else {
            layer_t L = {ix->layers[cur - 1], IDO_M, NULL};
            search_layer(s, query, &L, &P, num, 0);
            search_cull(s);                               /* :377-379 */
        }
        if (is_zero) break;
    }
    uint32_t cnt = (uint32_t)s->nearest.len;              /* :382 */
    for (uint32_t i = 0; i < cnt; i++) {
        out_pid[i] = s->nearest.a[i].pid;
        out_dist[i] = s->nearest.a[i].distance;
    }
    if (c) *c = s->ctr;
    return cnt;
}

typedef struct {
    const ido_index* ix;
    const float* queries;
    uint32_t q0, q1;
    uint32_t* out_pid; float* out_dist; uint32_t* out_count; uint32_t* out_counters;
} sb_t;
static void* search_batch_worker(void* arg) {
    sb_t* w = (sb_t*)arg;
    ido_search* s = ido_search_new();
    uint32_t ef = w->ix->cfg.ef_search;
    for (uint32_t q = w->q0; q < w->q1; q++) {
        ido_counters c;
        uint32_t cnt = ido_search_one(w->ix, s, w->queries + (size_t)q * w->ix->dim,
                                      w->out_pid + (size_t)q * ef, w->out_dist + (size_t)q * ef, &c);
        for (uint32_t i = cnt; i < ef; i++) {
            w->out_pid[(size_t)q * ef + i] = IDO_INVALID;
            w->out_dist[(size_t)q * ef + i] = INFINITY;
        }
        w->out_count[q] = cnt;
        if (w->out_counters) {
            w->out_counters[3 * (size_t)q + 0] = (uint32_t)c.n_dist;
            w->out_counters[3 * (size_t)q + 1] = (uint32_t)c.n_exp0;
            w->out_counters[3 * (size_t)q + 2] = (uint32_t)c.n_expU;
        }
    }
    ido_search_free(s);
    return NULL;
}
void ido_search_batch(const ido_index* ix, const float* queries, uint32_t nq, int threads,
                      uint32_t* out_pid, float* out_dist, uint32_t* out_count,
                      uint32_t* out_counters) {
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > nq) threads = nq ? (int)nq : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    sb_t* ws = (sb_t*)malloc(sizeof(sb_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        ws[t] = (sb_t){ix, queries, (uint32_t)((uint64_t)nq * t / threads),
                       (uint32_t)((uint64_t)nq * (t + 1) / threads),
                       out_pid, out_dist, out_count, out_counters};
        if (threads == 1) search_batch_worker(&ws[t]);
        else pthread_create(&th[t], NULL, search_batch_worker, &ws[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(ws);
}

/* ---- brute force ground truth ---- */
typedef struct {
    const float* points; uint32_t n, dim; int metric;
    const float* queries; uint32_t q0, q1, k;
    uint32_t* out_pid; float* out_dist;
} bf_t;
static void* bf_worker(void* arg) {
    bf_t* w = (bf_t*)arg;
    cand_t* top = (cand_t*)malloc(sizeof(cand_t) * (w->k + 1));
    for (uint32_t q = w->q0; q < w->q1; q++) {
        uint32_t len = 0;
        const float* qp = w->queries + (size_t)q * w->dim;
        for (uint32_t i = 0; i < w->n; i++) {
            cand_t c = {ido_distance(qp, w->points + (size_t)i * w->dim, w->dim, w->metric), i};
            if (len == w->k && cand_cmp(c, top[len - 1]) >= 0) continue;
            uint32_t p = len < w->k ? len++ : len - 1;
            while (p > 0 && cand_cmp(top[p - 1], c) > 0) { top[p] = top[p - 1]; p--; }
            top[p] = c;
        }
        for (uint32_t i = 0; i < w->k; i++) {
            w->out_pid[(size_t)q * w->k + i] = i < len ? top[i].pid : IDO_INVALID;
            w->out_dist[(size_t)q * w->k + i] = i < len ? top[i].distance : INFINITY;
        }
    }
    free(top);
    return NULL;
}
void ido_bruteforce(const float* points, uint32_t n, uint32_t dim, int metric,
                    const float* queries, uint32_t nq, uint32_t k, int threads,
                    uint32_t* out_pid, float* out_dist) {
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > nq) threads = nq ? (int)nq : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    bf_t* ws = (bf_t*)malloc(sizeof(bf_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        ws[t] = (bf_t){points, n, dim, metric, queries, (uint32_t)((uint64_t)nq * t / threads),
                       (uint32_t)((uint64_t)nq * (t + 1) / threads), k, out_pid, out_dist};
        if (threads == 1) bf_worker(&ws[t]);
        else pthread_create(&th[t], NULL, bf_worker, &ws[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(ws);
}
