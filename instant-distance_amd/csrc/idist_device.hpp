// idist_device.hpp — wave64 device routines of the HNSW hot path for gfx950.
//
// One wavefront owns one query / one insertion (narrow batches: a four-wave workgroup, its wave 0 is that wavefront).
// Nothing here is a translation of the reference's data structures; the
// reference semantics that must be reproduced bit-for-bit are cited inline
// (paths relative to /root/reference/, core/ = instant-distance/src/).
//
//   * canonical distance  : 8 lanes per point row, lane j runs FMA chain j of
//                           instant-distance-py/src/lib.rs:390-411; 8 rows per
//                           wave instruction, each row read as one 128-B line.
//   * W (Search.nearest)  : ONE sorted u64 array in LDS, key = dist_bits<<32 |
//                           pid, bit 63 = "already expanded".  The candidate
//                           heap of core/lib.rs:564 is not materialised: the
//                           live candidates are exactly the un-expanded entries
//                           of that array (DESIGN.md §3 proves the equivalence;
//                           `push` works on it in registers, w_push_merge).
//                           Live distance ties that do not fit behind it go to a
//                           bag in HBM (WState::spill): unbounded like that heap.
//   * visited             : exact set membership (core/types.rs:13-59): a
//                           two-bucket hash set in LDS per walk — full ids, or
//                           16-bit quotients of two bijective hashes (q16_*: twice
//                           the ids in the same LDS); what it cannot hold goes to
//                           one bit per point and slot in HBM.
#pragma once
#ifdef IDIST_EMU
#include "hip_emu.hpp"   // tests/simt: CPU lockstep emulation of one wave (test infrastructure)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#ifdef IDIST_EMU
#define IDIST_DYN_SMEM(name) uint8_t* name = ::emu::cur_smem()
#else
#define IDIST_DYN_SMEM(name) extern __shared__ __align__(16) uint8_t name[]
#endif

namespace idist {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr int kM = 32;
constexpr int kM2 = 64;
constexpr uint64_t kFlag = 1ull << 63;   // entry already expanded
constexpr uint64_t kKeyMask = ~kFlag;
constexpr int kTieCap = 64;              // live equidistant candidates kept beyond ef (default; idist_config.tie_capacity)
constexpr uint32_t kNanBits = 0x7fc00000u;
constexpr uint64_t kMaxKey = 0x7fffffffffffffffull;

enum : uint32_t { kStTieOverflow = 1u, kStGuard = 2u, kStBadRow = 4u, kStQueue = 8u };

// The walk's reject filter (search only): a second, compact copy of the point rows — one byte per stored float on a fixed
// lattice p^_k = lo + step * u_k, u_k in [0, 255], in the f32 rows' own (blocked) element order, 8 bytes of row metadata at the
// end of the row {|p - p^|_2 rounded up (f32), sum u_k^2 (u32)}.  `Search::push` (core/lib.rs:704-720) needs the distance of a
// candidate only when it ACCEPTS it (`Err(idx) if idx < ef`); for the others a proof that the canonical distance exceeds
// nearest[ef-1]'s is enough, and  |q - p| >= |q^ - p^| - |q - q^| - |p - p^|  gives one from the compact row alone: |q^ - p^|^2
// is an exact integer (v_dot4_u32_u8 of the row's bytes with the query's 16-bit lattice coordinates).  Rows it cannot reject are
// fetched in f32 as before; accepted candidates carry the canonical distance, n_dist counts every push — same ids, same order,
// same distance bits, same counters (filter_rounds below).
struct FilterView {
    const uint8_t* rows = nullptr;   // [n][fstride], nullptr = no filter
    uint32_t fstride = 0;            // bytes per compact row: stride codes + 8 bytes of metadata, rounded up to 64
    float lo = 0.0f, step256 = 0.0f; // lattice: p^ = lo + (256 u) * step256, q^ = lo + Q * step256 (Q: 16 bits)
    float qscale = 0.0f;             // 1 / step256
    float dscale = 0.0f;             // step256^2: |q^ - p^|^2 = I * dscale
    float up = 1.0f;                 // safety factor > 1 on the threshold side (rounding of every float step, of the canonical sum)
    float slack = 0.0f;              // absolute slack of a query's own lattice error (f32 evaluation), per sqrt(dim) * magnitude
};
constexpr uint32_t filt_stride(uint32_t stride_floats) { return (stride_floats + 8u + 63u) & ~63u; }
constexpr int kFiltRtChunks = 5;     // runtime-geometry rows on THIN filtered waves: a compact row of at most five 128-B chunks (dim <= 624: 512-d)
constexpr int kFiltRtChunksFat = 13; // ... on one fat filtered wave per SIMD (longer rows, like 768-d): thirteen chunks (dim <= 1656: 512-d, 1024-d, 1536-d)

// Device view of an index (plain pointers; lives in kernel arguments).
struct IndexView {
    const float* points;      // [n][stride] blocked rows
    uint32_t* zero;           // [n][64]
    uint32_t* upper;          // [sum layer_len][32]; layer l (1-based) starts at row layer_off[l-1]
    const uint64_t* layer_off;  // [n_upper]
    uint32_t n, dim, stride;  // stride in floats, multiple of 16
    uint32_t nb;              // full 32-float blocks (4 chain steps x 8 chains, transposed)
    uint32_t rs;              // remaining chain steps (0..3), natural order
    uint32_t tail;            // 1 if a 4-wide tail follows (padded dim % 8 == 4)
    uint32_t n_upper;
    uint32_t metric;          // 0 = squared L2, 1 = L2
    FilterView f;             // compact copy of the rows for the walk's reject filter (search kernels only)
};

// natural element of stored position `pos` (inverse of blocked_pos)
__host__ __device__ inline uint32_t natural_pos(uint32_t pos, uint32_t nb) {
    if (pos < 32u * nb) {
        const uint32_t t = pos >> 5, r = pos & 31u, j = r >> 2, c = r & 3u;
        return (t << 5) + (c << 3) + j;
    }
    return pos;
}

// Position of natural element e inside a blocked row.
__host__ __device__ inline uint32_t blocked_pos(uint32_t e, uint32_t nb) {
    if (e < 32u * nb) {
        uint32_t t = e >> 5, r = e & 31u, c = r >> 3, j = r & 7u;
        return (t << 5) + (j << 2) + c;
    }
    return e;
}

// one 16-B piece of a gathered point row.  -DIDIST_ROW_NT (measurement build `make nt`): with the non-temporal hint — rows are read
// once per walk, the visited bitmaps of long walks are re-read dozens of times, and both compete for the Infinity Cache
#ifdef IDIST_ROW_NT
typedef float idist_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_row4(const float* p) {
    const idist_v4f v = __builtin_nontemporal_load(reinterpret_cast<const idist_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#else
__device__ __forceinline__ float4 ldg_row4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// Wave-level sync: the lanes of ONE wavefront hand data to each other through LDS.  A wave's LDS instructions execute
// in issue order, so all the hardware needs is that the compiler keeps that order: a wavefront-scope fence, no
// s_barrier and no forced s_waitcnt.  (Single-wave workgroups used __syncthreads() here; the four-wave kernel of the
// narrow-batch walk needs its leader wave to run the same routines alone, so this must not be a workgroup barrier.)
__device__ __forceinline__ void wave_sync() {
#ifdef IDIST_EMU
    ::emu::wave_sync();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// Workgroup barrier of the multi-wave kernels (s_barrier + LDS fence)
__device__ __forceinline__ void block_sync() { __syncthreads(); }
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
    uint32_t lo = bcast_u32((uint32_t)v, src), hi = bcast_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// value of lane `src` (src must be wave-uniform): v_readlane_b32, no LDS round trip
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int src) {
#ifdef IDIST_EMU
    return bcast_u32(v, src);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
#endif
}

__device__ __forceinline__ uint32_t canon_bits(float r, uint32_t metric) {
    if (metric) r = __builtin_sqrtf(r);      // tests/all.rs:96; correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    uint32_t b = __float_as_uint(r);
    if (r != r) b = kNanBits;                // OrderedFloat: all NaN equal & greatest
    return b;
}

// An LDS-resident blocked vector: block t (32 floats) at blk + t*bstride, the
// natural-order remainder (rem steps + tail) at rem.
struct VecView {
    const float* blk;
    int bstride;
    const float* rem;
};
__device__ __forceinline__ VecView natural_view(const float* q, int nb) { return VecView{q, 32, q + nb * 32}; }

// ---------------------------------------------------------------------------
// Canonical distances of `na` rows (pids in act_pid[0..na)) to the LDS-resident
// blocked vector q.  Result bits -> act_dist[0..na).  Group g = lane>>3 takes
// row base+g, lane j = lane&7 runs chain j.  NB/RS/TAIL < 0 => runtime values.
// ---------------------------------------------------------------------------
// Cross-lane moves of the fold as DPP modifiers (VALU, no LDS crossbar round trip like ds_bpermute):
// CTRL 0x104 = row_shl:4 (lane i reads lane i+4 of its row of 16), 0x4E = quad_perm [2,3,0,1], 0xB1 = [1,0,3,2].
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}

// Fold of one row's 8 chains, held by the 8 lanes j = 0..7 of a group; the result is valid in lane j == 0:
// acc_4x = lo128 + hi128 (lanes 0-3), the 4-wide tail, (s0+s2)+(s1+s3) — py/lib.rs:398-411
__device__ __forceinline__ float fold_chains(float acc, bool has_tail, float tq, float tp) {
    float a4 = acc + dpp_move<0x104>(acc);        // lane j<4: chain j + chain j+4
    if (has_tail) {                               // :402-405
        const float d = tq - tp;
        a4 = __builtin_fmaf(d, d, a4);
    }
    const float s = a4 + dpp_move<0x4E>(a4);      // movehl+add, :407-408: lane 0 = s0+s2, lane 1 = s1+s3
    return s + dpp_move<0xB1>(s);                 // add_ss, :409-411
}

// Latency variant: RIF rounds (8 rows each) are requested before the first one is consumed, so a pass costs
// one HBM round trip per 8*RIF rows instead of one per 8 rows, and the query fragment of the lane sits in
// registers (there is one wave per SIMD in this mode, so nothing else would hide the LDS reads).  Same
// arithmetic, same results.
// RSTEP / first: rows between the rounds a wave keeps in flight and its first row.  The default (8, 0) walks the list
// front to back; the four-wave walk gives wave w the rounds w and w + 4 of at most 8 (RSTEP = 32, first = 8 w).
// mid(): work of the caller that does not depend on the distances — it runs once, after the first batch of row loads
// has been issued and before anything waits for them (the walk inserts the new ids into its visited set there).
struct NoMid { __device__ __forceinline__ void operator()() const {} };
constexpr uint32_t kAbandoned = 0xFFFFFFFFu;   // act_dist of a row given up early: above every canonical distance pattern
// EAK > 0 (measurement builds, `make probe`): partial-distance early abandon.  Only the first EAK blocks of every row are
// requested at first; the canonical fold of the partial chains is a LOWER bound of the row's canonical distance (every
// term is >= 0 and round-to-nearest is monotone, so each chain's prefix <= the chain and each add of the fold keeps the
// order), and a row whose bound already exceeds thr_bits — the furthest distance of a full `nearest` when the expansion
// began; `nearest` only improves during it — can only be rejected by `push` (core/lib.rs:712-714): its remaining blocks are
// never fetched and it is reported as kAbandoned.  Same decisions, same results, same n_dist; fewer bytes, but a second,
// dependent load stage for the rows that stay.
template <int NB, int RS, int TAIL, int RIF, bool QREGS = true, int RSTEP = 8, class Mid = NoMid, int EAK = 0>
__device__ __forceinline__ void dist_rounds_inflight(const IndexView& ix, const VecView qv, const uint32_t* act_pid,
                                                     uint32_t* act_dist, int na, int first = 0, Mid mid = Mid(),
                                                     uint32_t thr_bits = 0xFFFFFFFFu) {
    static_assert(NB >= 0 && RS >= 0 && TAIL >= 0 && RIF >= 1, "compile-time layout only");
    if constexpr (EAK > 0 && EAK < NB) {
        const int lane = lane_id();
        const int g = lane >> 3, j = lane & 7;
        constexpr int RSA = RS > 0 ? RS : 1;
        float4 qf[NB];                                                    // the lane's query fragment (one wave per SIMD: registers to spare)
#pragma unroll
        for (int u = 0; u < NB; u++) qf[u] = *reinterpret_cast<const float4*>(qv.blk + u * qv.bstride + j * 4);
        for (int base = first; base < na; base += RSTEP * RIF) {
            float4 p[RIF][NB];
            float pr[RIF][RSA], pt[RIF];
            const float* row[RIF];
            bool on[RIF], alive[RIF];
#pragma unroll
            for (int r = 0; r < RIF; r++) {
                const int k = base + RSTEP * r + g;
                on[r] = k < na;
                row[r] = ix.points + (size_t)act_pid[on[r] ? k : first] * ix.stride;
                if (on[r]) {
#pragma unroll
                    for (int u = 0; u < EAK; u++) p[r][u] = ldg_row4(row[r] + u * 32 + j * 4);
                }
            }
            if (base == first) mid();
            float acc[RIF];
#pragma unroll
            for (int r = 0; r < RIF; r++) {
                acc[r] = 0.0f;
                pt[r] = 0.0f;
                alive[r] = false;
                if (on[r]) {
#pragma unroll
                    for (int u = 0; u < EAK; u++) {
                        const float4 w = qf[u];
                        float d;
                        d = w.x - p[r][u].x; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.y - p[r][u].y; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.z - p[r][u].z; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.w - p[r][u].w; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    }
                }
                float lb = fold_chains(acc[r], false, 0.0f, 0.0f);        // lanes j < 4 of the group hold the bound
                const float lb_hi = dpp_move<0x114>(lb);                  // row_shr:4 — lanes j >= 4 take it from lane j - 4
                lb = j < 4 ? lb : lb_hi;
                alive[r] = on[r] && !(canon_bits(lb, ix.metric) > thr_bits);
                if (alive[r]) {                                           // the rest of the row, requested as soon as its bound is known
#pragma unroll
                    for (int u = EAK; u < NB; u++) p[r][u] = ldg_row4(row[r] + u * 32 + j * 4);
#pragma unroll
                    for (int c = 0; c < RS; c++) pr[r][c] = row[r][NB * 32 + c * 8 + j];
                    if (TAIL) pt[r] = row[r][NB * 32 + RS * 8 + (j & 3)];
                }
            }
            const float qt = TAIL ? qv.rem[RS * 8 + (j & 3)] : 0.0f;
#pragma unroll
            for (int r = 0; r < RIF; r++) {
                if (alive[r]) {
#pragma unroll
                    for (int u = EAK; u < NB; u++) {
                        const float4 w = qf[u];
                        float d;
                        d = w.x - p[r][u].x; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.y - p[r][u].y; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.z - p[r][u].z; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.w - p[r][u].w; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    }
#pragma unroll
                    for (int c = 0; c < RS; c++) {
                        const float d = qv.rem[c * 8 + j] - pr[r][c];
                        acc[r] = __builtin_fmaf(d, d, acc[r]);
                    }
                }
                const float res = fold_chains(acc[r], TAIL && alive[r], qt, pt[r]);
                if (on[r] && j == 0) act_dist[base + RSTEP * r + g] = alive[r] ? canon_bits(res, ix.metric) : kAbandoned;
            }
        }
        return;
    }
    const int lane = lane_id();
    const int g = lane >> 3, j = lane & 7;
    constexpr int NBA = NB > 0 ? NB : 1, RSA = RS > 0 ? RS : 1;
    constexpr bool QREG = QREGS && NB <= 12;
    float4 qf[QREG ? NBA : 1];
    float qr[RSA];
    if constexpr (QREG) {
#pragma unroll
        for (int u = 0; u < NB; u++) qf[u] = *reinterpret_cast<const float4*>(qv.blk + u * qv.bstride + j * 4);
    }
#pragma unroll
    for (int c = 0; c < RS; c++) qr[c] = qv.rem[c * 8 + j];
    const float qt = TAIL ? qv.rem[RS * 8 + (j & 3)] : 0.0f;
    for (int base = first; base < na; base += RSTEP * RIF) {
        float4 p[RIF][NBA];
        float pr[RIF][RSA], pt[RIF];
#pragma unroll
        for (int r = 0; r < RIF; r++) {
            const int k = base + RSTEP * r + g;
            pt[r] = 0.0f;
            if (k < na) {
                const float* row = ix.points + (size_t)act_pid[k] * ix.stride;
#pragma unroll
                for (int u = 0; u < NB; u++) p[r][u] = ldg_row4(row + u * 32 + j * 4);
#pragma unroll
                for (int c = 0; c < RS; c++) pr[r][c] = row[NB * 32 + c * 8 + j];
                if (TAIL) pt[r] = row[NB * 32 + RS * 8 + (j & 3)];
            }
        }
        if (base == first) mid();
        float acc[RIF];
#pragma unroll
        for (int r = 0; r < RIF; r++) {
            acc[r] = 0.0f;
            if (base + RSTEP * r + g < na) {
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    float4 w;
                    if constexpr (QREG) w = qf[u];
                    else w = *reinterpret_cast<const float4*>(qv.blk + u * qv.bstride + j * 4);
                    float d;
                    d = w.x - p[r][u].x; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    d = w.y - p[r][u].y; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    d = w.z - p[r][u].z; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    d = w.w - p[r][u].w; acc[r] = __builtin_fmaf(d, d, acc[r]);
                }
#pragma unroll
                for (int c = 0; c < RS; c++) {
                    const float d = qr[c] - pr[r][c];
                    acc[r] = __builtin_fmaf(d, d, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RIF; r++) {
            const int k = base + RSTEP * r + g;
            const bool on = k < na;
            const float res = fold_chains(acc[r], TAIL && on, qt, pt[r]);
            if (on && j == 0) act_dist[k] = canon_bits(res, ix.metric);
        }
    }
}

// The same for ANY dimension (`trait Point` is dimension-agnostic, core/lib.rs:780-782): the row geometry (nb full blocks,
// rs remaining chain steps, tail) is read at run time, and what is fixed at compile time is the register tile — RIF rounds
// of 8 rows in flight, each row cut into groups of CH blocks (CH float4 per lane).  Two groups are in flight at any time:
// group c + 1 of every round is requested before group c is consumed, so a wave keeps 2 * RIF * 8 rows * CH * 128 B on the
// wire whatever the row length (the compile-time geometries issue a whole row up front, which only fits the register file
// up to 768-d).  A chain still sees its operands in row order — group after group, block after block — so the arithmetic
// is the compile-time variants' bit for bit.  The query fragment comes from LDS (one ds_read_b128 per block, shared by
// the RIF rows of a lane).
template <int CH, int RIF, int RSTEP = 8, class Mid = NoMid>
__device__ __forceinline__ void dist_rounds_inflight_rt(const IndexView& ix, const VecView qv, const uint32_t* act_pid,
                                                        uint32_t* act_dist, int na, int first = 0, Mid mid = Mid()) {
    const int lane = lane_id();
    const int g = lane >> 3, j = lane & 7;
    const int nb = (int)ix.nb, rs = (int)ix.rs;
    const bool tail = ix.tail != 0;
    for (int base = first; base < na; base += RSTEP * RIF) {
        const float* row[RIF];
        bool on[RIF];
#pragma unroll
        for (int r = 0; r < RIF; r++) {
            const int k = base + RSTEP * r + g;
            on[r] = k < na;
            row[r] = ix.points + (size_t)act_pid[on[r] ? k : first] * ix.stride + j * 4;   // (idle groups re-read a live row: never used)
        }
        float4 A[RIF][CH], B[RIF][CH];
        auto load = [&](float4 (&dst)[RIF][CH], int t0) {
#pragma unroll
            for (int u = 0; u < CH; u++)
                if (t0 + u < nb) {                                   // wave-uniform: a scalar branch
#pragma unroll
                    for (int r = 0; r < RIF; r++)
                        if (on[r]) dst[r][u] = ldg_row4(row[r] + (t0 + u) * 32);
                }
        };
        float acc[RIF];
#pragma unroll
        for (int r = 0; r < RIF; r++) acc[r] = 0.0f;
        auto eat = [&](const float4 (&src)[RIF][CH], int t0) {
#pragma unroll
            for (int u = 0; u < CH; u++)
                if (t0 + u < nb) {
                    const float4 w = *reinterpret_cast<const float4*>(qv.blk + (t0 + u) * qv.bstride + j * 4);
#pragma unroll
                    for (int r = 0; r < RIF; r++) {
                        if (!on[r]) continue;
                        float d;
                        d = w.x - src[r][u].x; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.y - src[r][u].y; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.z - src[r][u].z; acc[r] = __builtin_fmaf(d, d, acc[r]);
                        d = w.w - src[r][u].w; acc[r] = __builtin_fmaf(d, d, acc[r]);
                    }
                }
        };
        load(A, 0);
        // the steps that do not fill a block and the 4-wide tail: at most four dwords per row, requested with the first group
        float pr[RIF][3], pt[RIF];
#pragma unroll
        for (int r = 0; r < RIF; r++) {
            const float* rem = row[r] - j * 4 + nb * 32;
#pragma unroll
            for (int c = 0; c < 3; c++) pr[r][c] = (on[r] && c < rs) ? rem[c * 8 + j] : 0.0f;
            pt[r] = (on[r] && tail) ? rem[rs * 8 + (j & 3)] : 0.0f;
        }
        if (nb > CH) load(B, CH);
        if (base == first) mid();
        for (int t0 = 0; t0 < nb; t0 += 2 * CH) {
            eat(A, t0);
            if (t0 + 2 * CH < nb) load(A, t0 + 2 * CH);
            if (t0 + CH < nb) {
                eat(B, t0 + CH);
                if (t0 + 3 * CH < nb) load(B, t0 + 3 * CH);
            }
        }
        float qr[3];
#pragma unroll
        for (int c = 0; c < 3; c++) qr[c] = c < rs ? qv.rem[c * 8 + j] : 0.0f;
        const float qt = tail ? qv.rem[rs * 8 + (j & 3)] : 0.0f;
#pragma unroll
        for (int r = 0; r < RIF; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (c < rs) {                                        // py/lib.rs:391-396, steps not filling a block
                    const float d = qr[c] - pr[r][c];
                    acc[r] = __builtin_fmaf(d, d, acc[r]);
                }
            const float res = fold_chains(acc[r], tail && on[r], qt, pt[r]);
            if (on[r] && j == 0) act_dist[base + RSTEP * r + g] = canon_bits(res, ix.metric);
        }
    }
}

template <int NB, int RS, int TAIL>
__device__ __forceinline__ void dist_rounds(const IndexView& ix, const float* q, const uint32_t* act_pid,
                                            uint32_t* act_dist, int na);
template <int NB, int RS, int TAIL>
__device__ __forceinline__ void dist_rounds(const IndexView& ix, const VecView qv, const uint32_t* act_pid,
                                            uint32_t* act_dist, int na, int first = 0, int step = 8) {
    const int lane = lane_id();
    const int g = lane >> 3, j = lane & 7;
    const int nb = NB >= 0 ? NB : (int)ix.nb;
    const int rs = RS >= 0 ? RS : (int)ix.rs;
    const int tail = TAIL >= 0 ? TAIL : (int)ix.tail;
    for (int base = first; base < na; base += step) {
        const int k = base + g;
        const bool on = k < na;
        const uint32_t pid = on ? act_pid[k] : 0u;
        const float* row = ix.points + (size_t)pid * ix.stride;
        float acc = 0.0f;
        if (on) {
            if constexpr (NB >= 0) {
                constexpr int CH = NB > 12 ? 8 : (NB > 0 ? NB : 1);
#pragma unroll
                for (int t0 = 0; t0 < NB; t0 += CH) {
                    float4 p[CH];
#pragma unroll
                    for (int u = 0; u < CH; u++)
                        if (t0 + u < NB) p[u] = ldg_row4(row + (t0 + u) * 32 + j * 4);
#pragma unroll
                    for (int u = 0; u < CH; u++) {
                        if (t0 + u < NB) {
                            const float4 w = *reinterpret_cast<const float4*>(qv.blk + (t0 + u) * qv.bstride + j * 4);
                            float d;
                            d = w.x - p[u].x; acc = __builtin_fmaf(d, d, acc);
                            d = w.y - p[u].y; acc = __builtin_fmaf(d, d, acc);
                            d = w.z - p[u].z; acc = __builtin_fmaf(d, d, acc);
                            d = w.w - p[u].w; acc = __builtin_fmaf(d, d, acc);
                        }
                    }
                }
            } else {
                // any dimension: eight blocks (1 KB of the row per 8-lane group) requested together, then consumed in order
                for (int t0 = 0; t0 < nb; t0 += 8) {
                    float4 p[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (t0 + u < nb) p[u] = ldg_row4(row + (t0 + u) * 32 + j * 4);
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (t0 + u < nb) {
                            const float4 w = *reinterpret_cast<const float4*>(qv.blk + (t0 + u) * qv.bstride + j * 4);
                            float d;
                            d = w.x - p[u].x; acc = __builtin_fmaf(d, d, acc);
                            d = w.y - p[u].y; acc = __builtin_fmaf(d, d, acc);
                            d = w.z - p[u].z; acc = __builtin_fmaf(d, d, acc);
                            d = w.w - p[u].w; acc = __builtin_fmaf(d, d, acc);
                        }
                }
            }
            for (int c = 0; c < rs; c++) {   // py/lib.rs:391-396, steps not filling a block
                const float d = qv.rem[c * 8 + j] - row[nb * 32 + c * 8 + j];
                acc = __builtin_fmaf(d, d, acc);
            }
        }
        float tq = 0.0f, tp = 0.0f;
        if (tail && on) {                    // 4-wide tail, py/lib.rs:402-405
            tq = qv.rem[rs * 8 + (j & 3)];
            tp = row[nb * 32 + rs * 8 + (j & 3)];
        }
        const float r = fold_chains(acc, tail && on, tq, tp);
        if (on && j == 0) act_dist[k] = canon_bits(r, ix.metric);
    }
}
template <int NB, int RS, int TAIL>
__device__ __forceinline__ void dist_rounds(const IndexView& ix, const float* q, const uint32_t* act_pid,
                                            uint32_t* act_dist, int na) {
    dist_rounds<NB, RS, TAIL>(ix, natural_view(q, NB >= 0 ? NB : (int)ix.nb), act_pid, act_dist, na);
}

// Four-wave walk: the share of wave `wv` of one expansion's distance pass — rounds wv and wv + 4 of the (at most 8)
// rounds of 8 rows, both in flight at once.  Same arithmetic per row as every other variant.
template <int NB, int RS, int TAIL, class Mid = NoMid>
__device__ __forceinline__ void dist_rounds_quad(const IndexView& ix, const float* q, const uint32_t* act_pid,
                                                 uint32_t* act_dist, int na, int wv, Mid mid = Mid()) {
    if constexpr (NB >= 0) {
        if (8 * wv >= na) mid();                               // no row for this wave: the loop body never runs
        dist_rounds_inflight<NB, RS, TAIL, 2, true, 32>(ix, natural_view(q, NB), act_pid, act_dist, na, 8 * wv, mid);
    } else {
        if (8 * wv >= na) mid();
        dist_rounds_inflight_rt<8, 2, 32>(ix, natural_view(q, (int)ix.nb), act_pid, act_dist, na, 8 * wv, mid);
    }
}

// Walk modes of the graph kernels (search_layer):
//   kWalkClassic  one distance round in flight, query tile in LDS: smallest register footprint
//   kWalkLatency  narrow batches: few waves per CU, so each wave overlaps as much as its registers allow
//                 (4 rounds of 300-d rows in flight, 32-KB Bloom filter)
//   kWalkOverlap  full batches: the same overlaps with 2 rounds in flight and the 8-KB Bloom filter — 12 waves
//                 per CU stay resident and each keeps twice the bytes in flight (+11 % over classic at C3)
enum : int { kWalkClassic = 0, kWalkLatency = 1, kWalkOverlap = 2 };
// A walk code may carry overrides on top of its mode (tuning builds instantiate several):
//   bits 0-1 mode, bits 4-7 rounds in flight (0 = the mode's default), bit 8 query fragment read from LDS
//   instead of registers, bits 12-14 waves per SIMD the kernel is compiled for (0 = compiler's choice)
//   bit 9 visited set kept on chip (LDS hash set, HBM bitmap only as overflow) instead of bitmap + Bloom filter
//   bit 10 four-wave workgroup per walk: wave 0 leads (this routine), all four share each distance pass (QuadCtl)
//   bit 11 the on-chip set stores 16-bit quotients instead of ids (twice the ids in the same LDS; search walks only)
constexpr int walk_code(int mode, int rif = 0, bool q_lds = false, int waves = 0, bool vis_lds = false, bool quad = false,
                        bool vis16 = false) {
    return mode | (rif << 4) | ((q_lds ? 1 : 0) << 8) | ((vis_lds ? 1 : 0) << 9) | ((quad ? 1 : 0) << 10) |
           ((vis16 ? 1 : 0) << 11) | (waves << 12);
}
constexpr bool walk_vis_lds(int code) { return ((code >> 9) & 1) != 0; }
constexpr bool walk_vis16(int code) { return ((code >> 11) & 1) != 0; }
constexpr bool walk_quad(int code) { return ((code >> 10) & 1) != 0; }
constexpr int walk_mode(int code) { return code & 3; }
constexpr int walk_rif(int code) { return (code >> 4) & 15; }
constexpr bool walk_q_lds(int code) { return ((code >> 8) & 1) != 0; }
constexpr int walk_waves(int code) { return (code >> 12) & 7; }
// bits 15-18 (measurement builds only): blocks of a row fetched before the early-abandon test, 0 = off
constexpr int walk_ea(int code) { return (code >> 15) & 15; }
constexpr int walk_with_ea(int code, int blocks) { return code | (blocks << 15); }
#ifndef IDIST_RIF9
#define IDIST_RIF9 4
#endif
#ifndef IDIST_RIF_OVERLAP
#define IDIST_RIF_OVERLAP 2
#endif
#ifndef IDIST_RIF24_OVERLAP
#define IDIST_RIF24_OVERLAP 1
#endif
// rounds in flight: bounded by the VGPRs one row fragment needs (a 300-d fragment is 38 dwords per lane)
template <int NB, int WALK>
constexpr int rounds_in_flight() {
    if (NB < 0 || walk_mode(WALK) == kWalkClassic) return 1;
    if (walk_rif(WALK)) return walk_rif(WALK);
    if (walk_vis_lds(WALK)) return NB <= 4 ? 8 : (NB <= 12 ? 4 : (NB <= 24 ? 3 : 1));   // one wave per SIMD: registers to spare
    if (walk_mode(WALK) == kWalkLatency) return NB <= 4 ? 8 : (NB <= 12 ? IDIST_RIF9 : (NB <= 24 ? 2 : 1));
    return NB <= 12 ? IDIST_RIF_OVERLAP : (NB <= 24 ? IDIST_RIF24_OVERLAP : 1);
}
// register tile of the runtime geometry (dist_rounds_inflight_rt) by the kernel's register budget: one fat wave per SIMD
// (512 registers) keeps 2 x 4 rounds x 8 blocks = 256 data registers = 64 KB on the wire; two waves per SIMD (256
// registers: the build's descents) 2 x 3 x 4 = 96; the many-small-waves bitmap walks (16 per CU: 128 registers each) 2 x 1 x 4 = 32
// (two waves per SIMD, 256 registers: <6 blocks, 3 rounds> = 2 x 3 x 6 float4 = 144 data registers, 36 KB on the wire per wave and a
//  WHOLE row of up to 12 blocks (384-d) in flight; the <4, 3> tile of round 4 kept 8 of 12 blocks in flight and built 1M x 384-d in
//  2.57-2.67 s where this one takes 1.75 s — profiles/r05/probe_r05f_build_rt_tiles_dim384.jsonl.  Measurement builds override both.)
#ifndef IDIST_RT2_ROUNDS
#define IDIST_RT2_ROUNDS 3
#endif
#ifndef IDIST_RT2_BLOCKS
#define IDIST_RT2_BLOCKS 6
#endif
constexpr bool walk_is_thin(int code) { return ((code >> 20) & 1) != 0; }   // (walk_thin, defined with the reject filter below)
#ifndef IDIST_THIN_LONG_CHUNKED
#define IDIST_THIN_LONG_CHUNKED 0   // blocks per group of a chunked f32 pass of thin walks on 768-d rows; 0: the row whole (see dist_rounds_walk)
#endif
template <int WALK> constexpr int rt_rounds() { return walk_is_thin(WALK) ? 1 : (walk_waves(WALK) == 1 ? 4 : (walk_waves(WALK) == 2 ? IDIST_RT2_ROUNDS : 1)); }
template <int WALK> constexpr int rt_blocks() { return walk_waves(WALK) == 1 ? 8 : (walk_waves(WALK) == 2 ? IDIST_RT2_BLOCKS : 4); }
template <int NB, int RS, int TAIL, int WALK, class Mid = NoMid>
__device__ __forceinline__ void dist_rounds_walk(const IndexView& ix, const float* q, const uint32_t* act_pid,
                                                 uint32_t* act_dist, int na, Mid mid = Mid(), uint32_t thr_bits = 0xFFFFFFFFu) {
    constexpr int RIF = rounds_in_flight<NB, WALK>();
    if constexpr (NB >= 0 && walk_ea(WALK) > 0) {
        if (na <= 0) mid();
        dist_rounds_inflight<NB, RS, TAIL, RIF, !walk_q_lds(WALK), 8, Mid, walk_ea(WALK)>(ix, natural_view(q, NB), act_pid, act_dist, na, 0, mid, thr_bits);
        return;
    }
    if constexpr (walk_is_thin(WALK) && NB > 12 && IDIST_THIN_LONG_CHUNKED && walk_mode(WALK) != kWalkClassic) {
        // thin filtered walks on long compile-time rows (the 768-d build descents) with a CHUNKED f32 pass, two groups of
        // IDIST_THIN_LONG_CHUNKED blocks on the wire — a measurement variant: the default fetches a surviving row whole (96 registers
        // per lane, 84 B of scratch) and builds 1M x 768 in 2.01 s against 2.30 / 2.17 / 2.08 s with groups of 6 / 8 / 12 blocks
        // (profiles/probe_r06u_thin768_whole_row.jsonl)
        if (na <= 0) mid();
        dist_rounds_inflight_rt<(IDIST_THIN_LONG_CHUNKED > 0 ? IDIST_THIN_LONG_CHUNKED : 1), 1>(ix, natural_view(q, (int)ix.nb), act_pid, act_dist, na, 0, mid);
    } else if constexpr (NB < 0 && walk_mode(WALK) != kWalkClassic) {
        if (na <= 0) mid();
        dist_rounds_inflight_rt<rt_blocks<WALK>(), rt_rounds<WALK>()>(ix, natural_view(q, (int)ix.nb), act_pid, act_dist, na, 0, mid);
    } else if constexpr (RIF > 1 || (walk_rif(WALK) == 1 && NB >= 0)) {
        if (na <= 0) mid();
        dist_rounds_inflight<NB, RS, TAIL, RIF, !walk_q_lds(WALK)>(ix, natural_view(q, NB), act_pid, act_dist, na, 0, mid);
    } else {
        mid();
        dist_rounds<NB, RS, TAIL>(ix, q, act_pid, act_dist, na);
    }
}

// ---------------------------------------------------------------------------
// The walk's reject filter (FilterView above).  Walk-code bit 19 compiles it in.
// ---------------------------------------------------------------------------
constexpr int kWalkFilterBit = 1 << 19;
constexpr bool walk_filter(int code) { return (code & kWalkFilterBit) != 0; }
constexpr int walk_with_filter(int code) { return code | kWalkFilterBit; }
// Thin filtered walks (bit 20).  With the filter in front an expansion moves ~21 KB instead of ~62 KB (C3) and the walk is bound by
// its dependent round trips, not by bytes: what helps is MORE walks per CU, each with less state — two waves per SIMD, the f32 pass
// one round of 8 rows at a time (a filtered expansion keeps ~4 rows), the query fragment read from LDS, no hand-over block of the
// four-wave walk in LDS, a smaller on-chip visited set.  (C3 ef 100, 10k queries: one fat wave per SIMD 7.6 ms, two thin ones 4.85,
// with all 64 compact rows of an expansion in flight 4.6; three per SIMD 4.97 — profiles/probe_r06c/d/j_*.jsonl.)
constexpr int kWalkThinBit = 1 << 20;
constexpr bool walk_thin(int code) { return (code & kWalkThinBit) != 0; }
constexpr int walk_thin_filter(int waves = 2) {  // the walk code of a thin filtered walk at `waves` per SIMD: quotient set, one f32 round in flight
    return walk_code(kWalkOverlap, 1, true, waves, true, false, true) | kWalkFilterBit | kWalkThinBit;
}
// 128-B chunks of a compact row (8 lanes x 16 B each; the last one may be half a chunk)
// WALK: the walk code of the kernel (runtime geometries only: the fat filtered walk's tile is larger than the thin one's)
template <int NB, int RS, int TAIL, int WALK = 0>
constexpr int filt_chunks() {
    if (NB < 0) return walk_is_thin(WALK) || ((WALK >> 19) & 1) == 0 ? kFiltRtChunks : kFiltRtChunksFat;
    const uint32_t used = 32u * (uint32_t)NB + 8u * (uint32_t)RS + 4u * (uint32_t)TAIL;
    const uint32_t stride = used < 16u ? 16u : ((used + 15u) & ~15u);
    return (int)((filt_stride(stride) + 127u) / 128u);
}
// The compile-time geometries' compact rows all end in HALF a chunk (320 = 2 x 128 + 64, 192, 832): that half is read as 8 bytes
// per lane by all 8 lanes of a group (T8) instead of 16 bytes by four of them — two registers of row data per lane and round less,
// which is what lets a thin wave keep all 64 rows of an expansion in flight at 300-d (8 x 10 registers).
template <int NB, int RS, int TAIL>
constexpr bool filt_tail8() {
    if (NB < 0) return false;
    const uint32_t used = 32u * (uint32_t)NB + 8u * (uint32_t)RS + 4u * (uint32_t)TAIL;
    const uint32_t stride = used < 16u ? 16u : ((used + 15u) & ~15u);
    return (filt_stride(stride) & 127u) == 64u;
}
// rows of one filter pass in flight: 8 * FR, bounded by the registers a row's bytes take per lane
template <int NCH, bool T8, int WALK> constexpr int filt_rounds() {
    const int per_round = 4 * (T8 ? NCH - 1 : NCH) + (T8 ? 2 : 0);
    // thin waves (256 registers): all 64 rows of an expansion at 300-d / 128-d (8 x 10 / 8 x 6 registers); long rows and the
    // runtime-geometry tile keep what leaves the kernel without scratch (768-d: 2 x 26, the query's byte planes take 52 more)
    // (fat waves: 8 x 26 at 768-d; the thirteen-chunk runtime tile keeps 2 x 52 beside its 104 registers of query bytes)
    const int budget = walk_waves(WALK) == 1 ? (NCH >= 8 ? 128 : 256) : (NCH >= 6 ? 56 : (T8 ? 80 : 64));
    const int r = budget / per_round;
    return r > 8 ? 8 : (r < 2 ? 2 : r);
}

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
#ifdef IDIST_EMU
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
#else
    return __builtin_amdgcn_udot4(a, b, c, false);        // v_dot4_u32_u8
#endif
}
// A query on the filter's lattice, held in registers for the whole walk: lane j of every 8-lane group keeps the bytes of the
// positions its loads of a compact row cover (chunk c: positions 128 c + 16 j ... + 15) — the 16-bit coordinate Q split into a
// high and a low byte plane, so that sum Q u = 256 (h . u) + (l . u) is two v_dot4_u32_u8 per dword of row.
template <int NCH, bool T8 = false>
struct FilterQ {
    static constexpr int NCF = T8 ? NCH - 1 : NCH;      // full 128-B chunks (16 bytes per lane); T8: + one half chunk (8 bytes per lane)
    uint4 h[NCF > 0 ? NCF : 1], l[NCF > 0 ? NCF : 1];
    uint2 ht, lt;
    float eq = 0.0f;        // |q - q^|_2, rounded up (NaN for a query with a NaN coordinate: nothing is ever rejected)
    uint64_t sq = 0;        // sum Q^2
    mutable bool on = false; // (a walk switches its own filter off when it is not paying, dist_pass_filtered)
    mutable uint32_t seen = 0, rejected = 0;   // per walk; search_kernel adds them to the context's counters when the query ends
};
template <int NCH, bool T8>
__device__ __forceinline__ void filter_stage_query(const IndexView& ix, const float* q, FilterQ<NCH, T8>& fq) {
    constexpr int NCF = FilterQ<NCH, T8>::NCF;
    fq.on = ix.f.rows != nullptr && ix.f.fstride <= 128u * (uint32_t)NCH && (!T8 || ix.f.fstride == 128u * (uint32_t)NCF + 64u);
    fq.ht = make_uint2(0u, 0u);
    fq.lt = make_uint2(0u, 0u);
#pragma unroll
    for (int c = 0; c < (NCF > 0 ? NCF : 1); c++) fq.h[c] = fq.l[c] = make_uint4(0u, 0u, 0u, 0u);
    if (!fq.on) return;
    const int j = lane_id() & 7;
    float e2 = 0.0f, mx = 0.0f;
    uint32_t sq_lo = 0, sq_hi = 0;
#pragma unroll
    for (int c = 0; c < NCF + (T8 ? 1 : 0); c++) {
        uint32_t hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
        const bool tail = T8 && c == NCF;               // the half chunk: 8 bytes per lane
#pragma unroll
        for (int b = 0; b < 16; b++) {
            if (tail && b >= 8) break;
            const uint32_t pos = 128u * (uint32_t)c + (tail ? 8u : 16u) * (uint32_t)j + (uint32_t)b;
            if (pos < ix.stride && natural_pos(pos, ix.nb) < ix.dim) {       // (padding: Q = u = 0 by construction, no error either side)
                const float v = q[pos];
                const float t = (v - ix.f.lo) * ix.f.qscale;
                const float r = __builtin_rintf(__builtin_fminf(__builtin_fmaxf(t, 0.0f), 65535.0f));   // fmax(NaN, 0) = 0
                const uint32_t Q = (uint32_t)r;
                const float e = v - __builtin_fmaf((float)Q, ix.f.step256, ix.f.lo);                     // NaN / inf for such a v
                e2 = __builtin_fmaf(e, e, e2);
                mx = __builtin_fmaxf(mx, __builtin_fabsf(v));
                const uint32_t qq = Q * Q;
                sq_hi += (sq_lo + qq < sq_lo) ? 1u : 0u;
                sq_lo += qq;
                hw[b >> 2] |= (Q >> 8) << (8 * (b & 3));
                lw[b >> 2] |= (Q & 255u) << (8 * (b & 3));
            }
        }
        if (tail) {
            fq.ht = make_uint2(hw[0], hw[1]);
            fq.lt = make_uint2(lw[0], lw[1]);
        } else {
            fq.h[c] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            fq.l[c] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    }
    // the 8 lanes of a group cover the row once: totals over the group (every group computes the same)
    uint64_t sq = ((uint64_t)sq_hi << 32) | sq_lo;
    for (int m = 1; m <= 4; m <<= 1) {
        e2 += __shfl_xor(e2, m, 64);
        const float o = __shfl_xor(mx, m, 64);
        mx = __builtin_fmaxf(mx, o);
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)sq, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(sq >> 32), m, 64);
        sq += ((uint64_t)hi << 32) | lo;
    }
    fq.sq = sq;
    // |q - q^| as evaluated in f32: each term is off by at most ~2^-22 of the magnitudes involved
    const float mag = mx + __builtin_fabsf(ix.f.lo) + 65536.0f * ix.f.step256;
    fq.eq = __builtin_sqrtf(e2) * 1.001f + ix.f.slack * mag;
}

// One filter pass over act_pid[0..na): act_dist[k] = kAbandoned where the compact row PROVES that the canonical distance
// exceeds the threshold (st = sqrt of the furthest distance of a full `nearest` for squared L2, the distance itself for L2),
// 0 otherwise.  8 lanes per row like the f32 gather; FR (<= 8) rounds of 8 rows are requested before the first is consumed.
// The dot products of a round are summed over the row's 8 lanes into ALL of them (three DPP adds), and lane j of a group keeps
// round j's: the integer arithmetic that follows runs ONCE per batch with every lane finishing another row — the row whose 8
// bytes of metadata that lane fetched itself — instead of once per round with one lane in eight at work.
// BOUND = true (idist_filter_bound_batch): act_dist[k] = the bits of the largest threshold the row would be rejected against —
// a lower bound of the canonical distance in the index's metric (st is not used).
__device__ __forceinline__ uint32_t group_total_u32(uint32_t v) {           // sum over the 8 lanes of a group, in every lane
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad's sum
    return v;
}
template <int NCH, bool T8, int FR, class Mid = NoMid, bool BOUND = false>
__device__ __forceinline__ void filter_rounds(const IndexView& ix, const FilterQ<NCH, T8>& fq, const uint32_t* act_pid, uint32_t* act_dist,
                                              int na, float st, Mid mid = Mid()) {
    static_assert(FR >= 1 && FR <= 8, "lane j of a group finishes round j");
    constexpr int NCF = FilterQ<NCH, T8>::NCF;
    constexpr int NCA = NCF > 0 ? NCF : 1;
    const int lane = lane_id();
    const int g = lane >> 3, j = lane & 7;
    const uint32_t fs = ix.f.fstride;
    for (int base = 0; base < na; base += 8 * FR) {
        uint4 p[FR][NCA];
        uint2 pt[FR];
#pragma unroll
        for (int r = 0; r < FR; r++) {
            const int k = base + 8 * r + g;
            const uint8_t* rowb = ix.f.rows + (size_t)act_pid[k < na ? k : 0] * fs;
            const uint8_t* row = rowb + 16u * (uint32_t)j;
#pragma unroll
            for (int c = 0; c < NCF; c++) {
                p[r][c] = make_uint4(0u, 0u, 0u, 0u);
                if (k < na && (T8 || 128u * (uint32_t)c + 16u * (uint32_t)j < fs)) p[r][c] = *reinterpret_cast<const uint4*>(row + 128u * (uint32_t)c);
            }
            pt[r] = make_uint2(0u, 0u);
            if constexpr (T8) { if (k < na) pt[r] = *reinterpret_cast<const uint2*>(rowb + 128u * (uint32_t)NCF + 8u * (uint32_t)j); }
        }
        // the row this lane finishes: round j of the batch, its metadata {|p - p^| (f32 bits), sum u^2} sits in the row's last 8 bytes
        const int kf = base + 8 * j + g;
        const bool fin = j < FR && kf < na;
        uint32_t ep = 0u, su = 0u;
        if (fin) {
            const uint32_t* m = reinterpret_cast<const uint32_t*>(ix.f.rows + (size_t)act_pid[kf] * fs + (fs - 8u));
            ep = m[0];
            su = m[1];
        }
        if (base == 0) mid();
        uint32_t ah_f = 0u, al_f = 0u;
#pragma unroll
        for (int r = 0; r < FR; r++) {
            uint32_t ah = 0u, al = 0u;
#pragma unroll
            for (int c = 0; c < NCF; c++) {                               // (the query's bytes are zero over the padding and the metadata)
                ah = udot4(fq.h[c].x, p[r][c].x, ah); al = udot4(fq.l[c].x, p[r][c].x, al);
                ah = udot4(fq.h[c].y, p[r][c].y, ah); al = udot4(fq.l[c].y, p[r][c].y, al);
                ah = udot4(fq.h[c].z, p[r][c].z, ah); al = udot4(fq.l[c].z, p[r][c].z, al);
                ah = udot4(fq.h[c].w, p[r][c].w, ah); al = udot4(fq.l[c].w, p[r][c].w, al);
            }
            if constexpr (T8) {
                ah = udot4(fq.ht.x, pt[r].x, ah); al = udot4(fq.lt.x, pt[r].x, al);
                ah = udot4(fq.ht.y, pt[r].y, ah); al = udot4(fq.lt.y, pt[r].y, al);
            }
            ah = group_total_u32(ah);
            al = group_total_u32(al);
            ah_f = j == r ? ah : ah_f;
            al_f = j == r ? al : al_f;
        }
        if (fin) {
            // I = sum (Q - 256 u)^2 = sum Q^2 - 512 sum Q u + 65536 sum u^2, exact
            const uint64_t A = ((uint64_t)ah_f << 8) + (uint64_t)al_f;
            const uint64_t I = fq.sq + ((uint64_t)su << 16) - (A << 9);
            const float dh = (float)I * ix.f.dscale;                      // |q^ - p^|^2
            if constexpr (BOUND) {
                // rejected iff dh > ((st + E) up)^2  <=>  st < sqrt(dh) / up - E: that supremum, rounded down
                float b = __builtin_sqrtf(dh) * 0.999999f / ix.f.up - (__uint_as_float(ep) + fq.eq) * 1.000001f;
                b = b > 0.0f ? b : 0.0f;                                  // (NaN: 0 — no bound)
                act_dist[kf] = __float_as_uint(ix.metric ? b : b * b * 0.999999f);
            } else {
                const float t = (st + (__uint_as_float(ep) + fq.eq)) * ix.f.up;
                act_dist[kf] = dh > t * t ? kAbandoned : 0u;              // (NaN anywhere: not rejected)
            }
        }
    }
}

// The distance pass of one expansion with the filter in front: the compact rows of all `na` new ids first (the caller's `mid` —
// its visited-set inserts — runs while they are in flight), then the f32 rows of the ids that were not rejected, through the
// walk's ordinary pass.  act_dist[k] ends up as the canonical distance bits of id k, or kAbandoned — above every key of a full
// `nearest`, so `push` turns it down exactly as it would have turned down the distance itself (core/lib.rs:712-714).
// thr_bits = 0xFFFFFFFF (nearest not full yet, or a build descent: its distance log needs every distance): no filter.
template <int NB, int RS, int TAIL, int WALK> using FilterQFor = FilterQ<filt_chunks<NB, RS, TAIL, WALK>(), filt_tail8<NB, RS, TAIL>()>;
template <int NB, int RS, int TAIL, int WALK, class Mid = NoMid>
__device__ __forceinline__ void dist_pass_filtered(const IndexView& ix, const float* q, const FilterQFor<NB, RS, TAIL, WALK>& fq,
                                                   uint32_t* act_pid, uint32_t* act_dist, int na, Mid mid, uint32_t thr_bits) {
    if constexpr (walk_filter(WALK)) {
        if (fq.on && thr_bits != 0xFFFFFFFFu && na > 0) {
            constexpr int NCH = filt_chunks<NB, RS, TAIL, WALK>();
            constexpr bool T8 = filt_tail8<NB, RS, TAIL>();
            const int lane = lane_id();
            const float thr = __uint_as_float(thr_bits);
            const float st = ix.metric ? thr : __builtin_sqrtf(thr);
            filter_rounds<NCH, T8, filt_rounds<NCH, T8, WALK>()>(ix, fq, act_pid, act_dist, na, st, mid);
            wave_sync();
            const bool in = lane < na;
            const uint32_t pid = in ? act_pid[lane] : 0u;
            const bool keep = in && act_dist[lane] != kAbandoned;
            const uint64_t km = __ballot(keep);
            const int pos = __popcll(km & ((1ull << lane) - 1ull));
            wave_sync();
            if (keep) act_pid[pos] = pid;                                  // slot order is kept
            wave_sync();
            const int ns = __popcll(km);
            fq.seen += (uint32_t)na;
            fq.rejected += (uint32_t)(na - ns);
            // A lattice that fits the data badly (heavy tails: most of a row's error in a few clamped coordinates) rejects little and
            // costs every candidate a second round trip: a walk that has examined 1024 candidates and spared fewer than a quarter of
            // their f32 rows goes on without the filter.  (Wave-uniform, a function of the query and the index only: deterministic.)
            if (fq.seen >= 1024u && fq.rejected * 4u < fq.seen) fq.on = false;
            if (ns) dist_rounds_walk<NB, RS, TAIL, WALK>(ix, q, act_pid, act_dist, ns);
            wave_sync();
            const uint32_t d = keep ? act_dist[pos] : kAbandoned;
            wave_sync();
            if (in) act_dist[lane] = d;
            return;
        }
    }
    dist_rounds_walk<NB, RS, TAIL, WALK>(ix, q, act_pid, act_dist, na, mid, thr_bits);
}

// ---------------------------------------------------------------------------
// Sorted-W state (Search.nearest + the live part of Search.candidates).
//   W[0..min(plen,ef))  = Search.nearest (core/lib.rs:568)
//   W[ef..plen)         = un-expanded candidates already truncated out of
//                         `nearest` whose distance EQUALS nearest.last()'s —
//                         the only truncated candidates the break test at
//                         core/lib.rs:600-604 (distance-only, strict >) can
//                         still expand.
// ---------------------------------------------------------------------------
struct WState {
    uint64_t* W;     // LDS, capacity ef_cap + 64 + kTieCap
    int plen;
    int ef;
    int cursor;      // lower bound of the first un-expanded entry
    uint32_t status;
    int tie_cap = kTieCap;   // capacity of the tie region behind W[ef)
    // Ties beyond that capacity (the reference's candidate heap is unbounded, core/lib.rs:564): an unsorted bag in HBM, all
    // of one distance (spill_fd = the furthest distance they tie with).  Invariant: every tie kept in LDS is smaller than
    // every key of the bag (spill_min), so the LDS region is popped first and refilled from the bag's smallest keys.
    uint64_t* spill = nullptr;          // [spill_cap] or nullptr: ties that do not fit are an error / dropped (tie policy)
    uint32_t spill_cap = 0, spill_n = 0, spill_fd = 0;
    uint64_t spill_min = ~0ull;
};

// number of entries with (masked) key < k  == Vec::binary_search Err(idx), core/lib.rs:712
__device__ __forceinline__ int w_rank(const WState& st, uint64_t k) {
    const int lane = lane_id();
    int cnt = 0;
    for (int i0 = 0; i0 < st.plen; i0 += 64) {
        const int i = i0 + lane;
        const bool lt = i < st.plen && (st.W[i] & kKeyMask) < k;
        const int c = __popcll(__ballot(lt));
        cnt += c;
        if (c < 64) break;   // sorted: first chunk that is not all-less ends the count
    }
    return cnt;
}

// Vec::insert(idx, new), core/lib.rs:718 (the matching candidates.push is implicit)
__device__ __forceinline__ void w_insert(WState& st, int idx, uint64_t k) {
    const int lane = lane_id();
    // shift [idx, plen) up by one, top chunk first so nothing is overwritten early
    int hi = st.plen;
    while (hi > idx) {
        int lo = hi - 64;
        if (lo < idx) lo = idx;
        const int i = lo + lane;
        uint64_t v = 0;
        const bool act = i < hi;
        if (act) v = st.W[i];
        wave_sync();
        if (act) st.W[i + 1] = v;
        wave_sync();
        hi = lo;
    }
    if (lane == 0) st.W[idx] = k;
    wave_sync();
    st.plen += 1;
    if (idx < st.cursor) st.cursor = idx;
}

// first un-expanded entry at or after cursor, -1 if none (== candidates.pop() of the
// minimum LIVE candidate, core/lib.rs:599; dead ones would only trigger the break)
__device__ __forceinline__ int w_pop(WState& st) {
    const int lane = lane_id();
    for (int i0 = st.cursor; i0 < st.plen; i0 += 64) {
        const int i = i0 + lane;
        const bool open = i < st.plen && !(st.W[i] & kFlag);
        const uint64_t m = __ballot(open);
        if (m) {
            const int idx = i0 + __builtin_ctzll(m);
            st.cursor = idx;
            return idx;
        }
    }
    st.cursor = st.plen;
    return -1;
}

// nearest.truncate(ef), core/lib.rs:612 — keeps, past ef, only live distance ties.
__device__ __forceinline__ void w_truncate(WState& st) {
    if (st.plen <= st.ef) return;
    const int lane = lane_id();
    if (st.ef == 0) { st.plen = 0; st.cursor = 0; return; }
    const uint32_t fd = (uint32_t)((st.W[st.ef - 1] & kKeyMask) >> 32);
    if (st.spill_n && fd != st.spill_fd) { st.spill_n = 0; st.spill_min = ~0ull; }   // the furthest distance fell: those ties are dead
    int out = st.ef;
    for (int i0 = st.ef; i0 < st.plen; i0 += 64) {
        const int i = i0 + lane;
        uint64_t v = 0;
        bool keep = false;
        if (i < st.plen) {
            v = st.W[i];
            keep = !(v & kFlag) && (uint32_t)(v >> 32) == fd;
        }
        const uint64_t m = __ballot(keep);
        wave_sync();
        if (keep) st.W[out + __popcll(m & ((1ull << lane) - 1ull))] = v;
        wave_sync();
        out += __popcll(m);
    }
    int keep_lds = out - st.ef;
    if (st.spill_n) {                                     // ties not below the bag's smallest key queue up behind it
        int below = 0;
        for (int i0 = st.ef; i0 < out; i0 += 64) {
            const int i = i0 + lane;
            below += __popcll(__ballot(i < out && st.W[i] < st.spill_min));   // (kept ties carry no flag)
        }
        keep_lds = below;
    }
    if (keep_lds > st.tie_cap) keep_lds = st.tie_cap;
    const int n_mv = out - st.ef - keep_lds;
    if (n_mv > 0) {
        if (st.spill && st.spill_n + (uint32_t)n_mv <= st.spill_cap) {
            const uint64_t first = st.W[st.ef + keep_lds];
            for (int i = lane; i < n_mv; i += 64) st.spill[st.spill_n + (uint32_t)i] = st.W[st.ef + keep_lds + i];
            st.spill_n += (uint32_t)n_mv;
            st.spill_fd = fd;
            if (first < st.spill_min) st.spill_min = first;
        } else {
            st.status |= kStTieOverflow;                  // no bag (or, impossibly, a full one): reported, never silent
        }
        out = st.ef + keep_lds;
    }
    st.plen = out;
    if (st.cursor > st.ef) st.cursor = st.ef;
}

// The LDS tie region is exhausted and the bag is not: move the bag's tie_cap smallest keys into W[ef..) (sorted), close the
// gap they leave.  Returns false if nothing could be refilled (the bag's ties died with the furthest distance).
__device__ __forceinline__ bool w_refill_ties(WState& st) {
    const int lane = lane_id();
    if (st.plen < st.ef || st.ef == 0) { st.spill_n = 0; st.spill_min = ~0ull; return false; }
    const uint32_t fd = (uint32_t)((st.W[st.ef - 1] & kKeyMask) >> 32);
    if (fd != st.spill_fd) { st.spill_n = 0; st.spill_min = ~0ull; return false; }
#ifndef IDIST_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the bag's own stores have landed before it is read back
#endif
    wave_sync();
    WState ts{st.W + st.ef, 0, st.tie_cap, 0, 0u};
    for (uint32_t base = 0; base < st.spill_n; base += 64u) {
        const bool on = base + (uint32_t)lane < st.spill_n;
        const uint64_t key = on ? st.spill[base + (uint32_t)lane] : kMaxKey;
        const uint64_t thr = ts.plen >= ts.ef ? (ts.W[ts.ef - 1] & kKeyMask) : kMaxKey + 1ull;
        uint64_t pm = __ballot(on && key < thr);
        while (pm) {
            const int i = __builtin_ctzll(pm);
            pm &= pm - 1ull;
            const uint64_t kk = bcast_u64(key, i);
            const int idx = w_rank(ts, kk);
            if (idx < ts.ef) w_insert(ts, idx, kk);
            if (ts.plen > ts.ef) ts.plen = ts.ef;
        }
        wave_sync();
    }
    const int cnt = ts.plen;
    if (cnt == 0) { st.spill_n = 0; st.spill_min = ~0ull; return false; }
    const uint64_t thr = st.W[st.ef + cnt - 1];
    uint32_t out = 0;
    uint64_t mn = ~0ull;
    for (uint32_t base = 0; base < st.spill_n; base += 64u) {
        const bool on = base + (uint32_t)lane < st.spill_n;
        const uint64_t key = on ? st.spill[base + (uint32_t)lane] : 0ull;
        const bool keepk = on && key > thr;
        const uint64_t m = __ballot(keepk);
        // in place: out + |kept| <= base + 64, and this chunk already sits in registers
        if (keepk) { st.spill[out + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = key; mn = key < mn ? key : mn; }
        out += (uint32_t)__popcll(m);
    }
    for (int sh = 32; sh >= 1; sh >>= 1) {                 // wave minimum of the surviving keys
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mn, sh, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mn >> 32), sh, 64);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        mn = o < mn ? o : mn;
    }
    st.spill_n = out;
    st.spill_min = out ? mn : ~0ull;
    st.plen = st.ef + cnt;
    st.cursor = st.ef;
    wave_sync();
    return true;
}

// Search::cull, core/lib.rs:729-737: candidates := nearest, visited := pids(nearest)
__device__ __forceinline__ void w_cull(WState& st) {
    const int lane = lane_id();
    if (st.plen > st.ef) st.plen = st.ef;
    for (int i = lane; i < st.plen; i += 64) st.W[i] &= kKeyMask;
    st.cursor = 0;
    st.spill_n = 0;
    st.spill_min = ~0ull;
    wave_sync();
}

// ---------------------------------------------------------------------------
// Visited (core/types.rs:13-59): exact set membership, one BIT per point and resident query slot in HBM.
//
// The reference stamps a generation byte per point and clears by bumping the generation (:48-58) — "an
// optimisation with no observable effect" (SURVEY App. A.13).  A byte per point and slot is 4 GB for 4096
// slots at 1M points: 8x the page footprint of the bitmap for the same random single-sector updates.  The bitmap
// has no generation, so `clear` has to zero what was set; the wave remembers WHERE it set bits in a small LDS
// bitmap over 64-B..512-B blocks of its slot ("dirty blocks") and zeroes exactly those with full-sector stores
// (no read).  Upper layers (ef = 1) dirty a few dozen blocks, the zero layer most of a 1M-point slot — never
// more than one streaming pass over n/8 bytes per query.
//
// All updates are L2 atomics (`global_atomic_or`): lanes of one wave may share a word, and the returning form is
// the test-and-set of Visited::insert (:32-40).  No plain load ever reads the bitmap, so the CU's L1 cannot
// serve a stale copy.  A slot belongs to one wave; slots never share a 64-B sector.
//
// In front of it an optional per-wave Bloom filter in LDS (two hashes): it has no false negatives, so "not in
// the filter" proves the node new — it is marked with the fire-and-forget form of the atomic and goes straight
// to the distance rounds, while a "maybe" waits for the returning form (the bitmap stays the ground truth).
// ---------------------------------------------------------------------------
#ifndef IDIST_BLOOM_LOG2_WORDS
#define IDIST_BLOOM_LOG2_WORDS 11
#endif
constexpr int kBloomLog2Words = IDIST_BLOOM_LOG2_WORDS;     // 2048 words = 8 KB: throughput mode (many waves per CU)
constexpr int kBloomWords = 1 << kBloomLog2Words;
#ifndef IDIST_BLOOM_LAT_LOG2_WORDS
#define IDIST_BLOOM_LAT_LOG2_WORDS 13
#endif
constexpr int kBloomLatLog2Words = IDIST_BLOOM_LAT_LOG2_WORDS;   // 8192 words = 32 KB: latency mode (few waves, LDS to spare)
constexpr int kBloomLatWords = 1 << kBloomLatLog2Words;

// Geometry of one slot: the bitmap is cut into blocks of 2^shift points (>= 512 points = one 64-B sector) such
// that the dirty-block bitmap fits kDirtyMaxWords dwords of LDS whatever n is.
constexpr uint32_t kDirtyMaxWords = 256;                    // 1 KB of LDS: 8192 blocks (128-B blocks at 10M points)
struct VisGeom {
    uint32_t shift;        // log2(points per block), >= 9
    uint32_t blocks;       // blocks per slot
    uint32_t dirty_words;  // LDS dwords of the dirty-block bitmap (multiple of 64: whole wave strides)
    uint32_t slot_words;   // dwords per slot in HBM (blocks << (shift - 5))
};
__host__ __device__ inline VisGeom vis_geometry(uint32_t n) {
    VisGeom g;
    g.shift = 9;
    while ((((uint64_t)n + (1ull << g.shift) - 1) >> g.shift) > (uint64_t)kDirtyMaxWords * 32u) g.shift++;
    g.blocks = (uint32_t)(((uint64_t)n + (1ull << g.shift) - 1) >> g.shift);
    if (g.blocks == 0) g.blocks = 1;
    g.dirty_words = ((g.blocks + 31u) / 32u + 63u) & ~63u;
    g.slot_words = g.blocks << (g.shift - 5);
    return g;
}

struct Visited {
    uint32_t* bits;     // HBM: this slot's bitmap, all-zero whenever no search is in progress on the slot
    uint32_t n;
    uint32_t* dirty;    // LDS: bit b set <=> block b of `bits` may hold a set bit
    uint32_t shift;     // log2(points per block)
    uint32_t dirty_words;
    uint32_t* bloom;    // LDS, 1 << blog2 words, or nullptr
    int blog2;          // log2(words)
    // On-chip variant: an exact hash set of the visited ids in LDS (open addressing, linear probing, EMPTY = INVALID).
    // While it has room the HBM bitmap is not touched at all; once `count` reaches `tlimit` the set is frozen for the
    // rest of the layer ("spill"): lookups still probe it, new ids go to the bitmap.
    uint32_t* tab = nullptr;   // LDS, tmask + 1 entries, or nullptr
    uint32_t tmask = 0, tshift = 0, tlimit = 0;   // buckets - 1, 32 - log2(buckets), ids the set may hold before it is frozen
    uint32_t count = 0;        // entries in tab (wave-uniform)
    bool spill = false;        // tab frozen, bitmap in use (wave-uniform)
    bool dirtied = false;      // the bitmap may hold set bits (wave-uniform)
    // Quotient form of the on-chip set (search walks): see q16_* below
    bool q16 = false;
    uint32_t ubits = 0, rbits = 0;   // bits of the id universe (2^ubits >= n), bits of a remainder
};
__device__ __forceinline__ uint32_t bloom_h1(const Visited& v, uint32_t pid) { return (pid * 0x9E3779B1u) >> (27 - v.blog2); }
__device__ __forceinline__ uint32_t bloom_h2(const Visited& v, uint32_t pid) { return (pid * 0x85EBCA6Bu + 0xC2B2AE35u) >> (27 - v.blog2); }
__device__ __forceinline__ bool bloom_maybe(const Visited& v, uint32_t pid) {
    const uint32_t a = bloom_h1(v, pid), b = bloom_h2(v, pid);
    return ((v.bloom[a >> 5] >> (a & 31u)) & (v.bloom[b >> 5] >> (b & 31u)) & 1u) != 0u;
}
__device__ __forceinline__ void bloom_set(const Visited& v, uint32_t pid) {
    const uint32_t a = bloom_h1(v, pid), b = bloom_h2(v, pid);
    atomicOr(&v.bloom[a >> 5], 1u << (a & 31u));
    atomicOr(&v.bloom[b >> 5], 1u << (b & 31u));
}
// everything the wave's earlier bitmap updates and zeroing stores did is performed before anything that follows
__device__ __forceinline__ void visited_drain() {
#ifndef IDIST_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// the LDS side of a freshly set bit: its block is dirty, the filter knows the node
__device__ __forceinline__ void visited_note(const Visited& v, uint32_t pid) {
    const uint32_t blk = pid >> v.shift;
    atomicOr(&v.dirty[blk >> 5], 1u << (blk & 31u));
    if (v.bloom) bloom_set(v, pid);
}
// The set is cut into buckets of four ids (one ds_read_b128).  Every id has TWO home buckets (two hashes): a lookup
// reads both at once — one LDS round trip, no probe loop — and an insert goes to the emptier one (first empty entry,
// ds_cmpst; lanes of a wave insert distinct ids concurrently, the loser of a race looks again).  Buckets fill front to
// back and never lose an entry before the next clear.  Only an id that finds BOTH home buckets full overflows: into
// the first bucket with room on the linear chain behind its second home bucket; so a lookup walks that chain only
// while both home buckets are full, and "this bucket has an empty entry" ends the walk.  (With one home bucket and
// linear probing the longest chain among the 64 lanes of an expansion set the pace: 5-8 dependent LDS round trips
// at 70 % load.)
__device__ __forceinline__ uint32_t tab_hash1(uint32_t pid, uint32_t shift) { return (pid * 0x9E3779B1u) >> shift; }
__device__ __forceinline__ uint32_t tab_hash2(uint32_t pid, uint32_t shift) { return ((pid ^ (pid >> 15)) * 0x85EBCA6Bu + 0xC2B2AE35u) >> shift; }
__device__ __forceinline__ int bucket_find(const uint4 e, uint32_t pid) {
    return e.x == pid ? 0 : (e.y == pid ? 1 : (e.z == pid ? 2 : (e.w == pid ? 3 : -1)));
}
__device__ __forceinline__ int bucket_fill(const uint4 e) {      // entries in use = index of the first empty one
    return e.x == kInvalid ? 0 : (e.y == kInvalid ? 1 : (e.z == kInvalid ? 2 : (e.w == kInvalid ? 3 : 4)));
}
// index (in words) of pid in a set stored at T (bmask + 1 buckets of BW words — the four ids first —, hashes shifted by
// bshift), -1 if it is not there.  BW = 4: the on-chip set; BW = 8: the published form, ids + their four distances
template <uint32_t BW>
__device__ __forceinline__ int tabset_index(const uint32_t* T, uint32_t bmask, uint32_t bshift, uint32_t pid) {
    const uint32_t b1 = tab_hash1(pid, bshift), b2 = tab_hash2(pid, bshift);
    const uint4 e1 = *reinterpret_cast<const uint4*>(T + BW * b1);
    if constexpr (BW == 8) {                                    // published form in HBM: the second home bucket only on a miss
        const int k1 = bucket_find(e1, pid);
        if (k1 >= 0) return (int)(BW * b1) + k1;
    }
    const uint4 e2 = *reinterpret_cast<const uint4*>(T + BW * b2);
    int k = bucket_find(e1, pid);
    if (k >= 0) return (int)(BW * b1) + k;
    k = bucket_find(e2, pid);
    if (k >= 0) return (int)(BW * b2) + k;
    if (e1.w == kInvalid || e2.w == kInvalid) return -1;       // a home bucket has room: the id never overflowed
    uint32_t b = (b2 + 1u) & bmask;
    for (uint32_t probe = 0; probe <= bmask; probe++) {
        const uint4 e = *reinterpret_cast<const uint4*>(T + BW * b);
        k = bucket_find(e, pid);
        if (k >= 0) return (int)(BW * b) + k;
        if (e.w == kInvalid) return -1;
        b = (b + 1u) & bmask;
    }
    return -1;
}
__device__ __forceinline__ int tab_index(const Visited& v, uint32_t pid) { return tabset_index<4>(v.tab, v.tmask, v.tshift, pid); }
__device__ __forceinline__ bool tab_find(const Visited& v, uint32_t pid) { return tab_index(v, pid) >= 0; }
// insert pid into the LDS set: its index if it was new, -1 if it was there already.  The set is never full
// (frozen at 7/8, visited_begin), so the overflow walk ends.
__device__ __forceinline__ int tab_insert(const Visited& v, uint32_t pid) {
    const uint32_t b1 = tab_hash1(pid, v.tshift), b2 = tab_hash2(pid, v.tshift);
    for (;;) {
        const uint4 e1 = *reinterpret_cast<const uint4*>(v.tab + 4u * b1);
        const uint4 e2 = *reinterpret_cast<const uint4*>(v.tab + 4u * b2);
        if (bucket_find(e1, pid) >= 0 || bucket_find(e2, pid) >= 0) return -1;
        const int f1 = bucket_fill(e1), f2 = bucket_fill(e2);
        uint32_t b = f2 < f1 ? b2 : b1;
        int k = f2 < f1 ? f2 : f1;
        if (k == 4) {                                           // both full: first bucket with room behind the second home
            b = (b2 + 1u) & v.tmask;
            for (;;) {
                const uint4 e = *reinterpret_cast<const uint4*>(v.tab + 4u * b);
                if (bucket_find(e, pid) >= 0) return -1;
                k = bucket_fill(e);
                if (k < 4) break;
                b = (b + 1u) & v.tmask;
            }
        }
        const uint32_t old = atomicCAS(&v.tab[4u * b + (uint32_t)k], kInvalid, pid);
        if (old == kInvalid) return (int)(4u * b) + k;
        if (old == pid) return -1;
        // another lane claimed the entry (a different id: a row never holds duplicates): look again
    }
}
// ---------------------------------------------------------------------------
// Quotient form of the on-chip set (search walks): 16-bit entries, twice the ids in the same LDS.
//
// Both hashes are BIJECTIONS of the id universe [0, 2^ubits) (an odd multiplier, an xor-shift, an added constant, all
// mod 2^ubits), so an id is determined by (which hash, bucket, remainder): the bucket index carries the top bits of the
// hashed id, the entry only stores {which hash : 1, remainder : rbits} in 16 bits (0xFFFF = empty; rbits <= 14 keeps
// bit 15 clear).  A bucket is eight entries = the same 16 B (one ds_read_b128) that hold four full ids in the plain form:
// 16384 ids in 32 KB.  Membership stays EXACT: an id is in the set iff its first home bucket holds {0, r1} or its second
// {1, r2}.
//
// No overflow chain and no freeze: an id that finds BOTH home buckets full goes to the HBM bitmap on its own (buckets
// fill front to back and never lose an entry before the next clear, so "both full" holds for that id until then: it
// is looked up in, and only in, the bitmap from then on; and an id that found room never was in the bitmap).  With two
// choices and eight-entry buckets the table takes ~90 % of its capacity before the first id overflows, and the share
// of ids that need the bitmap round trip then grows gradually instead of jumping to 100 % at a threshold.
// ---------------------------------------------------------------------------
enum : int { kQFound = 0, kQRoom = 1, kQFull = 2 };
struct Q16Keys { uint32_t b1, t1, b2, t2; };
__device__ __forceinline__ Q16Keys q16_keys(uint32_t ubits, uint32_t rbits, uint32_t pid) {
    const uint32_t um = (1u << ubits) - 1u, rm = (1u << rbits) - 1u;
    const uint32_t h1 = (pid * 0x9E3779B1u) & um;
    const uint32_t h2 = ((pid ^ (pid >> 7)) * 0x85EBCA6Bu + 0xC2B2AE35u) & um;
    return Q16Keys{h1 >> rbits, h1 & rm, h2 >> rbits, (h2 & rm) | (1u << rbits)};
}
__device__ __forceinline__ Q16Keys q16_keys(const Visited& v, uint32_t pid) { return q16_keys(v.ubits, v.rbits, pid); }
__device__ __forceinline__ bool q16_word_has(uint32_t w, uint32_t tt) {     // tt = t | t << 16
    const uint32_t x = w ^ tt;
    return (x & 0xFFFFu) == 0u || (x >> 16) == 0u;
}
__device__ __forceinline__ bool q16_has(const uint4 e, uint32_t t) {
    const uint32_t tt = t | (t << 16);
    return q16_word_has(e.x, tt) || q16_word_has(e.y, tt) || q16_word_has(e.z, tt) || q16_word_has(e.w, tt);
}
__device__ __forceinline__ int q16_fill(const uint4 e) {                     // entries in use = index of the first empty one
    auto wf = [](uint32_t w) { return (w & 0xFFFFu) == 0xFFFFu ? 0 : ((w >> 16) == 0xFFFFu ? 1 : 2); };
    const int a = wf(e.x), b = wf(e.y), c = wf(e.z), d = wf(e.w);
    return a < 2 ? a : (b < 2 ? 2 + b : (c < 2 ? 4 + c : 6 + d));
}
__device__ __forceinline__ int q16_lookup(const Visited& v, uint32_t pid) {
    const Q16Keys k = q16_keys(v, pid);
    const uint4 e1 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b1);
    const uint4 e2 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b2);
    if (q16_has(e1, k.t1) || q16_has(e2, k.t2)) return kQFound;
    return ((e1.w >> 16) == 0xFFFFu || (e2.w >> 16) == 0xFFFFu) ? kQRoom : kQFull;
}
// slot (0..7) of the entry t in a bucket, -1 if it is not there
__device__ __forceinline__ int q16_slot(const uint4 e, uint32_t t) {
    auto ws = [t](uint32_t w) { return (w & 0xFFFFu) == t ? 0 : ((w >> 16) == t ? 1 : -1); };
    const int a = ws(e.x), b = ws(e.y), c = ws(e.z), d = ws(e.w);
    return a >= 0 ? a : (b >= 0 ? 2 + b : (c >= 0 ? 4 + c : (d >= 0 ? 6 + d : -1)));
}
// index of pid in the set = 8 * bucket + slot, -1 if it is not there (ids that overflowed to the bitmap have none)
__device__ __forceinline__ int q16_index(const Visited& v, uint32_t pid) {
    const Q16Keys k = q16_keys(v, pid);
    const uint4 e1 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b1);
    const uint4 e2 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b2);
    const int s1 = q16_slot(e1, k.t1);
    if (s1 >= 0) return (int)(8u * k.b1) + s1;
    const int s2 = q16_slot(e2, k.t2);
    return s2 >= 0 ? (int)(8u * k.b2) + s2 : -1;
}
// insert: kQRoom = it went in (idx = where), kQFound = it was there, kQFull = both home buckets are full (the caller
// uses the bitmap).  Lanes of a wave insert distinct ids concurrently; the loser of a race for a word looks again.
__device__ __forceinline__ int q16_insert(const Visited& v, uint32_t pid, int& idx) {
    idx = -1;
    const Q16Keys k = q16_keys(v, pid);
    for (;;) {
        const uint4 e1 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b1);
        const uint4 e2 = *reinterpret_cast<const uint4*>(v.tab + 4u * k.b2);
        if (q16_has(e1, k.t1) || q16_has(e2, k.t2)) return kQFound;
        const int f1 = q16_fill(e1), f2 = q16_fill(e2);
        if (f1 == 8 && f2 == 8) return kQFull;
        const bool second = f2 < f1;
        const uint4 e = second ? e2 : e1;
        const int f = second ? f2 : f1;
        const uint32_t t = second ? k.t2 : k.t1, b = second ? k.b2 : k.b1;
        const int wi = f >> 1;
        const uint32_t cur = wi == 0 ? e.x : (wi == 1 ? e.y : (wi == 2 ? e.z : e.w));
        const uint32_t nw = (f & 1) ? ((cur & 0x0000FFFFu) | (t << 16)) : ((cur & 0xFFFF0000u) | t);
        if (atomicCAS(&v.tab[4u * b + (uint32_t)wi], cur, nw) == cur) { idx = (int)(8u * b) + f; return kQRoom; }
    }
}
__device__ __forceinline__ int q16_insert(const Visited& v, uint32_t pid) {
    int idx;
    return q16_insert(v, pid, idx);
}
// index of pid in the on-chip set, whichever form it has (-1: not there)
__device__ __forceinline__ int vis_index(const Visited& v, uint32_t pid) {
    return v.q16 ? q16_index(v, pid) : (v.tab ? tab_index(v, pid) : -1);
}
// Visited::clear (core/types.rs:48-58): empty the on-chip set / zero the dirty blocks.  Wave-uniform control flow.
__device__ __forceinline__ void visited_clear(Visited& v) {
    const int lane = lane_id();
    wave_sync();                                                // the other lanes' LDS updates are in
    if (v.bloom) {
        uint4* b = reinterpret_cast<uint4*>(v.bloom);
        for (int i = lane; i < (1 << v.blog2) / 4; i += 64) b[i] = make_uint4(0, 0, 0, 0);
    }
    if (v.tab) {
        uint4* t = reinterpret_cast<uint4*>(v.tab);
        for (uint32_t i = lane; i <= v.tmask; i += 64) t[i] = make_uint4(kInvalid, kInvalid, kInvalid, kInvalid);
        v.count = 0;
        v.spill = false;
    }
    if (!v.tab || v.dirtied || v.q16) {                         // (quotient form: single ids overflow, no wave-uniform flag)
        const uint32_t wpb = 1u << (v.shift - 5);              // dwords per block (>= 16)
        for (uint32_t w0 = 0; w0 < v.dirty_words; w0 += 64) {
            const uint32_t mine = v.dirty[w0 + lane];
            uint64_t nz = __ballot(mine != 0u);
            if (mine) v.dirty[w0 + lane] = 0u;
            while (nz) {                                        // one dirty word (32 blocks) per iteration
                const int src = __builtin_ctzll(nz);
                nz &= nz - 1ull;
                const uint32_t m = bcast_u32(mine, src);
                // two lanes per block: lane >> 1 = bit, lane & 1 = half of the block
                const uint32_t bit = (uint32_t)lane >> 1;
                if ((m >> bit) & 1u) {
                    const uint32_t blk = (w0 + (uint32_t)src) * 32u + bit;
                    uint4* p = reinterpret_cast<uint4*>(v.bits + ((size_t)blk << (v.shift - 5)) + (size_t)(lane & 1) * (wpb / 2));
                    for (uint32_t i = 0; i < wpb / 8; i++) p[i] = make_uint4(0, 0, 0, 0);
                }
            }
        }
        visited_drain();
        v.dirtied = false;
    }
    wave_sync();
}
// Wave-uniform bookkeeping of the on-chip set around one expansion: freeze it before it could fill up
// (an expansion adds at most 64 ids), count what was added.
__device__ __forceinline__ void visited_begin(Visited& v, uint32_t n_max = 64u) {
    if (v.tab && !v.spill && v.count + n_max > v.tlimit) v.spill = true;
}
__device__ __forceinline__ void visited_added(Visited& v, uint32_t n_new) {
    if (!v.tab) return;
    if (v.spill) { if (n_new) v.dirtied = true; }
    else v.count += n_new;
}
// the returning atomic is the test-and-set of Visited::insert (:32-40)
__device__ __forceinline__ bool visited_test_and_set(const Visited& v, uint32_t pid) {
    const uint32_t bit = 1u << (pid & 31u);
    const uint32_t old = atomicOr(&v.bits[pid >> 5], bit);
    return (old & bit) == 0u;
}
// Visited::extend with one pid per lane (core/types.rs:42-46; cull) / a node known to be new.
// (with the on-chip set the caller brackets it with visited_begin / visited_added like an expansion)
__device__ __forceinline__ void visited_mark(const Visited& v, uint32_t pid) {
    if (v.q16) {
        if (q16_insert(v, pid) != kQFull) return;
    } else if (v.tab) {
        if (!v.spill) { (void)tab_insert(v, pid); return; }
        if (tab_find(v, pid)) return;
    }
    atomicOr(&v.bits[pid >> 5], 1u << (pid & 31u));            // result unused: fire and forget
    visited_note(v, pid);
}
// Visited::insert for one lane's pid (core/types.rs:32-40): true if it was new; tab_idx = where it sits in the on-chip
// set (-1: not there)
__device__ __forceinline__ bool visited_insert(const Visited& v, uint32_t pid, int& tab_idx) {
    tab_idx = -1;
    if (v.q16) {
        const int r = q16_insert(v, pid, tab_idx);
        if (r != kQFull) return r == kQRoom;
    } else if (v.tab) {
        if (!v.spill) { tab_idx = tab_insert(v, pid); return tab_idx >= 0; }
        if (tab_find(v, pid)) return false;
    } else if (v.bloom && !bloom_maybe(v, pid)) {
        visited_mark(v, pid);
        return true;
    }
    const bool fresh = visited_test_and_set(v, pid);
    if (fresh) visited_note(v, pid);
    return fresh;
}

// ---------------------------------------------------------------------------
// Distance log of one insertion (build only): the neighbour re-selection of step B needs d(new, r) for the
// members r of the rows the new point chose — distances its own descent already computed.  They are handed over
// through HBM without a single read-modify-write: while it walks, the wave APPENDS (distance, index of the id in
// its on-chip visited set) pairs to a per-insertion log — contiguous 8-B stores — and when the descent is over it
// publishes (a) the on-chip set itself (ids, one coalesced pass) and (b) the distances scattered to the same
// indices, in one burst.  Step B then probes the published id array exactly like the on-chip set (dlog_find).
// Only the last layer's distances survive (the set is emptied at every layer transition), ids that went to the
// overflow bitmap are not logged: a miss is always legal, the caller recomputes.
// ---------------------------------------------------------------------------
constexpr uint32_t kDlogMiss = 0xFFFFFFFFu;            // never a canonical distance pattern
// Descents with the reject filter (round 6): a candidate the filter turned down has no distance to log — what is known is that it
// lies BEYOND the threshold of that expansion (d > thr, strictly: the furthest distance of a full `nearest`, which is at least the
// descent's final one).  It is logged in bound form, the sign bit (free: canonical distance patterns are non-negative or the
// positive canonical NaN) over the threshold's bits; step B resolves it against the one comparand it needs (dlog_resolve).
constexpr uint32_t kDlogBound = 0x80000000u;
__device__ __forceinline__ bool dlog_is_bound(uint32_t v) { return (v & kDlogBound) != 0u && v != kDlogMiss; }
// `v` will be compared as `d < x` (strict, core/lib.rs:678): a bound-form entry with thr >= x decides it (d > thr >= x: false) and is
// replaced by thr itself — any value >= x gives the same verdict; one with thr < x decides nothing: a miss, the caller recomputes.
__device__ __forceinline__ uint32_t dlog_resolve(uint32_t v, uint32_t x) {
    if (!dlog_is_bound(v)) return v;
    const uint32_t thr = v & ~kDlogBound;
    return thr >= x ? thr : kDlogMiss;
}
struct DistLog {
    uint64_t* log;      // HBM [ids the set holds]: dist_bits << 32 | index in the set; nullptr = nothing is logged
    uint32_t n;         // entries (wave-uniform)
    // Quotient form: the set holds twice as many entries as it has dwords, so its distances are sorted through its LDS in
    // two halves (dlog_publish_q16).  The log is kept in two halves as well — entries with index < half at log[0 ..), the
    // others at log[half ..) (an index is logged at most once per layer, so neither half can overflow) — and each pass of
    // the publication reads only the entries it scatters.  half = 0: one log (the id form).
    uint32_t half = 0, n_hi = 0;
    __device__ __forceinline__ void reset() { n = 0; n_hi = 0; }
};
// wave-collective: lanes with idx >= 0 (the id's index in the on-chip set) log their distance
__device__ __forceinline__ void dlog_append(DistLog& L, int idx, uint32_t dist_bits) {
    const uint64_t m = __ballot(idx >= 0);
    if (!m) return;
    const uint64_t below = (1ull << lane_id()) - 1ull;
    const uint64_t e = ((uint64_t)dist_bits << 32) | (uint32_t)idx;
    if (L.half) {
        const bool hi = idx >= (int)L.half;
        const uint64_t mh = __ballot(idx >= 0 && hi), ml = m & ~mh;
        if (idx >= 0) {
            if (hi) L.log[L.half + L.n_hi + (uint32_t)__popcll(mh & below)] = e;
            else L.log[L.n + (uint32_t)__popcll(ml & below)] = e;
        }
        L.n += (uint32_t)__popcll(ml);
        L.n_hi += (uint32_t)__popcll(mh);
        return;
    }
    if (idx >= 0) L.log[L.n + (uint32_t)__popcll(m & below)] = e;
    L.n += (uint32_t)__popcll(m);
}
// End of the descent: the set goes out in its published form, one 32-B record per bucket = its four ids followed by
// their four distances, so that a lookup of step B touches one sector per home bucket and finds the distance in the
// same one.  First the ids (all buckets, empty entries included: nothing to clear beforehand); then the set's LDS is
// reused to sort the logged distances by index, and they go out into the other half of every record (entries of
// empty slots are never read).  The on-chip set is destroyed: the caller clears it (visited_clear) before the next descent.
__device__ __forceinline__ void dlog_publish(const DistLog& L, const Visited& v, uint32_t* out_pd) {
    const int lane = lane_id();
    wave_sync();
    uint4* t = reinterpret_cast<uint4*>(v.tab);
    uint4* o = reinterpret_cast<uint4*>(out_pd);
    for (uint32_t i = lane; i <= v.tmask; i += 64) o[2u * i] = t[i];
    visited_drain();                                       // the log's own stores have landed before it is read back
    wave_sync();
    constexpr int kDeep = 16;                              // log entries per lane in flight: one round trip per 1024 entries
    for (uint32_t base = 0; base < L.n; base += 64u * kDeep) {
        uint64_t e[kDeep];
#pragma unroll
        for (int u = 0; u < kDeep; u++) {
            const uint32_t i = base + 64u * (uint32_t)u + (uint32_t)lane;
            e[u] = L.log[i < L.n ? i : 0u];
        }
#pragma unroll
        for (int u = 0; u < kDeep; u++)
            if (base + 64u * (uint32_t)u + (uint32_t)lane < L.n) v.tab[(uint32_t)e[u]] = (uint32_t)(e[u] >> 32);
    }
    wave_sync();
    for (uint32_t i = lane; i <= v.tmask; i += 64) o[2u * i + 1u] = t[i];
}
// step B: d(new, pid) from the published set of `new`'s descent, kDlogMiss if it is not there
__device__ __forceinline__ uint32_t dlog_find(const uint32_t* PD, uint32_t bmask, uint32_t bshift, uint32_t pid) {
    const int i = tabset_index<8>(PD, bmask, bshift, pid);
    return i >= 0 ? PD[i + 4] : kDlogMiss;
}
// The same for the quotient form of the set: one 64-B record per bucket = its eight 16-bit entries (16 B), their eight
// distances (32 B), 16 B unused — a lookup touches one record per home bucket, the distance sits in the line the
// entries came from.  The set holds twice as many entries as it has dwords, so the distances are sorted through its LDS
// in two halves (each half re-reads the log, which is still in L2).
__device__ __forceinline__ void dlog_publish_q16(const DistLog& L, const Visited& v, uint32_t* out_pd) {
    const int lane = lane_id();
    wave_sync();
    uint4* t = reinterpret_cast<uint4*>(v.tab);
    uint4* o = reinterpret_cast<uint4*>(out_pd);
    const uint32_t nbuck = v.tmask + 1u, half = 4u * nbuck;          // entries per half = dwords of the set
    for (uint32_t i = lane; i < nbuck; i += 64) o[4u * i] = t[i];
    visited_drain();                                       // the log's own stores have landed before it is read back
    wave_sync();
    // log entries per lane in flight: one round trip per 1024 entries (a descent logs ~5k: three trips per half where eight
    // entries per lane over the whole log, twice, took twenty — a tenth of a descent's time with nothing else in flight)
    constexpr int kDeep = 16;
    for (uint32_t h = 0; h < 2u; h++) {
        const uint64_t* lg = L.log + (L.half ? h * L.half : 0u);
        const uint32_t cnt = L.half ? (h ? L.n_hi : L.n) : L.n;
        for (uint32_t base = 0; base < cnt; base += 64u * kDeep) {
            uint64_t e[kDeep];
#pragma unroll
            for (int u = 0; u < kDeep; u++) {
                const uint32_t i = base + 64u * (uint32_t)u + (uint32_t)lane;
                e[u] = lg[i < cnt ? i : 0u];
            }
#pragma unroll
            for (int u = 0; u < kDeep; u++) {
                const uint32_t idx = (uint32_t)e[u];
                if (base + 64u * (uint32_t)u + (uint32_t)lane < cnt && idx / half == h) v.tab[idx - h * half] = (uint32_t)(e[u] >> 32);
            }
        }
        wave_sync();
        // dword j of the set's LDS now is the distance of entry h * half + j: four of them = half a bucket
        for (uint32_t i = lane; i < nbuck; i += 64) {
            const uint32_t ent = h * half + 4u * i;                    // first entry of this group of four
            o[4u * (ent >> 3) + 1u + ((ent >> 2) & 1u)] = t[i];
        }
        wave_sync();
    }
}
__device__ __forceinline__ uint32_t dlog_find_q16(const uint32_t* PD, uint32_t ubits, uint32_t rbits, uint32_t pid) {
    const Q16Keys k = q16_keys(ubits, rbits, pid);
    // entries and distances of a record come in ONE round trip (three 16-B loads of one 64-B line), not entries first
    auto probe = [](const uint32_t* r, uint32_t t) -> uint32_t {
        const uint4 e = *reinterpret_cast<const uint4*>(r);
        const uint4 d0 = *reinterpret_cast<const uint4*>(r + 4), d1 = *reinterpret_cast<const uint4*>(r + 8);
        const int s = q16_slot(e, t);
        return s < 0 ? kDlogMiss : (s == 0 ? d0.x : (s == 1 ? d0.y : (s == 2 ? d0.z : (s == 3 ? d0.w : (s == 4 ? d1.x : (s == 5 ? d1.y : (s == 6 ? d1.z : d1.w)))))));
    };
    const uint32_t v1 = probe(PD + 16u * k.b1, k.t1);
    if (v1 != kDlogMiss) return v1;
    return probe(PD + 16u * k.b2, k.t2);
}

struct Counters {
    uint32_t n_dist, n_exp0, n_expU;
    uint32_t f_seen = 0, f_rej = 0;   // build descents: what the reject filter examined / turned down (idist_build_stats)
};

// ---------------------------------------------------------------------------
// Four waves per walk (narrow batches: Hnsw::search is one query per call, core/lib.rs:352-356).  A single wave
// walking the graph waits for two or three dependent HBM round trips per expansion and runs the FMA chains of ~50
// rows on one SIMD.  Here a workgroup of four waves — one per SIMD of a CU — owns the query: wave 0 (the leader)
// runs the walk exactly as the single-wave kernels do (pop, adjacency row, visited set, pushes in slot order), but
// every distance pass is shared: round r of 8 rows goes to wave r & 3, so the up to 64 rows of an expansion are
// requested at once and cost ONE round trip and a quarter of the arithmetic per SIMD.  Protocol, two workgroup
// barriers per expansion:
//     leader: act_pid[0..na) and ctl->na written | B1 | its share of the pass | B2 | reads act_dist, pushes, ...
//     helper:                         (waits)     | B1 | its share of the pass | B2 | (waits at the next B1)
// kQuadExit releases the helpers when the leader's work queue is empty.  Decisions, their order and the
// arithmetic per row are those of the other variants: results are bit-identical.
// ---------------------------------------------------------------------------
// Round 4 — the helpers also work AHEAD.  The leader's share of an expansion that is not a distance pass (pop, visited set,
// push: ~1.9 us of ~4.1) used to leave three SIMDs idle.  Now the leader names the candidate it expects to pop next (the first
// un-expanded entry behind the current one) and hands its adjacency row over; while the leader pushes, the helpers look that
// row's ids up in the visited set (read only), compact the new ones in slot order and compute their distances.  If the next
// pop is that candidate, the leader takes ids and distances as they are — nothing changed the visited set in between (ids
// enter it inside distance passes only), so they are exactly what its own look-up and pass would produce — inserts the ids,
// pushes in slot order, and no distance pass stands between two pushes.  Otherwise the results are ignored.  Speculation is
// only asked for while the on-chip set answers every look-up alone (no id in the bitmap: narrow batches at ef_search ~100
// never get there).
// Measured (make probe + scripts/probe_r04_quad.py, C3 index, one query per call, ef 100): the guess is asked for on 98 % of the
// expansions and is right on 67 % of them (the others pop a key the expansion itself just pushed); per expansion the leader
// spends ~0.3 us popping, ~1.0 on the adjacency row and the peek (a cold row on the 33 %), ~0.45 waiting for the helpers,
// ~0.55 inserting, ~0.85 in its own passes, ~1.0 pushing; the helpers need ~0.4 us for the look-up and 1.6-1.9 for the rows.
// The leader's serial chain bounds the walk, so what the protocol buys is small: 0.478 ms against 0.487 without it, same box
// (profiles/r04/probe_r04_quad_variants_same_box.jsonl).  Three richer protocols were built and measured on that box and dropped:
// knowing the next candidate before the merge (smallest new key against the first old un-expanded entry: every guess right,
// new candidates' rows requested during the push) costs the leader more per expansion than the saved waits return (0.497);
// adjacency two expansions ahead (0.485); helpers computing every slot of the row without looking at the set (0.494 in its
// own session, and the 100k x 128 build loses 3 % to the extra rows).
// Commands, each behind one workgroup barrier A (helpers loop: A, read command, act):
//     kQuadPass(na)  all four waves take their share of act_pid[0..na), then barrier B
//     kQuadSpec      helpers: speculate on ctl->spec_row (the row of ctl->spec_pid)
//     kQuadTake      nothing — the barrier itself is the hand-over: helpers arrive at A when their speculation is done
//     kQuadExit      helpers leave
// Measurement build (make probe): where the leader of a four-wave walk and its helpers spend a zero-layer walk — 10-ns ticks
// per segment, summed over all walks since the last reset (idist_probe_quad in idist_capi.hip).
#ifdef IDIST_PROBE
__device__ unsigned long long g_quad_probe[24];
#define QP_DECL unsigned long long qp[12] = {}; unsigned long long qp_t = wall_clock64();
#define QP_MARK(i) { const unsigned long long t_ = wall_clock64(); qp[i] += t_ - qp_t; qp_t = t_; }
#define QP_CNT(i, v) { qp[i] += (unsigned long long)(v); }
#define QP_FLUSH(on) { if ((on) && lane_id() == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_quad_probe[i_], qp[i_]); }
#else
#define QP_DECL
#define QP_MARK(i)
#define QP_CNT(i, v)
#define QP_FLUSH(on)
#endif
enum : uint32_t { kQuadPass = 0, kQuadSpec = 1, kQuadTake = 2, kQuadExit = 3 };
constexpr uint32_t kSpecAbort = 0xFFFFFFFFu;            // spec_n: the row holds an id only the bitmap can answer for
// The command word and the row count are kept TWICE, by the parity of the barrier they belong to: after barrier k the helpers
// read cmd[k & 1] while the leader may already be writing cmd[(k + 1) & 1] for its next command (nothing but that barrier
// stands between "the helpers were released" and "the leader posts again").  Every wave counts the A barriers it passed.
struct QuadCtl {
    uint32_t cmd[2], na[2], spec_n, pad[3];
    uint32_t spec_row[2][64];                            // leader -> helpers: the predicted candidate's adjacency row (slot order), kept
                                                         // by the parity of its barrier like cmd / na: after a wrong guess and an own pass
                                                         // of one round no barrier separates one speculation's row from the next one's
    uint32_t spec_new[64];                               // helpers -> leader: its new ids, compacted in slot order
    uint32_t spec_dist[64];                              // ... and their canonical distance bits
};
// the leader's end of the protocol: the control block and the number of A barriers passed so far (wave-uniform, in registers)
struct QuadLead {
    QuadCtl* ctl = nullptr;
    uint32_t seq = 0;
    __device__ __forceinline__ void post(uint32_t cmd, uint32_t na = 0u) {
        if (lane_id() == 0) { ctl->cmd[seq & 1u] = cmd; ctl->na[seq & 1u] = na; }
        block_sync();                                                  // A
        seq++;
    }
};
template <int NB, int RS, int TAIL, class Mid = NoMid>
__device__ __forceinline__ void quad_dist_pass(const IndexView& ix, const float* q, QuadLead& ql, const uint32_t* act_pid,
                                               uint32_t* act_dist, int na, Mid mid = Mid()) {
    if (na <= 8) {                                                     // a single round: not worth two barriers
        dist_rounds_quad<NB, RS, TAIL>(ix, q, act_pid, act_dist, na, 0, mid);
        return;
    }
    ql.post(kQuadPass, (uint32_t)na);
    dist_rounds_quad<NB, RS, TAIL>(ix, q, act_pid, act_dist, na, 0, mid);
    block_sync();                                                      // B
}
// leader: hand the predicted candidate's row to the helpers (row_id: this lane's slot of it, kInvalid beyond its end)
__device__ __forceinline__ void quad_post_spec(QuadLead& ql, uint32_t row_id) {
    ql.ctl->spec_row[ql.seq & 1u][lane_id()] = row_id;
    ql.post(kQuadSpec);
}
// leader: wait for the speculation it asked for; number of new ids in ctl->spec_new / spec_dist, or kSpecAbort
__device__ __forceinline__ uint32_t quad_take_spec(QuadLead& ql) {
    ql.post(kQuadTake);                                                // the helpers arrive at A when they are done
    return uniform_u32(ql.ctl->spec_n);
}
// the helpers' three-way split of a speculated list: wave wv (1..3) takes rounds wv - 1, wv + 2, ... of 8 rows
template <int NB, int RS, int TAIL>
__device__ __forceinline__ void dist_rounds_spec(const IndexView& ix, const float* q, const uint32_t* pids, uint32_t* dists, int na, int wv) {
    if constexpr (NB >= 0) {
        constexpr int RIF = NB <= 12 ? 3 : 1;
        dist_rounds_inflight<NB, RS, TAIL, RIF, true, 24>(ix, natural_view(q, NB), pids, dists, na, 8 * (wv - 1));
    } else {
        dist_rounds_inflight_rt<8, 2, 24>(ix, natural_view(q, (int)ix.nb), pids, dists, na, 8 * (wv - 1));
    }
}
template <int NB, int RS, int TAIL, int LAT>
__device__ __forceinline__ void quad_helper_loop(const IndexView& ix, const float* q, QuadCtl* ctl, const uint32_t* act_pid,
                                                 uint32_t* act_dist, int wv, const Visited& vis) {
    const int lane = lane_id();
    for (uint32_t seq = 0;; seq++) {
        block_sync();                                                  // A number `seq`
        const uint32_t cmd = uniform_u32(ctl->cmd[seq & 1u]);
        if (cmd == kQuadExit) break;
        if (cmd == kQuadPass) {
            dist_rounds_quad<NB, RS, TAIL>(ix, q, act_pid, act_dist, (int)uniform_u32(ctl->na[seq & 1u]), wv);
            block_sync();                                              // B
        } else if (cmd == kQuadSpec) {
#ifdef IDIST_PROBE
            const unsigned long long hp0 = wall_clock64();
#endif
            const uint32_t id = ctl->spec_row[seq & 1u][lane];
            const uint64_t inval = __ballot(id == kInvalid);
            const int nv = inval ? __builtin_ctzll(inval) : 64;
            bool fresh = false, other = false;                         // other: only the bitmap knows
            if (lane < nv && id < ix.n) {
                if constexpr (walk_vis16(LAT)) {
                    const int st = q16_lookup(vis, id);
                    fresh = st == kQRoom;
                    other = st == kQFull;
                } else {
                    fresh = !tab_find(vis, id);
                }
            }
            if (__ballot(other)) {
                if (wv == 1 && lane == 0) ctl->spec_n = kSpecAbort;
                continue;
            }
            const uint64_t fm = __ballot(fresh);
            if (fresh) ctl->spec_new[__popcll(fm & ((1ull << lane) - 1ull))] = id;   // (the same values from all three helpers)
            if (wv == 1 && lane == 0) ctl->spec_n = (uint32_t)__popcll(fm);
            wave_sync();
#ifdef IDIST_PROBE
            const unsigned long long hp1 = wall_clock64();
#endif
            dist_rounds_spec<NB, RS, TAIL>(ix, q, ctl->spec_new, ctl->spec_dist, __popcll(fm), wv);
#ifdef IDIST_PROBE
            if (lane == 0) {
                const unsigned long long hp2 = wall_clock64();
                atomicAdd(&g_quad_probe[12 + 3 * (wv - 1)], hp1 - hp0);
                atomicAdd(&g_quad_probe[13 + 3 * (wv - 1)], hp2 - hp1);
                atomicAdd(&g_quad_probe[14 + 3 * (wv - 1)], 1ull);
            }
#endif
        }
    }
}
__device__ __forceinline__ void quad_release_helpers(QuadLead& ql) { ql.post(kQuadExit); }   // A of the helpers' last iteration

// Search::push for the very first entry point (core/lib.rs:364, :444)
template <int NB, int RS, int TAIL>
__device__ __forceinline__ void push_entry(const IndexView& ix, const float* q, WState& st, Visited& vis,
                                           uint32_t* act_pid, uint32_t* act_dist, Counters& ctr, DistLog& dlog) {
    const int lane = lane_id();
    visited_begin(vis);
    if (lane == 0) { act_pid[0] = 0u; visited_mark(vis, 0u); }
    visited_added(vis, 1u);
    wave_sync();
    dist_rounds<NB, RS, TAIL>(ix, q, act_pid, act_dist, 1);
    wave_sync();
    if (lane == 0) st.W[0] = ((uint64_t)act_dist[0] << 32);  // pid 0
    if (dlog.log) dlog_append(dlog, lane == 0 ? vis_index(vis, 0u) : -1, act_dist[0]);
    wave_sync();
    st.plen = 1;
    st.cursor = 0;
    ctr.n_dist += 1;
}

// ---------------------------------------------------------------------------
// Search::search (core/lib.rs:598-614) on one layer.  rows/row_stride: the
// adjacency array (UpperNode: 32, ZeroNode: 64); links: `.take(links)`.
// ---------------------------------------------------------------------------
// first un-expanded entry after index `after`, -1 if none (does not move the cursor)
__device__ __forceinline__ int w_peek_next(const WState& st, int after) {
    const int lane = lane_id();
    for (int i0 = after + 1; i0 < st.plen; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t m = __ballot(i < st.plen && !(st.W[i] & kFlag));
        if (m) return i0 + __builtin_ctzll(m);
    }
    return -1;
}

// The insertion step shared by both variants: lanes holding a fresh neighbour (key = dist<<32|pid) push it in
// slot order, core/lib.rs:606-608 + :712-719.  `push` accepts key i iff fewer than ef entries of `nearest` —
// the list as of that moment, i.e. the old one plus the keys accepted earlier in this expansion — are smaller;
// nothing is truncated before the expansion ends (:612).  So acceptance is decided from the rank in the old
// list plus a count over the earlier accepted lanes, and the accepted keys are merged into W in one pass
// instead of one shifted insertion each.
constexpr int kPushChunks = 8;   // one-pass merge for W up to 512 entries, sequential insertion beyond
// One pass over the candidate keys in slot order with the old list held in registers (NC chunks of 64 entries, entry
// c * 64 + lane in w[c]); no LDS access inside the loop:
//   r0   rank of the key in the OLD list (Vec::binary_search, :712) = number of old entries below it, counted with one
//        ballot per chunk;
//   cnt  per lane: accepted keys so far that are smaller than the lane's own key.  When slot j is decided it holds
//        exactly the earlier-slot keys in front of key j; after the loop, all accepted keys in front of the lane's key;
//   sh   per old entry: accepted keys below it (entry t lies above key j iff t >= r0_j) = how far it moves up.
template <int NC>
__device__ __forceinline__ void w_push_merge(WState& st, uint64_t key, uint64_t pm) {
    const int lane = lane_id();
    const int plen = st.plen;
    uint64_t w[NC];
    int sh[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int t = c * 64 + lane;
        sh[c] = 0;
        w[c] = t < plen ? st.W[t] : ~0ull;                 // beyond the list: above every key
    }
    const uint32_t k_lo = (uint32_t)key, k_hi = (uint32_t)(key >> 32);
    uint32_t cnt = 0;
    bool acc = false;
    int my_r0 = 0, first = plen, nacc = 0;
    for (uint64_t m = pm; m; m &= m - 1ull) {              // slot order
        const int j = __builtin_ctzll(m);
        const uint64_t kj = ((uint64_t)readlane_u32(k_hi, j) << 32) | readlane_u32(k_lo, j);
        int r0 = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) r0 += __popcll(__ballot((w[c] & kKeyMask) < kj));
        const int cj = (int)readlane_u32(cnt, j);
        if (r0 + cj < st.ef) {                             // idx < ef, :713
            cnt += kj < key ? 1u : 0u;
#pragma unroll
            for (int c = 0; c < NC; c++) sh[c] += c * 64 + lane >= r0 ? 1 : 0;
            if (lane == j) { acc = true; my_r0 = r0; }
            first = r0 < first ? r0 : first;
            nacc++;
        }
    }
    if (!nacc) return;
    // merge: an old entry moves up by the number of accepted keys below it, an accepted key lands at its old
    // rank plus the number of accepted keys below it
    wave_sync();
#pragma unroll
    for (int c = 0; c < NC; c++)
        if (sh[c] > 0 && c * 64 + lane < plen) st.W[c * 64 + lane + sh[c]] = w[c];
    if (acc) st.W[my_r0 + (int)cnt] = key;
    wave_sync();
    st.plen = plen + nacc;
    if (first < st.cursor) st.cursor = first;
}
// MAXC: chunks of 64 entries the one-pass merge may hold in registers (two dwords + a counter per chunk and lane).  The fat
// on-chip waves (512 registers) take 16 — a `nearest` of up to 1024 entries, i.e. ef_search up to ~900, is merged in one pass
// instead of one ranked, shifted insertion per accepted key (at ef_search 800 that was most of an expansion's time).
template <int MAXC = kPushChunks>
__device__ __forceinline__ void w_push_keys(WState& st, uint64_t key, bool has) {
    // entries that cannot have rank < ef even now never will (W only improves)
    const uint64_t thr = st.plen >= st.ef ? (st.ef ? (st.W[st.ef - 1] & kKeyMask) : 0ull) : kMaxKey + 1ull;
    const bool cand = has && key < thr;
    uint64_t pm = __ballot(cand);
    if (!pm) return;
    const int plen = st.plen;
    if (plen <= 64) return w_push_merge<1>(st, key, pm);
    if (plen <= 128) return w_push_merge<2>(st, key, pm);
    if (plen <= 256) return w_push_merge<4>(st, key, pm);
    if (plen <= 64 * kPushChunks) return w_push_merge<kPushChunks>(st, key, pm);
    if constexpr (MAXC >= 16) { if (plen <= 64 * 16) return w_push_merge<16>(st, key, pm); }
    while (pm) {
        const int i = __builtin_ctzll(pm);
        pm &= pm - 1ull;
        const uint64_t k = bcast_u64(key, i);
        const int idx = w_rank(st, k);             // :712
        if (idx < st.ef) w_insert(st, idx, k);     // :713-719
    }
}

// LAT = kWalkClassic: one thing at a time per wave (many waves per CU hide the latencies).
// LAT = kWalkLatency / kWalkOverlap: the same decisions in the same order, but the dependent HBM round trips
//          of one expansion are overlapped:
//   * the adjacency row of the candidate most likely to be expanded next is requested one expansion ahead;
//   * neighbours the Bloom filter proves new go straight to the distance rounds while the visited bytes of
//     the "maybe" ones are still in flight (those that turn out new get a second, usually empty, pass);
//   * all rounds of a pass are requested together (dist_rounds_inflight).
template <int WALK> constexpr int push_chunks() { return walk_vis_lds(WALK) && (walk_waves(WALK) == 1 || walk_waves(WALK) == 2) && !walk_quad(WALK) ? 16 : kPushChunks; }
template <int NB, int RS, int TAIL, int LAT = 0>
__device__ __forceinline__ void search_layer(const IndexView& ix, const uint32_t* rows, int row_stride, int links,
                                             const float* q, WState& st, Visited& vis, uint32_t* act_pid,
                                             uint32_t* act_dist, Counters& ctr, bool is_zero, DistLog& dlog,
                                             QuadLead* quad = nullptr,
                                             const FilterQFor<NB, RS, TAIL, LAT>& fq = FilterQFor<NB, RS, TAIL, LAT>()) {
    const int lane = lane_id();
    uint32_t guard = 0;
    const bool row_lane = lane < row_stride && lane < links;
    constexpr bool PFA = walk_mode(LAT) != kWalkClassic;   // adjacency requested one expansion ahead
    // visited test-and-set in flight during the first distance pass (bitmap + Bloom filter only: the on-chip set answers at once)
    constexpr bool OVL = walk_mode(LAT) != kWalkClassic && !walk_vis_lds(LAT);
    uint32_t pf_pid = kInvalid, pf_row = kInvalid;
    // four-wave walk with the visited set on chip: the helpers work one expansion ahead (QuadCtl)
    // the helpers work one expansion ahead — for the compile-time row geometries only: with the runtime-geometry tile the work-ahead
    // COSTS a scalar call 9-26 % (1M x 200 / 384-d, 500k x 1024-d: 0.526 / 0.656 / 2.40 ms without it against 0.576 / 0.780 / 3.23 ms
    // with it; 300-d: 0.478 with, 0.479 without — profiles/r05/probe_r05l_single_query_rt_rows_ab.jsonl; round 4 measured it at 300-d and
    // 768-d only)
    constexpr bool kSpec = walk_quad(LAT) && walk_vis_lds(LAT) && PFA && NB >= 0;
    [[maybe_unused]] uint32_t sq_pid = kInvalid;          // the candidate whose row the helpers were given (wave-uniform)
    [[maybe_unused]] bool sq_off = false;                 // this layer met an id only the bitmap answers for: no more guesses
    QP_DECL
    for (;;) {
        int ci = w_pop(st);                               // :599-604
        if (ci < 0 && st.spill_n && w_refill_ties(st)) ci = w_pop(st);   // live ties that did not fit the LDS region
        if (ci < 0) break;
        const uint64_t c = st.W[ci];
        wave_sync();
        if (lane == 0) st.W[ci] = c | kFlag;
        const uint32_t cpid = (uint32_t)c;
        if (is_zero) ctr.n_exp0++; else ctr.n_expU++;
        QP_MARK(0) QP_CNT(6, 1)

        // layer.nearest_iter(pid).take(links): stop at first INVALID (core/types.rs:183-187)
        uint32_t nb_pid = kInvalid;
        if constexpr (PFA) {
            if (cpid == pf_pid) nb_pid = pf_row;
            else if (row_lane) nb_pid = rows[(size_t)cpid * row_stride + lane];
            const int c2 = w_peek_next(st, ci);
            pf_pid = c2 >= 0 ? (uint32_t)st.W[c2] : kInvalid;
            pf_row = kInvalid;
            if (pf_pid != kInvalid && row_lane) pf_row = rows[(size_t)pf_pid * row_stride + lane];
        } else {
            if (row_lane) nb_pid = rows[(size_t)cpid * row_stride + lane];
        }
        const uint64_t inval = __ballot(nb_pid == kInvalid);
        const int nvalid = inval ? __builtin_ctzll(inval) : 64;
        const bool is_nb = lane < nvalid;

        // What the branches below hand to the common tail: this lane's key (slot order = lane order), whether it is a new id,
        // how many there are, and (build) where the id sits in the on-chip set.
        uint64_t key = kMaxKey;
        bool fresh = false;
        int na = 0, tab_idx = -1;
        uint32_t my_d = 0, my_id = nb_pid;
        // reject filter: the furthest distance of a full `nearest` as this expansion begins (`nearest` only improves during it)
        [[maybe_unused]] uint32_t thr_bits = 0xFFFFFFFFu;
        if constexpr (walk_ea(LAT) > 0 || walk_filter(LAT)) {
            if (st.ef > 0 && st.plen >= st.ef && (walk_filter(LAT) || !dlog.log)) thr_bits = (uint32_t)((st.W[st.ef - 1] & kKeyMask) >> 32);
        }
        if (is_nb && nb_pid >= ix.n) st.status |= kStBadRow;
        const bool ok_nb = is_nb && nb_pid < ix.n;

        // four-wave walk: did the helpers work this expansion out in advance (QuadCtl)?
        [[maybe_unused]] bool took = false;
        QP_MARK(1)
        if constexpr (kSpec) {
            if (sq_pid != kInvalid && sq_pid == cpid) {
                const uint32_t ns = quad_take_spec(*quad);
                QP_MARK(2)
                if (ns != kSpecAbort) {
                    QP_CNT(7, 1)
                    took = true;
                    na = (int)ns;
                    fresh = lane < na;
                    my_id = fresh ? quad->ctl->spec_new[lane] : kInvalid;
                    my_d = fresh ? quad->ctl->spec_dist[lane] : 0u;
                    // the inserts a distance pass would have made while its rows were in flight
                    if constexpr (walk_vis16(LAT)) {
                        if (fresh && q16_insert(vis, my_id, tab_idx) == kQFull) {       // filled up by this very expansion
                            atomicOr(&vis.bits[my_id >> 5], 1u << (my_id & 31u));
                            visited_note(vis, my_id);
                        }
                    } else {
                        visited_begin(vis);
                        if (fresh) tab_idx = tab_insert(vis, my_id);
                        visited_added(vis, (uint32_t)na);
                    }
                } else {
                    sq_off = true;                                    // an id of the bitmap class: no more guesses on this layer
                    QP_CNT(8, 1)
                }
            }
            sq_pid = kInvalid;
        }
        QP_MARK(11)

        if (took) {
            // (nothing: ids and distances came from the helpers)
        } else if constexpr (walk_vis16(LAT)) {
            // visited.insert(pid), core/lib.rs:705 / core/types.rs:32-40, on the quotient set: one LDS round trip tells
            // every neighbour apart — known / surely new (a home bucket has room: it never went to the bitmap) / both
            // home buckets full (the bitmap decides).  The surely new ones go to the distance pass at once and enter the
            // set while their rows are in flight; the test-and-set of the others is in flight during that pass, and
            // those that turn out new get a second pass.  Keys are pushed in slot order afterwards, whichever pass
            // computed them.
            int stt = kQFound;
            uint32_t vold = 0;
            const uint32_t vbit = 1u << (nb_pid & 31u);
            if (ok_nb) {
                stt = q16_lookup(vis, nb_pid);
                if (stt == kQFull) vold = atomicOr(&vis.bits[nb_pid >> 5], vbit);
            }
            const bool sure = ok_nb && stt == kQRoom, maybe = ok_nb && stt == kQFull;
            if constexpr (kSpec) { if (__ballot(maybe)) sq_off = true; }
            const uint64_t sm = __ballot(sure);
            wave_sync();
            if (sm) {
                const int my = __popcll(sm & ((1ull << lane) - 1ull));
                if (sure) act_pid[my] = nb_pid;                                         // keeps slot order
                wave_sync();
                auto mid = [&]() {
                    if (sure && q16_insert(vis, nb_pid, tab_idx) == kQFull) {           // filled up by this very expansion
                        atomicOr(&vis.bits[nb_pid >> 5], vbit);
                        visited_note(vis, nb_pid);
                    }
                };
                if constexpr (walk_quad(LAT)) quad_dist_pass<NB, RS, TAIL>(ix, q, *quad, act_pid, act_dist, __popcll(sm), mid);
                else dist_pass_filtered<NB, RS, TAIL, LAT>(ix, q, fq, act_pid, act_dist, __popcll(sm), mid, thr_bits);
                wave_sync();
                if (sure) my_d = act_dist[my];
            }
            const bool late = maybe && (vold & vbit) == 0u;
            if (late) visited_note(vis, nb_pid);
            const uint64_t lm = __ballot(late);
            wave_sync();
            if (lm) {
                const int my = __popcll(lm & ((1ull << lane) - 1ull));
                if (late) act_pid[my] = nb_pid;
                wave_sync();
                if constexpr (walk_quad(LAT)) quad_dist_pass<NB, RS, TAIL>(ix, q, *quad, act_pid, act_dist, __popcll(lm));
                else dist_pass_filtered<NB, RS, TAIL, LAT>(ix, q, fq, act_pid, act_dist, __popcll(lm), NoMid(), thr_bits);
                wave_sync();
                if (late) my_d = act_dist[my];
            }
            fresh = sure || late;                                                       // (ids that went to the bitmap have no index)
            na = __popcll(sm) + __popcll(lm);
        } else if constexpr (!OVL) {
            // visited.insert(pid), core/lib.rs:705 / core/types.rs:32-40
            visited_begin(vis);
            // While the on-chip set takes new ids, an expansion only LOOKS its neighbours up here; the new ones are
            // inserted while their rows are in flight (`mid` below) — rows never hold duplicates (validated on import,
            // impossible in a built graph), so "not in the set" is final.  Every other configuration inserts at once.
            const bool defer = vis.tab != nullptr && !vis.spill;
            if (ok_nb) {
                if (defer) fresh = !tab_find(vis, nb_pid);
                else fresh = visited_insert(vis, nb_pid, tab_idx);
            }
            const uint64_t fm = __ballot(fresh);
            na = __popcll(fm);
            visited_added(vis, (uint32_t)na);
            wave_sync();
            if (na) {
                const int my = __popcll(fm & ((1ull << lane) - 1ull));
                if (fresh) act_pid[my] = nb_pid;                                        // keeps slot order
                wave_sync();
                auto mid = [&]() { if (defer && fresh) tab_idx = tab_insert(vis, nb_pid); };
                // early abandon (measurement builds): the furthest distance of a full `nearest` as this expansion begins
                if constexpr (walk_quad(LAT)) quad_dist_pass<NB, RS, TAIL>(ix, q, *quad, act_pid, act_dist, na, mid);
                else dist_pass_filtered<NB, RS, TAIL, LAT>(ix, q, fq, act_pid, act_dist, na, mid, thr_bits);     // :709-710
                wave_sync();
                if (fresh) my_d = act_dist[my];
            }
        } else {
            bool sure = false, maybe = false;
            uint32_t vold = 0;
            const uint32_t vbit = 1u << (nb_pid & 31u);
            if (ok_nb) {
                if (vis.bloom && !bloom_maybe(vis, nb_pid)) sure = true;
                else { maybe = true; vold = atomicOr(&vis.bits[nb_pid >> 5], vbit); }   // test-and-set, in flight during the first pass
            }
            if (sure) visited_mark(vis, nb_pid);
            const uint64_t sm = __ballot(sure);
            wave_sync();
            if (sm) {
                const int my = __popcll(sm & ((1ull << lane) - 1ull));
                if (sure) act_pid[my] = nb_pid;
                wave_sync();
                dist_rounds_walk<NB, RS, TAIL, LAT>(ix, q, act_pid, act_dist, __popcll(sm));
                wave_sync();
                if (sure) my_d = act_dist[my];
            }
            const bool late = maybe && (vold & vbit) == 0u;
            if (late) visited_note(vis, nb_pid);
            const uint64_t lm = __ballot(late);
            wave_sync();
            if (lm) {
                const int my = __popcll(lm & ((1ull << lane) - 1ull));
                if (late) act_pid[my] = nb_pid;
                wave_sync();
                dist_rounds_walk<NB, RS, TAIL, LAT>(ix, q, act_pid, act_dist, __popcll(lm));
                wave_sync();
                if (late) my_d = act_dist[my];
            }
            fresh = sure || late;
            na = __popcll(sm) + __popcll(lm);
            if (dlog.log && fresh) tab_idx = vis_index(vis, nb_pid);
        }

        QP_MARK(3)
        // four-wave walk: hand the row of the candidate expected next to the helpers — they work on it while this wave pushes
        if constexpr (kSpec) {
            bool ask = !sq_off && pf_pid != kInvalid;
            if constexpr (!walk_vis16(LAT)) ask = ask && vis.tab != nullptr && !vis.spill && vis.count + 128u <= vis.tlimit;
            if (ask) {
                quad_post_spec(*quad, pf_row);
                sq_pid = pf_pid;
                QP_CNT(9, 1)
            }
        }
        QP_MARK(4)

        if (na) {                                                                       // Search::push in slot order, :606-608
            ctr.n_dist += (uint32_t)na;
            if (fresh) key = ((uint64_t)my_d << 32) | my_id;
            if (dlog.log) {
                // (a candidate the filter turned down is logged in bound form: beyond this expansion's threshold)
                uint32_t log_d = my_d;
                if constexpr (walk_filter(LAT)) { if (my_d == kAbandoned) log_d = kDlogBound | thr_bits; }
                dlog_append(dlog, fresh ? tab_idx : -1, log_d);
            }
            // (thin walks on long compile-time rows — the 768-d build descents — merge `nearest` in eight register chunks: W of up to
            //  512 entries in one pass, ef_construction far below; the sixteen-chunk merge's registers are the whole f32 row's)
            constexpr int kPC = (walk_is_thin(LAT) && NB > 12) ? kPushChunks : push_chunks<LAT>();
            w_push_keys<kPC>(st, key, fresh);
        }
        w_truncate(st);                                    // :612
        QP_MARK(5) QP_CNT(10, na)
        if (++guard > ix.n + 64u) { st.status |= kStGuard; break; }
    }
    QP_FLUSH(kSpec && is_zero)
}

// n_dist / n_rows: work as executed here.  n_ref: distance calls as the REFERENCE makes them — `any` stops at the first
// closer member (core/lib.rs:676-679) — kept by the reference-order kernels only (idist_build_stats.n_heur_ref)
struct HeurCounters { uint32_t n_dist, n_rows; uint32_t n_ref = 0; };

// ---------------------------------------------------------------------------
// Search::select_heuristic with extend_candidates = false (core/lib.rs:636-698), selected rows kept on
// chip.  Input: the first `nw` entries of Wsrc (sorted, = Search.nearest).  Output: sel[0..return) =
// selected-then-backfilled keys (NOT re-sorted); n_selected = the split point;
// out_aux[i] = pruner pid of back-filled entry i (0 for selected ones).
// Tile layout (LDS): blk[t][slot][32] for the full 128-B blocks, rem[slot][32]
// for the natural-order remainder.  Slots [0, rt) hold R (the selected set, in
// selection order), slots [rt, rt+fc) stage the next fc candidates, fetched
// together so their HBM latencies overlap.  Consecutive slots are 32 dwords
// apart => the 8 row-groups of a ds_read_b128 round hit disjoint banks.
// Selected rows beyond rt are compared through the global gather path.
// ---------------------------------------------------------------------------
struct Tile {
    float* blk;
    float* rem;
    int slots;   // rt + fc
    int rt, fc;
};
__device__ __forceinline__ VecView tile_view(const Tile& t, int slot) {
    return VecView{t.blk + slot * 32, t.slots * 32, t.rem + slot * 32};
}
__host__ __device__ inline size_t tile_floats(uint32_t nb, uint32_t slots) { return (size_t)(nb + 1) * slots * 32; }

// float4 #f of a stored row -> its place in slot `slot` (branch-free: rem == blk + nb*slots*32)
__device__ __forceinline__ float* tile_addr(const Tile& t, int nb, int slot, int f) {
    const int off_blk = (((f >> 3) * t.slots + slot) << 5) + ((f & 7) << 2);
    const int off_rem = ((nb * t.slots + slot) << 5) + ((f - 8 * nb) << 2);
    return t.blk + (f < 8 * nb ? off_blk : off_rem);
}

// (kept out of line: inlined into the selection loop hipcc demotes the staging registers to scratch)
template <int NB, int RS, int TAIL>
__device__ __attribute__((noinline)) void tile_stage(const float* __restrict__ points, uint32_t stride, uint32_t nb_rt,
                                                     float* tblk, int tslots, int first_slot, const uint32_t* pids, int cnt) {
    const int lane = lane_id();
    const int nb = NB >= 0 ? NB : (int)nb_rt;
    const int nf4 = (int)stride >> 2;
    Tile t;
    t.blk = tblk;
    t.rem = tblk + (size_t)nb * tslots * 32;
    t.slots = tslots;
    t.rt = first_slot;
    t.fc = 8;
    struct { const float* points; uint32_t stride; } ix{points, stride};
    if constexpr (NB >= 0) {
        constexpr int USED = 32 * NB + 8 * RS + 4 * TAIL;
        constexpr int STRIDE = USED <= 16 ? 16 : ((USED + 15) & ~15);
        constexpr int NF = (STRIDE / 4 + 63) / 64;
        constexpr int FC = 8;
        float4 v[FC][NF];
        // branch-free: slots past cnt re-fetch the last candidate (never read), so every load of the
        // chunk is in flight before the first LDS store
#pragma unroll
        for (int u = 0; u < FC; u++) {
            const int uu = u < cnt ? u : cnt - 1;
            const float* row = ix.points + (size_t)pids[uu] * ix.stride;
#pragma unroll
            for (int k = 0; k < NF; k++) {
                const int f = lane + 64 * k;
                v[u][k] = *reinterpret_cast<const float4*>(row + 4 * (f < STRIDE / 4 ? f : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < FC; u++) {
#pragma unroll
            for (int k = 0; k < NF; k++) {
                const int f = lane + 64 * k;
                if (f < STRIDE / 4) *reinterpret_cast<float4*>(tile_addr(t, NB, first_slot + u, f)) = v[u][k];
            }
        }
    } else {
        // any dimension: the same 16 B of all (up to eight) candidates are requested together, 64 lanes x 8 rows = 8 KB per trip
        const float* row[8];
#pragma unroll
        for (int u = 0; u < 8; u++) row[u] = ix.points + (size_t)pids[u < cnt ? u : cnt - 1] * ix.stride;
        for (int f = lane; f < nf4; f += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(row[u] + 4 * f);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (u < cnt) *reinterpret_cast<float4*>(tile_addr(t, nb, first_slot + u, f)) = v[u];
        }
    }
}

__device__ __forceinline__ void tile_copy(const IndexView& ix, const Tile& t, int nb, int dst, int src) {
    const int lane = lane_id();
    const int nf4 = (int)ix.stride >> 2;
    for (int f = lane; f < nf4; f += 64)
        *reinterpret_cast<float4*>(tile_addr(t, nb, dst, f)) = *reinterpret_cast<const float4*>(tile_addr(t, nb, src, f));
}

// any R[b..b+cnt) (slots b..) closer to the candidate in slot `cslot` than cd?  (core/lib.rs:676-679)
template <int NB, int RS, int TAIL>
__device__ __forceinline__ uint64_t tile_any_closer(const IndexView& ix, const Tile& t, int cslot, int b, int cnt, uint32_t cd) {
    const int lane = lane_id();
    const int g = lane >> 3, j = lane & 7;
    const int nb = NB >= 0 ? NB : (int)ix.nb;
    const int rs = RS >= 0 ? RS : (int)ix.rs;
    const int tail = TAIL >= 0 ? TAIL : (int)ix.tail;
    const bool on = g < cnt;
    const int rslot = on ? b + g : cslot;
    const float* cb = t.blk + (cslot << 5) + (j << 2);
    const float* rb = t.blk + (rslot << 5) + (j << 2);
    const int bs = t.slots << 5;
    float acc = 0.0f;
    if constexpr (NB >= 0) {
#pragma unroll
        for (int tt = 0; tt < NB; tt++) {
            const float4 w = *reinterpret_cast<const float4*>(cb + tt * bs);
            const float4 p = *reinterpret_cast<const float4*>(rb + tt * bs);
            float d;
            d = w.x - p.x; acc = __builtin_fmaf(d, d, acc);
            d = w.y - p.y; acc = __builtin_fmaf(d, d, acc);
            d = w.z - p.z; acc = __builtin_fmaf(d, d, acc);
            d = w.w - p.w; acc = __builtin_fmaf(d, d, acc);
        }
    } else {
        for (int tt = 0; tt < nb; tt++) {
            const float4 w = *reinterpret_cast<const float4*>(cb + tt * bs);
            const float4 p = *reinterpret_cast<const float4*>(rb + tt * bs);
            float d;
            d = w.x - p.x; acc = __builtin_fmaf(d, d, acc);
            d = w.y - p.y; acc = __builtin_fmaf(d, d, acc);
            d = w.z - p.z; acc = __builtin_fmaf(d, d, acc);
            d = w.w - p.w; acc = __builtin_fmaf(d, d, acc);
        }
    }
    const float* cr = t.rem + (cslot << 5);
    const float* rr = t.rem + (rslot << 5);
    for (int c = 0; c < rs; c++) {
        const float d = cr[c * 8 + j] - rr[c * 8 + j];
        acc = __builtin_fmaf(d, d, acc);
    }
    const float r = fold_chains(acc, tail != 0, tail ? cr[rs * 8 + (j & 3)] : 0.0f, tail ? rr[rs * 8 + (j & 3)] : 0.0f);
    const bool closer = on && j == 0 && canon_bits(r, ix.metric) < cd;   // strict <, core/lib.rs:678
    return __ballot(closer);   // bit 8*g set <=> R[b+g] is closer
}

template <int NB, int RS, int TAIL>
__device__ __forceinline__ int select_heuristic_tiled(const IndexView& ix, const uint64_t* Wsrc, int nw, bool keep_pruned,
                                                      const Tile& t, uint64_t* sel, uint64_t* disc, uint32_t* act_pid,
                                                      uint32_t* act_dist, HeurCounters& hc, int& n_selected,
                                                      uint32_t* dprn, uint32_t* out_aux) {
    const int lane = lane_id();
    const int nb = NB >= 0 ? NB : (int)ix.nb;
    int nsel = 0, ndis = 0;
    for (int w0 = 0; w0 < nw && nsel < kM2; w0 += t.fc) {           // core/lib.rs:668-671
        const int cnt = nw - w0 < t.fc ? nw - w0 : t.fc;
        if (lane < cnt) act_pid[lane] = (uint32_t)Wsrc[w0 + lane];
        wave_sync();
        tile_stage<NB, RS, TAIL>(ix.points, ix.stride, ix.nb, t.blk, t.slots, t.rt, act_pid, cnt);   // points[candidate.pid], :675
        hc.n_rows += (uint32_t)cnt;
        wave_sync();
        for (int u = 0; u < cnt && nsel < kM2; u++) {
            const uint64_t c = Wsrc[w0 + u] & kKeyMask;
            const uint32_t cd = (uint32_t)(c >> 32);
            const int cslot = t.rt + u;
            bool pruned = false;
            uint32_t pr_pid = 0;
            const int nl = nsel < t.rt ? nsel : t.rt;                // `any`, :676-679, early exit per 8
            for (int b = 0; b < nl && !pruned; b += 8) {
                const int c8 = nl - b < 8 ? nl - b : 8;
                const uint64_t cm = tile_any_closer<NB, RS, TAIL>(ix, t, cslot, b, c8, cd);
                pruned = cm != 0ull;
                if (pruned) pr_pid = (uint32_t)sel[b + (__builtin_ctzll(cm) >> 3)];
                hc.n_dist += (uint32_t)c8;
                hc.n_ref += pruned ? (uint32_t)(__builtin_ctzll(cm) >> 3) + 1u : (uint32_t)c8;
            }
            for (int b = t.rt; b < nsel && !pruned; b += 8) {        // selected rows that did not fit on chip
                const int c8 = nsel - b < 8 ? nsel - b : 8;
                wave_sync();
                if (lane < c8) act_pid[8 + lane] = (uint32_t)sel[b + lane];
                wave_sync();
                dist_rounds<NB, RS, TAIL>(ix, tile_view(t, cslot), act_pid + 8, act_dist, c8);
                wave_sync();
                hc.n_dist += (uint32_t)c8;
                const uint64_t cm = __ballot(lane < c8 && act_dist[lane] < cd);
                pruned = cm != 0ull;
                if (pruned) pr_pid = (uint32_t)sel[b + __builtin_ctzll(cm)];
                hc.n_ref += pruned ? (uint32_t)__builtin_ctzll(cm) + 1u : (uint32_t)c8;
            }
            if (!pruned) {                                           // :681-684
                if (lane == 0) sel[nsel] = c;
                if (nsel < t.rt) tile_copy(ix, t, nb, nsel, cslot);
                nsel++;
            } else {
                if (lane == 0 && ndis < kM2) { disc[ndis] = c; dprn[ndis] = pr_pid; }
                ndis++;
            }
            wave_sync();
        }
    }
    n_selected = nsel;
    out_aux[lane] = 0u;
    wave_sync();
    if (keep_pruned) {                                               // :687-695
        if (ndis > kM2) ndis = kM2;
        int take = kM2 - nsel;
        if (take > ndis) take = ndis;
        if (lane < take) { sel[nsel + lane] = disc[lane]; out_aux[nsel + lane] = dprn[lane]; }
        if (take > 0) nsel += take;
        wave_sync();
    }
    return nsel;
}

}  // namespace idist
