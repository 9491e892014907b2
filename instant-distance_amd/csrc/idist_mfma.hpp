// idist_mfma.hpp — the wide-batch distance path: -2*Q*P^T on the f32 matrix cores.
//
// BASELINE config C4 / north star: "the -2QP^T term taken to MFMA only when the candidate
// batch is wide enough to be a real dense contraction".  A (queries x all points) scan IS a
// dense contraction; the graph search's <= 64-row expansions are not, and they also define the
// result order, which needs the canonical FMA-chain distance — so MFMA is used
// here only as an exact-recall FILTER:
//
//   1. d~(q,p) = |q|^2 + |p|^2 - 2 q.p  by v_mfma_f32_32x32x2_f32 tiles (exact f32, k-ordered
//      fma chain; rounds differently from sum (a-b)^2, so it never decides an order),
//   2. a sample pass over the first S points gives each query a threshold (k-th smallest d~),
//   3. the full pass appends every point with d~ <= threshold + slack to the query's candidate list,
//   4. the candidates (a superset of the true top-k, typically k*n/S + k of them) are re-ranked
//      with the canonical distance by the same top-k machinery as the scan kernel.
//
// The result is bit-identical to bruteforce_kernel's (tests/test_parity.py::test_bruteforce_mfma*).
#pragma once
#include "idist_device.hpp"
#include "idist_kernels.hpp"

namespace idist {

#ifndef IDIST_EMU
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

// tile, K chunk, LDS pitch: 36 floats = 16-B aligned rows, and 36 = 4 * 9 with 9 odd, so 16 lanes reading float4s of rows
// that differ mod 16 hit 64 distinct banks (ds_read_b128 / ds_write_b128 conflict-free)
#ifndef IDIST_MFMA_KC
#define IDIST_MFMA_KC 32
#endif
constexpr int kTM = 128, kTN = 128, kKC = IDIST_MFMA_KC, kLDP = kKC + 4;   // (36 and 68 = 4 * odd)
constexpr int kKF4 = kKC / 4;                         // float4s per row and chunk
constexpr int kStageF4 = kTM * kKF4 / 256;            // float4s a thread stages per matrix and chunk

// |row|^2 in storage order (one wave per row)
__global__ __launch_bounds__(64) void row_norms_kernel(const float* __restrict__ rows, uint32_t n, uint32_t stride,
                                                      float* __restrict__ out) {
    const int lane = lane_id();
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float* row = rows + (size_t)r * stride;
        float acc = 0.0f;
        for (uint32_t o = lane * 4; o < stride; o += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + o);
            acc = __builtin_fmaf(v.x, v.x, acc);
            acc = __builtin_fmaf(v.y, v.y, acc);
            acc = __builtin_fmaf(v.z, v.z, acc);
            acc = __builtin_fmaf(v.w, v.w, acc);
        }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) out[r] = acc;
    }
}

struct MfmaArgs {
    const float* Q;        // [nq_pad][stride] blocked rows (zero rows past nq)
    const float* P;        // [n][stride] blocked rows
    const float* qn;       // [nq_pad]
    const float* pn;       // [n]
    uint32_t nq, n, stride;
    uint32_t p_begin, p_end;   // point range of this launch
    int mode;              // 0: dense output, 1: threshold filter
    float* dense;          // [nq_pad][dense_ld]  (mode 0), column = pid - p_begin
    uint32_t dense_ld;
    const float* thr;      // [nq_pad] (mode 1)
    uint32_t* cand;        // [nq][cap] (mode 1)
    uint32_t* cnt;         // [nq]
    uint32_t cap;
};

// 128 x 128 tile per 256-thread workgroup; wave w computes the 64 x 64 quadrant (w>>1, w&1)
// as 2 x 2 blocks of v_mfma_f32_32x32x2_f32.
__global__ __launch_bounds__(256) void mfma_dist_kernel(MfmaArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);
    float* Bs = As + kTM * kLDP;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const uint32_t nqt = (a.nq + kTM - 1) / kTM;
    const uint32_t q0 = (blockIdx.x % nqt) * kTM;                     // query tiles fastest: neighbours share the P tile in L2
    const uint32_t p0 = a.p_begin + (blockIdx.x / nqt) * kTN;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // K loop: one kKC-float chunk of the 128 query rows and the 128 point rows per iteration.  The next chunk is requested
    // (global -> registers) before the current one's MFMAs (2 * kKC per wave), operands move as float4s, and inside a
    // chunk instruction #j contracts elements j (lanes 0-31) and kKC / 2 + j (lanes 32-63), so a lane's operands of a row
    // are contiguous.  (Stored rows are a multiple of 16 floats: the last chunk may be partly zero.)
    auto fetch = [&](uint32_t kc, float4 (&va)[kStageF4], float4 (&vb)[kStageF4]) {
#pragma unroll
        for (int u = 0; u < kStageF4; u++) {
            const int idx = tid + 256 * u;                                // 128 rows x kKF4 float4
            const int row = idx / kKF4, part = (idx % kKF4) << 2;
            va[u].x = va[u].y = va[u].z = va[u].w = 0.0f;
            vb[u] = va[u];
            if (kc + (uint32_t)part < a.stride) {
                // rows past the end re-read the last valid row (masked in the epilogue); Q is padded to a tile multiple
                uint32_t gp = p0 + row;
                if (gp >= a.n) gp = a.n - 1;
                va[u] = *reinterpret_cast<const float4*>(a.Q + (size_t)(q0 + row) * a.stride + kc + part);
                vb[u] = *reinterpret_cast<const float4*>(a.P + (size_t)gp * a.stride + kc + part);
            }
        }
    };
    float4 pa[kStageF4], pb[kStageF4];
    fetch(0u, pa, pb);
    for (uint32_t kc = 0; kc < a.stride; kc += kKC) {
#pragma unroll
        for (int u = 0; u < kStageF4; u++) {
            const int idx = tid + 256 * u;
            const int off = (idx / kKF4) * kLDP + ((idx % kKF4) << 2);
            *reinterpret_cast<float4*>(As + off) = pa[u];
            *reinterpret_cast<float4*>(Bs + off) = pb[u];
        }
        __syncthreads();
        if (kc + kKC < a.stride) fetch(kc + kKC, pa, pb);
        const float* ap = As + (wr * 64 + (lane & 31)) * kLDP + (kKC / 2) * (lane >> 5);
        const float* bp = Bs + (wc * 64 + (lane & 31)) * kLDP + (kKC / 2) * (lane >> 5);
#pragma unroll
        for (int q = 0; q < kKC / 8; q++) {
            const float4 a0 = *reinterpret_cast<const float4*>(ap + 4 * q);
            const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * kLDP + 4 * q);
            const float4 b0 = *reinterpret_cast<const float4*>(bp + 4 * q);
            const float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * kLDP + 4 * q);
#define IDIST_MFMA4(X_)                                                                      \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X_, b0.X_, acc[0][0], 0, 0, 0);  \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X_, b1.X_, acc[0][1], 0, 0, 0);  \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X_, b0.X_, acc[1][0], 0, 0, 0);  \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X_, b1.X_, acc[1][1], 0, 0, 0);
            IDIST_MFMA4(x) IDIST_MFMA4(y) IDIST_MFMA4(z) IDIST_MFMA4(w)
#undef IDIST_MFMA4
        }
        __syncthreads();
    }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The 128 query rows' norms and thresholds go
    // through LDS once (the staging buffers are free now): 64 accumulators per lane would otherwise each wait for two
    // dependent global loads — measured as a sixth of the kernel's time.
    float* sQ = As;                                                      // [128] |q|^2
    float* sT = As + kTM;                                                // [128] threshold (mode 1)
    if (tid < kTM) {
        const uint32_t q = q0 + (uint32_t)tid;
        sQ[tid] = q < a.nq ? a.qn[q] : 0.0f;
        sT[tid] = (a.mode == 1 && q < a.nq) ? a.thr[q] : 0.0f;
    }
    __syncthreads();
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
            const uint32_t p = p0 + (uint32_t)(wc * 64 + j * 32 + (lane & 31));
            const bool pok = p < a.p_end && p < a.n;
            const float pnv = pok ? a.pn[p] : 0.0f;
            for (int r = 0; r < 16; r++) {
                const int ql = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const uint32_t q = q0 + (uint32_t)ql;
                if (q >= a.nq || !pok) continue;
                float d = sQ[ql] + pnv - 2.0f * acc[i][j][r];
                if (!(d > 0.0f)) d = 0.0f;                           // cancellation can dip below zero
                if (a.mode == 0) {
                    a.dense[(size_t)q * a.dense_ld + (p - a.p_begin)] = d;
                } else if (d <= sT[ql]) {
                    const uint32_t slot = atomicAdd(&a.cnt[q], 1u);
                    if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = p;
                }
            }
        }
    }
}

// threshold of query q = k-th smallest of its dense sample row (+ slack): one wave per query
__global__ __launch_bounds__(64) void kth_threshold_kernel(const float* __restrict__ dense, uint32_t dense_ld, uint32_t ncols,
                                                          uint32_t nq, uint32_t k, uint32_t wcap, const float* __restrict__ qn,
                                                          float pn_max, float* __restrict__ thr) {
    IDIST_DYN_SMEM(smem_raw);
    uint64_t* W = reinterpret_cast<uint64_t*>(smem_raw);
    const int lane = lane_id();
    (void)wcap;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        WState st{W, 0, (int)k, 0, 0u};
        wave_sync();
        for (uint32_t base = 0; base < ncols; base += 64) {
            const uint32_t c = base + lane;
            uint64_t key = kMaxKey;
            if (c < ncols) key = ((uint64_t)__float_as_uint(dense[(size_t)q * dense_ld + c]) << 32) | c;
            const uint64_t t = st.plen >= st.ef ? (st.W[st.ef - 1] & kKeyMask) : kMaxKey + 1ull;
            uint64_t pm = __ballot(c < ncols && key < t);
            while (pm) {
                const int i = __builtin_ctzll(pm);
                pm &= pm - 1ull;
                const uint64_t kk = bcast_u64(key, i);
                const int idx = w_rank(st, kk);
                if (idx < st.ef) w_insert(st, idx, kk);
            }
            if (st.plen > st.ef) st.plen = st.ef;
            wave_sync();
        }
        if (lane == 0) {
            // fewer than k sample points: no bound.  Slack covers |d~ - d*| of both the sample's k-th and the candidate.
            float t = __uint_as_float(0x7f800000u);
            if (st.plen >= (int)k) t = __uint_as_float((uint32_t)((st.W[k - 1] & kKeyMask) >> 32)) + 1e-4f * (qn[q] + pn_max) + 1e-30f;
            thr[q] = t;
        }
        wave_sync();
    }
}

// exact canonical top-k of each query's candidate list (same machinery as bruteforce_kernel)
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(64) void rerank_kernel(IndexView ix, const float* __restrict__ queries, uint32_t nq, uint32_t k,
                                                   uint32_t wcap, const uint32_t* __restrict__ cand, const uint32_t* __restrict__ cnt,
                                                   uint32_t cap, uint32_t* out_pid, float* out_dist, uint32_t* overflow) {
    IDIST_DYN_SMEM(smem_raw);
    float* q = reinterpret_cast<float*>(smem_raw);
    uint64_t* W = reinterpret_cast<uint64_t*>(q + ix.stride);
    uint32_t* act_pid = reinterpret_cast<uint32_t*>(W + wcap);
    uint32_t* act_dist = act_pid + 64;
    const int lane = lane_id();
    const uint32_t nb = NB >= 0 ? (uint32_t)NB : ix.nb;
    for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        wave_sync();
        for (uint32_t o = lane; o < ix.stride; o += 64) q[o] = 0.0f;
        wave_sync();
        for (uint32_t e = lane; e < ix.dim; e += 64) q[blocked_pos(e, nb)] = queries[(size_t)qi * ix.dim + e];
        wave_sync();
        uint32_t nc = cnt[qi];
        if (nc > cap) { nc = cap; if (lane == 0) atomicAdd(overflow, 1u); }
        WState st{W, 0, (int)k, 0, 0u};
        for (uint32_t base = 0; base < nc; base += 64) {
            const int na = nc - base < 64u ? (int)(nc - base) : 64;
            if (lane < na) act_pid[lane] = cand[(size_t)qi * cap + base + lane];
            wave_sync();
            dist_rounds<NB, RS, TAIL>(ix, q, act_pid, act_dist, na);
            wave_sync();
            uint64_t key = kMaxKey;
            if (lane < na) key = ((uint64_t)act_dist[lane] << 32) | act_pid[lane];
            const uint64_t t = st.plen >= st.ef ? (st.W[st.ef - 1] & kKeyMask) : kMaxKey + 1ull;
            uint64_t pm = __ballot(lane < na && key < t);
            while (pm) {
                const int i = __builtin_ctzll(pm);
                pm &= pm - 1ull;
                const uint64_t kk = bcast_u64(key, i);
                const int idx = w_rank(st, kk);
                if (idx < st.ef) w_insert(st, idx, kk);
            }
            if (st.plen > st.ef) st.plen = st.ef;
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
// Build step A2 on the matrix cores: Search::select_heuristic for a new point (core/lib.rs:636-698, extend_candidates
// = false) — the one place on the build path where the distances ARE a dense contraction: the verdict on candidate i
// needs d(c_i, c_j) for the candidates j < i that were selected, i.e. entries of the Gram matrix of the <= 128
// candidate rows (100 x 100 x 300 at C3, per inserted point).
//
// The order-defining comparison is `d(c_i, c_j) < d(c_i, q)` (:676-679) with the CANONICAL distance; a product-form
// distance |c_i|^2 + |c_j|^2 - 2 c_i.c_j rounds differently, so it is used as a filter with a margin, never as the
// answer: with G~ from v_mfma_f32_32x32x2_f32 and eps_ij = kGramEps * K * (|c_i|^2 + |c_j|^2) (K = stored row length).
// Error budget: a K-term fma chain is off by at most ~K u sum|a_k b_k| <= K u |a||b| (u = 2^-24), so the three chains of G~
// are within 2 K u (|c_i|^2 + |c_j|^2) of the true value, the canonical 8-chain sum within (K/8 + 3) u d <= 0.3 K u (...)
// of it: |G~ - canonical| <= 2.3 K u (|c_i|^2 + |c_j|^2), and eps is 4 K u (...) plus an absolute floor of 1e-30 for
// inputs so small that a flushed denormal could matter.  With it,
//      G~ + eps <  d(c_i,q)   =>  c_j is closer for certain          (bit in closer[i])
//      G~ - eps >= d(c_i,q)   =>  it is not, for certain
//      otherwise              =>  uncertain: the canonical distance is computed for that pair (bit in unsure[i]).
// The sequential part of the heuristic then is bit arithmetic on 128-bit masks; ~0.01 % of the pairs (C3 data) take the exact
// path.  Results are identical to build_select_kernel's (same selected set, same order, a valid pruner per discarded
// entry); tests run both (IDIST_BUILD_A2=tile selects the tile kernel).  Squared-L2 metric and ef_construction <= 128.
//
// One 256-thread workgroup per new point: the four waves share the staged K-chunks (128 rows x 32 floats) and own
// the ten lower-triangle 32 x 32 tiles 3/3/2/2; wave 0 runs the selection.  28 KB of LDS, < 128 VGPRs.
// ---------------------------------------------------------------------------
// pitch 36 floats: 16-B aligned rows, and 36 = 4 * 9 with 9 odd => any 16 lanes reading float4s of rows that differ mod 16 hit
// 64 distinct banks (ds_read_b128 / ds_write_b128 without conflicts)
constexpr int kGramRows = 128, kGramChunk = 32, kGramPitch = 36;
constexpr float kGramEps = 4.0f * 5.9604645e-8f;    // 4 * 2^-24 per stored element, times the row length at run time

__host__ __device__ inline size_t smem_bytes_select_mfma(uint32_t stride) {
    return (size_t)kGramRows * kGramPitch * 4 + 2 * kGramRows * 4 * 4 + 2 * kGramRows * 4 + kGramRows * 8 + kGramRows * 4 +
           2 * 64 * 8 + 2 * 64 * 4 + 2 * 64 * 4 + (size_t)stride * 4 + 16;
}

// (compiled for >= 4 waves per SIMD, i.e. <= 128 registers: its waves must fit next to the 1-wave-per-SIMD descents)
#ifdef IDIST_EMU
#define IDIST_A2M_ATTR
#else
#define IDIST_A2M_ATTR __attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
template <int NB, int RS, int TAIL>
__global__ __launch_bounds__(256) IDIST_A2M_ATTR void build_select_mfma_kernel(IndexView ix, BuildArgs a) {
    IDIST_DYN_SMEM(smem_raw);
    float* Cs = reinterpret_cast<float*>(smem_raw);                      // [128][33] one K-chunk of every candidate row
    uint32_t* closer = reinterpret_cast<uint32_t*>(Cs + kGramRows * kGramPitch);   // [128][4]
    uint32_t* unsure = closer + kGramRows * 4;                           // [128][4]
    float* norms = reinterpret_cast<float*>(unsure + kGramRows * 4);     // [128]
    float* cdf = norms + kGramRows;                                      // [128] d(c_i, q)
    uint64_t* keys = reinterpret_cast<uint64_t*>(cdf + kGramRows);       // [128] Search.nearest
    uint32_t* pids = reinterpret_cast<uint32_t*>(keys + kGramRows);      // [128]
    uint64_t* sel = reinterpret_cast<uint64_t*>(pids + kGramRows);       // [64]
    uint64_t* disc = sel + 64;                                           // [64]
    uint32_t* dprn = reinterpret_cast<uint32_t*>(disc + 64);             // [64]
    uint32_t* out_aux = dprn + 64;                                       // [64]
    uint32_t* act_pid = out_aux + 64;                                    // [64]
    uint32_t* act_dist = act_pid + 64;                                   // [64]
    float* qrow = reinterpret_cast<float*>(act_dist + 64);               // [stride] one candidate row (exact path)
    uint32_t* ctl = reinterpret_cast<uint32_t*>(qrow + ix.stride);       // [4]
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = (int)uniform_u32((uint32_t)tid >> 6);
    // lower-triangle tiles (ti, tj) of the 4 x 4 grid, by wave: w0 (0,0) (3,0) (3,1); w1 (1,0) (1,1) (3,2); w2 (2,0) (2,1);
    // w3 (2,2) (3,3) — one nibble per tile
    const uint32_t ti_pack = wv == 0 ? 0x330u : (wv == 1 ? 0x311u : (wv == 2 ? 0x022u : 0x032u));
    const uint32_t tj_pack = wv == 0 ? 0x100u : (wv == 1 ? 0x210u : (wv == 2 ? 0x010u : 0x032u));
    const int ntile = wv < 2 ? 3 : 2;
    int kTi[3], kTj[3];
#pragma unroll
    for (int t = 0; t < 3; t++) { kTi[t] = (int)((ti_pack >> (4 * t)) & 15u); kTj[t] = (int)((tj_pack >> (4 * t)) & 15u); }
    const float eps_k = kGramEps * (float)ix.stride;
    HeurCounters hc{0, 0};
    for (;;) {
        if (tid == 0) ctl[0] = atomicAdd(&a.queue[4], 1u);
        block_sync();
        const uint32_t item = uniform_u32(ctl[0]);
        if (item >= a.count) break;
        const uint32_t nw_pid = a.start + item;
        const int nw = (int)a.wcount[item];
        for (int i = tid; i < kGramRows; i += 256) {
            const uint64_t k = i < nw ? a.wbuf[(size_t)item * a.efc + i] : 0ull;
            keys[i] = k;
            pids[i] = i < nw ? (uint32_t)k : 0u;                          // padding rows re-read point 0 (never used)
            cdf[i] = __uint_as_float((uint32_t)(k >> 32));
        }
        for (int i = tid; i < kGramRows * 4; i += 256) { closer[i] = 0u; unsure[i] = 0u; }
        block_sync();
        f32x16 acc[3];
        for (int t = 0; t < 3; t++)
            for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
        // ---- Gram matrix: G = C * C^T over the stored row length, one 32-float chunk at a time.  The next chunk's rows are
        //      requested (global -> registers) before the current chunk's MFMAs, so HBM latency hides behind the matrix
        //      pipe; operands move as float4s (ds_write_b128 / ds_read_b128).  K order inside a chunk: the instruction
        //      v_mfma_f32_32x32x2_f32 #j contracts elements j (lanes 0-31) and 16 + j (lanes 32-63) — a lane's sixteen
        //      operands are contiguous in its row.  (Any fixed order of the sum is a valid Gram entry for the filter.)
        auto fetch = [&](uint32_t kc, float4 (&v)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = tid + 256 * u;                           // 1024 float4 = 128 rows x 8
                const int row = idx >> 3, part = (idx & 7) << 2;
                v[u].x = v[u].y = v[u].z = v[u].w = 0.0f;
                if (kc + (uint32_t)part < ix.stride && row < nw)
                    v[u] = *reinterpret_cast<const float4*>(ix.points + (size_t)pids[row] * ix.stride + kc + part);
            }
        };
        float4 pre[4];
        fetch(0u, pre);
        for (uint32_t kc = 0; kc < ix.stride; kc += kGramChunk) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = tid + 256 * u;
                *reinterpret_cast<float4*>(Cs + (idx >> 3) * kGramPitch + ((idx & 7) << 2)) = pre[u];
            }
            block_sync();
            if (kc + kGramChunk < ix.stride) fetch(kc + kGramChunk, pre);
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (t < ntile && kTi[t] * 32 < nw) {
                    const float* ap = Cs + (kTi[t] * 32 + (lane & 31)) * kGramPitch + 16 * (lane >> 5);
                    const float* bp = Cs + (kTj[t] * 32 + (lane & 31)) * kGramPitch + 16 * (lane >> 5);
                    float4 af[4], bf[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        af[q] = *reinterpret_cast<const float4*>(ap + 4 * q);
                        bf[q] = *reinterpret_cast<const float4*>(bp + 4 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].w, acc[t], 0, 0, 0);
                    }
                }
            }
            block_sync();
        }
        // ---- |c_i|^2 from the diagonal tiles: lane l holds G[x][x], x = l & 31, iff ((x >> 2) & 1) == (l >> 5), in
        //      register ((x >> 3) << 2) | (x & 3)
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (t < ntile && kTi[t] == kTj[t]) {
                const int x = lane & 31;
                const bool has = ((x >> 2) & 1) == (lane >> 5);
                const int rsel = ((x >> 3) << 2) | (x & 3);
                float nv = 0.0f;
                for (int r = 0; r < 16; r++) nv = r == rsel ? acc[t][r] : nv;
                if (has) norms[kTi[t] * 32 + x] = nv;
            }
        }
        block_sync();
        // ---- verdict masks: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (t < ntile && kTi[t] * 32 < nw) {
                const int j = kTj[t] * 32 + (lane & 31);
                const float nj = norms[j];
                for (int r = 0; r < 16; r++) {
#ifndef IDIST_EMU
                    asm volatile("" ::: "memory");                        // one row's operands at a time (keeps the kernel under 128 registers)
#endif
                    const int i = kTi[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float ni = norms[i], cd = cdf[i];
                    const float g = ni + nj - 2.0f * acc[t][r];
                    const float eps = eps_k * (ni + nj) + 1e-30f;         // + a floor far above any flushed-denormal product sum
                    const bool valid = j < i && i < nw;
                    const bool cl = valid && (g + eps < cd);
                    const bool ncl = g - eps >= cd;
                    const bool un = valid && !cl && !ncl;                 // also every NaN / inf case
                    const uint64_t mc = __ballot(cl), mu = __ballot(un);
                    const int i0 = kTi[t] * 32 + (r & 3) + 8 * (r >> 2);
                    if (lane == 0) { closer[i0 * 4 + kTj[t]] = (uint32_t)mc; unsure[i0 * 4 + kTj[t]] = (uint32_t)mu; }
                    if (lane == 32) { closer[(i0 + 4) * 4 + kTj[t]] = (uint32_t)(mc >> 32); unsure[(i0 + 4) * 4 + kTj[t]] = (uint32_t)(mu >> 32); }
                }
            }
        }
        block_sync();
        // ---- the heuristic itself (wave 0): core/lib.rs:666-695 on the masks.  The verdict masks of all candidates sit in
        //      registers (lane l: candidates l and 64 + l), so one step of the sequential loop is four readlanes and a few
        //      scalar ops — no LDS round trip on the path from one verdict to the next (the loop used to read closer[] /
        //      keys[] / pids[] from LDS every iteration: ~100 dependent round trips per new point with three waves idle).
        //      Who pruned whom is kept as a candidate INDEX per lane; keys and pids are gathered after the loop, in parallel.
        if (wv == 0) {
            const uint4 cl0 = *reinterpret_cast<const uint4*>(closer + lane * 4), cl1 = *reinterpret_cast<const uint4*>(closer + (64 + lane) * 4);
            const uint4 un0 = *reinterpret_cast<const uint4*>(unsure + lane * 4), un1 = *reinterpret_cast<const uint4*>(unsure + (64 + lane) * 4);
            uint32_t selm[4] = {0u, 0u, 0u, 0u};
            int nsel = 0, nproc = 0;
            uint32_t pr0 = 0u, pr1 = 0u;                                 // lane l: index of the candidate that pruned l / 64 + l
            for (int i = 0; i < nw && nsel < kM2; i++) {
                const int li = i & 63;
                uint32_t m[4], u[4];
                if (i < 64) {
                    m[0] = readlane_u32(cl0.x, li); m[1] = readlane_u32(cl0.y, li); m[2] = readlane_u32(cl0.z, li); m[3] = readlane_u32(cl0.w, li);
                    u[0] = readlane_u32(un0.x, li); u[1] = readlane_u32(un0.y, li); u[2] = readlane_u32(un0.z, li); u[3] = readlane_u32(un0.w, li);
                } else {
                    m[0] = readlane_u32(cl1.x, li); m[1] = readlane_u32(cl1.y, li); m[2] = readlane_u32(cl1.z, li); m[3] = readlane_u32(cl1.w, li);
                    u[0] = readlane_u32(un1.x, li); u[1] = readlane_u32(un1.y, li); u[2] = readlane_u32(un1.z, li); u[3] = readlane_u32(un1.w, li);
                }
                bool pruned = false;
                uint32_t pr_idx = 0;
#pragma unroll
                for (int w = 3; w >= 0; w--) {                           // (lowest word last: the first closer member wins, as before)
                    const uint32_t mm = m[w] & selm[w];
                    if (mm) { pruned = true; pr_idx = (uint32_t)(w * 32 + __builtin_ctz(mm)); }
                }
                if (!pruned && (((u[0] & selm[0]) | (u[1] & selm[1]) | (u[2] & selm[2]) | (u[3] & selm[3])) != 0u)) {
                    // pairs the filter could not decide: canonical distance (rare: ~0.01 % of the pairs on float data)
                    const uint32_t cd = (uint32_t)(keys[i] >> 32);
                    bool staged = false;
                    for (int w = 0; w < 4 && !pruned; w++) {
                        const uint32_t mm = u[w] & selm[w];
                        if (!mm) continue;
                        wave_sync();
                        if (!staged) {
                            const float* prow = ix.points + (size_t)pids[i] * ix.stride;
                            for (uint32_t o = lane * 4; o < ix.stride; o += 256)
                                *reinterpret_cast<float4*>(qrow + o) = *reinterpret_cast<const float4*>(prow + o);
                            staged = true;
                        }
                        const int cnt = __builtin_popcount(mm);
                        if (lane < 32 && ((mm >> lane) & 1u)) act_pid[__builtin_popcount(mm & ((1u << lane) - 1u))] = pids[w * 32 + lane];
                        wave_sync();
                        dist_rounds<-1, -1, -1>(ix, qrow, act_pid, act_dist, cnt);     // rare: the small runtime-geometry form
                        wave_sync();
                        hc.n_dist += (uint32_t)cnt;
                        const uint64_t cm = __ballot(lane < cnt && act_dist[lane] < cd);   // strict <, :678
                        if (cm) {
                            uint32_t rest = mm;                              // the ctz(cm)-th member of the list = that set bit of mm
                            for (int k = __builtin_ctzll(cm); k > 0; k--) rest &= rest - 1u;
                            pruned = true;
                            pr_idx = (uint32_t)(w * 32 + __builtin_ctz(rest));
                        }
                    }
                    wave_sync();
                }
                if (!pruned) {                                           // :681-684
                    selm[i >> 5] |= 1u << (i & 31);
                    nsel++;
                } else if (lane == li) {
                    if (i < 64) pr0 = pr_idx; else pr1 = pr_idx;
                }
                nproc = i + 1;
            }
            hc.n_rows += (uint32_t)nw;
            hc.n_dist += (uint32_t)(nw * (nw - 1) / 2);                  // pairs decided (by the filter or exactly)
            // selected-then-discarded lists, every lane placing its (up to) two candidates: position = rank among its kind
            const int n_selected = nsel;
            int ndis = nproc - nsel;
            out_aux[lane] = 0u;
            wave_sync();
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = h * 64 + lane;
                if (i < nproc) {
                    int below = 0;                                       // selected candidates in front of i
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const int lo = w * 32;
                        const uint32_t msk = i >= lo + 32 ? 0xFFFFFFFFu : (i > lo ? ((1u << (i - lo)) - 1u) : 0u);
                        below += __builtin_popcount(selm[w] & msk);
                    }
                    const bool is_sel = ((selm[i >> 5] >> (i & 31)) & 1u) != 0u;
                    if (is_sel) sel[below] = keys[i];
                    else if (i - below < kM2) { disc[i - below] = keys[i]; dprn[i - below] = pids[h ? pr1 : pr0]; }
                }
            }
            wave_sync();
            if (a.keep_pruned) {                                         // :687-695
                if (ndis > kM2) ndis = kM2;
                int take = kM2 - nsel;
                if (take > ndis) take = ndis;
                if (lane < take) { sel[nsel + lane] = disc[lane]; out_aux[nsel + lane] = dprn[lane]; }
                if (take > 0) nsel += take;
                wave_sync();
            }
            if (lane == 0) a.row_nsel[nw_pid] = (uint32_t)n_selected;
            a.nbr_aux[(size_t)nw_pid * kM2 + lane] = out_aux[lane];
            emit_new_node(ix, a, item, nw_pid, sel, nsel);
        }
        block_sync();
    }
    if (tid == 0 && (hc.n_dist | hc.n_rows)) {
        atomicAdd(&a.stats[3], (unsigned long long)hc.n_dist);
        atomicAdd(&a.stats[4], (unsigned long long)hc.n_rows);
    }
}

}  // namespace idist
