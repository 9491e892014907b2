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

namespace idist {

#ifndef IDIST_EMU
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

constexpr int kTM = 128, kTN = 128, kKC = 16, kLDP = kKC + 1;   // tile, K chunk, padded LDS pitch

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

    const int srow = tid >> 2, scol = (tid & 3) << 2;                 // staging: 4 threads x float4 per 64-B row chunk
    for (uint32_t kc = 0; kc < a.stride; kc += kKC) {
        for (int pass = 0; pass < 2; pass++) {
            const int row = pass * 64 + srow;
            // rows past the end re-read the last valid row (masked in the epilogue)
            const uint32_t gq = q0 + row;                             // Q is padded to a tile multiple
            uint32_t gp = p0 + row;
            if (gp >= a.n) gp = a.n - 1;
            const float4 va = *reinterpret_cast<const float4*>(a.Q + (size_t)gq * a.stride + kc + scol);
            const float4 vb = *reinterpret_cast<const float4*>(a.P + (size_t)gp * a.stride + kc + scol);
            float* da = As + row * kLDP + scol;
            float* db = Bs + row * kLDP + scol;
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kKC / 2; kk++) {
            const int kcol = kk * 2 + (lane >> 5);
            const float a0 = As[(wr * 64 + (lane & 31)) * kLDP + kcol];
            const float a1 = As[(wr * 64 + 32 + (lane & 31)) * kLDP + kcol];
            const float b0 = Bs[(wc * 64 + (lane & 31)) * kLDP + kcol];
            const float b1 = Bs[(wc * 64 + 32 + (lane & 31)) * kLDP + kcol];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    for (int i = 0; i < 2; i++) {
        for (int j = 0; j < 2; j++) {
            const uint32_t p = p0 + (uint32_t)(wc * 64 + j * 32 + (lane & 31));
            const bool pok = p < a.p_end && p < a.n;
            const float pnv = pok ? a.pn[p] : 0.0f;
            for (int r = 0; r < 16; r++) {
                const uint32_t q = q0 + (uint32_t)(wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                if (q >= a.nq || !pok) continue;
                float d = a.qn[q] + pnv - 2.0f * acc[i][j][r];
                if (!(d > 0.0f)) d = 0.0f;                           // cancellation can dip below zero
                if (a.mode == 0) {
                    a.dense[(size_t)q * a.dense_ld + (p - a.p_begin)] = d;
                } else if (d <= a.thr[q]) {
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

}  // namespace idist
