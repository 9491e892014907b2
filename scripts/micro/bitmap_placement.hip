// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/bp scripts/micro/bitmap_placement.hip && /tmp/bp
// Why does a long walk's time depend on which allocation its visited bitmaps got (profiles/r05/probe_r05b_placement_*)?
// K allocations of 2048 slots x 125,056 B (the C3 geometry), each driven by 1536 waves doing DEPENDENT random test-and-set atomics on
// their own slot (the long walk's pattern: ~36 touches per 128-B line and query), (a) alone, (b) beside waves that stream random
// 1216-B rows out of a 1.2-GB array (the walk's row gathers, which compete for the Infinity Cache).  One JSON line per allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr uint32_t kSlotWords = 31264, kSlots = 2048, kActive = 1536, kRowFloats = 304, kRows = 1000000;

__global__ __launch_bounds__(64) void walk(uint32_t* __restrict__ bitmaps, const float* __restrict__ rows, uint32_t iters, uint32_t rows_per_iter,
                                           float* __restrict__ sink) {
    const uint32_t slot = blockIdx.x;
    uint32_t* bm = bitmaps + (size_t)slot * kSlotWords;
    uint32_t x = slot * 2654435761u + threadIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (uint32_t i = 0; i < iters; i++) {
        // 64 lanes: one test-and-set each on a random word of the slot; the next address depends on the returned word
        x = x * 1664525u + 1013904223u;
        const uint32_t w = (x >> 8) % kSlotWords;
        const uint32_t old = atomicOr(&bm[w], 1u << (x & 31u));
        x += old & 1u;
        // the "new" rows of this expansion: rows_per_iter random rows, 8 lanes per row, 16 B per lane and pass
        for (uint32_t r = 0; r < rows_per_iter; r += 8) {
            x = x * 1664525u + 1013904223u;
            const uint32_t row = (__shfl(x, (threadIdx.x / 8) * 8) >> 4) % kRows;
            const float* p = rows + (size_t)row * kRowFloats;
            for (uint32_t o = (threadIdx.x % 8) * 4; o < kRowFloats; o += 32) {
                const float4 v = *reinterpret_cast<const float4*>(p + o);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 123.456f) sink[slot] = acc;
}

int main() {
    const size_t vb = (size_t)kSlots * kSlotWords * 4;
    float* d_rows; float* d_sink;
    hipMalloc(&d_rows, (size_t)kRows * kRowFloats * 4); hipMemset(d_rows, 0, (size_t)kRows * kRowFloats * 4);
    hipMalloc(&d_sink, 1 << 20);
    const int K = 8;
    std::vector<uint32_t*> bm(K);
    std::vector<void*> spacer(K);
    for (int k = 0; k < K; k++) {
        hipMalloc(&bm[k], vb); hipMemset(bm[k], 0, vb);
        hipMalloc(&spacer[k], (size_t)(97 + 61 * k) << 20);            // odd-sized neighbours: the next candidate lands somewhere else
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++)
        for (int k = 0; k < K; k++) {
            float t[2];
            for (int mode = 0; mode < 2; mode++) {
                const uint32_t rows_per_iter = mode ? 8 : 0;
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    walk<<<kActive, 64>>>(bm[k], d_rows, 4000, rows_per_iter, d_sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                t[mode] = best;
            }
            printf("{\"probe\": \"bitmap_placement_micro\", \"pass\": %d, \"allocation\": %d, \"address\": \"%p\", \"atomics_alone_ms\": %.3f, \"atomics_beside_row_gathers_ms\": %.3f}\n",
                   pass, k, (void*)bm[k], t[0], t[1]);
        }
    return 0;
}
