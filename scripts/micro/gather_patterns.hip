// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/gp scripts/micro/gather_patterns.hip && /tmp/gp
// Microbenchmark: random-row gather rate of 1216-B rows (C3's row) under different lane-to-row mappings.
//   P8 : 8 lanes per row, 8 rows per wave instruction (the engine's dist_rounds mapping), 16 B per lane
//   P16: 16 lanes per row, 4 rows per instruction
//   P64: 64 lanes per row (one row per instruction group), 16 B per lane, 1216 B = 76 lanes*16 -> 2 instructions
// Each wave sums what it loads so the loads cannot be dropped.  Rows are a random permutation (every row once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

constexpr int kStride = 304;   // floats per row (1216 B)

template <int LPR>   // lanes per row
__global__ __launch_bounds__(64) void gather(const float* __restrict__ pts, const uint32_t* __restrict__ order, uint32_t n,
                                             float* __restrict__ out, int rows_in_flight) {
    const int lane = threadIdx.x;
    const int rpi = 64 / LPR;                 // rows per instruction
    const int g = lane / LPR, j = lane % LPR;
    float acc = 0.f;
    const uint32_t waves = gridDim.x;
    for (uint32_t base = blockIdx.x * rpi; base < n; base += waves * rpi) {
        const uint32_t r = base + g;
        if (r < n) {
            const float* row = pts + (size_t)order[r] * kStride;
            for (int o = j * 4; o < kStride; o += LPR * 4) {
                const float4 v = *reinterpret_cast<const float4*>(row + o);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main() {
    const uint32_t n = 1000000;
    float* d_pts; uint32_t* d_ord; float* d_out;
    hipMalloc(&d_pts, (size_t)n * kStride * 4); hipMemset(d_pts, 0, (size_t)n * kStride * 4);
    std::vector<uint32_t> ord(n); for (uint32_t i = 0; i < n; i++) ord[i] = i;
    std::mt19937 rng(1); std::shuffle(ord.begin(), ord.end(), rng);
    hipMalloc(&d_ord, n * 4); hipMemcpy(d_ord, ord.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)n * kStride * 4;
    for (int wpc : {8, 16, 32}) {
        const int grid = 256 * wpc;
        for (int pat : {8, 16, 64}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                hipEventRecord(e0);
                if (pat == 8) gather<8><<<grid, 64>>>(d_pts, d_ord, n, d_out, 0);
                else if (pat == 16) gather<16><<<grid, 64>>>(d_pts, d_ord, n, d_out, 0);
                else gather<64><<<grid, 64>>>(d_pts, d_ord, n, d_out, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("{\"lanes_per_row\": %d, \"waves_per_cu\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", pat, wpc, best, bytes / best / 1e9);
        }
    }
    // sequential rows (streaming) for reference
    for (uint32_t i = 0; i < n; i++) ord[i] = i;
    hipMemcpy(d_ord, ord.data(), n * 4, hipMemcpyHostToDevice);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0); gather<8><<<256 * 16, 64>>>(d_pts, d_ord, n, d_out, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("{\"lanes_per_row\": 8, \"waves_per_cu\": 16, \"order\": \"sequential rows\", \"ms\": %.4f, \"TBps\": %.3f}\n", best, bytes / best / 1e9);
    return 0;
}
