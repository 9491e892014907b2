// build + run under the profiler on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fc scripts/micro/fetch_calib.hip && rocprofv3 --pmc FETCH_SIZE -d /tmp/fc_out -o fc -- /tmp/fc
// FETCH_SIZE on gfx950 tallies 128-B fabric requests at 64 B (MI355X_MICROARCH.md §HBM) — for WIDE requests.  The build's kernels mix
// access patterns: 1216-B row gathers (128-B requests), 48-B probes of the published distance logs (one 64-B sector each), 256-B
// adjacency rows.  One kernel per pattern with a KNOWN byte count, each touching every byte of a buffer far beyond the Infinity Cache
// exactly once, so that reported / known gives the correction per pattern.  Prints the known bytes; the counters come from rocprofv3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

// every lane reads PIECES consecutive 16-B pieces at a random RECORD-byte-aligned record; records are visited exactly once
template <int RECORD, int PIECES>
__global__ __launch_bounds__(64) void calib_probe(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ order, uint32_t n_rec, float* __restrict__ sink) {
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * 64 + threadIdx.x; i < n_rec; i += gridDim.x * 64) {
        const float4* p = reinterpret_cast<const float4*>(buf + (size_t)order[i] * RECORD);
#pragma unroll
        for (int u = 0; u < PIECES; u++) { const float4 v = p[u]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}
// the walk's row gather: 8 lanes per row, 16 B per lane and pass, rows of ROWB bytes in random order, every row once
template <int ROWB>
__global__ __launch_bounds__(64) void calib_rows(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ order, uint32_t n_rows, float* __restrict__ sink) {
    float acc = 0.f;
    const int g = threadIdx.x >> 3, j = threadIdx.x & 7;
    for (uint32_t base = blockIdx.x * 8; base < n_rows; base += gridDim.x * 8) {
        const uint32_t r = base + g;
        if (r < n_rows) {
            const uint8_t* row = buf + (size_t)order[r] * ROWB;
            for (int o = j * 16; o < ROWB; o += 128) { const float4 v = *reinterpret_cast<const float4*>(row + o); acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ buf, size_t n16, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = buf[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;                     // 2 GiB: eight times the Infinity Cache
    uint8_t* d_buf; float* d_sink; uint32_t* d_ord;
    hipMalloc(&d_buf, bytes); hipMemset(d_buf, 0, bytes);
    hipMalloc(&d_sink, 1 << 20);
    const uint32_t max_rec = (uint32_t)(bytes / 32);
    hipMalloc(&d_ord, (size_t)max_rec * 4);
    std::mt19937 rng(7);
    auto shuffle_to = [&](uint32_t n) { std::vector<uint32_t> o(n); for (uint32_t i = 0; i < n; i++) o[i] = i; std::shuffle(o.begin(), o.end(), rng);
                                         hipMemcpy(d_ord, o.data(), (size_t)n * 4, hipMemcpyHostToDevice); };
    const int grid = 256 * 16;
    calib_stream<<<grid, 256>>>(reinterpret_cast<const float4*>(d_buf), bytes / 16, d_sink);
    printf("{\"kernel\": \"calib_stream\", \"known_bytes\": %zu}\n", bytes);
    uint32_t n;
    n = (uint32_t)(bytes / 1216); shuffle_to(n); calib_rows<1216><<<grid, 64>>>(d_buf, d_ord, n, d_sink);
    printf("{\"kernel\": \"calib_rows<1216>\", \"known_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 1216, (size_t)n * 4);
    n = (uint32_t)(bytes / 512); shuffle_to(n); calib_rows<512><<<grid, 64>>>(d_buf, d_ord, n, d_sink);
    printf("{\"kernel\": \"calib_rows<512>\", \"known_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 512, (size_t)n * 4);
    n = (uint32_t)(bytes / 256); shuffle_to(n); calib_probe<256, 16><<<grid, 64>>>(d_buf, d_ord, n, d_sink);   // (one lane reads a whole 256-B adjacency row)
    printf("{\"kernel\": \"calib_probe<256, 16>\", \"known_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 256, (size_t)n * 4);
    n = (uint32_t)(bytes / 64); shuffle_to(n); calib_probe<64, 3><<<grid, 64>>>(d_buf, d_ord, n, d_sink);      // the q16 distance-log probe: 48 B of a 64-B record
    printf("{\"kernel\": \"calib_probe<64, 3>\", \"known_bytes\": %zu, \"known_sector_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 48, (size_t)n * 64, (size_t)n * 4);
    n = (uint32_t)(bytes / 64); shuffle_to(n); calib_probe<64, 4><<<grid, 64>>>(d_buf, d_ord, n, d_sink);
    printf("{\"kernel\": \"calib_probe<64, 4>\", \"known_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 64, (size_t)n * 4);
    n = (uint32_t)(bytes / 32); shuffle_to(n); calib_probe<32, 2><<<grid, 64>>>(d_buf, d_ord, n, d_sink);      // the id-form log record: 32 B
    printf("{\"kernel\": \"calib_probe<32, 2>\", \"known_bytes\": %zu, \"index_bytes\": %zu}\n", (size_t)n * 32, (size_t)n * 4);
    hipDeviceSynchronize();
    return 0;
}
