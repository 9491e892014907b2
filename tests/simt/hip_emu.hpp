// hip_emu.hpp — TEST INFRASTRUCTURE: a CPU lockstep emulator for the single-wave
// kernels of instant-distance_amd/csrc, so that the REAL kernel source and the REAL
// C-ABI host code can be exercised by the parity tests in a container without a GPU.
//
//  * never shipped, never loaded by the product package: tests compile the csrc
//    sources with -DIDIST_EMU into tests/simt/_build/libidist_emu.so;
//  * each workgroup is a set of fibers (one per lane); a lane runs until it reaches
//    a collective (__syncthreads, __ballot, __shfl*, readfirstlane), then the next lane
//    runs; when all live lanes wait at the SAME collective it is resolved.  Lanes reaching
//    different collectives = divergent collective = abort (a real bug on hardware);
//  * lane scheduling order between collectives is configurable (IDIST_EMU_ORDER=
//    forward|reverse) — results must not depend on it, which catches cross-lane LDS
//    hand-offs that lack a barrier;
//  * "device" memory is host memory, poisoned with 0xCD at allocation.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define IDIST_WAVES_ATTR(...)   /* register-allocation hint of the device compiler */
#define __restrict__ __restrict

struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct uint2 { uint32_t x, y; } __attribute__((aligned(8)));
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

namespace emu {

struct Dim3 { uint32_t x, y, z; };

enum Op { OP_NONE = 0, OP_SYNC, OP_WSYNC, OP_BALLOT, OP_SHFL, OP_SHFL_XOR, OP_READFIRST, OP_MFMA_32x32x2, OP_DPP };

struct Lane {
    void* sp = nullptr;          // saved stack pointer
    uint8_t* stack = nullptr;
    Dim3 tid{0, 0, 0};
    bool alive = false, waiting = false;
    int op = OP_NONE;
    uint64_t val = 0, res = 0;
    int arg = 0;
    float a = 0, b = 0, c[16] = {0}, d[16] = {0};   // MFMA operands / result
};

extern "C" void idist_emu_switch(void** save_sp, void* new_sp);

struct State {
    std::vector<Lane> lanes;
    Lane* cur = nullptr;
    void* sched_sp = nullptr;
    Dim3 blk{0, 0, 0}, bdim{64, 1, 1}, gdim{1, 1, 1};
    uint8_t* smem = nullptr;
    const std::function<void()>* body = nullptr;
    bool reverse = false;
    uint64_t n_collectives = 0;
};
State& S();

inline uint8_t* cur_smem() { return S().smem; }

inline void yield_to_sched() {
    Lane* l = S().cur;
    idist_emu_switch(&l->sp, S().sched_sp);
}

inline uint64_t collective(int op, uint64_t v, int arg) {
    Lane* l = S().cur;
    l->op = op;
    l->val = v;
    l->arg = arg;
    l->waiting = true;
    yield_to_sched();
    return l->res;
}

void launch(uint32_t grid, uint32_t block, size_t smem_bytes, const std::function<void()>& body);

// wave-level sync of the kernels (idist::wave_sync): a rendezvous of the 64 lanes of ONE wave — on hardware the
// lanes of a wave run in lockstep and their LDS accesses execute in order; here every lane is a fiber
inline void wave_sync() { collective(OP_WSYNC, 0, 0); }

}  // namespace emu

#define threadIdx (::emu::S().cur->tid)
#define blockIdx (::emu::S().blk)
#define blockDim (::emu::S().bdim)
#define gridDim (::emu::S().gdim)

// ---- collectives ----
static inline void __syncthreads() { ::emu::collective(::emu::OP_SYNC, 0, 0); }
static inline unsigned long long __ballot(int pred) { return ::emu::collective(::emu::OP_BALLOT, pred ? 1 : 0, 0); }
static inline int __shfl(int v, int src, int /*width*/ = 64) {
    return (int)(uint32_t)::emu::collective(::emu::OP_SHFL, (uint32_t)v, src & 63);
}
static inline int __shfl_xor(int v, int mask, int /*width*/ = 64) {
    return (int)(uint32_t)::emu::collective(::emu::OP_SHFL_XOR, (uint32_t)v, mask);
}
static inline float __shfl_xor(float v, int mask, int /*width*/ = 64) {
    uint32_t b;
    memcpy(&b, &v, 4);
    b = (uint32_t)::emu::collective(::emu::OP_SHFL_XOR, b, mask);
    memcpy(&v, &b, 4);
    return v;
}
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) {
    return (uint32_t)::emu::collective(::emu::OP_READFIRST, v, 0);
}
static inline void __threadfence_block() {}
// v_mov_b32_dpp: quad_perm (ctrl 0x00-0xFF), row_shl:n / row_shr:n (0x101-0x10F / 0x111-0x11F), row_mirror / row_half_mirror (0x140 / 0x141), all rows and
// banks enabled; a source lane outside the row reads 0 with bound_ctrl, keeps `old` without.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (row_mask != 0xF || bank_mask != 0xF) { fprintf(stderr, "[hip_emu] dpp row/bank masks not emulated\n"); abort(); }
    const uint64_t r = ::emu::collective(::emu::OP_DPP, (uint32_t)src, ctrl);
    if (r >> 32) return bound_ctrl ? 0 : old;   // invalid source lane
    return (int)(uint32_t)r;
}

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact f32, k-ordered fma chain.
// lane l: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; C/D reg r: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
struct f32x16 {
    float v[16];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    ::emu::Lane* l = ::emu::S().cur;
    l->a = a;
    l->b = b;
    for (int i = 0; i < 16; i++) l->c[i] = c[i];
    ::emu::collective(::emu::OP_MFMA_32x32x2, 0, 0);
    f32x16 d;
    for (int i = 0; i < 16; i++) d[i] = l->d[i];
    return d;
}

// ---- scalar helpers ----
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline uint32_t __float_as_uint(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static inline float __uint_as_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

// ---- atomics (one lane runs at a time) ----
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { auto o = *p; *p = o + v; return o; }
static inline uint32_t atomicExch(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
static inline uint32_t atomicAnd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o & v; return o; }
static inline uint32_t atomicCAS(uint32_t* p, uint32_t cmp, uint32_t v) {
    uint32_t o = *p;
    if (o == cmp) *p = v;
    return o;
}
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) {
    unsigned long long o = *p;
    if (o == cmp) *p = v;
    return o;
}

// ---- a minimal HIP runtime over host memory ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };

static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* c) { *c = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    strcpy(p->gcnArchName, "gfx950:emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    void* m = nullptr;
    if (posix_memalign(&m, 256, n ? n : 256)) return hipErrorOutOfMemory;
    memset(m, 0xCD, n ? n : 256);
    *p = m;
    return hipSuccess;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline void __threadfence_system() {}
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
