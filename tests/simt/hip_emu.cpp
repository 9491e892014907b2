// hip_emu.cpp — scheduler of the CPU lockstep emulator (TEST INFRASTRUCTURE, see hip_emu.hpp).
#include "hip_emu.hpp"

#include <algorithm>

namespace emu {

State& S() {
    static thread_local State s;   // one emulated device context per host thread (sharded searches run in threads)
    return s;
}

// void idist_emu_switch(void** save_sp, void* new_sp): save callee-saved registers, swap stacks
asm(R"(
.text
.globl idist_emu_switch
.type idist_emu_switch,@function
idist_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size idist_emu_switch, .-idist_emu_switch
)");

static void lane_entry() {
    State& s = S();
    (*s.body)();
    s.cur->alive = false;
    yield_to_sched();
    abort();  // a dead lane is never resumed
}

static constexpr size_t kStack = 256 * 1024;

static void init_lane(Lane& l, uint32_t tid) {
    if (!l.stack) {
        void* m = nullptr;
        if (posix_memalign(&m, 64, kStack)) abort();
        l.stack = (uint8_t*)m;
    }
    uintptr_t top = ((uintptr_t)l.stack + kStack) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)top;
    *--sp = 0;                        // fake return address of lane_entry
    *--sp = (uint64_t)&lane_entry;    // `ret` target
    for (int i = 0; i < 6; i++) *--sp = 0;  // rbp rbx r12 r13 r14 r15
    l.sp = sp;
    l.tid = Dim3{tid, 0, 0};
    l.alive = true;
    l.waiting = false;
    l.op = OP_NONE;
}

static const char* op_name(int op) {
    switch (op) {
        case OP_SYNC: return "__syncthreads";
        case OP_WSYNC: return "wave_sync";
        case OP_BALLOT: return "__ballot";
        case OP_SHFL: return "__shfl";
        case OP_SHFL_XOR: return "__shfl_xor";
        case OP_READFIRST: return "readfirstlane";
        case OP_DPP: return "dpp";
        default: return "none";
    }
}

void launch(uint32_t grid, uint32_t block, size_t smem_bytes, const std::function<void()>& body) {
    State& s = S();
    const char* ord = getenv("IDIST_EMU_ORDER");
    s.reverse = ord && ord[0] == 'r';
    if (s.lanes.size() < block) s.lanes.resize(block);
    void* sm = nullptr;
    if (posix_memalign(&sm, 256, smem_bytes + 256)) abort();
    s.smem = (uint8_t*)sm;
    s.body = &body;
    s.bdim = Dim3{block, 1, 1};
    s.gdim = Dim3{grid, 1, 1};
    for (uint32_t b = 0; b < grid; b++) {
        s.blk = Dim3{b, 0, 0};
        memset(s.smem, 0xCD, smem_bytes + 256);
        for (uint32_t i = 0; i < block; i++) init_lane(s.lanes[i], i);
        for (;;) {
            for (uint32_t k = 0; k < block; k++) {
                Lane& l = s.lanes[s.reverse ? block - 1 - k : k];
                if (l.alive && !l.waiting) {
                    s.cur = &l;
                    idist_emu_switch(&s.sched_sp, l.sp);
                }
            }
            // every live lane now waits at a collective: wave collectives resolve per wave (64 lanes),
            // __syncthreads when the whole workgroup has arrived
            uint32_t n_alive = 0;
            for (uint32_t i = 0; i < block; i++) n_alive += s.lanes[i].alive ? 1u : 0u;
            if (!n_alive) break;
            bool progressed = false;
            const uint32_t n_waves = (block + 63) / 64;
            bool all_sync = true;
            for (uint32_t w = 0; w < n_waves; w++) {
                const uint32_t lo = w * 64, hi = std::min(block, lo + 64);
                int op = OP_NONE;
                uint32_t first = hi, alive = 0;
                for (uint32_t i = lo; i < hi; i++) {
                    Lane& l = s.lanes[i];
                    if (!l.alive) continue;
                    if (first == hi) first = i;
                    alive++;
                    if (op == OP_NONE) op = l.op;
                    else if (op != l.op) {
                        fprintf(stderr, "[hip_emu] DIVERGENT COLLECTIVE in block %u wave %u: lane %u at %s, lane %u at %s\n", b, w,
                                first, op_name(op), i, op_name(l.op));
                        abort();
                    }
                }
                if (!alive) continue;
                if (op == OP_SYNC) continue;
                all_sync = false;
                if (alive != hi - lo && getenv("IDIST_EMU_STRICT")) {
                    fprintf(stderr, "[hip_emu] collective %s with %u/%u lanes alive\n", op_name(op), alive, hi - lo);
                    abort();
                }
                s.n_collectives++;
                uint64_t ballot = 0;
                if (op == OP_BALLOT)
                    for (uint32_t i = lo; i < hi; i++)
                        if (s.lanes[i].alive && s.lanes[i].val) ballot |= 1ull << (i - lo);
                for (uint32_t i = lo; i < hi; i++) {
                    Lane& l = s.lanes[i];
                    if (!l.alive) continue;
                    const uint32_t li = i - lo;
                    switch (op) {
                        case OP_BALLOT: l.res = ballot; break;
                        case OP_SHFL: { const Lane& o = s.lanes[lo + ((uint32_t)l.arg & 63u)]; l.res = (lo + ((uint32_t)l.arg & 63u) < hi && o.alive) ? o.val : 0; break; }
                        case OP_SHFL_XOR: { const uint32_t j = lo + (li ^ (uint32_t)l.arg); l.res = (j < hi && s.lanes[j].alive) ? s.lanes[j].val : l.val; break; }
                        case OP_READFIRST: l.res = s.lanes[first].val; break;
                        case OP_DPP: {
                            const uint32_t ctrl = (uint32_t)l.arg;
                            int64_t src = -1;
                            if (ctrl <= 0xFFu) src = (int64_t)((li & ~3u) + ((ctrl >> (2u * (li & 3u))) & 3u));   // quad_perm
                            else if (ctrl >= 0x101u && ctrl <= 0x10Fu) {                                          // row_shl:n
                                const uint32_t j = li + (ctrl & 15u);
                                if ((j >> 4) == (li >> 4)) src = j;
                            } else if (ctrl >= 0x111u && ctrl <= 0x11Fu) {                                        // row_shr:n
                                const uint32_t n = ctrl & 15u;
                                if ((li & 15u) >= n) src = li - n;
                            } else if (ctrl == 0x140u) src = (int64_t)((li & ~15u) + (15u - (li & 15u)));         // row_mirror
                            else if (ctrl == 0x141u) src = (int64_t)((li & ~7u) + (7u - (li & 7u)));              // row_half_mirror
                            else { fprintf(stderr, "[hip_emu] dpp ctrl 0x%x not emulated\n", ctrl); abort(); }
                            if (src < 0 || lo + (uint32_t)src >= hi || !s.lanes[lo + (uint32_t)src].alive) l.res = 1ull << 32;
                            else l.res = (uint32_t)s.lanes[lo + (uint32_t)src].val;
                            break;
                        }
                        case OP_MFMA_32x32x2: {
                            const uint32_t col = li & 31u;
                            for (int r = 0; r < 16; r++) {
                                const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (li >> 5);
                                float acc = l.c[r];
                                for (uint32_t k = 0; k < 2; k++)
                                    acc = fmaf(s.lanes[lo + k * 32 + row].a, s.lanes[lo + k * 32 + col].b, acc);
                                l.d[r] = acc;
                            }
                            break;
                        }
                        default: break;
                    }
                    l.waiting = false;
                }
                progressed = true;
            }
            if (!progressed) {
                if (!all_sync) { fprintf(stderr, "[hip_emu] deadlock in block %u\n", b); abort(); }
                s.n_collectives++;
                for (uint32_t i = 0; i < block; i++) { s.lanes[i].res = 0; s.lanes[i].waiting = false; }
            }
        }
    }
    free(sm);
    s.smem = nullptr;
    s.body = nullptr;
}

}  // namespace emu
