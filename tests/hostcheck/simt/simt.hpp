// simt.hpp -- TEST-ONLY lockstep emulation of one 64-lane wavefront on the host, so that the SIMT-style product headers
// (csrc/azul_wave.hpp, azul_core.hpp, azul_selfplay2.hpp: per-lane scalar code + __builtin_amdgcn_* cross-lane builtins) compile
// UNMODIFIED with g++ and run in the build container: the benchmarked two-games-per-wave self-play step gets the same pre-GPU
// logic check against the oracle (and UBSan / ASan coverage) the one-game-per-wave core has.  Never part of libazulhip.so.
//
// Model: every lane is a fiber (ucontext) that runs the per-lane code until it reaches a cross-lane operation (ballot, readlane,
// ds_bpermute, DPP, permlane swap, wave barrier), where it parks with its operands.  When every live lane is parked, the scheduler
// releases ONE group: the lanes parked at the EARLIEST program point -- the call path (return addresses, outermost first)
// compared lexicographically; compiled -O0, code addresses inside a function follow source order, so "earliest" is the side of a
// divergent branch, or the body of a loop, that the hardware would also run before the lanes reconverge.  The released lanes are
// the active set (EXEC) of the operation: inactive lanes contribute nothing (a ballot sees 0, a read from one returns 0).
// A group whose members disagree on the operation is a bug in this emulation (or divergence the rule does not cover): abort.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#define SIMT_ASAN 1
#else
#define SIMT_ASAN 0
#endif

namespace simt {

enum { W = 64, MAXPATH = 48, STACK_BYTES = 1 << 20 };
enum Op { OP_NONE = 0, OP_BALLOT, OP_READLANE, OP_BPERMUTE, OP_DPP, OP_SWAP16, OP_BARRIER, OP_READFIRST };

struct Lane {
    ucontext_t ctx;
    char *stack;
    void *fake;                 // ASan fake-stack handle
    int state;                  // 0 runnable, 1 parked, 2 finished
    Op op;
    uintptr_t path[MAXPATH];
    int depth;
    uint64_t a, b, c, d;        // operands
    uint64_t r0, r1;            // results
    void *entry_fp;
};

struct Wave {
    Lane lane[W];
    ucontext_t main;
    void *main_fake;
    int cur;                    // lane running now (-1: scheduler)
    void (*fn)(void *);
    void *arg;
    uint64_t collectives;
};

static Wave *g_wave = nullptr;

static inline unsigned lane_id() { return (unsigned)g_wave->cur; }

static void to_scheduler()
{
    Wave *w = g_wave;
    Lane &me = w->lane[w->cur];
#if SIMT_ASAN
    __sanitizer_start_switch_fiber(&me.fake, nullptr, 0);       // (the scheduler runs on the thread's own stack; ASan looks it up)
#endif
    swapcontext(&me.ctx, &w->main);
#if SIMT_ASAN
    __sanitizer_finish_switch_fiber(me.fake, nullptr, nullptr);
#endif
}

static void trampoline()
{
    Wave *w = g_wave;
    Lane &me = w->lane[w->cur];
#if SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, nullptr, nullptr);
#endif
    me.entry_fp = __builtin_frame_address(0);
    w->fn(w->arg);
    me.state = 2;
#if SIMT_ASAN
    __sanitizer_start_switch_fiber(nullptr, nullptr, 0);        // this fiber never resumes
#endif
    swapcontext(&me.ctx, &w->main);
}

// park the calling lane at a cross-lane operation; returns when the scheduler has executed it for the lane's group
static __attribute__((noinline)) void park(Op op, uint64_t a, uint64_t b = 0, uint64_t c = 0, uint64_t d = 0)
{
    Lane &me = g_wave->lane[g_wave->cur];
    me.op = op; me.a = a; me.b = b; me.c = c; me.d = d;
    // call path, innermost first, then reversed: frames between this function and the fiber's entry
    uintptr_t tmp[MAXPATH];
    int n = 0;
    void **fp = (void **)__builtin_frame_address(0);
    while (fp && (void *)fp != me.entry_fp && n < MAXPATH) {
        tmp[n++] = (uintptr_t)fp[1];
        fp = (void **)fp[0];
    }
    if (n == MAXPATH) { fprintf(stderr, "simt: call path deeper than %d frames\n", MAXPATH); abort(); }
    me.depth = n;
    for (int i = 0; i < n; i++) me.path[i] = tmp[n - 1 - i];
    me.state = 1;
    to_scheduler();
}

static int path_cmp(const Lane &x, const Lane &y)
{
    int n = x.depth < y.depth ? x.depth : y.depth;
    for (int i = 0; i < n; i++)
        if (x.path[i] != y.path[i]) return x.path[i] < y.path[i] ? -1 : 1;
    return x.depth == y.depth ? 0 : (x.depth < y.depth ? -1 : 1);
}

static uint32_t dpp_source(unsigned l, unsigned ctrl, bool &valid)
{
    const unsigned row = l & ~15u, i = l & 15u;
    valid = true;
    if (ctrl <= 0xffu) return (l & ~3u) | ((ctrl >> (2u * (l & 3u))) & 3u);          // quad_perm
    if (ctrl >= 0x111u && ctrl <= 0x11fu) { unsigned n = ctrl - 0x110u; valid = i >= n; return row | (i - n); }     // row_shr:n
    if (ctrl >= 0x101u && ctrl <= 0x10fu) { unsigned n = ctrl - 0x100u; valid = i + n < 16u; return row | (i + n); }   // row_shl:n
    if (ctrl == 0x140u) return row | (15u - i);                                        // row_mirror
    if (ctrl == 0x141u) return row | ((i & 8u) | (7u - (i & 7u)));                      // row_half_mirror
    if (ctrl == 0x142u) { valid = row != 0u; return (row - 16u) | 15u; }                // row_bcast:15 (lane 15 of the previous row)
    if (ctrl == 0x143u) { valid = l >= 32u; return 31u; }                               // row_bcast:31
    fprintf(stderr, "simt: DPP control 0x%x is not emulated\n", ctrl);
    abort();
}

static void execute_group(Wave *w, const bool *in)
{
    Op op = OP_NONE;
    for (int l = 0; l < W; l++) if (in[l]) { op = w->lane[l].op; break; }
    for (int l = 0; l < W; l++)
        if (in[l] && w->lane[l].op != op) { fprintf(stderr, "simt: lanes of one group disagree on the operation (%d vs %d)\n", (int)op, (int)w->lane[l].op); abort(); }
    w->collectives++;
    switch (op) {
    case OP_BALLOT: {
        uint64_t m = 0;
        for (int l = 0; l < W; l++) if (in[l] && w->lane[l].a) m |= 1ull << l;
        for (int l = 0; l < W; l++) if (in[l]) w->lane[l].r0 = m;
    } break;
    case OP_READLANE:           // a = value, b = lane to read (the register of an inactive lane reads as 0 here)
        for (int l = 0; l < W; l++) if (in[l]) { unsigned s = (unsigned)w->lane[l].b & 63u; w->lane[l].r0 = in[s] ? w->lane[s].a : 0; }
        break;
    case OP_READFIRST: {
        int f = 0;
        while (!in[f]) f++;
        for (int l = 0; l < W; l++) if (in[l]) w->lane[l].r0 = w->lane[f].a;
    } break;
    case OP_BPERMUTE:           // a = byte address, b = value
        for (int l = 0; l < W; l++) if (in[l]) { unsigned s = ((unsigned)w->lane[l].a >> 2) & 63u; w->lane[l].r0 = in[s] ? w->lane[s].b : 0; }
        break;
    case OP_DPP:                // a = old, b = src, c = ctrl | row_mask << 16 | bank_mask << 20 | bound_ctrl << 24
        for (int l = 0; l < W; l++) if (in[l]) {
            Lane &x = w->lane[l];
            const unsigned ctrl = (unsigned)x.c & 0xffffu, rm = ((unsigned)x.c >> 16) & 15u, bm = ((unsigned)x.c >> 20) & 15u;
            const bool bound = (((unsigned)x.c >> 24) & 1u) != 0u;
            bool valid;
            unsigned s = dpp_source((unsigned)l, ctrl, valid);
            const bool enabled = ((rm >> (l >> 4)) & 1u) && ((bm >> ((l >> 2) & 3)) & 1u);
            if (!enabled) x.r0 = x.a;                               // row / bank masked off: the destination keeps `old`
            else if (valid && in[s]) x.r0 = w->lane[s].b;
            else x.r0 = bound ? 0 : x.a;                             // no source lane: 0 with bound_ctrl, else `old`
        }
        break;
    case OP_SWAP16:             // v_permlane16_swap: odd rows of the first operand <-> even rows of the second; a = vdst, b = vsrc
        for (int l = 0; l < W; l++) if (in[l]) {
            Lane &x = w->lane[l];
            const unsigned rowi = (unsigned)l >> 4, p = (unsigned)l ^ 16u;
            if (rowi & 1u) { x.r0 = in[p] ? w->lane[p].b : 0; x.r1 = x.b; }     // odd row: new vdst = vsrc of the even row below it
            else { x.r0 = x.a; x.r1 = in[p] ? w->lane[p].a : 0; }               // even row: new vsrc = vdst of the odd row above it
        }
        break;
    case OP_BARRIER: break;
    default: fprintf(stderr, "simt: empty group\n"); abort();
    }
    for (int l = 0; l < W; l++) if (in[l]) w->lane[l].state = 0;
}

// run fn(arg) on 64 lanes in lockstep; returns the number of cross-lane operations executed
static uint64_t run_wave(void (*fn)(void *), void *arg)
{
    Wave *w = (Wave *)calloc(1, sizeof(Wave));
    g_wave = w;
    w->fn = fn; w->arg = arg; w->cur = -1;
    for (int l = 0; l < W; l++) {
        Lane &x = w->lane[l];
        x.stack = (char *)malloc(STACK_BYTES);
        getcontext(&x.ctx);
        x.ctx.uc_stack.ss_sp = x.stack;
        x.ctx.uc_stack.ss_size = STACK_BYTES;
        x.ctx.uc_link = &w->main;
        makecontext(&x.ctx, trampoline, 0);
    }
    for (;;) {
        bool any = false;
        for (int l = 0; l < W; l++) {
            Lane &x = w->lane[l];
            if (x.state != 0) continue;
            any = true;
            w->cur = l;
#if SIMT_ASAN
            __sanitizer_start_switch_fiber(&w->main_fake, x.stack, STACK_BYTES);
#endif
            swapcontext(&w->main, &x.ctx);
#if SIMT_ASAN
            __sanitizer_finish_switch_fiber(w->main_fake, nullptr, nullptr);
#endif
            w->cur = -1;
        }
        if (any) continue;
        int first = -1;
        for (int l = 0; l < W; l++) if (w->lane[l].state == 1 && (first < 0 || path_cmp(w->lane[l], w->lane[first]) < 0)) first = l;
        if (first < 0) break;                                        // every lane has finished
        bool in[W];
        for (int l = 0; l < W; l++) in[l] = w->lane[l].state == 1 && path_cmp(w->lane[l], w->lane[first]) == 0;
        execute_group(w, in);
    }
    uint64_t n = w->collectives;
    for (int l = 0; l < W; l++) free(w->lane[l].stack);
    free(w);
    g_wave = nullptr;
    return n;
}

// ---- the cross-lane builtins of gfx950, as seen by one lane -----------------------------------------------------------------
static inline uint64_t ballot(bool p) { park(OP_BALLOT, p ? 1 : 0); return g_wave->lane[g_wave->cur].r0; }
static inline int readlane(int v, int l) { park(OP_READLANE, (uint32_t)v, (uint32_t)l); return (int)(uint32_t)g_wave->lane[g_wave->cur].r0; }
static inline int readfirstlane(int v) { park(OP_READFIRST, (uint32_t)v); return (int)(uint32_t)g_wave->lane[g_wave->cur].r0; }
static inline int ds_bpermute(int addr, int v) { park(OP_BPERMUTE, (uint32_t)addr, (uint32_t)v); return (int)(uint32_t)g_wave->lane[g_wave->cur].r0; }
static inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    park(OP_DPP, (uint32_t)old, (uint32_t)src, (uint32_t)ctrl | ((uint32_t)row_mask << 16) | ((uint32_t)bank_mask << 20) | ((uint32_t)bound_ctrl << 24));
    return (int)(uint32_t)g_wave->lane[g_wave->cur].r0;
}
struct Pair { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
static inline Pair permlane16_swap(uint32_t vdst, uint32_t vsrc, bool, bool)
{
    park(OP_SWAP16, vdst, vsrc);
    Pair p = {{(uint32_t)g_wave->lane[g_wave->cur].r0, (uint32_t)g_wave->lane[g_wave->cur].r1}};
    return p;
}
static inline void wave_barrier() { park(OP_BARRIER, 0); }
static inline uint32_t mbcnt_lo(uint32_t mask, uint32_t base) { unsigned l = lane_id(); return base + (uint32_t)__builtin_popcount(mask & (l >= 32u ? 0xffffffffu : ((1u << l) - 1u))); }
static inline uint32_t mbcnt_hi(uint32_t mask, uint32_t base) { unsigned l = lane_id(); return base + (l > 32u ? (uint32_t)__builtin_popcount(mask & ((1u << (l - 32u)) - 1u)) : 0u); }

} // namespace simt
