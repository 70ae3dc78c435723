// simt.hpp -- TEST-ONLY lockstep emulation of one 64-lane wavefront on the host, so that the SIMT-style product headers
// (csrc/azul_common.hpp, azul_selfplay2.hpp, ...: per-lane scalar code + __builtin_amdgcn_* cross-lane builtins) compile
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
#include <math.h>
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
enum Op { OP_NONE = 0, OP_BALLOT, OP_READLANE, OP_BPERMUTE, OP_DPP, OP_SWAP16, OP_BARRIER, OP_READFIRST, OP_MFMA16X4, OP_WGBARRIER };

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
    int id;                     // wave of its workgroup (run_workgroup); 0 for run_wave
    bool at_wg_barrier;         // every live lane is parked at s_barrier: the wave waits for the workgroup's other waves
};

static Wave *g_wave = nullptr;

static inline unsigned lane_id() { return (unsigned)g_wave->cur; }
static inline unsigned wave_id() { return (unsigned)g_wave->id; }
static inline unsigned thread_id() { return 64u * wave_id() + lane_id(); }      // threadIdx.x of a one-dimensional workgroup

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
    case OP_MFMA16X4: {
        // v_mfma_f32_16x16x4_f32: D = A (16 x 4) * B (4 x 16) + C.  Lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16] and holds
        // C / D[4 (l / 16) + r][l % 16] for r = 0..3 (CDNA3 ISA guide, "MFMA 16x16x4 F32" register layout); the four products of an
        // element are added in ascending k with fused multiply-adds.  Executed with all 64 lanes (EXEC is ignored by the hardware too).
        for (int l = 0; l < W; l++) if (!in[l]) { fprintf(stderr, "simt: MFMA issued with lane %d inactive\n", l); abort(); }
        float A[16][4], B[4][16];
        for (int l = 0; l < W; l++) {
            uint32_t ua = (uint32_t)w->lane[l].a, ub = (uint32_t)w->lane[l].b;
            memcpy(&A[l % 16][l / 16], &ua, 4);
            memcpy(&B[l / 16][l % 16], &ub, 4);
        }
        for (int l = 0; l < W; l++) {
            Lane &x = w->lane[l];
            uint32_t cu[4] = {(uint32_t)x.c, (uint32_t)(x.c >> 32), (uint32_t)x.d, (uint32_t)(x.d >> 32)}, du[4];
            for (int r = 0; r < 4; r++) {
                float acc;
                memcpy(&acc, &cu[r], 4);
                const int i = 4 * (l / 16) + r, j = l % 16;
                for (int k = 0; k < 4; k++) acc = fmaf(A[i][k], B[k][j], acc);
                memcpy(&du[r], &acc, 4);
            }
            x.r0 = (uint64_t)du[0] | ((uint64_t)du[1] << 32);
            x.r1 = (uint64_t)du[2] | ((uint64_t)du[3] << 32);
        }
    } break;
    case OP_WGBARRIER:
        // s_barrier: handled by run_workgroup (the wave waits for the others); a PARTIAL group here is a divergent barrier
        for (int l = 0; l < W; l++) if (!in[l] && w->lane[l].state != 2) { fprintf(stderr, "simt: s_barrier reached by part of wave %d only\n", w->id); abort(); }
        w->at_wg_barrier = true;
        return;                                                      // (the lanes stay parked)
    default: fprintf(stderr, "simt: empty group\n"); abort();
    }
    for (int l = 0; l < W; l++) if (in[l]) w->lane[l].state = 0;
}

static Wave *wave_create(void (*fn)(void *), void *arg, int id, size_t stack_bytes)
{
    Wave *w = (Wave *)calloc(1, sizeof(Wave));
    w->fn = fn; w->arg = arg; w->cur = -1; w->id = id;
    for (int l = 0; l < W; l++) {
        Lane &x = w->lane[l];
        x.stack = (char *)malloc(stack_bytes);
        getcontext(&x.ctx);
        x.ctx.uc_stack.ss_sp = x.stack;
        x.ctx.uc_stack.ss_size = stack_bytes;
        x.ctx.uc_link = &w->main;
        g_wave = w;
        makecontext(&x.ctx, trampoline, 0);
    }
    return w;
}

static void wave_destroy(Wave *w)
{
    for (int l = 0; l < W; l++) free(w->lane[l].stack);
    free(w);
}

// run the wave until every lane has finished or the wave waits at a workgroup barrier; returns false when it has finished
static bool wave_advance(Wave *w, size_t stack_bytes)
{
    g_wave = w;
    for (;;) {
        bool any = false;
        for (int l = 0; l < W; l++) {
            Lane &x = w->lane[l];
            if (x.state != 0) continue;
            any = true;
            w->cur = l;
#if SIMT_ASAN
            __sanitizer_start_switch_fiber(&w->main_fake, x.stack, stack_bytes);
#endif
            swapcontext(&w->main, &x.ctx);
#if SIMT_ASAN
            __sanitizer_finish_switch_fiber(w->main_fake, nullptr, nullptr);
#endif
            w->cur = -1;
        }
        if (any) continue;
        // the earliest group -- but lanes waiting at s_barrier go LAST: a half of the wave that skipped a divergent region and ran on to
        // the barrier at the top of the NEXT loop iteration sits at a lower code address than the lanes still inside the region, and the
        // hardware reaches that barrier only after the region (address order is execution order only up to a loop's back edge)
        int first = -1;
        bool others = false;
        for (int l = 0; l < W; l++) if (w->lane[l].state == 1 && w->lane[l].op != OP_WGBARRIER) others = true;
        for (int l = 0; l < W; l++) {
            const Lane &x = w->lane[l];
            if (x.state != 1 || (others && x.op == OP_WGBARRIER)) continue;
            if (first < 0 || path_cmp(x, w->lane[first]) < 0) first = l;
        }
        if (first < 0) return false;                                 // every lane has finished
        bool in[W];
        for (int l = 0; l < W; l++) in[l] = w->lane[l].state == 1 && w->lane[l].op == w->lane[first].op && path_cmp(w->lane[l], w->lane[first]) == 0;
        execute_group(w, in);
        if (w->at_wg_barrier) return true;
    }
}

// "LDS" of the emulated kernels: every __shared__ array of the translation unit lives in the section simt_lds (hip/hip_runtime.h).  A
// workgroup starts with UNDEFINED LDS contents on the hardware: here with 0xA5 bytes.
extern "C" char __start_simt_lds[] __attribute__((weak, visibility("hidden")));
extern "C" char __stop_simt_lds[] __attribute__((weak, visibility("hidden")));
static void poison_lds()
{
    if (__start_simt_lds && __stop_simt_lds > __start_simt_lds) memset(__start_simt_lds, 0xA5, (size_t)(__stop_simt_lds - __start_simt_lds));
}

// run fn(arg) on 64 lanes in lockstep; returns the number of cross-lane operations executed
static uint64_t run_wave(void (*fn)(void *), void *arg)
{
    poison_lds();
    Wave *w = wave_create(fn, arg, 0, STACK_BYTES);
    if (wave_advance(w, STACK_BYTES)) { fprintf(stderr, "simt: s_barrier in a single-wave run\n"); abort(); }
    uint64_t n = w->collectives;
    wave_destroy(w);
    g_wave = nullptr;
    return n;
}

// run fn(arg) as ONE WORKGROUP of n_waves wavefronts (threadIdx.x = 64 wave + lane): the waves take turns, each running until it
// finishes or waits at s_barrier; when every unfinished wave waits there, all are released.  (Waves only meet at barriers: LDS
// traffic between them is ordered by those, like on the hardware.)  Returns the number of cross-lane operations executed.
static uint64_t run_workgroup(void (*fn)(void *), void *arg, int n_waves, size_t stack_bytes = 256u << 10)
{
    enum { MAXWAVES = 16 };
    if (n_waves < 1 || n_waves > MAXWAVES) { fprintf(stderr, "simt: 1..16 waves per workgroup\n"); abort(); }
    poison_lds();
    Wave *ws[MAXWAVES];
    bool done[MAXWAVES];
    for (int i = 0; i < n_waves; i++) { ws[i] = wave_create(fn, arg, i, stack_bytes); done[i] = false; }
    for (;;) {
        int waiting = 0, live = 0;
        for (int i = 0; i < n_waves; i++) {
            if (done[i]) continue;
            if (!ws[i]->at_wg_barrier) done[i] = !wave_advance(ws[i], stack_bytes);
            if (!done[i]) { live++; waiting += ws[i]->at_wg_barrier ? 1 : 0; }
        }
        if (live == 0) break;
        if (waiting != live) { fprintf(stderr, "simt: workgroup neither finished nor at a barrier\n"); abort(); }
        for (int i = 0; i < n_waves; i++) {
            if (done[i]) continue;
            ws[i]->at_wg_barrier = false;
            for (int l = 0; l < W; l++) if (ws[i]->lane[l].state == 1) ws[i]->lane[l].state = 0;     // (all of them are parked at the barrier)
        }
    }
    uint64_t n = 0;
    for (int i = 0; i < n_waves; i++) { n += ws[i]->collectives; wave_destroy(ws[i]); }
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
static inline void wg_barrier() { park(OP_WGBARRIER, 0); }
template <class V4>
static inline V4 mfma_f32_16x16x4f32(float a, float b, V4 c)
{
    uint32_t ua, ub, cu[4];
    memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    for (int r = 0; r < 4; r++) { float f = c[r]; memcpy(&cu[r], &f, 4); }
    park(OP_MFMA16X4, ua, ub, (uint64_t)cu[0] | ((uint64_t)cu[1] << 32), (uint64_t)cu[2] | ((uint64_t)cu[3] << 32));
    const Lane &me = g_wave->lane[g_wave->cur];
    const uint32_t du[4] = {(uint32_t)me.r0, (uint32_t)(me.r0 >> 32), (uint32_t)me.r1, (uint32_t)(me.r1 >> 32)};
    V4 d = c;
    for (int r = 0; r < 4; r++) { float f; memcpy(&f, &du[r], 4); d[r] = f; }
    return d;
}
// buffer resources: base + size; an access outside [0, size) reads 0 like the hardware's range check -- and is COUNTED, so that a
// test can require that the kernel never relies on it
struct Rsrc { const char *base; uint32_t bytes; };
static uint64_t g_buffer_oob = 0;
template <int N>
struct Words { uint32_t v[N]; };
template <int N>
static inline Words<N> buffer_load(Rsrc r, uint32_t voff, uint32_t soff)
{
    Words<N> out;
    const uint64_t at = (uint64_t)voff + soff;
    if (at + 4u * N > r.bytes) { g_buffer_oob++; memset(&out, 0, sizeof(out)); return out; }
    memcpy(&out, r.base + at, sizeof(out));
    return out;
}
static inline uint32_t mbcnt_lo(uint32_t mask, uint32_t base) { unsigned l = lane_id(); return base + (uint32_t)__builtin_popcount(mask & (l >= 32u ? 0xffffffffu : ((1u << l) - 1u))); }
static inline uint32_t mbcnt_hi(uint32_t mask, uint32_t base) { unsigned l = lane_id(); return base + (l > 32u ? (uint32_t)__builtin_popcount(mask & ((1u << (l - 32u)) - 1u)) : 0u); }

} // namespace simt
