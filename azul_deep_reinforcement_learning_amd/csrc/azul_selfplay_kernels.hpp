// azul_selfplay_kernels.hpp -- the small kernels of the self-play path as a header: the per-game stream seeding, the discounted-returns
// scans (one window / a ring of windows), the one-game-per-wave self-play kernel of round 1 (AZUL_SELFPLAY_KERNEL=1: the A/B partner) and
// THE BENCHMARKED KERNEL, azul_selfplay2_kernel (two games per wavefront, csrc/azul_selfplay2.hpp).  azul_kernels.hip includes this file;
// tests/hostcheck/simt_selfplay2.cpp compiles it UNMODIFIED with g++ and runs these kernels under the lockstep wave emulation, so the CPU
// check and the sanitizer passes cover the kernels themselves (XCD-aware game placement, staging, the move loop), not a restatement.
#pragma once

__global__ void __launch_bounds__(64) azul_seed_kernel(BatchDev b, u64 seed_base, const u64 *seeds)
{
    u32 g = blockIdx.x * 64u + threadIdx.x;
    if (g >= b.n) return;
    u64 seed = seeds ? seeds[g] : seed_base + (u64)g;
    seed_stream(b.mt + (size_t)g * 624u, seed);
    b.mtpos[g] = 624u;
}

template <bool LID>
__global__ void __launch_bounds__(64) azul_op_kernel(BatchDev b, OpArgs a)
{
    __shared__ u32 mt_lds[624];
    __shared__ double fr_lds[T_ROWS * T_BINADES];
    op_body<LID>(b, a, blockIdx.x, mt_lds, fr_lds);
}

// Discounted returns over the time-major trajectory of one launch window (reference loop: nn_runner.py:70-76,
// qval = reward + gamma * qval backwards within an episode).  One thread per game walks its column backwards;
// `done[t][g] != 0` ends an episode at move t; `carry[g]` holds the return flowing in from the NEXT window
// (0 for a window that ends with finished episodes), and receives the value flowing out of this window's start.
__global__ void __launch_bounds__(256) azul_returns_kernel(const i32 *reward, const uint8_t *done, float *out, float *carry,
                                                           float gamma, int n_steps, u32 n)
{
    u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n) return;
    float q = carry ? carry[g] : 0.f;
    for (int t = n_steps - 1; t >= 0; t--) {
        size_t i = (size_t)t * n + g;
        if (done[i]) q = 0.f;
        q = (float)reward[i] + gamma * q;
        out[i] = q;
    }
    if (carry) carry[g] = q;
}


// The same scan over a RING of time slots (absolute step s lives in slot s % ring_steps): one launch walks from the newest step
// s_end - 1 back over `span` steps, so the return flowing out of a window's first step chains into the window before it.
__global__ void __launch_bounds__(64) azul_returns_ring_kernel(const i32 *__restrict__ reward, const uint8_t *__restrict__ done, float *__restrict__ out,
                                                               float gamma, int ring_steps, int s_end, int span, u32 n)
{
    u32 g = blockIdx.x * 64u + threadIdx.x;
    if (g >= n) return;
    float q = 0.f;
    int slot = (s_end - 1) % ring_steps;
    // the scan itself is a short dependent chain; what costs is memory latency: sixteen steps' rewards and flags are requested
    // together (unconditionally: past the span the last slot is read again), then folded in
    constexpr int RB = 16;
#pragma unroll 1
    for (int j = 0; j < span; j += RB) {
        i32 r[RB];
        u32 d[RB], at[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            at[b] = (u32)slot * n + g;
            r[b] = reward[at[b]];
            d[b] = done[at[b]];
            if (j + b + 1 < span) slot = slot == 0 ? ring_steps - 1 : slot - 1;
        }
#pragma unroll
        for (int b = 0; b < RB; b++) {
            if (j + b < span) {
                q = (float)r[b] + gamma * (d[b] ? 0.f : q);
                out[at[b]] = q;
            }
        }
    }
}

struct TrajArgs {
    int n_steps;
    uint8_t *mask;     // [T][N][180]
    u64 *maskbits;     // [T][N][3]
    i32 *action;       // [T][N]
    i32 *reward;       // [T][N]
    uint8_t *done;     // [T][N]
    uint8_t *rec;      // [T][N][128]
    u32 *packed;       // [T][N]
};

template <bool LID, int OUT>
__global__ void __launch_bounds__(64) azul_selfplay_kernel(BatchDev b, TrajArgs t)
{
    __shared__ u32 mt_lds[624];
    __shared__ double fr_lds[T_ROWS * T_BINADES];
    const u32 gi = blockIdx.x;
    const size_t N = b.n;
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES;
    LaneConst k;
    lane_consts(k);
    SampleTab tab;
    sample_tab_load(tab, b.T, fr_lds);
    Game g;
    game_load(g, rec);
    game_prime<LID>(g, k);
    Rng r;
    rng_open(r, b.mt + (size_t)gi * 624u, mt_lds, b.mtpos[gi]);
    r.margin = b.draw_margin;
    Counters cnt = {b.episodes + gi, b.stuck + gi, b.stat_sum + (size_t)gi * 10};
    OutV ov;
    OutS os = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t sm = 0, sb = 0, sa = 0, sr = 0, sd = 0, sc = 0, sp = 0;
    if (OUT == 1) {
        outv_open(ov, gi, b.n, t.mask, t.maskbits, t.action, t.reward, t.done, t.packed);
    } else {
        outv_open(ov, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (OUT == 2) {
            os.mask = t.mask ? t.mask + (size_t)gi * AZUL_NUM_ACTIONS : nullptr;
            os.maskbits = t.maskbits ? t.maskbits + (size_t)gi * 3 : nullptr;
            os.action = t.action ? t.action + gi : nullptr;
            os.reward = t.reward ? t.reward + gi : nullptr;
            os.done = t.done ? t.done + gi : nullptr;
            os.rec = t.rec ? t.rec + (size_t)gi * AZUL_RECORD_BYTES : nullptr;
            os.packed = t.packed ? t.packed + gi : nullptr;
            sp = os.packed ? N : 0;
            sm = os.mask ? N * AZUL_NUM_ACTIONS : 0; sb = os.maskbits ? N * 3 : 0; sa = os.action ? N : 0;
            sr = os.reward ? N : 0; sd = os.done ? N : 0; sc = os.rec ? N * AZUL_RECORD_BYTES : 0;   // a NULL stream stays NULL
        }
    }
#if defined(AZ_PROFILE_SEGMENTS)
    SegProf prof;
    for (int q = 0; q < SEG_COUNT; q++) prof.acc[q] = 0;
    prof.last = __builtin_amdgcn_s_memtime();
    SegProf *pp = &prof;
#else
    SegProf *pp = nullptr;
#endif
#pragma unroll 1
    for (int s = 0; s < t.n_steps; s++) {
        u32 f = selfplay_step<LID, OUT>(g, b.rules.first_player, k, r, tab, cnt, ov, os, pp);
        if (f & 0x100u) break;      // rule error (box empty): leave the game as it is
        if (OUT == 1) outv_next(ov);
        if (OUT == 2) { os.mask += sm; os.maskbits += sb; os.action += sa; os.reward += sr; os.done += sd; os.rec += sc; os.packed += sp; }
    }
#if defined(AZ_PROFILE_SEGMENTS)
    if (wv::lane() == 0) for (int q = 0; q < SEG_COUNT; q++) atomicAdd((unsigned long long *)(b.prof + q), (unsigned long long)prof.acc[q]);
#endif
    game_store(g, rec);
    rng_close(r, b.mtpos + gi);
}


#include "azul_selfplay2.hpp"

// Flat self-play, TWO GAMES PER WAVEFRONT (azul_selfplay2.hpp): grid = ceil(N / 2) one-wave workgroups; lanes 0..31 own game
// 2 b, lanes 32..63 game 2 b + 1.  Same semantics and outputs as azul_selfplay_kernel; `mask_stride` is the byte distance between
// the mask rows of consecutive games (180, or 192 to keep every row 64-byte aligned).
template <bool LID, int OUT, bool PAD, bool BITS>
__global__ void __launch_bounds__(64) azul_selfplay2_kernel(BatchDev b, TrajArgs t, u32 mask_stride)
{
    __shared__ u32 mt_lds[2][624];
    __shared__ u32 mtt_lds[2][624];                        // the same words tempered (az2::Rng2::tlds)
    __shared__ double tab_lds[T_WORDS];
    __shared__ double2 tabfs_lds[T_ROWS * T_BINADES];      // {Fr[J][b], S[J]}: both table values of a decision in one 16-byte read
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    for (u32 i = lane; i < (u32)T_WORDS; i += 64u) tab_lds[i] = b.T[i];
    for (u32 i = lane; i < (u32)(T_ROWS * T_BINADES); i += 64u) tabfs_lds[i] = make_double2(b.T[i], b.T[T_ROWS * T_BINADES + i / T_BINADES]);
    az2::lds_sync();
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b runs on XCD b % 8.
    // Give every XCD a CONTIGUOUS range of games: the waves that share a cache line of a time-major stream (32 games of an int32
    // stream, 2 / 3 of a mask row pair) then write it through ONE L2, which merges them into whole-line HBM writes.
    const u32 nb = gridDim.x, xcd = blockIdx.x & 7u, q8 = nb >> 3, rem = nb & 7u;
    const u32 wave_id = xcd * q8 + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const u32 gi = wave_id * 2u + half;
    if (gi >= b.n) return;                               // odd batch: the last wave plays one game
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES;
    az2::K2 k;
    az2::k2_init(k);
    az2::Tab2 tab = {tab_lds, tab_lds + T_ROWS * T_BINADES, tabfs_lds};
    az2::G2 g;
    az2::g2_load(g, rec, l);
    az2::prime2(g, k);
    az2::Rng2 r;
    u32 *gmt = b.mt + (size_t)gi * 624u;
    az2::rng2_open(r, gmt, mt_lds[half], b.mtpos[gi], l);
    az2::rng2_attach_tempered(r, mtt_lds[half], l);
    const u64 margin = b.draw_margin;
    az2::Counters2 cnt;
    az2::counters2_open(cnt, b.episodes + gi, b.stuck + gi, b.stat_sum + (size_t)gi * 10, l);
    az2::Out2 o = {t.mask, t.maskbits, t.action, t.reward, t.done, t.packed, t.rec, mask_stride, gi,
                   l == 0u ? (u32 *)t.action : (l == 1u ? (u32 *)t.reward : t.packed)};
#if defined(AZ_PROFILE_SEGMENTS)
    SegProf prof;
    for (int q = 0; q < SEG_COUNT; q++) prof.acc[q] = 0;
    prof.last = __builtin_amdgcn_s_memtime();
    SegProf *pp = &prof;
#else
    SegProf *pp = nullptr;
#endif
#if !defined(AZ2_ROTATED_LOOP)    // default: one selfplay_step2 per move; -DAZ2_ROTATED_LOOP: the rotated loop (measured 2 % slower)
    // A uniform counted loop (scalar loop control: a per-game `break` costs ~16 exec-mask instructions per move).  A game stopped by a
    // rule error (box and lid empty when a round has to be dealt: crafted states only) stays as it is: its lanes skip the later moves.
    bool dead = false;               // (set inside the rare blocks only: the common path carries no test for it)
#pragma unroll 1
    for (int s = 0; s < t.n_steps; s++) {
        if (!dead) az2::selfplay_step2<LID, OUT, PAD, BITS>(g, b.rules.first_player, k, r, tab, margin, cnt, o, pp, dead);
        o.e += b.n;
    }
#else
    az2::Prep2 P;
    az2::prepare2(g, k, r, tab, P);
#pragma unroll 1
    for (int s = 0; s < t.n_steps; s++) {
        u32 f = az2::selfplay_rotated2<LID, OUT, PAD, BITS>(g, P, b.rules.first_player, k, r, tab, margin, cnt, o, pp);
        if (f & 0x100u) break;      // rule error (box empty): leave the game as it is
        o.e += b.n;
    }
#endif
#if defined(AZ_PROFILE_SEGMENTS)
    if (lane == 0u) for (int q = 0; q < SEG_COUNT; q++) atomicAdd((unsigned long long *)(b.prof + q), (unsigned long long)prof.acc[q]);
#endif
    az2::g2_store(g, rec, l);
    az2::rng2_close(r, gmt, b.mtpos + gi, l);
    az2::counters2_close(cnt, l);
}
