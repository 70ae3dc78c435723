// azul_selfplay_kernels.hpp -- the kernels of the self-play path as a header: the per-game stream seeding, the two-player rule kernel
// (azul_op_kernel on azul_ops2.hpp), the discounted-returns scans (one window / a ring of windows) and THE BENCHMARKED KERNEL,
// azul_selfplay2_kernel (two games per wavefront, csrc/azul_selfplay2.hpp).  azul_kernels.hip includes this file;
// tests/hostcheck/simt_selfplay2.cpp compiles it UNMODIFIED with g++ and runs these kernels under the lockstep wave emulation, so the CPU
// check and the sanitizer passes cover the kernels themselves (XCD-aware game placement, staging, the move loop), not a restatement.
#pragma once
#include "azul_ops2.hpp"

__global__ void __launch_bounds__(64) azul_seed_kernel(BatchDev b, u64 seed_base, const u64 *seeds)
{
    u32 g = blockIdx.x * 64u + threadIdx.x;
    if (g >= b.n) return;
    u64 seed = seeds ? seeds[g] : seed_base + (u64)g;
    seed_stream(b.mt + (size_t)g * 624u, seed);
    b.mtpos[g] = 624u;
}

// One rule call per game: grid = ceil(a.count / 2) one-wave workgroups, two games per wavefront (azul_ops2.hpp).
template <bool LID>
__global__ void __launch_bounds__(64) azul_op_kernel(BatchDev b, OpArgs a)
{
    __shared__ u32 mt_lds[2][az2::MT_LDS_WORDS];
    __shared__ double2 tabfs_lds[T_PAIRS];
    __shared__ float obs_lds[2][OP2_OBS_STRIDE];
    op_body2<LID>(b, a, blockIdx.x, mt_lds, tabfs_lds, obs_lds);
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

// DIAGNOSTIC: the shader clock the device really runs at, measured ON the device: one wave runs a fixed chain of dependent vector operations
// between two readings of s_memtime (shader-clock cycles) and s_memrealtime (a constant 100 MHz counter); out[0] / out[1] x 100 MHz is the
// clock the wave saw.  bench.py launches it between the blocks of its sustained phase, beside the driver's reported clock.
__global__ void __launch_bounds__(64) azul_clock_probe_kernel(u64 *out, u32 iters)
{
    const u64 r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    u32 v = threadIdx.x;
#pragma unroll 1
    for (u32 i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    asm volatile("" : "+v"(v));
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (u64)v; }
}

// The C1 wire record of the opt-in trajectory all-gather (BASELINE configs[4]; what NNRunner.run_episode keeps per agent step, nn_runner.py:17-47,
// and NNRunner.train concatenates, nn_runner.py:59-78), 184 bytes = 46 dwords per (step, game) cell -- layout in include/azul_hip.h.  One
// thread per OUTPUT dword: consecutive threads write consecutive dwords (coalesced 256-byte rows per wave) and read consecutive 16-byte
// pieces of the observation; HBM-bound: 544 + 180 + 26 bytes in, 184 out per cell.
struct PackC1Args {
    const float *obs;        // [cells][136]
    const uint8_t *mask;     // [cells][180]
    const uint8_t *player;   // [cells]
    const i32 *action;       // [cells]
    const i32 *reward;
    const uint8_t *done;
    const float *value, *logp, *entropy, *returns;
    u32 *out;                // [cells][46]
    u32 cells;
};

__global__ void __launch_bounds__(256) azul_pack_c1_kernel(PackC1Args a)
{
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.cells * 46u) return;
    const u32 cell = i / 46u, d = i - 46u * cell;
    u32 w = 0;
    if (d < 34u) {                                       // bytes 0..135: the observation's integers (0..255 for every state the rules reach)
        const float4 v = *(const float4 *)(a.obs + (size_t)cell * 136u + 4u * d);
        w = ((u32)(i32)v.x & 0xffu) | (((u32)(i32)v.y & 0xffu) << 8) | (((u32)(i32)v.z & 0xffu) << 16) | (((u32)(i32)v.w & 0xffu) << 24);
    } else if (d < 40u) {                                // bytes 136..159: the 180 legal-move bits, bit a & 7 of byte a >> 3
        const u32 k = d - 34u, nd = k < 5u ? 8u : 5u;    // actions 32 k .. 32 k + 31 = dwords 8 k .. of the 45-dword mask row
        const u32 *row = (const u32 *)(a.mask + (size_t)cell * 180u) + 8u * k;
#pragma unroll
        for (u32 j = 0; j < 8u; j++) {
            const u32 m = j < nd ? row[j] : 0u;
            w |= ((m & 0xffu) != 0u ? 1u : 0u) << (4u * j);
            w |= ((m & 0xff00u) != 0u ? 1u : 0u) << (4u * j + 1u);
            w |= ((m & 0xff0000u) != 0u ? 1u : 0u) << (4u * j + 2u);
            w |= ((m & 0xff000000u) != 0u ? 1u : 0u) << (4u * j + 3u);
        }
    } else if (d == 40u) {
        const i32 av = a.action[cell];
        w = (av < 0 ? 0xffu : (u32)av & 0xffu) | ((u32)a.done[cell] << 8) | ((u32)a.player[cell] << 16);
    } else if (d == 41u) w = (u32)a.reward[cell];
    else {
        const float *src = d == 42u ? a.value : d == 43u ? a.logp : d == 44u ? a.entropy : a.returns;
        w = __builtin_bit_cast(u32, src[cell]);
    }
    a.out[i] = w;
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


// Flat self-play, TWO GAMES PER WAVEFRONT (azul_selfplay2.hpp): grid = ceil(N / 2) one-wave workgroups; lanes 0..31 own game
// 2 b, lanes 32..63 game 2 b + 1.  `mask_stride` is the byte distance between the mask rows of consecutive games (180, or 192 to
// keep every row 64-byte aligned).
template <bool LID, int OUT, bool PAD, bool BITS, bool LIM = false /* the batch has a move limit: az2::after_move2 */>
__global__ void __launch_bounds__(64) azul_selfplay2_kernel(BatchDev b, TrajArgs t, u32 mask_stride)
{
    __shared__ u32 mt_lds[2][az2::MT_LDS_WORDS];          // (+ the move limit: az2::rng2_set_move_limit)
    __shared__ u32 mtt_lds[2][624];                        // the same words tempered (az2::Rng2::tlds)
    __shared__ double2 tabfs_lds[T_PAIRS];                 // {Fr[J][b], S[J]} + the floor-only pairs: both table values of a decision in one 16-byte read
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    for (u32 i = lane; i < (u32)T_PAIRS; i += 64u) tabfs_lds[i] = b.tab[i];
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
    az2::rng2_set_move_limit(mt_lds[half], b.move_limit, l);
    az2::Tab2 tab = {tabfs_lds};
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
    // A uniform counted loop (scalar loop control: a per-game `break` costs ~16 exec-mask instructions per move).  A game stopped by a
    // rule error (box and lid empty when a round has to be dealt) stays as it is: its lanes skip the later moves.  Such a state cannot be
    // reached by play -- 100 tiles, at most 50 on the walls and 30 in the pattern lines when a round is dealt leave 20 for box + lid -- only
    // handed in; the LIM instantiation also marks the skipped slots like stuck slots (action -1, done 2, counted in `stuck`), the
    // default one does not: every form of that bookkeeping tried cost the benchmarked kernel 0.8 .. 1.5 % (profiles/round6_headline_ab.txt).
    // The host therefore launches the LIM instantiation (limit 0 = none) for every batch it has written records into (azul_kernels.hip:
    // azul_batch::handed_in), so the default one only ever sees states that play produced.
    bool dead = false;               // (set inside the rare blocks only: the common path carries no test for it)
    if (LIM) {
        u32 skipped = 0;
#pragma unroll 1
        for (int s = 0; s < t.n_steps; s++) {
            if (!dead) az2::selfplay_step2<LID, OUT, PAD, BITS, true>(g, b.rules.first_player, k, r, tab, margin, cnt, o, pp, dead);
            else skipped += 1u;
            o.e += b.n;
        }
        if (AZ_UNLIKELY(az2::wave_any(dead))) {
            if (dead) az2::dead_slots2<OUT, PAD, BITS>(g, o, cnt, b.n, o.e, skipped, l);
        }
    } else {
#pragma unroll 1
        for (int s = 0; s < t.n_steps; s++) {
            if (!dead) az2::selfplay_step2<LID, OUT, PAD, BITS, false>(g, b.rules.first_player, k, r, tab, margin, cnt, o, pp, dead);
            o.e += b.n;
        }
    }
#if defined(AZ_PROFILE_SEGMENTS)
    if (lane == 0u) for (int q = 0; q < SEG_COUNT; q++) atomicAdd((unsigned long long *)(b.prof + q), (unsigned long long)prof.acc[q]);
#endif
    az2::g2_store(g, rec, l);
    az2::rng2_close(r, gmt, b.mtpos + gi, l);
    az2::counters2_close(cnt, l);
}
