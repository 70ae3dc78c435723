// azul_rollout2.hpp -- the persistent policy rollout (row N1 / N2: the batched NNRunner.run_episode loop, nn_runner.py:17-47) with the
// ENV SIDE ON THE VECTOR PIPE: a workgroup of 8 waves owns 16 games for a whole window, wave w plays games 2w and 2w + 1 in its two
// 32-lane halves with azul_selfplay2.hpp's rules (state in VGPRs, half-uniform), and the same 8 waves run the network on the f32
// matrix cores between the env phases.  Included by azul_kernels.hip after azul_policy.hpp.
//
// Why on the vector pipe: rounds 1-2 ran sixteen one-game waves per workgroup whose rule code sat on the CU's ONE scalar ALU, and every
// move waited for the slowest of the sixteen games (env step 6.2 k + waiting 7.2 k of 33.2 k cycles per move; LABNOTES.md).  Here a move's
// env work is ~500 vector instructions per PAIR of games.
//
// Arithmetic per output element: the same k-ordered v_mfma_f32_16x16x4_f32 chain per hidden unit / logit, the same critic summation and
// the same head as azul_policy_forward_kernel, and the rules are azul_selfplay2.hpp's: the trajectories are bit-identical to the
// two-launches-per-move path (tests/test_policy_bridge.py, tests/test_full_size_configs.py).
// Three opponents (template parameter OPP): 0 the policy moves for both players; 1 GameRunner with its default RandomAgent inside the env
// phase; 2 GameRunner(opponent=Agent(...)) (game_runner.py:27-30) -- a SECOND network: after the agent's move the workgroup runs matrix phases
// on the opponent's weights while any of its 16 games owes an opponent_move() (azul_env2.hpp: the NET_* protocol).
// Reference lines: azulnet/azul.py:296-313 (step), azulnet/game_runner.py:27-30, 37-55 (opponent_move, GameRunner.step), :56-72 (get_state),
// :76-85 (reset), :87-97 (RandomAgent); azulnet/agent.py:64-81; azulnet/nn_runner.py:17-47.
#pragma once

#include "azul_env2.hpp"

constexpr u32 PR2_WAVES = 8;
constexpr u32 PR2_AHEAD = 6;       // k-steps of layer-1 weights in flight (3..6 measured alike, 8 and 12 slower: profiles/round3_policy_rollout_phases.txt)
constexpr int PR2_ADEPTH = 4;      // A fragments read from LDS this many k-steps ahead

// The first weight fragments of a matrix phase are REQUESTED A PHASE EARLIER (layer 1's before the env step, layer 2's before layer 1's
// epilogue) and stay in flight across the LDS-only barriers: the matrix pipe does not wait for L2 after each barrier.
constexpr int PR2_HOIST1 = 4;
constexpr int PR2_HOIST2 = 6;      // (2..6 measured alike, 8 and 12 slower: too many layer-2 fragments in flight hold up layer 1's tail)
#define PR2_LOAD1(vo, s) __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs1, vo, (4 * (s)) * PF_H2 * 4, 0))
// waves 0..3: FOUR adjacent hidden columns per lane and k-step in one 16-byte load (two column pairs)
#define PR2_LOAD1Q(vo, s) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, (4 * (s)) * PF_H2 * 4, 0))
#define PR2_LOAD2(NT_, s) ((NT_) == 2 ? __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs2, voff, (4 * (s)) * PF_ACT * 4, 0)) \
                                      : make_float2(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, voff, (4 * (s)) * PF_ACT * 4, 0)), 0.f))

__device__ __forceinline__ void pr2_request1(bool two, const __amdgpu_buffer_rsrc_t rs1, u32 voffA, u32 voffB, float2 (&preA)[PR2_HOIST1],
                                             float2 (&preB)[PR2_HOIST1])
{
    if (two) {
#pragma unroll
        for (int s = 0; s < PR2_HOIST1; s++) { const float4 t = PR2_LOAD1Q(voffA, s); preA[s] = make_float2(t.x, t.y); preB[s] = make_float2(t.z, t.w); }
    } else {
#pragma unroll
        for (int s = 0; s < PR2_HOIST1; s++) preA[s] = PR2_LOAD1(voffA, s);
    }
    (void)voffB;
}

__device__ __forceinline__ void pr2_request2(bool two, const __amdgpu_buffer_rsrc_t rs2, u32 voff, float2 (&pre)[PR2_HOIST2])
{
    if (two) {
#pragma unroll
        for (int s = 0; s < PR2_HOIST2; s++) pre[s] = PR2_LOAD2(2, s);
    } else {
#pragma unroll
        for (int s = 0; s < PR2_HOIST2; s++) pre[s] = PR2_LOAD2(1, s);
    }
}

// layer 1 of one wave: NP pairs of adjacent hidden columns per lane (8-byte loads), 136-deep, one 16-row tile
template <int NP>
__device__ __forceinline__ void pr2_layer1(const __amdgpu_buffer_rsrc_t rs1, u32 voffA, u32 voffB, const float *ap, const float2 (&preA)[PR2_HOIST1],
                                           const float2 (&preB)[PR2_HOIST1], pf_f32x4 (&acc)[4], u64 *t_loop = nullptr /* diagnostic build: stamp at the loop's entry */)
{
    float2 bwA[PF_IN / 4], bwB[PF_IN / 4];
#pragma unroll
    for (int s = 0; s < PR2_HOIST1; s++) { bwA[s] = preA[s]; if (NP == 2) bwB[s] = preB[s]; }
#pragma unroll
    for (int s = PR2_HOIST1; s < (int)PR2_AHEAD; s++) {
        if (NP == 2) { const float4 t = PR2_LOAD1Q(voffA, s); bwA[s] = make_float2(t.x, t.y); bwB[s] = make_float2(t.z, t.w); }
        else bwA[s] = PR2_LOAD1(voffA, s);
    }
    // A fragments PR2_ADEPTH k-steps ahead: a step is only 64..128 cycles of matrix pipe per wave, an LDS round trip is longer
    float af[PF_IN / 4];
#pragma unroll
    for (int s = 0; s < PR2_ADEPTH; s++) af[s] = ap[4 * s];
    __builtin_amdgcn_sched_barrier(0);
    if (t_loop) *t_loop = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < PF_IN / 4; s++) {
        if (s + (int)PR2_AHEAD < PF_IN / 4) {
            if (NP == 2) { const float4 t = PR2_LOAD1Q(voffA, s + PR2_AHEAD); bwA[s + PR2_AHEAD] = make_float2(t.x, t.y); bwB[s + PR2_AHEAD] = make_float2(t.z, t.w); }
            else bwA[s + PR2_AHEAD] = PR2_LOAD1(voffA, s + PR2_AHEAD);
        }
        if (s + PR2_ADEPTH < PF_IN / 4) af[s + PR2_ADEPTH] = ap[4 * (s + PR2_ADEPTH)];
        const float av = af[s];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bwA[s].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bwA[s].y, acc[1], 0, 0, 0);
        if (NP == 2) {
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bwB[s].x, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bwB[s].y, acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// layer 2 of one wave: NT logit columns per lane (NT == 2: one 8-byte load, NT == 1: one 4-byte load per k-step), 180-deep
// (`draw`: the wave also draws its head rows' uniforms -- Philox4x32-10 does not depend on the logits.  Its ten rounds are spread over
// the first twenty k-steps, half a round each: the quarter-rate 32-bit multiplies issue in the shadow of the MFMAs instead of holding
// up the head phase; philox_u32's arithmetic, statement for statement)
template <int NT>
__device__ __forceinline__ void pr2_layer2(const __amdgpu_buffer_rsrc_t rs2, u32 voff, const float *ap, const float2 (&pre)[PR2_HOIST2], pf_f32x4 &acc0,
                                           pf_f32x4 &acc1, bool draw, u64 seed, u64 ctr, u32 game, float &u_out, u64 *t_loop = nullptr)
{
    float2 bw[PF_HID / 4];
#pragma unroll
    for (int s = 0; s < PR2_HOIST2; s++) bw[s] = pre[s];
    float af[PF_HID / 4];
#pragma unroll
    for (int s = 0; s < PR2_ADEPTH; s++) af[s] = ap[4 * s];
    u32 c0 = (u32)ctr, c1 = (u32)(ctr >> 32), c2 = game, c3 = 0x415A554Cu, k0 = (u32)seed, k1 = (u32)(seed >> 32), hi0 = 0, lo0 = 0;
    __builtin_amdgcn_sched_barrier(0);
    if (t_loop) *t_loop = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < PF_HID / 4; s++) {
        if (s + PR2_HOIST2 < PF_HID / 4) bw[s + PR2_HOIST2] = PR2_LOAD2(NT, s + PR2_HOIST2);
        if (s + PR2_ADEPTH < PF_HID / 4) af[s + PR2_ADEPTH] = ap[4 * (s + PR2_ADEPTH)];
        const float av = af[s];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[s].x, acc0, 0, 0, 0);
        if (NT == 2 && s < 20) {
            if (draw) {
                if ((s & 1) == 0) { hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0; }
                else {
                    const u32 hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
                    const u32 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
                    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
                    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
                }
            }
        }
        if (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[s].y, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NT == 2 && draw) u_out = (float)(c0 >> 8) * (1.0f / 16777216.0f);
}


// ---- network opponent (OPP == 2; game_runner.py:27-30 GameRunner(opponent=Agent(...))) ---------------------------------------------------
// The opponent only ever runs forward_actor (agent.py:73-81 -> model.py:28-41): layer 1 is the ACTOR half alone, 180 hidden columns
// (columns 180..359 of its k-major w1t), spread over the eight waves like layer 2's 180 logits -- waves 0..3 a column pair per lane
// (32w + 2c + j, one 8-byte load per k-step), waves 4..7 one column (128 + 16 (w - 4) + c): three 16x16 tiles per SIMD, 136-deep.  Each
// hidden unit is the same k-ordered v_mfma_f32_16x16x4_f32 chain as in azul_policy_forward_kernel, so a reply equals the per-move path's.
constexpr int PR2_HOISTO = 4;
constexpr u32 NET_MAX_REPLIES = 64;   // an opponent_move() takes tiles off the table: a chain of replies ends with the round (<= 21 moves)
#define PR2_LOADO(NT_, s) ((NT_) == 2 ? __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rso, voff, (4 * (s)) * PF_H2 * 4, 0)) \
                                      : make_float2(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rso, voff, (4 * (s)) * PF_H2 * 4, 0)), 0.f))

__device__ __forceinline__ void pr2_request_opp1(bool two, const __amdgpu_buffer_rsrc_t rso, u32 voff, float2 (&pre)[PR2_HOISTO])
{
    if (two) {
#pragma unroll
        for (int s = 0; s < PR2_HOISTO; s++) pre[s] = PR2_LOADO(2, s);
    } else {
#pragma unroll
        for (int s = 0; s < PR2_HOISTO; s++) pre[s] = PR2_LOADO(1, s);
    }
}

template <int NT>
__device__ __forceinline__ void pr2_opp_layer1(const __amdgpu_buffer_rsrc_t rso, u32 voff, const float *ap, const float2 (&pre)[PR2_HOISTO], pf_f32x4 &acc0,
                                               pf_f32x4 &acc1)
{
    float2 bw[PF_IN / 4];
#pragma unroll
    for (int s = 0; s < PR2_HOISTO; s++) bw[s] = pre[s];
    float af[PF_IN / 4];
#pragma unroll
    for (int s = 0; s < PR2_ADEPTH; s++) af[s] = ap[4 * s];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PF_IN / 4; s++) {
        if (s + PR2_HOISTO < PF_IN / 4) bw[s + PR2_HOISTO] = PR2_LOADO(NT, s + PR2_HOISTO);
        if (s + PR2_ADEPTH < PF_IN / 4) af[s + PR2_ADEPTH] = ap[4 * (s + PR2_ADEPTH)];
        const float av = af[s];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[s].x, acc0, 0, 0, 0);
        if (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[s].y, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// OPP: 0 the policy moves for both players, 1 GameRunner with the RandomAgent opponent, 2 GameRunner with a NETWORK opponent (a.Wopp)
template <bool LID, int OPP>
__global__ void __launch_bounds__(64 * PR2_WAVES) azul_policy_rollout2_kernel(BatchDev b, PolicyWeights W, RolloutArgs a)
{
    __shared__ float obsS[PF_GAMES * PF_OBS_STRIDE];
    __shared__ float hidS[PF_GAMES * PF_HID_STRIDE];
    __shared__ float lgS[PF_GAMES * PF_LOG_STRIDE];
    __shared__ float w2cS[PF_HID];
    __shared__ float b1S[PF_H2 + 24], b2aS[PF_ACT + 12];
    __shared__ u32 mtS[PF_GAMES][az2::MT_LDS_WORDS];
    __shared__ double2 tabfs_lds[T_PAIRS];
    __shared__ u64 maskS[PF_GAMES][4];
    __shared__ i32 actS[PF_GAMES];
    __shared__ float b1oS[OPP == 2 ? PF_HID + 12 : 1], b2aoS[OPP == 2 ? PF_ACT + 12 : 1];      // the opponent's actor biases
    __shared__ u32 oweS[PF_GAMES];                       // OPP == 2: does game i of the workgroup owe an opponent_move()?
    const u32 tid = threadIdx.x, lane = tid & 63u, l = lane & 31u, half = lane >> 5, c = lane & 15u, q = (lane >> 4) & 3u;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const u32 n = b.n, g0 = blockIdx.x * PF_GAMES, gl = 2u * w + half, gi = g0 + gl;
    const bool live = gi < n;
    const u32 gic = live ? gi : n - 1u;
    u64 counter = a.counter;
    if (a.counter_dev) counter += a.counter_dev[0];
    if (tid < (u32)PF_HID) w2cS[tid] = W.w2c[tid];
    if (tid < (u32)(PF_H2 + 24)) b1S[tid] = tid < (u32)PF_H2 ? W.b1[tid] : 0.f;
    if (tid < (u32)(PF_ACT + 12)) b2aS[tid] = tid < (u32)PF_ACT ? W.b2a[tid] : 0.f;
    if (OPP == 1) {
        for (u32 i = tid; i < (u32)T_PAIRS; i += 64u * PR2_WAVES) tabfs_lds[i] = b.tab[i];
    }
    if (OPP == 2) {
        if (tid < (u32)(PF_HID + 12)) b1oS[tid] = tid < (u32)PF_HID ? a.Wopp.b1[PF_HID + tid] : 0.f;
        if (tid < (u32)(PF_ACT + 12)) b2aoS[tid] = tid < (u32)PF_ACT ? a.Wopp.b2a[tid] : 0.f;
        if (tid < (u32)PF_GAMES) oweS[tid] = 0u;
    }
    const float b2c_v = W.b2c[0];

    // matrix-phase constants of this wave: layer 2 columns 32w + 2c + j (waves 0..3) or 128 + 16 (w - 4) + c (waves 4..7) -- three
    // 16x16 tiles per SIMD (waves w and w + 4 share one); layer 1 likewise with twice the tiles (see there)
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)W.w1t, 0, PF_IN * PF_H2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)W.w2a_t, 0, PF_HID * PF_ACT * 4, 0x00020000);
    const bool two = w < 4u;
#define PR2_L2COL0 (two ? 32u * w + 2u * c : 128u + 16u * (w - 4u) + c)

    // env state of this half's game
    az2::K2 k;
    az2::k2_init(k);
    az2::rng2_set_move_limit(mtS[gl], b.move_limit, l);
    az2::Tab2 tab = {tabfs_lds};
    az2::G2 g;
    uint8_t *rec = b.state + (size_t)gic * AZUL_RECORD_BYTES;
    az2::g2_load(g, rec, l);
    az2::prime2(g, k);
    az2::Rng2 r;
    u32 *gmt = b.mt + (size_t)gic * 624u;
    az2::rng2_open(r, gmt, mtS[gl], b.mtpos[gic], l);
    const u64 margin = b.draw_margin;
    az2::Counters2 cnt;
    az2::counters2_open(cnt, b.episodes + gic, b.stuck + gic, b.stat_sum + (size_t)gic * 10, l);
    u32 st_last = ST_OK;
    i32 win_rew = 0;
    u32 win_done = 0;
    float *orow = obsS + gl * PF_OBS_STRIDE;
    __syncthreads();                                     // tables / biases staged

    // observation + legal mask of the current state -> LDS (network / head) and trajectory slot `slot`
    az2::Mask2 m;                                        // legal mask of the published state: the next env step tests the action against it
    // The env phase only writes LDS (observation row, packed mask bits) + the player byte; the trajectory slot's 544-byte observation and
    // 180-byte mask of every game are copied out of LDS by flush() below, on waves that idle during the head phase: eleven vector
    // stores per game pair leave the env phase, which every other phase of the move waits for.
    // (publish_rows: the mask `m` already holds, seen from `persp`)
    auto publish_rows = [&](u32 persp) {
        if (l == 0u) {                                   // the 180 bits packed: the six 30-bit row words concatenated
            maskS[gl][0] = (u64)m.m[0] | ((u64)m.m[1] << 30) | ((u64)m.m[2] << 60);
            maskS[gl][1] = ((u64)m.m[2] >> 4) | ((u64)m.m[3] << 26) | ((u64)m.m[4] << 56);
            maskS[gl][2] = ((u64)m.m[4] >> 8) | ((u64)m.m[5] << 22);
        }
        az2::observe2(g, persp, orow, nullptr, l);
    };
    auto publish = [&](u32 slot, bool mask_current) {
        if (!mask_current) az2::legal_mask2(g, k, m);
        if (l == 0u) a.player[(size_t)slot * n + gi] = (uint8_t)g.cur;
        publish_rows(OPP ? 0u : az2::me2(g));
    };
    // trajectory slot `slot` of the workgroup's 16 games <- the LDS rows: the 16 x 136 floats are contiguous in the [T+1][N][136]
    // array (16-byte chunks), the 16 x 180 mask bytes in [T+1][N][180] (dwords: four bits of the packed mask spread into four bytes).
    // `idx` in 0..191 over the three copying waves.
    auto flush = [&](u32 slot, u32 idx) {
        const size_t cell0 = (size_t)slot * n + g0;
        float *og = a.obs + cell0 * PF_IN;
#pragma unroll
        for (u32 rep = 0; rep < 3u; rep++) {
            const u32 j = idx + 192u * rep;              // 16-byte chunk: row j / 34, floats 4 (j % 34) ..
            const u32 row = j / 34u, c4 = j - 34u * row;
            if (j < 16u * 34u && g0 + row < n) *(float4 *)(og + 4u * j) = *(const float4 *)(obsS + row * PF_OBS_STRIDE + 4u * c4);
        }
        u32 *mg = (u32 *)(a.mask + cell0 * AZUL_NUM_ACTIONS);
#pragma unroll
        for (u32 rep = 0; rep < 4u; rep++) {
            const u32 d = idx + 192u * rep;              // dword: row d / 45, actions 4 (d % 45) ..
            const u32 row = d / 45u, a0 = 4u * (d - 45u * row);
            if (d < 16u * 45u && g0 + row < n) {
                const u32 nib = (u32)(maskS[row][a0 >> 6] >> (a0 & 63u)) & 0xfu;
                mg[d] = (nib * 0x00204081u) & 0x01010101u;
            }
        }
    };
    if (live) publish(0u, false);
    else {
        orow[l] = 0.f; orow[l + 32u] = 0.f; orow[l + 64u] = 0.f; orow[l + 96u] = 0.f;
        if (l < 8u) orow[l + 128u] = 0.f;
        if (l == 0u) { maskS[gl][0] = 0; maskS[gl][1] = 0; maskS[gl][2] = 0; }
    }

    // (waves 0..3: lane c owns the FOUR adjacent columns 64w + 4c .. + 3 -- one 16-byte load per k-step; waves 4..7: a pair, 8 bytes)
    const u32 colA = two ? 64u * w + 4u * c : 256u + 32u * (w - 4u) + 2u * c, colB = colA + 2u;
    const bool liveA = colA < (u32)PF_H2, liveB = two && colB < (u32)PF_H2;
    const u32 voffA = ((liveA ? colA : 0u) + q * (u32)PF_H2) * 4u, voffB = ((liveB ? colB : 0u) + q * (u32)PF_H2) * 4u;
    const u32 voff2 = ((PR2_L2COL0 < (u32)PF_ACT ? PR2_L2COL0 : 0u) + q * (u32)PF_ACT) * 4u;
    float2 preA[PR2_HOIST1], preB[PR2_HOIST1], pre2[PR2_HOIST2];
    pr2_request1(two, rs1, voffA, voffB, preA, preB);
    // the opponent's matrices (OPP == 2): the actor half of its layer 1 and its layer 2, both in layer 2's column mapping
    const __amdgpu_buffer_rsrc_t rso1 = __builtin_amdgcn_make_buffer_rsrc((void *)(OPP == 2 ? a.Wopp.w1t : W.w1t), 0, PF_IN * PF_H2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso2 = __builtin_amdgcn_make_buffer_rsrc((void *)(OPP == 2 ? a.Wopp.w2a_t : W.w2a_t), 0, PF_HID * PF_ACT * 4, 0x00020000);
    const u32 voffo1 = ((u32)PF_HID + (PR2_L2COL0 < (u32)PF_HID ? PR2_L2COL0 : 0u) + q * (u32)PF_H2) * 4u;
    float2 preo[PR2_HOISTO];
    az2::NetStep ns;
    ns.pending = az2::NET_READY; ns.replies = 0; ns.rew = 0; ns.dn = 0; ns.closed = false; ns.st = ST_OK;
#if defined(AZ_PROFILE_SEGMENTS)
    u64 pr_acc[6] = {0, 0, 0, 0, 0, 0}, pr_last = __builtin_amdgcn_s_memtime();
    const u64 pr_t0 = pr_last, pr_r0 = __builtin_amdgcn_s_memrealtime();
#define PR2_STAMP(i) do { u64 now_ = __builtin_amdgcn_s_memtime(); pr_acc[i] += now_ - pr_last; pr_last = now_; } while (0)
    // matrix sub-phases of the two waves of SIMD 1 (w == 1: four / two tiles, w == 5: two / one): prologue (barrier release -> loop entry), the
    // MFMA loop's issue span, the epilogue (waits for the last results, bias + relu, LDS writes), the wait at the closing barrier
    u64 sub[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts[4] = {0, 0, 0, 0}, tl = 0;
    u64 *const tlp = &tl;
#define PR2_SUB(i) do { ts[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define PR2_SUBACC(o) do { sub[(o)] += tl - ts[0]; sub[(o) + 1] += ts[1] - tl; sub[(o) + 2] += ts[2] - ts[1]; sub[(o) + 3] += ts[3] - ts[2]; } while (0)
#else
#define PR2_STAMP(i) do { } while (0)
#define PR2_SUB(i) do { } while (0)
#define PR2_SUBACC(o) do { } while (0)
    u64 *const tlp = nullptr;
#endif
#pragma unroll 1
    for (int t = 0; t < a.n_steps; t++) {
        const size_t row_t = (size_t)t * n;
        float u_head = 0.f;
        PR2_STAMP(0);                                    // own env step + publish
        lds_barrier();                                   // observations and mask bits of all 16 games are in LDS
        PR2_STAMP(1);                                    // waiting for the slowest env wave
        PR2_SUB(0);
        {
            // layer 1: pairs of adjacent hidden columns per lane (one 8-byte load per k-step and pair: a 16-lane group reads 128 contiguous
            // bytes of a k-row).  Waves 0..3 own two pairs, FOUR adjacent columns 64w + 4c + j (one 16-byte load per k-step), waves 4..7 one (256 + 32 (w - 4) +
            // 2c + j): 24 tiles over 8 waves, six per SIMD.
            pf_f32x4 acc[4];
            for (int j = 0; j < 4; j++) acc[j] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
            const float *ap = obsS + c * PF_OBS_STRIDE + q;
            if (two) pr2_layer1<2>(rs1, voffA, voffB, ap, preA, preB, acc, tlp);
            else pr2_layer1<1>(rs1, voffA, voffB, ap, preA, preB, acc, tlp);
            PR2_SUB(1);
            pr2_request2(two, rs2, voff2, pre2);         // layer 2's first fragments: in flight across the epilogue and the barrier
            for (int rr = 0; rr < 4; rr++) {             // C layout: column = lane & 15, row = 4 (lane >> 4) + rr
                float *hp = hidS + (4u * q + rr) * PF_HID_STRIDE;
                if (liveA) {
                    const float h0 = acc[0][rr] + b1S[colA], h1 = acc[1][rr] + b1S[colA + 1u];
                    hp[colA] = h0 > 0.f ? h0 : 0.f;      // F.relu, model.py:24/30
                    hp[colA + 1u] = h1 > 0.f ? h1 : 0.f;
                }
                if (liveB) {
                    const float h0 = acc[2][rr] + b1S[colB], h1 = acc[3][rr] + b1S[colB + 1u];
                    hp[colB] = h0 > 0.f ? h0 : 0.f;
                    hp[colB + 1u] = h1 > 0.f ? h1 : 0.f;
                }
            }
        }
        PR2_SUB(2);
        lds_barrier();
        PR2_STAMP(2);                                    // layer 1 (incl. barrier)
        PR2_SUB(3);
        PR2_SUBACC(0);
        PR2_SUB(0);
        {
            // layer 2 (actor)
            const u32 col0 = PR2_L2COL0;
            const bool live2 = col0 < (u32)PF_ACT;
            pf_f32x4 acc0 = (pf_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            const float *ap = hidS + c * PF_HID_STRIDE + PF_HID + q;
            const u32 hg_ = g0 + 4u * w + q;             // waves 0..3: the head row of this 16-lane group
            if (two) pr2_layer2<2>(rs2, voff2, ap, pre2, acc0, acc1, true, a.seed, counter + (u64)t, b.id_base + (hg_ < n ? hg_ : n - 1u), u_head, tlp);
            else pr2_layer2<1>(rs2, voff2, ap, pre2, acc0, acc1, false, 0, 0, 0, u_head, tlp);
            PR2_SUB(1);
            if (live2)
                for (int rr = 0; rr < 4; rr++) {
                    float *lp = lgS + (4u * q + rr) * PF_LOG_STRIDE + col0;
                    lp[0] = acc0[rr] + b2aS[col0];
                    if (two) lp[1] = acc1[rr] + b2aS[col0 + 1u];
                }
        }
        PR2_SUB(2);
        lds_barrier();
        PR2_STAMP(3);                                    // layer 2 + critic (incl. barrier)
        PR2_SUB(3);
        PR2_SUBACC(4);
        if (w < 4u) {
            // head: waves 0..3 (one per SIMD) sample four games each, 16 lanes per game
            const u32 hrow = 4u * w + q, hg = g0 + hrow;
            float x[HEAD_PER_LANE];
            const float *lg = lgS + hrow * PF_LOG_STRIDE + (c < 15u ? 12u * c : 0u);
            for (int j = 0; j < HEAD_PER_LANE; j++) x[j] = lg[j];
            const u64 M0 = maskS[hrow][0], M1 = maskS[hrow][1], M2 = maskS[hrow][2];
            const u32 bitpos = 12u * c, word = bitpos >> 6, off = bitpos & 63u;
            const u64 lo = word == 0u ? M0 : (word == 1u ? M1 : M2), hi = word == 0u ? M1 : M2;
            u64 field = lo >> off;
            if (off > 52u) field |= hi << (64u - off);
            const u32 okbits = c < 15u ? (u32)field & 0xfffu : 0u;
            policy_head_rows(x, lgS + hrow * PF_LOG_STRIDE, okbits, a.seed, counter + (u64)t, hg < n ? hg : n - 1u, lane, hg < n, a.action + row_t, a.logp + row_t,
                             a.entropy + row_t, b.id_base, actS + 4u * w, &u_head);
        } else if (w < 7u) {
            flush((u32)t, 64u * (w - 4u) + lane);        // waves 4..6: the trajectory slot of this move's decision
        } else {
            // the critic, on a wave that idles during the head, summed exactly like azul_policy_forward_kernel: lane (row c, quarter q)
            // sums k = q (mod 4), then the quarters are added (model.py:22-26)
            float sum = 0.f;
            const float *hp = hidS + c * PF_HID_STRIDE;
#pragma unroll
            for (int s = 0; s < PF_HID / 4; s++) sum = fmaf(hp[4 * s + q], w2cS[4 * s + q], sum);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (q == 0u && g0 + c < n) a.value[row_t + g0 + c] = sum + b2c_v;
        }
        lds_barrier();
        PR2_STAMP(4);                                    // head (incl. barrier)
        if constexpr (OPP != 2) {
            pr2_request1(two, rs1, voffA, voffB, preA, preB);    // the next move's layer 1: in flight during the env step
            if (live) {
                const i32 av = actS[gl];
                i32 rew = 0;
                u32 dn = 0;
                if constexpr (OPP == 1) st_last = az2::agent_step2<LID>(g, av, m, b.rules.first_player, r, tab, margin, cnt, k, rew, dn);
                else st_last = az2::policy_step2<LID>(g, av, m, b.rules.first_player, r, margin, cnt, k, rew, dn);
                if (l == 0u) { a.reward[row_t + gi] = rew; a.done[row_t + gi] = (uint8_t)dn; }
                win_rew = l == ((u32)t & 31u) ? rew : win_rew;       // lane t of the half keeps step t (the returns scan below)
                win_done = l == ((u32)t & 31u) ? dn : win_done;
                publish((u32)t + 1u, false);
            }
        } else {
            // GameRunner.step against the NETWORK opponent (game_runner.py:43-55 with :37-42 answered by a.Wopp): the agent's move, then
            // matrix phases on the opponent's weights WHILE ANY of the workgroup's 16 games owes an opponent_move() -- its own replies and
            // player 1's forced moves (:46), after an episode end the opening moves of the next (:84) -- the other games sit masked
            // ONE env site per pass: pass 0 plays the agent's action, pass j > 0 reply j - 1 of the games that owed it
#pragma unroll 1
            for (u32 j = 0;; j++) {
                pr2_request_opp1(two, rso1, voffo1, preo);       // (almost every step has a reply: requested ahead like layer 1; speculative after that)
                if (live && (j == 0u || ns.pending != az2::NET_READY)) {
                    az2::net_move2<LID>(g, actS[gl], j == 0u, m, b.rules.first_player, r, margin, cnt, k, ns);
                    if (ns.pending != az2::NET_READY) publish_rows(az2::me2(g));   // what opponent_move hands the opponent (:38-39)
                }
                if (l == 0u) oweS[gl] = live && ns.pending != az2::NET_READY ? 1u : 0u;
                PR2_STAMP(0);                                    // (diagnostic build: the reply rounds' phases are added to the agent pass's)
                lds_barrier();                                   // every game's debt, observation row and mask bits are in LDS
                PR2_STAMP(1);
                const bool any = __builtin_amdgcn_ballot_w64(oweS[lane & 15u] != 0u) != 0ull;      // (the same 16 words in every wave)
                if (!any || j >= NET_MAX_REPLIES) break;
                const u64 okey = a.opp_seed == AZUL_POLICY_ARGMAX ? a.opp_seed : a.opp_seed + (u64)j;      // reply j of a step: its own Philox key
                const u32 col0 = PR2_L2COL0;
                {
                    pf_f32x4 acc0 = (pf_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    const float *ap = obsS + c * PF_OBS_STRIDE + q;
                    if (two) pr2_opp_layer1<2>(rso1, voffo1, ap, preo, acc0, acc1);
                    else pr2_opp_layer1<1>(rso1, voffo1, ap, preo, acc0, acc1);
                    pr2_request2(two, rso2, voff2, pre2);
                    if (col0 < (u32)PF_HID)
                        for (int rr = 0; rr < 4; rr++) {
                            float *hp = hidS + (4u * q + rr) * PF_HID_STRIDE + PF_HID + col0;
                            const float h0 = acc0[rr] + b1oS[col0];
                            hp[0] = h0 > 0.f ? h0 : 0.f;         // F.relu, model.py:30
                            if (two) { const float h1 = acc1[rr] + b1oS[col0 + 1u]; hp[1] = h1 > 0.f ? h1 : 0.f; }
                        }
                }
                lds_barrier();
                PR2_STAMP(2);
                {
                    pf_f32x4 acc0 = (pf_f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                    const float *ap = hidS + c * PF_HID_STRIDE + PF_HID + q;
                    const u32 hg_ = g0 + 4u * w + q;
                    if (two) pr2_layer2<2>(rso2, voff2, ap, pre2, acc0, acc1, true, okey, counter + (u64)t, b.id_base + (hg_ < n ? hg_ : n - 1u), u_head);
                    else pr2_layer2<1>(rso2, voff2, ap, pre2, acc0, acc1, false, 0, 0, 0, u_head);
                    if (col0 < (u32)PF_ACT)
                        for (int rr = 0; rr < 4; rr++) {
                            float *lp = lgS + (4u * q + rr) * PF_LOG_STRIDE + col0;
                            lp[0] = acc0[rr] + b2aoS[col0];
                            if (two) lp[1] = acc1[rr] + b2aoS[col0 + 1u];
                        }
                }
                lds_barrier();
                PR2_STAMP(3);
                if (w < 4u) {                                    // the opponent's sampling head (agent.py:76-80), waves 0..3, four games each
                    const u32 hrow = 4u * w + q, hg = g0 + hrow;
                    float x[HEAD_PER_LANE];
                    const float *lg = lgS + hrow * PF_LOG_STRIDE + (c < 15u ? 12u * c : 0u);
                    for (int jj = 0; jj < HEAD_PER_LANE; jj++) x[jj] = lg[jj];
                    const u64 M0 = maskS[hrow][0], M1 = maskS[hrow][1], M2 = maskS[hrow][2];
                    const u32 bitpos = 12u * c, word = bitpos >> 6, off = bitpos & 63u;
                    const u64 lo = word == 0u ? M0 : (word == 1u ? M1 : M2), hi = word == 0u ? M1 : M2;
                    u64 field = lo >> off;
                    if (off > 52u) field |= hi << (64u - off);
                    const u32 okbits = c < 15u ? (u32)field & 0xfffu : 0u;
                    const bool tr = hg < n && oweS[hrow] != 0u && a.opp_action && j < (u32)a.opp_slots;
                    const size_t trow = ((size_t)t * (size_t)(a.opp_slots > 0 ? a.opp_slots : 1) + (j < (u32)a.opp_slots ? j : 0u)) * n;
                    policy_head_rows(x, lgS + hrow * PF_LOG_STRIDE, okbits, okey, counter + (u64)t, hg < n ? hg : n - 1u, lane, tr,
                                     a.opp_action ? a.opp_action + trow : nullptr, a.opp_logp ? a.opp_logp + trow : nullptr, nullptr, b.id_base, actS + 4u * w,
                                     &u_head);
                }
                lds_barrier();
                PR2_STAMP(4);
            }
            pr2_request1(two, rs1, voffA, voffB, preA, preB);    // the next step's layer 1 on the agent's weights
            if (live) {
                if (ns.pending != az2::NET_READY) {              // (NET_MAX_REPLIES hit: cannot happen with legal replies; the slot reports it and moves on)
                    ns.pending = az2::NET_READY;
                    if (!ns.st) ns.st = ST_STUCK;
                    az2::legal_mask2(g, k, m);
                }
                st_last = ns.st;
                if (l == 0u) {
                    a.reward[row_t + gi] = ns.rew; a.done[row_t + gi] = (uint8_t)ns.dn;
                    if (a.opp_replies) a.opp_replies[row_t + gi] = (uint8_t)(ns.replies < 255u ? ns.replies : 255u);
                }
                win_rew = l == ((u32)t & 31u) ? ns.rew : win_rew;
                win_done = l == ((u32)t & 31u) ? ns.dn : win_done;
                publish((u32)t + 1u, true);                      // net_settle2 left the mask of this state in `m`
            }
        }
    }
    lds_barrier();                                       // the rows of the state after the last move
    if (w >= 4u && w < 7u) flush((u32)a.n_steps, 64u * (w - 4u) + lane);
#if defined(AZ_PROFILE_SEGMENTS)
    if (lane == 0u && w == 5u) {
        for (int i = 0; i < 5; i++) atomicAdd((unsigned long long *)(b.prof + i), (unsigned long long)pr_acc[i]);
        const u64 pr_r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd((unsigned long long *)(b.prof + 6), (unsigned long long)(__builtin_amdgcn_s_memtime() - pr_t0));
        atomicAdd((unsigned long long *)(b.prof + 7), (unsigned long long)(pr_r1 - pr_r0));
        atomicMax((unsigned long long *)(b.prof + 5), (unsigned long long)((1ull << 62) - pr_r0));
        atomicMax((unsigned long long *)(b.prof + 8), (unsigned long long)pr_r1);
    }
    if (lane == 0u && (w == 1u || w == 5u))
        for (int i = 0; i < 8; i++) atomicAdd((unsigned long long *)(b.prof + 16 + (w == 5u ? 8 : 0) + i), (unsigned long long)sub[i]);
#endif
    if (live && a.returns && a.n_steps <= 32) {
        // the window's discounted returns (azul_returns_kernel's scan, statement for statement: q = reward + gamma * q backwards, an
        // episode end resets q; nn_runner.py:70-76) from the rewards / done flags this half kept in its lanes: no second launch
        float q = 0.f;
        float mine = 0.f;
#pragma unroll 1
        for (int t = a.n_steps - 1; t >= 0; t--) {
            const i32 rt = (i32)az2::hread((u32)win_rew, (u32)t);
            const u32 dt = az2::hread(win_done, (u32)t);
            if (dt) q = 0.f;
            q = (float)rt + a.gamma * q;
            mine = l == (u32)t ? q : mine;
        }
        if (l < (u32)a.n_steps) a.returns[(size_t)l * n + gi] = mine;
    }
    if (live) {
        az2::g2_store(g, rec, l);
        az2::rng2_close(r, gmt, b.mtpos + gi, l);
        az2::counters2_close(cnt, l);
        if (a.status && l == 0u) a.status[gi] = (uint8_t)st_last;
    }
    if (a.counter_dev && tid == 0u) {
        __threadfence();
        u64 done_blocks = atomicAdd((unsigned long long *)(a.counter_dev + 1), 1ull);
        if (done_blocks == (u64)gridDim.x - 1ull) {
            a.counter_dev[1] = 0ull;
            a.counter_dev[0] += (u64)a.n_steps;
            __threadfence();
        }
    }
#undef PR2_L2COL0
#undef PR2_LOAD1
#undef PR2_LOAD1Q
#undef PR2_LOAD2
#undef PR2_LOADO
}
