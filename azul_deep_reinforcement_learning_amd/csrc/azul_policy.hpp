// azul_policy.hpp -- device code of row N1 (policy-driven self-play, BASELINE configs[2]); included by azul_kernels.hip.
//
//   policy_head_rows        masked softmax + ONE categorical sample + log-prob + entropy term over 180 logits, four games per
//                           wave (Agent.get_ac_output, agent.py:64-72; NNRunner.run_episode, nn_runner.py:32-40)
//   azul_policy_head_kernel the head alone, for logits produced elsewhere
//   azul_policy_forward_kernel   the WHOLE ActorCritic forward of model.py:22-41 for 16 games per workgroup on the f32 matrix
//                           cores (v_mfma_f32_16x16x4_f32: exact f32, k-ordered fma chain) + the head, in one launch:
//                               hidden = relu(obs @ [critic_linear1 | actor_linear1]^T + b1)          16 x 360, K = 136
//                               value  = hidden[:, :180] . critic_linear2 + b                         16
//                               logits = hidden[:, 180:] @ actor_linear2^T + b                        16 x 180, K = 180
//                           observation tile, hidden and logits live in LDS; weights stream from L2 (k-major, coalesced).
// fp32 throughout (the reference computes in fp32 torch); randomness: Philox4x32-10 keyed by (seed, GLOBAL game id) with the
// step counter as the block index -- its own documented stream, not torch's or numpy's.  The global id is
// `id_base` + the game's index in the launch, so a game draws the same numbers however the games are sharded over GPUs.
#pragma once

__device__ __forceinline__ void philox_round(u32 &c0, u32 &c1, u32 &c2, u32 &c3, u32 k0, u32 k1)
{
    const u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    u32 hi0 = __umulhi(M0, c0), lo0 = M0 * c0, hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    u32 n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ __forceinline__ u32 philox_u32(u64 seed, u64 counter, u32 game)
{
    u32 c0 = (u32)counter, c1 = (u32)(counter >> 32), c2 = game, c3 = 0x415A554Cu;      // "AZUL"
    u32 k0 = (u32)seed, k1 = (u32)(seed >> 32);
    for (int i = 0; i < 10; i++) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    return c0;
}

__device__ __forceinline__ float policy_uniform(u64 seed, u64 counter, u32 game)      // [0, 1), 24 bits
{
    return (float)(philox_u32(seed, counter, game) >> 8) * (1.0f / 16777216.0f);
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for its global loads and stores, so
// weight fragments requested ahead and trajectory stores stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- the head: FOUR games per wave, 16 lanes per game, lane c of a group owns actions 12c .. 12c+11 (c < 15) -----------------
// All reductions stay inside a 16-lane DPP row (quad_perm / row_half_mirror / row_mirror / row_shr): no LDS round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)          // lanes without a source read 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ u32 dpp_u(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_SHR1 = 0x111, DPP_SHR2 = 0x112,
              DPP_SHR4 = 0x114, DPP_SHR8 = 0x118;

__device__ __forceinline__ float row_max(float v)
{
    v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float row_sum(float v)
{
    v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HALF_MIRROR>(v); v += dpp_f<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ u32 row_sum_u(u32 v)
{
    v += dpp_u<DPP_XOR1>(v); v += dpp_u<DPP_XOR2>(v); v += dpp_u<DPP_HALF_MIRROR>(v); v += dpp_u<DPP_MIRROR>(v);
    return v;
}

constexpr int HEAD_PER_LANE = 12;

// x[j]: the lane's 12 logits; row: the game's 180 logits (LDS or global: the chosen action's logit is read back from there); okbits: bit
// j set when action 12c+j is legal; g: the lane's game (uniform inside a 16-lane row).
// Lane c == 0 of every row whose `store` is true writes the three results of its game.
// The per-lane loops are BRANCH-FREE (round 3): the first version's `if (ok && pick < 0 && target < cum)` became twelve divergent
// regions per lane (26 exec-mask saves, 58 register copies in 670 instructions).  Now the twelve crossing tests are the sign bits of
// target - cum_j collected into a bit mask, the pick is its lowest bit among the legal ones, and the chosen action's logit is read
// back from `row` instead of being tracked through the loop.  Same additions in the same order: the same numbers as before.
__device__ __forceinline__ void policy_head_rows(const float (&x)[HEAD_PER_LANE], const float *row, u32 okbits, u64 seed, u64 counter, u32 g, u32 l,
                                                 bool store, i32 *action, float *logp, float *entropy, u32 id_base, i32 *lds_action = nullptr,
                                                 const float *u_ready = nullptr /* the game's uniform, when the caller drew it ahead of time */)
{
    const u32 c = l & 15u, grp = l >> 4;
    const float NEG = -3.0e38f;
    float m = NEG;
#pragma unroll
    for (int j = 0; j < HEAD_PER_LANE; j++) m = fmaxf(m, ((okbits >> j) & 1u) ? x[j] : NEG);
    m = row_max(m);
    const u32 cnt = row_sum_u((u32)__popc(okbits));
    float e[HEAD_PER_LANE];
    float mine = 0.f, zs = 0.f;
#pragma unroll
    for (int j = 0; j < HEAD_PER_LANE; j++) {
        const bool ok = (okbits >> j) & 1u;
        const float z = ok ? x[j] - m : 0.f;
        e[j] = ok ? __expf(z) : 0.f;
        mine += e[j];
        zs += z;
    }
    const float S = row_sum(mine);
    const float logS = __logf(S);
    const float zsum = row_sum(zs);
    const float ent = -(zsum / (float)(cnt ? cnt : 1u) - logS);          // -mean(log p over legal actions), nn_runner.py:36-40
    // inverse CDF in ascending action order (agent.py:69): exclusive prefix of the lane sums inside the row, then the lane's 12 steps
    float incl = mine;
    incl += dpp_f<DPP_SHR1>(incl); incl += dpp_f<DPP_SHR2>(incl); incl += dpp_f<DPP_SHR4>(incl); incl += dpp_f<DPP_SHR8>(incl);
    const bool argmax = seed == AZUL_POLICY_ARGMAX;      // action_selection == "Max" (agent.py:70-71): np.argmax, first maximum
    const float u = u_ready ? *u_ready : policy_uniform(seed, counter, id_base + g);
    const float target = u * S;
    float cum = incl - mine;
    u32 cross = 0;                                       // bit j: target < cum_j (sampling) / x_j == m (argmax)
    if (argmax) {                                        // (wave-uniform)
#pragma unroll
        for (int j = 0; j < HEAD_PER_LANE; j++) cross |= (x[j] == m ? 1u : 0u) << j;
    } else {
#pragma unroll
        for (int j = 0; j < HEAD_PER_LANE; j++) {
            cum += e[j];
            cross |= (__builtin_bit_cast(u32, target - cum) >> 31) << j;      // the sign of target - cum_j (no NaNs: finite logits)
        }
    }
    const u32 pickmask = cross & okbits;
    const int pick = pickmask ? (int)__builtin_ctz(pickmask) : -1;
    const int lastok = okbits ? 31 - (int)__builtin_clz(okbits) : 0;
    const u64 hit = __ballot(pickmask != 0u), any = __ballot(okbits != 0u);
    const u32 hit16 = (u32)(hit >> (16u * grp)) & 0xffffu, any16 = (u32)(any >> (16u * grp)) & 0xffffu;
    // fp32 round-off can push the target past the last cumulative sum: then the last legal action is taken
    const u32 lane_sel = hit16 ? (u32)__builtin_ctz(hit16) : (any16 ? 31u - (u32)__builtin_clz(any16) : 0u);
    const int jmine = hit16 ? pick : lastok;
    const int src = (int)((16u * grp + lane_sel) << 2);
    const int jsel = __builtin_amdgcn_ds_bpermute(src, jmine);
    const i32 chosen = cnt == 0u ? -1 : (i32)(HEAD_PER_LANE * lane_sel) + jsel;     // no legal action (stuck game): the rollout sends -1
    if (lds_action && c == 0u) lds_action[grp] = chosen;
    if (store && c == 0u) {
        const float zsel = row[chosen < 0 ? 0 : chosen] - m;                         // log-softmax numerator of the chosen action
        if (action) action[g] = chosen;
        if (logp) logp[g] = cnt == 0u ? 0.f : zsel - logS;
        if (entropy) entropy[g] = cnt == 0u ? 0.f : ent;
    }
}

// legal-mask bytes 12c .. 12c+11 of one game as a 12-bit field (three aligned dword loads; lane c == 15 owns no action)
__device__ __forceinline__ u32 head_mask_bits(const uint8_t *mk_row, u32 c)
{
    const u32 *p = (const u32 *)(mk_row + (c < 15u ? 12u * c : 0u));
    u32 bits = 0;
    for (int d = 0; d < 3; d++) {
        u32 wv = p[d];
        for (int b = 0; b < 4; b++) bits |= (((wv >> (8 * b)) & 0xffu) != 0u ? 1u : 0u) << (4 * d + b);
    }
    return c < 15u ? bits : 0u;
}

__global__ void __launch_bounds__(64) azul_policy_head_kernel(const float *logits, const uint8_t *mask, u64 seed, u64 counter,
                                                              const u64 *counter_dev, u32 n, i32 *action, float *logp, float *entropy, u32 id_base)
{
    const u32 l = threadIdx.x, c = l & 15u;
    const u32 g = blockIdx.x * 4u + (l >> 4);
    const u32 gc = g < n ? g : n - 1u;
    if (counter_dev) counter += *counter_dev;            // device-resident step counter: graph replays draw fresh numbers
    float x[HEAD_PER_LANE];
    const float *lg = logits + (size_t)gc * AZUL_NUM_ACTIONS + (c < 15u ? 12u * c : 0u);
    for (int j = 0; j < HEAD_PER_LANE; j++) x[j] = lg[j];
    u32 okbits = head_mask_bits(mask + (size_t)gc * AZUL_NUM_ACTIONS, c);
    policy_head_rows(x, logits + (size_t)gc * AZUL_NUM_ACTIONS, okbits, seed, counter, gc, l, g < n, action, logp, entropy, id_base);
}

// ---- fused ActorCritic forward + head ------------------------------------------------------------------------------
constexpr int PF_IN = 136, PF_HID = 180, PF_H2 = 360, PF_ACT = 180;
constexpr int PF_GAMES = 16;                 // games per workgroup = the M of v_mfma_f32_16x16x4_f32
constexpr int PF_OBS_STRIDE = 164;           // LDS row strides = 4 (mod 32): the A-fragment read (16 rows x 4 k) hits every bank twice
constexpr int PF_HID_STRIDE = 388;
constexpr int PF_LOG_STRIDE = 196;           // 4 (mod 32): the four games a wave samples together sit in different banks

struct PolicyWeights {
    const float *w1t;      // [136][360] k-major: columns 0..179 = critic_linear1.weight^T, 180..359 = actor_linear1.weight^T
    const float *b1;       // [360]
    const float *w2c;      // [180]       critic_linear2.weight
    const float *b2c;      // [1]
    const float *w2a_t;    // [180][180]  actor_linear2.weight^T (k-major)
    const float *b2a;      // [180]
};

typedef float pf_f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) azul_policy_forward_kernel(const float *obs, const uint8_t *mask, PolicyWeights W, u64 seed, u64 counter,
                                                                  u64 *counter_dev, int advance, u32 n, float *value, i32 *action,
                                                                  float *logp, float *entropy, float *logits_out, u32 id_base)
{
    __shared__ float obsS[PF_GAMES * PF_OBS_STRIDE];
    __shared__ float hidS[PF_GAMES * PF_HID_STRIDE];
    __shared__ float lgS[PF_GAMES * PF_LOG_STRIDE];
    __shared__ float w2cS[PF_HID];
    const u32 tid = threadIdx.x, w = tid >> 6, l = tid & 63u, c = l & 15u, q = l >> 4;
    const u32 g0 = blockIdx.x * PF_GAMES;
    if (counter_dev) counter += counter_dev[0];
#if defined(AZ_PF_PROFILE)
    u64 pf_t[8]; int pf_i = 0;
#define PF_STAMP() do { __builtin_amdgcn_s_waitcnt(0); pf_t[pf_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PF_STAMP() do { } while (0)
#endif
    PF_STAMP();

    // The small inputs go first (memory returns in order): the 16 observations of the tile (rows past the batch repeat the last
    // game; their results are never stored), critic_linear2's row, and the legal-mask bits of the four games this wave samples.
    constexpr int OBS_PER_THREAD_ = (PF_GAMES * PF_IN + 255) / 256;
    float ob[OBS_PER_THREAD_];
#pragma unroll
    for (int j = 0; j < OBS_PER_THREAD_; j++) {
        u32 i = tid + 256u * (u32)j;
        u32 ii = i < (u32)(PF_GAMES * PF_IN) ? i : 0u;
        u32 row = ii / PF_IN, k = ii - row * PF_IN;
        u32 g = g0 + row < n ? g0 + row : n - 1u;
        ob[j] = obs[(size_t)g * PF_IN + k];
    }
    const float w2c_v = W.w2c[tid < (u32)PF_HID ? tid : 0u];
    const u32 l1col0 = 96u * w + 6u * c;
    const bool l1live = l1col0 < (u32)PF_H2;                     // wave 3: lanes c >= 12 fall past column 359
    const u32 l2col0 = 48u * w + 3u * c;
    const bool l2live = l2col0 < (u32)PF_ACT;
    float bias1[6], bias2[3];
    for (int j = 0; j < 6; j++) bias1[j] = W.b1[(l1live ? l1col0 : 0u) + j];
    for (int j = 0; j < 3; j++) bias2[j] = W.b2a[(l2live ? l2col0 : 0u) + j];
    const float b2c_v = W.b2c[0];
    const u32 hrow = 4u * w + q, hg = g0 + hrow, hgc = hg < n ? hg : n - 1u;
    const u32 okbits = head_mask_bits(mask + (size_t)hgc * AZUL_NUM_ACTIONS, c);

    // Weights stream straight from L2 into registers, software-pipelined PF_AHEAD k-steps ahead of the matrix pipe (memory returns
    // in order and s_waitcnt counts at most 63 loads, so "request everything, then multiply" would idle the pipe for most of the
    // stream).  Layer 2's rows are requested during the last layer-1 steps.  The kernel owns the SIMD's whole register file.
    constexpr int S1 = PF_IN / 4, S2 = PF_HID / 4, PF_AHEAD = 16, L2_PER_STEP = (S2 + PF_AHEAD - 1) / PF_AHEAD;
    float2 bw[S1][3];
    float bw2[S2][3];
    const float *bp = W.w1t + (l1live ? l1col0 : 0u) + (size_t)q * PF_H2;
    const float *bp2 = W.w2a_t + (l2live ? l2col0 : 0u) + (size_t)q * PF_ACT;
#define PF_LOAD1(s) do { const float *kp = bp + (size_t)(4 * (s)) * PF_H2; \
        bw[s][0] = *(const float2 *)(kp); bw[s][1] = *(const float2 *)(kp + 2); bw[s][2] = *(const float2 *)(kp + 4); } while (0)
#define PF_LOAD2(s) do { const float *kp = bp2 + (size_t)(4 * (s)) * PF_ACT; bw2[s][0] = kp[0]; bw2[s][1] = kp[1]; bw2[s][2] = kp[2]; } while (0)
#pragma unroll
    for (int s = 0; s < PF_AHEAD; s++) PF_LOAD1(s);
    __builtin_amdgcn_sched_barrier(0);                           // keep the loads up here (the scheduler would sink them to their uses)

    constexpr int OBS_PER_THREAD = (PF_GAMES * PF_IN + 255) / 256;
#pragma unroll
    for (int j = 0; j < OBS_PER_THREAD; j++) {
        u32 i = tid + 256u * (u32)j;
        if (i < (u32)(PF_GAMES * PF_IN)) { u32 row = i / PF_IN, k = i - row * PF_IN; obsS[row * PF_OBS_STRIDE + k] = ob[j]; }
    }
    if (tid < (u32)PF_HID) w2cS[tid] = w2c_v;
    __syncthreads();
    PF_STAMP();

    // layer 1: wave w owns hidden columns 96w + 6c + j (j = 0..5): six 16x16 tiles whose columns are strided by 6, so a lane's
    // six B values are 24 contiguous bytes and a 16-lane group reads 384 contiguous bytes of one k-row of w1t
    {
        pf_f32x4 acc[6];
        for (int j = 0; j < 6; j++) acc[j] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
        const u32 col0 = l1col0;
        const bool live = l1live;
        const float *ap = obsS + c * PF_OBS_STRIDE + q;
        float av[S1];                                            // the A operands (observations) of all k-steps: one LDS burst
#pragma unroll
        for (int s = 0; s < S1; s++) av[s] = ap[4 * s];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < S1; s++) {
            if (s + PF_AHEAD < S1) PF_LOAD1(s + PF_AHEAD);
            else {
#pragma unroll
                for (int t = 0; t < L2_PER_STEP; t++) {
                    const int s2 = (s + PF_AHEAD - S1) * L2_PER_STEP + t;
                    if (s2 < S2) PF_LOAD2(s2);
                }
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][0].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][0].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][1].x, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][1].y, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][2].x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[s][2].y, acc[5], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (live) {
            for (int j = 0; j < 6; j++) {
                float bias = bias1[j];
                for (int r = 0; r < 4; r++) {                    // C layout: column = lane & 15, row = 4 (lane >> 4) + r
                    float h = acc[j][r] + bias;
                    hidS[(4u * q + r) * PF_HID_STRIDE + col0 + j] = h > 0.f ? h : 0.f;       // F.relu, model.py:24/30
                }
            }
        }
    }
    __syncthreads();
    PF_STAMP();

    // layer 2 (actor): wave w owns logit columns 48w + 3c + j (j = 0..2), K = the 180 actor hidden units
    {
        pf_f32x4 acc[3];
        for (int j = 0; j < 3; j++) acc[j] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
        const u32 col0 = l2col0;
        const bool live = l2live;
        const float *ap = hidS + c * PF_HID_STRIDE + PF_HID + q;
        float av[S2];
#pragma unroll
        for (int s = 0; s < S2; s++) av[s] = ap[4 * s];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < S2; s++) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw2[s][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw2[s][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw2[s][2], acc[2], 0, 0, 0);
        }
        if (live) {
            for (int j = 0; j < 3; j++) {
                float bias = bias2[j];
                for (int r = 0; r < 4; r++) {
                    u32 row = 4u * q + r;
                    float v = acc[j][r] + bias;
                    lgS[row * PF_LOG_STRIDE + col0 + j] = v;
                    if (logits_out && g0 + row < n) logits_out[(size_t)(g0 + row) * PF_ACT + col0 + j] = v;
                }
            }
        }
        // critic (model.py:22-26): wave 3 has a quarter of its layer-2 lanes idle; lane (row c, quarter q) sums k = q (mod 4)
        if (w == 3u) {
            float sum = 0.f;
            const float *hp = hidS + c * PF_HID_STRIDE;
#pragma unroll
            for (int s = 0; s < PF_HID / 4; s++) sum = fmaf(hp[4 * s + q], w2cS[4 * s + q], sum);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (q == 0u && g0 + c < n) value[g0 + c] = sum + b2c_v;
        }
    }
    __syncthreads();
    PF_STAMP();

    // head: wave w samples games 4w .. 4w+3 of the tile, 16 lanes each
    {
        float x[HEAD_PER_LANE];
        const float *lg = lgS + hrow * PF_LOG_STRIDE + (c < 15u ? 12u * c : 0u);
        for (int j = 0; j < HEAD_PER_LANE; j++) x[j] = lg[j];
        policy_head_rows(x, lgS + hrow * PF_LOG_STRIDE, okbits, seed, counter, hgc, l, hg < n, action, logp, entropy, id_base);
    }
    PF_STAMP();
#if defined(AZ_PF_PROFILE)
    if (tid == 0u && logits_out == nullptr) for (int i = 0; i + 1 < pf_i; i++) atomicAdd((unsigned long long *)(entropy + n) + i, (unsigned long long)(pf_t[i + 1] - pf_t[i]));
#endif
    // the launch advances the device-resident step counter itself: the LAST workgroup to finish does it (every workgroup read
    // the counter before it could finish); counter_dev[1] is the completion ticket and returns to zero
    if (counter_dev && advance && tid == 0u) {
        __threadfence();
        u64 done = atomicAdd((unsigned long long *)(counter_dev + 1), 1ull);
        if (done == (u64)gridDim.x - 1ull) {
            counter_dev[1] = 0ull;
            counter_dev[0] += (u64)advance;
            __threadfence();
        }
    }
}


// ---- arguments of the persistent policy rollout (azul_rollout2.hpp: a whole window of moves in ONE launch) ---------------------------
struct RolloutArgs {
    int n_steps;
    float *obs;          // [T+1][N][136]  slot t = what the policy saw at move t (slot 0 is written from the current state)
    uint8_t *mask;       // [T+1][N][180]
    uint8_t *player;     // [T+1][N]
    i32 *action;         // [T][N]
    i32 *reward;         // [T][N]
    uint8_t *done;       // [T][N]
    float *value;        // [T][N]
    float *logp;         // [T][N]
    float *entropy;      // [T][N]
    uint8_t *status;     // [N]  status of the last move
    float *returns;      // [T][N] optional: the window's discounted returns (nn_runner.py:70-76), written by the kernel itself when T <= 32
    float gamma;
    u64 seed, counter;
    u64 *counter_dev;    // optional [2]: [0] added to `counter`, advanced by n_steps; [1] completion ticket
    // ---- network opponent (azul_policy_rollout2_kernel<LID, 2>; game_runner.py:27-30 GameRunner(opponent=Agent(...))) ----
    PolicyWeights Wopp;  // the opponent's net in the same layouts (only forward_actor's half is read: w1t columns 180..359, b1[180..359], w2a_t, b2a)
    u64 opp_seed;        // reply j of a step samples with Philox key opp_seed + j (AZUL_POLICY_ARGMAX: action_selection "Max", agent.py:79-80)
    i32 *opp_action;     // optional [T][opp_slots][N]: the opponent's answers in the order they were played (slots beyond a step's replies untouched)
    float *opp_logp;     // optional [T][opp_slots][N]: log-probability of each answer under the opponent's masked softmax
    uint8_t *opp_replies;// optional [T][N]: opponent moves played inside the step (the next episode's opening moves included)
    int opp_slots;
};
