// azul_learner.hpp -- device code of row N2: the A2C update of agent.py:39-62 as hand-written gfx950 kernels; included by
// azul_kernels.hip after azul_policy.hpp (shares its constants, PolicyWeights and the 16-lane row reductions).
//
//   azul_a2c_grad_kernel     forward + backward of the reference's loss for tiles of 16 samples, everything GEMM-shaped on the f32
//                            matrix cores (v_mfma_f32_16x16x4_f32).  A workgroup of 8 waves walks over its share of the sample tiles
//                            and keeps its partial weight gradients IN REGISTERS for the whole launch (dW1: 207 16x16 tiles, dW2: 144
//                            tiles -> 176 accumulator registers per lane), then writes one partial gradient vector per workgroup.
//   azul_a2c_reduce_kernel   sums the per-workgroup partials in a fixed order (deterministic) into one flat gradient + loss terms.
//
// Loss per sample i (agent.py:45-57; n = number of samples of the whole batch, all ranks):
//     adv = q - v                                      (NOT detached in the actor term, like the reference)
//     L_i = ( -logp[a] * adv  +  0.5 * adv^2  +  0.1 * ( -mean_{j legal} logp[j] ) ) / n
// hence   dL/dv       = ( logp[a] - adv ) / n
//         dL/dlogp_j  = ( -adv [j == a]  -  0.1 / |legal| [j legal] ) / n
//         dL/dlogit_k = dL/dlogp_k - softmax_k * sum_j dL/dlogp_j            (masked log-softmax; 0 for illegal k)
// Gradient vector layout (k-major like the forward weights): dw1t [136][360] | db1 [360] | dw2c [180] | db2c [1]
// | (1 pad) | dw2a_t [180][180] | db2a [180]  = 82082 floats, followed by the four loss sums (actor, critic, entropy, count of
// samples used).  The same layout holds the k-major master copy of the parameters and Adam's moments (azul_a2c_apply_kernel).
#pragma once

constexpr int LG_P_W1 = 0, LG_P_B1 = LG_P_W1 + PF_IN * PF_H2, LG_P_W2C = LG_P_B1 + PF_H2, LG_P_B2C = LG_P_W2C + PF_HID,
              LG_P_W2A = LG_P_B2C + 2 /* one pad float: the matrix starts 8-byte aligned */, LG_P_B2A = LG_P_W2A + PF_HID * PF_ACT,
              LG_P_PARAMS = LG_P_B2A + PF_ACT, LG_P_LOSS = LG_P_PARAMS, LG_P_TOTAL = LG_P_PARAMS + 4;
static_assert(LG_P_PARAMS == 82082 && LG_P_W2A % 2 == 0, "ActorCritic(136, 180, 180): 82081 parameters + 1 pad");

constexpr u32 LG_WAVES = 8, LG_AHEAD = 6 /* k-steps of weights in flight; 4..6 measured alike, 8 and 12 slower */;
constexpr int LG_ADEPTH = 3;
constexpr int LG_SUB = 2, LG_M = PF_GAMES * LG_SUB;           // samples per pass: two 16-row MFMA tiles share every streamed weight fragment
constexpr int LG_F_TILES = 9, LG_C_TILES = 23;                 // dW1: 136 -> 9 feature tiles, 360 -> 23 column tiles
constexpr int LG_C_PER_WAVE = (LG_C_TILES + (int)LG_WAVES - 1) / (int)LG_WAVES;      // 3: wave w owns column tiles w, w + 8, w + 16

struct LearnerArgs {
    const float *obs;        // [n][136]
    const uint8_t *mask;     // [n][180]
    const i32 *action;       // [n]
    const float *qvals;      // [n]
    u32 n;                   // samples of this launch (this rank)
    float inv_n;             // 1 / (samples of the whole batch)
    const float *w2a;        // [180 actions][180 hidden]: actor_linear2.weight as PyTorch stores it (for dh = dlogits @ W)
    float *partial;          // [gridDim.x][LG_P_TOTAL]
    const i32 *index;        // optional [n]: sample s lives in row index[s] of obs / mask / action / qvals (a device-built selection)
    const i32 *n_dev;        // optional: the sample count in device memory (overrides n; no host round trip to learn it)
    const float *inv_n_dev;  // optional: 1 / (samples of the whole batch) in device memory (overrides inv_n)
};

// One 180-deep GEMM pass of the gradient kernel: NT column tiles of this wave (NT == 2: columns col0, col0 + 1 of a lane, one 8-byte
// load per k-step; NT == 1: column col0, one 4-byte load) times the two 16-row tiles of the pass.  B fragments stream from L2
// through the buffer descriptor `rs` (k-major rows of PF_ACT floats), A fragments come from LDS (ap, rows 16 u + c).  The first
// LG_AHEAD k-steps' fragments are requested by the caller a phase earlier (lg_request180), which hides the L2 latency of the ramp.
#define LG_LOAD_B(NT_, s) ((NT_) == 2 ? __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (4 * (s)) * PF_ACT * 4, 0)) \
                                      : make_float2(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (4 * (s)) * PF_ACT * 4, 0)), 0.f))
__device__ __forceinline__ void lg_request180(bool two, const __amdgpu_buffer_rsrc_t rs, u32 voff, float2 (&pre)[LG_AHEAD])
{
    if (two) {
#pragma unroll
        for (int s = 0; s < (int)LG_AHEAD; s++) pre[s] = LG_LOAD_B(2, s);
    } else {
#pragma unroll
        for (int s = 0; s < (int)LG_AHEAD; s++) pre[s] = LG_LOAD_B(1, s);
    }
}

template <int NT>
__device__ __forceinline__ void lg_gemm180(const __amdgpu_buffer_rsrc_t rs, u32 voff, const float *ap, int sub_stride, const float2 (&pre)[LG_AHEAD],
                                           pf_f32x4 (&acc)[LG_SUB][2])
{
    for (int u = 0; u < LG_SUB; u++) acc[u][0] = acc[u][1] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
    float2 bw[PF_HID / 4];
#pragma unroll
    for (int s = 0; s < (int)LG_AHEAD; s++) bw[s] = pre[s];
    // A fragments LG_ADEPTH k-steps ahead: a step is 64..128 cycles of matrix pipe per wave, an LDS round trip is longer
    float af[PF_HID / 4][LG_SUB];
#pragma unroll
    for (int s = 0; s < LG_ADEPTH; s++) for (int u = 0; u < LG_SUB; u++) af[s][u] = ap[u * sub_stride + 4 * s];
#pragma unroll
    for (int s = 0; s < PF_HID / 4; s++) {
        if (s + (int)LG_AHEAD < PF_HID / 4) bw[s + LG_AHEAD] = LG_LOAD_B(NT, s + LG_AHEAD);
        if (s + LG_ADEPTH < PF_HID / 4) for (int u = 0; u < LG_SUB; u++) af[s + LG_ADEPTH][u] = ap[u * sub_stride + 4 * (s + LG_ADEPTH)];
        float av[LG_SUB];
        for (int u = 0; u < LG_SUB; u++) av[u] = af[s][u];
        for (int u = 0; u < LG_SUB; u++) {
            acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[s].x, acc[u][0], 0, 0, 0);
            if (NT == 2) acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[s].y, acc[u][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

#if defined(AZ_LG_PROFILE)
// diagnostic build only (tools/grad_profile.py): s_memtime ticks of wave 0 per phase, summed over passes and workgroups
__device__ unsigned long long lg_prof_dev[16];
#define LG_STAMP(i) do { if (w == 0u) { __builtin_amdgcn_s_waitcnt(0); u64 t_ = __builtin_amdgcn_s_memtime(); lg_acc[i] += t_ - lg_t; lg_t = t_; } } while (0)
#else
#define LG_STAMP(i) do { } while (0)
#endif

__global__ void __launch_bounds__(64 * LG_WAVES) azul_a2c_grad_kernel(PolicyWeights W, LearnerArgs a)
{
    __shared__ float obsS[LG_M * PF_OBS_STRIDE];         // x      [32][136 (+pad, zero)]
    __shared__ float hidS[LG_M * PF_HID_STRIDE];         // relu h [32][360 (+pad, zero)]
    __shared__ float lgS[LG_M * PF_LOG_STRIDE];          // logits, then dL/dlogits [32][180 (+pad, zero)]
    __shared__ float dzS[LG_M * PF_HID_STRIDE];          // dL/dz  [32][360 (+pad, zero)]
    __shared__ float w2cS[PF_HID];
    __shared__ float valS[LG_M], dvS[LG_M];
    __shared__ float b1S[PF_H2 + 24], b2aS[PF_ACT + 12]; // biases (+ zero pads: the dead columns of the last waves)
    __shared__ float gw2cS[2][PF_HID];                   // dw2c partial sums of the two 16-sample halves (accumulated in LDS: the
                                                         // register file belongs to the weight-gradient tiles)
    __shared__ u32 idxS[2][LG_M];                        // source rows of this pass / the next one (0xffffffff: past the batch)
    __shared__ float lossS[5][LG_M];                     // loss-term (+ dL/dv) partials of the 32 head row-groups, summed in a fixed order
    // lane constants are RE-DERIVED at the start of every phase from an opaque copy of threadIdx.x (LG_LANE): otherwise the compiler
    // hoists every per-lane address out of the pass loop, and those ~30 loop-invariant registers push the weight-gradient
    // accumulators (180 of the 256 registers) into scratch memory
    u32 tid = threadIdx.x, c = tid & 15u, q = (tid >> 4) & 3u;
#define LG_LANE() do { u32 t_ = threadIdx.x; asm volatile("" : "+v"(t_)); tid = t_; c = t_ & 15u; q = (t_ >> 4) & 3u; } while (0)
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const u32 n = a.n_dev ? (u32)*a.n_dev : a.n, n_tiles = (n + LG_M - 1) / LG_M;
    const float inv_n = a.inv_n_dev ? *a.inv_n_dev : a.inv_n;

    // one-time LDS state: zero everything (the pad columns must stay zero: they feed the padded gradient tiles)
    for (u32 i = tid; i < (u32)(LG_M * PF_OBS_STRIDE); i += 64u * LG_WAVES) obsS[i] = 0.f;
    for (u32 i = tid; i < (u32)(LG_M * PF_HID_STRIDE); i += 64u * LG_WAVES) { hidS[i] = 0.f; dzS[i] = 0.f; }
    for (u32 i = tid; i < (u32)(LG_M * PF_LOG_STRIDE); i += 64u * LG_WAVES) lgS[i] = 0.f;
    if (tid < (u32)PF_HID) { w2cS[tid] = W.w2c[tid]; gw2cS[0][tid] = 0.f; gw2cS[1][tid] = 0.f; }
    if (tid < (u32)(PF_H2 + 24)) b1S[tid] = tid < (u32)PF_H2 ? W.b1[tid] : 0.f;
    if (tid < (u32)(PF_ACT + 12)) b2aS[tid] = tid < (u32)PF_ACT ? W.b2a[tid] : 0.f;
    if (tid < 5u * LG_M) lossS[tid / LG_M][tid % LG_M] = 0.f;
    lds_barrier();
    // a column of ones next to the observations / the actor's hidden units: the weight-gradient MFMAs then produce the bias gradients
    // as one more row (db1 = row 136 of dW1t, db2a = row 180 of dW2a_t) -- no separate column sums
    if (tid < (u32)LG_M) { obsS[tid * PF_OBS_STRIDE + PF_IN] = 1.0f; hidS[tid * PF_HID_STRIDE + PF_H2] = 1.0f; }

    // buffer descriptors: per-lane byte offset + literal k-step offset (see azul_policy_rollout_kernel)
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)W.w1t, 0, PF_IN * PF_H2 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2t = __builtin_amdgcn_make_buffer_rsrc((void *)W.w2a_t, 0, PF_HID * PF_ACT * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2a, 0, PF_ACT * PF_HID * 4, 0x00020000);
    // forward layer 1: wave w owns hidden columns 48w + 3c + j (j = 0..2)
#define f1col0 (48u * w + 3u * c)
#define f1live (f1col0 < (u32)PF_H2)
#define voff1 (((f1live ? f1col0 : 0u) + q * (u32)PF_H2) * 4u)
    // forward layer 2 and dh, 180 columns = 12 tiles over 8 waves, three per SIMD (waves w and w + 4 share one): waves 0..3 own two
    // tiles (columns 32w + 2c + j, j = 0, 1), waves 4..7 one (columns 128 + 16 (w - 4) + c)
    const bool two = w < 4u;
#define f2col0 (two ? 32u * w + 2u * c : 128u + 16u * (w - 4u) + c)
#define f2live (f2col0 < (u32)PF_ACT)
#define voff2 (((f2live ? f2col0 : 0u) + q * (u32)PF_ACT) * 4u)
    const float b2c_v = W.b2c[0];                        // (wave-uniform: a scalar register)
#define LG_LOAD_W1(s, j) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, voff1 + 4u * (j), (4 * (s)) * PF_H2 * 4, 0))
// (the lane's three adjacent columns as one 8-byte and one 4-byte load: two vector-memory instructions per k-step instead of three)
#define LG_LOAD_W1P(s) __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs1, voff1, (4 * (s)) * PF_H2 * 4, 0))
#define LG_LOAD_W1_3(dst, s) do { const float2 p_ = LG_LOAD_W1P(s); (dst)[0] = p_.x; (dst)[1] = p_.y; (dst)[2] = LG_LOAD_W1(s, 2); } while (0)

    // register-resident partial gradients of this workgroup
    pf_f32x4 gW2[3][6];                                  // dW2a_t tiles: hidden tiles 3 (w & 3) + i, action tiles 6 (w >> 2) + j
    pf_f32x4 gW1[LG_C_PER_WAVE][LG_F_TILES];             // dW1t tiles: column tiles w + 8 i (i = 0..2), all nine feature tiles
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) gW2[i][j] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < LG_C_PER_WAVE; i++) for (int f = 0; f < LG_F_TILES; f++) gW1[i][f] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
    const u32 wk = w & 3u, wj = w >> 2;
    lds_barrier();
#if defined(AZ_LG_PROFILE)
    u64 lg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lg_t = __builtin_amdgcn_s_memtime();
#endif

    // Global-memory latency stays off the critical path: the source rows of a pass (the selection's indirection) are staged in LDS
    // one pass ahead, its observations are requested before the previous pass's last MFMA phase and land in LDS after it, and the
    // per-sample inputs of the loss (mask bits, action, return) are requested before layer 2 and used after it.
    // (observation row tid / 16, features tid % 16 + 16 j: one source row and one address per thread, literal offsets for the nine loads)
    constexpr int LG_OBS_PER_THREAD = (PF_IN + 15) / 16;           // 9; the last one covers features 128..135 (lanes 0..7 of the group)
#define LG_FETCH_IDX(tile_) ([&]() -> u32 { const u32 s_ = (tile_) * LG_M + tid; return s_ < n ? (a.index ? (u32)a.index[s_] : s_) : 0xffffffffu; }())
#define LG_FETCH_OBS(buf) do {                                                                                                  \
        const u32 src_ = idxS[buf][tid >> 4];                                                                                   \
        const float *op_ = a.obs + (size_t)(src_ != 0xffffffffu ? src_ : 0u) * PF_IN + c;                                       \
        _Pragma("unroll") for (int j = 0; j < LG_OBS_PER_THREAD - 1; j++) ob[j] = op_[16 * j];                                  \
        ob[LG_OBS_PER_THREAD - 1] = op_[c < 8u ? 128 : 0]; } while (0)
#define LG_STORE_OBS(buf) do {                                                                                                  \
        const bool ok_ = idxS[buf][tid >> 4] != 0xffffffffu;                                                                    \
        float *xp_ = obsS + (tid >> 4) * PF_OBS_STRIDE + c;                                                                     \
        _Pragma("unroll") for (int j = 0; j < LG_OBS_PER_THREAD - 1; j++) xp_[16 * j] = ok_ ? ob[j] : 0.f;                      \
        if (c < 8u) xp_[128] = ok_ ? ob[LG_OBS_PER_THREAD - 1] : 0.f; } while (0)
    float ob[LG_OBS_PER_THREAD];
    float pre1[LG_AHEAD][3];                             // layer 1's first weight fragments, requested at the end of the previous pass
    float2 pre2[LG_AHEAD];                               // the same for the two 180-wide GEMMs (layer 2, dh)
#define LG_REQUEST_W1() do { _Pragma("unroll") for (int s = 0; s < (int)LG_AHEAD; s++) LG_LOAD_W1_3(pre1[s], s); } while (0)
    LG_REQUEST_W1();
    if (tid < (u32)LG_M) idxS[0][tid] = LG_FETCH_IDX(blockIdx.x);
    lds_barrier();
    LG_FETCH_OBS(0);
    LG_STORE_OBS(0);
    lds_barrier();
    u32 par = 0;

#pragma unroll 1
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1u) {
        // (this pass's observations are in LDS: rows past the batch are zero and contribute nothing anywhere)
        LG_LANE();
        u32 nidx = 0xffffffffu;
        if (tid < (u32)LG_M) nidx = LG_FETCH_IDX(tile + gridDim.x);       // a tile past the end: every row invalid
        LG_STAMP(0);
        // ---- P1: hidden = relu(x @ w1t + b1): every weight fragment feeds both 16-row tiles
        {
            pf_f32x4 acc[LG_SUB][3];
            for (int u = 0; u < LG_SUB; u++) for (int j = 0; j < 3; j++) acc[u][j] = (pf_f32x4){0.f, 0.f, 0.f, 0.f};
            const float *ap = obsS + c * PF_OBS_STRIDE + q;
            float bw[PF_IN / 4][3];
#pragma unroll
            for (int s = 0; s < (int)LG_AHEAD; s++) for (int j = 0; j < 3; j++) bw[s][j] = pre1[s][j];
            float af[PF_IN / 4][LG_SUB];
#pragma unroll
            for (int s = 0; s < LG_ADEPTH; s++) for (int u = 0; u < LG_SUB; u++) af[s][u] = ap[u * PF_GAMES * PF_OBS_STRIDE + 4 * s];
#pragma unroll
            for (int s = 0; s < PF_IN / 4; s++) {
                if (s + (int)LG_AHEAD < PF_IN / 4) LG_LOAD_W1_3(bw[s + LG_AHEAD], s + LG_AHEAD);
                if (s + LG_ADEPTH < PF_IN / 4) for (int u = 0; u < LG_SUB; u++) af[s + LG_ADEPTH][u] = ap[u * PF_GAMES * PF_OBS_STRIDE + 4 * (s + LG_ADEPTH)];
                float av[LG_SUB];
                for (int u = 0; u < LG_SUB; u++) av[u] = af[s][u];
                for (int j = 0; j < 3; j++)
                    for (int u = 0; u < LG_SUB; u++) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bw[s][j], acc[u][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            lg_request180(two, rs2t, voff2, pre2);             // layer 2's first weight fragments: in flight across the barrier
            if (f1live)
                for (int u = 0; u < LG_SUB; u++)
                    for (int j = 0; j < 3; j++)
                        for (int rr = 0; rr < 4; rr++) {
                            float h = acc[u][j][rr] + b1S[f1col0 + j];
                            hidS[(16u * u + 4u * q + rr) * PF_HID_STRIDE + f1col0 + j] = h > 0.f ? h : 0.f;
                        }
        }
        lds_barrier();
        LG_STAMP(1);
        LG_LANE();
        // ---- P2: value = h_critic . w2c + b2c (four rows per wave, 16 lanes each), logits = h_actor @ w2a_t + b2a
        const u32 hsrc = idxS[par][4u * w + q];                   // P3's sample of this 16-lane group; its inputs are requested now
        const bool hvalid = hsrc != 0xffffffffu;
        const u32 hsc = hvalid ? hsrc : 0u;
        u32 mraw[3];
        {
            const u32 *mp = (const u32 *)(a.mask + (size_t)hsc * AZUL_NUM_ACTIONS + (c < 15u ? 12u * c : 0u));
            for (int d = 0; d < 3; d++) mraw[d] = mp[d];
        }
        const i32 hact = a.action[hsc];
        const float hq = a.qvals[hsc];
        {
            const u32 row = 4u * w + q;
            const float *hp = hidS + row * PF_HID_STRIDE + c;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 12; i++) { const u32 k = c + 16u * (u32)i; sum = fmaf(k < (u32)PF_HID ? hp[16 * i] : 0.f, w2cS[k < (u32)PF_HID ? k : 0u], sum); }
            sum = row_sum(sum);
            if (c == 0u) valS[row] = sum + b2c_v;
            pf_f32x4 acc[LG_SUB][2];
            const float *ap = hidS + c * PF_HID_STRIDE + PF_HID + q;
            if (two) lg_gemm180<2>(rs2t, voff2, ap, PF_GAMES * PF_HID_STRIDE, pre2, acc);
            else lg_gemm180<1>(rs2t, voff2, ap, PF_GAMES * PF_HID_STRIDE, pre2, acc);
            if (f2live)
                for (int u = 0; u < LG_SUB; u++)
                    for (int rr = 0; rr < 4; rr++) {
                        float *lp = lgS + (16u * u + 4u * q + rr) * PF_LOG_STRIDE + f2col0;
                        lp[0] = acc[u][0][rr] + b2aS[f2col0];
                        if (two) lp[1] = acc[u][1][rr] + b2aS[f2col0 + 1u];
                    }
        }
        lds_barrier();
        LG_STAMP(2);
        LG_LANE();
        // ---- P3: per sample (16 lanes each, four per wave): masked log-softmax, loss terms, dL/dlogits (in place), dL/dv
        {
            const u32 row = 4u * w + q;
            const bool valid = hvalid;
            float *lg = lgS + row * PF_LOG_STRIDE + 12u * c;       // lane c == 15 owns the pad columns 180..191
            float x[HEAD_PER_LANE];
            for (int j = 0; j < HEAD_PER_LANE; j++) x[j] = lg[j];
            u32 okbits = 0;
            for (int d = 0; d < 3; d++) {                          // byte b != 0 -> bit b: fold every byte onto its bit 0, gather with one multiply
                u32 t = mraw[d] | (mraw[d] >> 4);
                t |= t >> 2; t |= t >> 1;
                okbits |= ((((t & 0x01010101u) * 0x01020408u) >> 24) & 15u) << (4 * d);
            }
            okbits = (valid && c < 15u) ? okbits : 0u;
            const float NEG = -3.0e38f;
            float m = NEG;
            for (int j = 0; j < HEAD_PER_LANE; j++) m = fmaxf(m, ((okbits >> j) & 1u) ? x[j] : NEG);
            m = row_max(m);
            const u32 cnt = row_sum_u((u32)__popc(okbits));
            float z[HEAD_PER_LANE], e[HEAD_PER_LANE], mine = 0.f, zs = 0.f;
            for (int j = 0; j < HEAD_PER_LANE; j++) {
                bool ok = (okbits >> j) & 1u;
                z[j] = ok ? x[j] - m : 0.f;
                e[j] = ok ? __expf(z[j]) : 0.f;
                mine += e[j];
                zs += z[j];
            }
            const float S = row_sum(mine), logS = __logf(S), zsum = row_sum(zs);
            const bool use = valid && cnt != 0u;                   // rows without a legal action carry no sample
            const i32 act = hact;
            const i32 aj = act - (i32)(12u * c);                   // index of the chosen action inside this lane, if any
            float mine_lpa = 0.f;
            for (int j = 0; j < HEAD_PER_LANE; j++) if (j == aj && ((okbits >> j) & 1u)) mine_lpa = z[j] - logS;
            const float logp_a = row_sum(mine_lpa);
            const float adv = hq - valS[row];
            const float ent_w = 0.1f / (float)(cnt ? cnt : 1u);
            const float G = -adv - 0.1f;                           // sum_j dL/dlogp_j (times n)
            const float invS = 1.0f / S;
            for (int j = 0; j < HEAD_PER_LANE; j++) {
                bool ok = (okbits >> j) & 1u;
                float gj = (j == aj ? -adv : 0.f) - ent_w;
                float d = (gj - e[j] * invS * G) * inv_n;
                lg[j] = (use && ok) ? d : 0.f;
            }
            if (c == 0u) {
                const float dv = use ? (logp_a - adv) * inv_n : 0.f;
                dvS[row] = dv;
                lossS[4][row] += dv;                     // (slot `row` belongs to this lane alone: fixed summation order)
                if (use) {
                    lossS[0][row] += -logp_a * adv;
                    lossS[1][row] += adv * adv;
                    lossS[2][row] += -(zsum / (float)cnt - logS);
                    lossS[3][row] += 1.f;
                }
            }
            if (tid < (u32)LG_M) idxS[par ^ 1u][tid] = nidx;
        }
        lds_barrier();
        LG_STAMP(3);
        LG_LANE();
        lg_request180(two, rs2, voff2, pre2);                // P4c's first weight fragments
        // ---- P4a: dw2c[k] += sum_s dv[s] * h_critic[s][k]: thread (k = tid & 255, half = tid >> 8) over its 16 samples
        {
            const u32 k = tid & 255u, half = tid >> 8;
            if (k < (u32)PF_HID) {
                const float *hp = hidS + 16u * half * PF_HID_STRIDE + k;
                const float *dp = dvS + 16u * half;
                float sw = 0.f;
#pragma unroll
                for (int s = 0; s < 16; s++) sw = fmaf(dp[s], hp[s * PF_HID_STRIDE], sw);
                gw2cS[half][k] += sw;
            }
        }
        // ---- P4b: dW2a_t[k][j] += sum_s h_actor[s][k] * dlogits[s][j]   (samples are the MFMA's k: chunks of four; hidden "unit" 180
        //      is the column of ones: that row is db2a)
        const float *ha = hidS + q * PF_HID_STRIDE + PF_HID + 48u * wk + c;
        const float *la = lgS + q * PF_LOG_STRIDE + 96u * wj + c;
#pragma unroll
        for (int mch = 0; mch < LG_M / 4; mch++) {
            float af[3], bf[6];
            for (int i = 0; i < 3; i++) af[i] = ha[4 * mch * PF_HID_STRIDE + 16 * i];
            for (int j = 0; j < 6; j++) bf[j] = la[4 * mch * PF_LOG_STRIDE + 16 * j];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 6; j++) gW2[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], gW2[i][j], 0, 0, 0);
        }
        LG_STAMP(4);
        LG_LANE();
        // ---- P4c: dz.  Critic half: dv * w2c * relu' (row tid / 16, units tid % 16 + 16 i); actor half: dh = dlogits @ W2a, times relu'
        {
            const u32 row = tid >> 4;
            const float dvr = dvS[row];
            const float *hp = hidS + row * PF_HID_STRIDE + c;
            float *zp = dzS + row * PF_HID_STRIDE + c;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const u32 k = c + 16u * (u32)i;
                if (k < (u32)PF_HID) zp[16 * i] = hp[16 * i] > 0.f ? dvr * w2cS[k] : 0.f;
            }
            pf_f32x4 acc[LG_SUB][2];
            const float *ap = lgS + c * PF_LOG_STRIDE + q;
            if (two) lg_gemm180<2>(rs2, voff2, ap, PF_GAMES * PF_LOG_STRIDE, pre2, acc);
            else lg_gemm180<1>(rs2, voff2, ap, PF_GAMES * PF_LOG_STRIDE, pre2, acc);
            if (f2live)
                for (int u = 0; u < LG_SUB; u++)
                    for (int rr = 0; rr < 4; rr++) {
                        const u32 o = (16u * u + 4u * q + rr) * PF_HID_STRIDE + PF_HID + f2col0;
                        dzS[o] = hidS[o] > 0.f ? acc[u][0][rr] : 0.f;
                        if (two) dzS[o + 1u] = hidS[o + 1u] > 0.f ? acc[u][1][rr] : 0.f;
                    }
        }
        lds_barrier();
        LG_STAMP(5);
        LG_LANE();
        // ---- P5: dW1t[f][col] += sum_s x[s][f] * dz[s][col]   ("feature" 136 is the column of ones: that row is db1)
        LG_FETCH_OBS(par ^ 1u);                          // the next pass's observations: in flight during the MFMAs below
        {
            // per-lane bases + compile-time offsets: the wave's column tiles are a constant stride apart
            const float *xa = obsS + q * PF_OBS_STRIDE + c;
            const float *za = dzS + q * PF_HID_STRIDE + 16u * w + c;
#pragma unroll
            for (int mch = 0; mch < LG_M / 4; mch++) {
                float af[LG_F_TILES], bf[LG_C_PER_WAVE];
                for (int f = 0; f < LG_F_TILES; f++) af[f] = xa[4 * mch * PF_OBS_STRIDE + 16 * f];
                for (int i = 0; i < LG_C_PER_WAVE; i++) bf[i] = za[4 * mch * PF_HID_STRIDE + 16 * (int)LG_WAVES * i];
                for (int i = 0; i < LG_C_PER_WAVE; i++)
                    for (int f = 0; f < LG_F_TILES; f++) gW1[i][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[f], bf[i], gW1[i][f], 0, 0, 0);
            }
        }
        lds_barrier();                                 // the next pass overwrites obsS / hidS / lgS / dzS
        LG_STAMP(6);
        LG_LANE();
        LG_REQUEST_W1();
        LG_STORE_OBS(par ^ 1u);
        lds_barrier();
    }
#if defined(AZ_LG_PROFILE)
    if (tid == 0u) for (int i = 0; i < 7; i++) atomicAdd(&lg_prof_dev[i], (unsigned long long)lg_acc[i]);
#endif

    LG_LANE();
    // ---- this workgroup's partial gradient vector
    float *out = a.partial + (size_t)blockIdx.x * LG_P_TOTAL;
    for (int i = 0; i < LG_C_PER_WAVE; i++)
        for (int f = 0; f < LG_F_TILES; f++) {
            const u32 col = 16u * (w + LG_WAVES * (u32)i) + c;
            for (int rr = 0; rr < 4; rr++) {
                const u32 ff = 16u * (u32)f + 4u * q + rr;
                if (ff < (u32)PF_IN && col < (u32)PF_H2) out[LG_P_W1 + ff * PF_H2 + col] = gW1[i][f][rr];
            }
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++) {
            const u32 col = 16u * (6u * wj + j) + c;
            for (int rr = 0; rr < 4; rr++) {
                const u32 k = 16u * (3u * wk + i) + 4u * q + rr;
                if (k < (u32)PF_HID && col < (u32)PF_ACT) out[LG_P_W2A + k * PF_ACT + col] = gW2[i][j][rr];
            }
        }
    // the bias rows of the weight-gradient tiles: feature 136 = tile 8, row 4 * 2 + 0; hidden unit 180 = tile 11 (wk == 3, i == 2), row 4 * 1 + 0
    if (q == 2u)
        for (int i = 0; i < LG_C_PER_WAVE; i++) {
            const u32 col = 16u * (w + LG_WAVES * (u32)i) + c;
            if (col < (u32)PF_H2) out[LG_P_B1 + col] = gW1[i][8][0];
        }
    if (q == 1u && wk == 3u)
        for (int j = 0; j < 6; j++) {
            const u32 col = 16u * (6u * wj + j) + c;
            if (col < (u32)PF_ACT) out[LG_P_B2A + col] = gW2[2][j][0];
        }
    // (the loss terms were accumulated one slot per contributing lane, no atomics: the logged losses are bit-reproducible)
    if (tid < (u32)PF_HID) out[LG_P_W2C + tid] = gw2cS[0][tid] + gw2cS[1][tid];
    if (tid < 5u) {
        float sum = 0.f;
        for (int i = 0; i < LG_M; i++) sum += lossS[tid][i];
        if (tid < 4u) out[LG_P_LOSS + tid] = sum;
        else { out[LG_P_B2C] = sum; out[LG_P_B2C + 1] = 0.f; }
    }
}
#undef f1col0
#undef f1live
#undef voff1
#undef f2col0
#undef f2live
#undef voff2
#undef LG_LANE

// sum of the per-workgroup partials in workgroup order (deterministic); optionally scaled loss sums stay raw (caller divides)
__global__ void __launch_bounds__(256) azul_a2c_reduce_kernel(const float *partial, u32 n_parts, float *grad /* [LG_P_TOTAL] */)
{
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (u32)LG_P_TOTAL) return;
    float s = 0.f;
    u32 i = 0;
    for (; i + 8u <= n_parts; i += 8u) {                 // eight loads in flight, added in workgroup order (the order is what is fixed)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = partial[(size_t)(i + j) * LG_P_TOTAL + p];
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
    }
    for (; i < n_parts; i++) s += partial[(size_t)i * LG_P_TOTAL + p];
    grad[p] = s;
}


// Which steps of a window feed the update: those whose episode ends inside the window (exact Monte-Carlo returns, nn_runner.py:70-76)
// and that carry an action.  done / action are time-major [T][N]; the selection is written game by game, steps ascending, as flat
// indices t * N + g, and its length to count[0].  One workgroup; deterministic order.
__global__ void __launch_bounds__(1024) azul_select_complete_kernel(const uint8_t *done, const i32 *action, int T, u32 N, i32 *index, i32 *count)
{
    __shared__ u32 scanS[1024];
    __shared__ u32 baseS;
    const u32 tid = threadIdx.x;
    if (tid == 0u) baseS = 0u;
    __syncthreads();
    for (u32 g0 = 0; g0 < N; g0 += 1024u) {
        const u32 g = g0 + tid;
        int last = -1;
        u32 kept = 0;
        u64 okmask = 0, okmask_at_done = 0;              // bit t: step t carries an action (windows of up to 64 steps need no second read)
        if (g < N) {
            // one pass without a data-dependent exit (the loads pipeline): running count of usable steps, latched at every `done`
            u32 run = 0;
#pragma unroll 8
            for (int t = 0; t < T; t++) {
                const bool ok = action[(size_t)t * N + g] >= 0;
                run += ok ? 1u : 0u;
                if (t < 64) okmask |= (u64)(ok ? 1u : 0u) << t;
                if (done[(size_t)t * N + g] != 0) { kept = run; last = t; okmask_at_done = okmask; }
            }
        }
        // inclusive scan of `kept` over the 1024 threads (Hillis-Steele in LDS)
        scanS[tid] = kept;
        __syncthreads();
        for (u32 o = 1; o < 1024u; o <<= 1) {
            u32 v = tid >= o ? scanS[tid - o] : 0u;
            __syncthreads();
            scanS[tid] += v;
            __syncthreads();
        }
        u32 pos = baseS + scanS[tid] - kept;
        if (g < N) {
            if (last < 64) {
                for (u64 mbits = okmask_at_done; mbits != 0; mbits &= mbits - 1)       // stores only: ascending steps of this game
                    index[pos++] = (i32)((u32)__builtin_ctzll(mbits) * N + g);
            } else {
                for (int t = 0; t <= last; t++)
                    if (action[(size_t)t * N + g] >= 0) index[pos++] = (i32)((u32)t * N + g);
            }
        }
        __syncthreads();
        if (tid == 1023u) baseS += scanS[1023];
        __syncthreads();
    }
    if (tid == 0u) count[0] = (i32)baseS;
}


// ---- selection over a RING of windows: every step of every episode is trained exactly once ---------------------------------
// The trajectory arrays are rings of R = D * T time slots (D windows of T agent steps); absolute step s lives in slot s mod R.
// After window k (absolute steps kT .. (k+1)T - 1) has been played, game g contributes the steps from `pend[g]` -- the first
// step of its oldest episode that has not been trained yet -- up to its LAST episode end inside window k: episodes that ended
// in this window, including their opening steps recorded in earlier windows (whose returns the caller has chained backwards
// through the ring with azul_discounted_returns' carry).  Steps that have already fallen out of the ring are counted in
// count[1] ("dropped").  Two launches, deterministic order (game by game, steps ascending): counts + block sums, then offsets +
// index writes.  scratch: int32 [3 N + blocks].
// (one WAVE per game: the lanes read 64 time slots at once and ballots replace the per-thread loops over steps)
__global__ void __launch_bounds__(256) azul_select_ring_count_kernel(const uint8_t *done, const i32 *action, int T, int R, u32 N, i32 s_end,
                                                                   const i32 *pend, i32 *scratch, i32 *count)
{
    __shared__ i32 cntS[4];
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const u32 g = blockIdx.x * 4u + w;
    i32 *cntA = scratch, *startA = scratch + N, *lastA = scratch + 2 * (size_t)N, *blockA = scratch + 3 * (size_t)N;
    i32 cnt = 0, start = 0, last = -1, dropped = 0;
    if (g < N) {
        const i32 s_win = s_end - T;
        for (i32 t0 = 0; t0 < T; t0 += 64) {              // the newest window: the LAST episode end of this game inside it
            const i32 t = t0 + (i32)lane;
            const bool d = t < T && done[(size_t)((s_win + t) % R) * N + g] != 0;
            const u64 m = __ballot(d);
            if (m) last = s_win + t0 + 63 - (i32)__builtin_clzll(m);
        }
        if (last >= 0) {
            // oldest step that is still intact in the ring.  The rollout kernel also writes the state AFTER the window into time slot
            // T of the window's view (the next window's slot 0: obs / mask / player only).  For the ring's last window that is the spare
            // slot R; for every other window it is the physical slot of absolute step s_end - R, whose observation and mask are
            // therefore gone (its action / reward / return are not): that step counts as fallen out of the ring.
            const i32 lo_ = s_end - R + ((s_end % R) != 0 ? 1 : 0);
            const i32 lo = lo_ > 0 ? lo_ : 0;
            const i32 p0 = pend[g];
            start = p0 > lo ? p0 : lo;
            dropped = start - p0;
            for (i32 s0 = start; s0 <= last; s0 += 64) {
                const i32 sx = s0 + (i32)lane;
                const bool ok = sx <= last && action[(size_t)(sx % R) * N + g] >= 0;
                cnt += (i32)__builtin_popcountll(__ballot(ok));
            }
        }
        if (lane == 0u) { cntA[g] = cnt; startA[g] = start; lastA[g] = last; }
    }
    if (lane == 0u) cntS[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blockA[blockIdx.x] = cntS[0] + cntS[1] + cntS[2] + cntS[3];
    if (g == 0 && lane == 0u) count[0] = 0;
    if (dropped && lane == 0u) atomicAdd(&count[1], dropped);           // integer: order does not matter
}

__global__ void __launch_bounds__(256) azul_select_ring_write_kernel(const i32 *action, int R, u32 N, i32 *pend, const i32 *scratch, i32 *index,
                                                                   i32 *count, float *countf)
{
    __shared__ i32 redS[256];
    __shared__ i32 baseS;
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const u32 g = blockIdx.x * 4u + w;
    const i32 *cntA = scratch, *startA = scratch + N, *lastA = scratch + 2 * (size_t)N, *blockA = scratch + 3 * (size_t)N;
    // this block's base: the sum of the earlier blocks' totals (integers: any order)
    i32 part = 0;
    for (u32 b = tid; b < blockIdx.x; b += 256u) part += blockA[b];
    redS[tid] = part;
    __syncthreads();
    for (u32 o = 128; o > 0; o >>= 1) { if (tid < o) redS[tid] += redS[tid + o]; __syncthreads(); }
    if (tid == 0) baseS = redS[0];
    __syncthreads();
    const u32 g0 = blockIdx.x * 4u;
    i32 pos = baseS;
    for (u32 v = 0; v < w; v++) pos += (g0 + v < N) ? cntA[g0 + v] : 0;      // the games of this block before mine
    if (g < N) {
        const i32 last = lastA[g];
        if (last >= 0) {
            for (i32 s0 = startA[g]; s0 <= last; s0 += 64) {
                const i32 sx = s0 + (i32)lane;
                const i32 slot = sx % R;
                const bool ok = sx <= last && action[(size_t)slot * N + g] >= 0;
                const u64 m = __ballot(ok);
                if (ok) index[pos + (i32)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (i32)((u32)slot * N + g);
                pos += (i32)__builtin_popcountll(m);
            }
            if (lane == 0u) pend[g] = last + 1;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        i32 total = baseS;
        for (u32 v = 0; v < 4u; v++) total += (g0 + v < N) ? cntA[g0 + v] : 0;
        count[0] = total;
        if (countf) { countf[0] = (float)total; countf[1] = 1.0f / (float)(total > 0 ? total : 1); }     // for the learner: n and 1 / max(n, 1)
    }
}


// Adam (torch.optim.Adam's defaults and arithmetic: lerp for the first moment, bias corrections, eps added to the corrected root)
// on the flat k-major master copy of the parameters, one thread per parameter; the step also lands in the eight PyTorch parameter
// tensors (nn.Linear layouts, i.e. transposed), so the module, the rollout kernels (which read the master copy) and the optimiser
// state never disagree and no re-layout kernels run after an update.
struct ModuleParams {
    float *c1w, *c1b, *c2w, *c2b, *a1w, *a1b, *a2w, *a2b;      // critic_linear1/2, actor_linear1/2: weight [out][in], bias [out]
};

// step counter of the optimiser in device memory: advanced only by an update that has samples (an empty window must not
// move the parameters along the stale moments, nor change the bias corrections of later updates)
__global__ void azul_a2c_step_kernel(i32 *step_dev, const float *n_total_dev)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && (!n_total_dev || *n_total_dev > 0.f)) step_dev[0] += 1;
}

__global__ void __launch_bounds__(256) azul_a2c_apply_kernel(const float *grad, float *flat, float *m, float *v, float lr, float beta1,
                                                             float beta2, float eps, float bias_c1, float bias_c2_sqrt, ModuleParams P,
                                                             const i32 *step_dev, const float *n_total_dev, float n_total_host, float *stats_out)
{
    if (stats_out && blockIdx.x == 0 && threadIdx.x == 0) {
        // the update's loss terms as the reference logs them (agent.py:51-58): means over the batch, ac_loss = 1 a + 0.5 c + 0.1 e
        const float nn = n_total_dev ? *n_total_dev : n_total_host, inv = 1.0f / (nn > 0.f ? nn : 1.0f);
        const float la = grad[LG_P_LOSS] * inv, lc = grad[LG_P_LOSS + 1] * inv, le = grad[LG_P_LOSS + 2] * inv;
        stats_out[0] = la; stats_out[1] = lc; stats_out[2] = le; stats_out[3] = 1.0f * la + 0.5f * lc + 0.1f * le; stats_out[4] = nn;
    }
    if (n_total_dev && !(*n_total_dev > 0.f)) return;    // a window without a finished episode: no samples, no step
    if (step_dev) {                                      // bias corrections from the device-resident step (torch.optim.Adam: 1 - beta^step)
        __shared__ float bcS[2];
        if (threadIdx.x == 0) {
            const double st = (double)step_dev[0];
            bcS[0] = (float)(1.0 - pow((double)beta1, st));
            bcS[1] = (float)sqrt(1.0 - pow((double)beta2, st));
        }
        __syncthreads();
        bias_c1 = bcS[0];
        bias_c2_sqrt = bcS[1];
    }
    const u32 p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (u32)LG_P_PARAMS || p == (u32)LG_P_B2C + 1u) return;
    const float g = grad[p];
    const float m1 = m[p] + (g - m[p]) * (1.0f - beta1);               // exp_avg.lerp_(grad, 1 - beta1)
    const float v1 = v[p] * beta2 + (1.0f - beta2) * g * g;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[p] = m1;
    v[p] = v1;
    const float denom = sqrtf(v1) / bias_c2_sqrt + eps;
    const float w = flat[p] - (lr / bias_c1) * (m1 / denom);           // param.addcdiv_(exp_avg, denom, value = -step_size)
    flat[p] = w;
    if (p < (u32)LG_P_B1) {
        const u32 k = p / (u32)PF_H2, col = p - k * (u32)PF_H2;
        if (col < (u32)PF_HID) P.c1w[col * PF_IN + k] = w; else P.a1w[(col - PF_HID) * PF_IN + k] = w;
    } else if (p < (u32)LG_P_W2C) {
        const u32 col = p - (u32)LG_P_B1;
        if (col < (u32)PF_HID) P.c1b[col] = w; else P.a1b[col - PF_HID] = w;
    } else if (p < (u32)LG_P_B2C) {
        P.c2w[p - (u32)LG_P_W2C] = w;
    } else if (p == (u32)LG_P_B2C) {
        P.c2b[0] = w;
    } else if (p < (u32)LG_P_B2A) {
        const u32 i = p - (u32)LG_P_W2A, k = i / (u32)PF_ACT, j = i - k * (u32)PF_ACT;
        P.a2w[j * PF_HID + k] = w;
    } else {
        P.a2b[p - (u32)LG_P_B2A] = w;
    }
}
