// simt_learner.cpp -- TEST-ONLY: the A2C gradient kernel (csrc/azul_learner.hpp: azul_a2c_grad_kernel -- forward + backward of the
// reference's loss on the f32 matrix cores, weight-gradient tiles in registers) and the ActorCritic forward + head kernel
// (csrc/azul_policy.hpp: azul_policy_forward_kernel), UNMODIFIED, as workgroups of emulated wavefronts (simt/simt.hpp) -- a CPU check of
// their arithmetic against torch autograd and, under ASan / UBSan, of every LDS and global index they form.
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
using namespace az;
#include "azul_selfplay_kernels.hpp"       // (includes azul_selfplay2.hpp; the returns scans live here)
#include "azul_policy.hpp"
#include "azul_rollout2.hpp"
#include "azul_learner.hpp"

struct GradJob { PolicyWeights W; LearnerArgs a; };
static void grad_lane(void *arg) { GradJob *j = (GradJob *)arg; azul_a2c_grad_kernel(j->W, j->a); }

struct FwdJob { const float *obs; const uint8_t *mask; PolicyWeights W; u64 seed, counter; u32 n; float *value; i32 *action; float *logp, *entropy, *logits; u32 id_base; };
static void fwd_lane(void *arg)
{
    FwdJob *j = (FwdJob *)arg;
    azul_policy_forward_kernel(j->obs, j->mask, j->W, j->seed, j->counter, nullptr, 0, j->n, j->value, j->action, j->logp, j->entropy, j->logits, j->id_base);
}

extern "C" {

unsigned long long sl_buffer_oob() { return simt::g_buffer_oob; }
void sl_layout(int *out) { out[0] = LG_P_W1; out[1] = LG_P_B1; out[2] = LG_P_W2C; out[3] = LG_P_B2C; out[4] = LG_P_W2A; out[5] = LG_P_B2A; out[6] = LG_P_LOSS; out[7] = LG_P_TOTAL; }

// the gradient kernel on n samples with `parts` workgroups, then azul_a2c_reduce_kernel's sum in workgroup order -> grad [LG_P_TOTAL]
long long sl_gradients(int n, int parts, const float *obs, const uint8_t *mask, const i32 *action, const float *qvals, const i32 *index, float inv_n,
                       const float *w1t, const float *b1, const float *w2c, const float *b2c, const float *w2a_t, const float *b2a,
                       const float *w2a, float *partial /* [parts][LG_P_TOTAL] */, float *grad)
{
    GradJob j;
    memset(&j, 0, sizeof(j));
    j.W = {w1t, b1, w2c, b2c, w2a_t, b2a};
    j.a.obs = obs; j.a.mask = mask; j.a.action = action; j.a.qvals = qvals; j.a.n = (u32)n; j.a.inv_n = inv_n; j.a.w2a = w2a;
    j.a.partial = partial; j.a.index = index;
    simt::g_grid_dim = {(unsigned)parts, 1, 1};
    long long ops = 0;
    for (int blk = 0; blk < parts; blk++) {
        simt::g_block_idx = {(unsigned)blk, 0, 0};
        ops += (long long)simt::run_workgroup(grad_lane, &j, (int)LG_WAVES, 512u << 10);
    }
    for (int p = 0; p < LG_P_TOTAL; p++) {                 // (azul_a2c_reduce_kernel: the partials added in workgroup order)
        float s = 0.f;
        for (int i = 0; i < parts; i++) s += partial[(size_t)i * LG_P_TOTAL + p];
        grad[p] = s;
    }
    return ops;
}

long long sl_forward(int n, const float *obs, const uint8_t *mask, const float *w1t, const float *b1, const float *w2c, const float *b2c,
                     const float *w2a_t, const float *b2a, unsigned long long seed, unsigned long long counter, unsigned id_base,
                     float *value, i32 *action, float *logp, float *entropy, float *logits)
{
    FwdJob j = {obs, mask, {w1t, b1, w2c, b2c, w2a_t, b2a}, seed, counter, (u32)n, value, action, logp, entropy, logits, id_base};
    const unsigned blocks = ((unsigned)n + PF_GAMES - 1u) / PF_GAMES;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 0;
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(fwd_lane, &j, 4);
    }
    return ops;
}


// azul_select_episode_samples' two launches (one wave per game, four games per workgroup) on host memory
struct SelJob { const uint8_t *done; const i32 *action; int T, R; u32 N; i32 s_end; i32 *pend, *scratch, *index, *count; float *countf; int phase; };
static void sel_lane(void *arg)
{
    SelJob *j = (SelJob *)arg;
    if (j->phase == 0) azul_select_ring_count_kernel(j->done, j->action, j->T, j->R, j->N, j->s_end, j->pend, j->scratch, j->count);
    else azul_select_ring_write_kernel(j->action, j->R, j->N, j->pend, j->scratch, j->index, j->count, j->countf);
}
long long sl_select_ring(const uint8_t *done, const i32 *action, int T, int D, int n, int steps_played, i32 *pend, i32 *index, i32 *count,
                         float *countf, i32 *scratch)
{
    SelJob j = {done, action, T, T * D, (u32)n, steps_played, pend, scratch, index, count, countf, 0};
    const unsigned blocks = ((unsigned)n + 3u) / 4u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 0;
    for (j.phase = 0; j.phase < 2; j.phase++)
        for (unsigned blk = 0; blk < blocks; blk++) {
            simt::g_block_idx = {blk, 0, 0};
            ops += (long long)simt::run_workgroup(sel_lane, &j, 4);
        }
    return ops;
}


// azul_discounted_returns_ring's launch: the scan over a ring of time slots, 64 games per one-wave workgroup
struct RetJob { const i32 *reward; const uint8_t *done; float *out; float gamma; int ring_steps, s_end, span; u32 n; };
static void ret_lane(void *arg)
{
    RetJob *j = (RetJob *)arg;
    azul_returns_ring_kernel(j->reward, j->done, j->out, j->gamma, j->ring_steps, j->s_end, j->span, j->n);
}
long long sl_returns_ring(const i32 *reward, const uint8_t *done, float *out, float gamma, int ring_steps, long long steps_played, int span, int n)
{
    RetJob j = {reward, done, out, gamma, ring_steps, (int)(steps_played % ring_steps == 0 ? ring_steps : steps_played % ring_steps), span, (u32)n};
    const unsigned blocks = ((unsigned)n + 63u) / 64u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 1;
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(ret_lane, &j, 1);
    }
    return ops;
}


// azul_a2c_apply_adam's launches: the step counter, then Adam on the flat k-major master copy + the eight nn.Linear tensors
struct AdamJob { const float *grad; float *flat, *m, *v; float lr, b1, b2, eps; ModuleParams mp; i32 *step; const float *n_total; float *stats; int phase; };
static void adam_lane(void *arg)
{
    AdamJob *j = (AdamJob *)arg;
    if (j->phase == 0) azul_a2c_step_kernel(j->step, j->n_total);
    else azul_a2c_apply_kernel(j->grad, j->flat, j->m, j->v, j->lr, j->b1, j->b2, j->eps, 1.f, 1.f, j->mp, j->step, j->n_total, 0.f, j->stats);
}
long long sl_adam(const float *grad, float *flat, float *m, float *v, float lr, float beta1, float beta2, float eps, float *c1w, float *c1b,
                  float *c2w, float *c2b, float *a1w, float *a1b, float *a2w, float *a2b, i32 *step, const float *n_total, float *stats5)
{
    AdamJob j = {grad, flat, m, v, lr, beta1, beta2, eps, {c1w, c1b, c2w, c2b, a1w, a1b, a2w, a2b}, step, n_total, stats5, 0};
    long long ops = 1;
    simt::g_grid_dim = {1, 1, 1};
    simt::g_block_idx = {0, 0, 0};
    ops += (long long)simt::run_workgroup(adam_lane, &j, 1);
    j.phase = 1;
    const unsigned blocks = ((unsigned)LG_P_PARAMS + 255u) / 256u;
    simt::g_grid_dim = {blocks, 1, 1};
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(adam_lane, &j, 4, 128u << 10);
    }
    return ops;
}


// ---- the remaining small kernels, each as its launch performs it ----------------------------------------------------------------
struct HeadJob { const float *logits; const uint8_t *mask; u64 seed, counter; u32 n; i32 *action; float *logp, *ent; u32 id_base; };
static void head_lane(void *arg)
{
    HeadJob *j = (HeadJob *)arg;
    azul_policy_head_kernel(j->logits, j->mask, j->seed, j->counter, nullptr, j->n, j->action, j->logp, j->ent, j->id_base);
}
long long sl_head(int n, const float *logits, const uint8_t *mask, unsigned long long seed, unsigned long long counter, unsigned id_base,
                  i32 *action, float *logp, float *entropy)
{
    HeadJob j = {logits, mask, seed, counter, (u32)n, action, logp, entropy, id_base};
    const unsigned blocks = ((unsigned)n + 3u) / 4u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 1;
    for (unsigned blk = 0; blk < blocks; blk++) { simt::g_block_idx = {blk, 0, 0}; ops += (long long)simt::run_workgroup(head_lane, &j, 1); }
    return ops;
}

struct SelcJob { const uint8_t *done; const i32 *action; int T; u32 N; i32 *index, *count; };
static void selc_lane(void *arg) { SelcJob *j = (SelcJob *)arg; azul_select_complete_kernel(j->done, j->action, j->T, j->N, j->index, j->count); }
long long sl_select_complete(const uint8_t *done, const i32 *action, int T, int n, i32 *index, i32 *count)
{
    SelcJob j = {done, action, T, (u32)n, index, count};
    simt::g_grid_dim = {1, 1, 1};
    simt::g_block_idx = {0, 0, 0};
    return 1 + (long long)simt::run_workgroup(selc_lane, &j, 16, 128u << 10);
}

struct Ret1Job { const i32 *reward; const uint8_t *done; float *out, *carry; float gamma; int n_steps; u32 n; };
static void ret1_lane(void *arg) { Ret1Job *j = (Ret1Job *)arg; azul_returns_kernel(j->reward, j->done, j->out, j->carry, j->gamma, j->n_steps, j->n); }
long long sl_returns(const i32 *reward, const uint8_t *done, float *out, float *carry, float gamma, int n_steps, int n)
{
    Ret1Job j = {reward, done, out, carry, gamma, n_steps, (u32)n};
    const unsigned blocks = ((unsigned)n + 255u) / 256u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 1;
    for (unsigned blk = 0; blk < blocks; blk++) { simt::g_block_idx = {blk, 0, 0}; ops += (long long)simt::run_workgroup(ret1_lane, &j, 4, 128u << 10); }
    return ops;
}

struct SeedJob { BatchDev b; u64 base; const u64 *seeds; };
static void seed_lane(void *arg) { SeedJob *j = (SeedJob *)arg; azul_seed_kernel(j->b, j->base, j->seeds); }
long long sl_seed(int n, u32 *mt, u32 *mtpos, unsigned long long seed_base, const u64 *seeds)
{
    SeedJob j;
    memset(&j, 0, sizeof(j));
    j.b.mt = mt; j.b.mtpos = mtpos; j.b.n = (u32)n; j.base = seed_base; j.seeds = seeds;
    const unsigned blocks = ((unsigned)n + 63u) / 64u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 1;
    for (unsigned blk = 0; blk < blocks; blk++) { simt::g_block_idx = {blk, 0, 0}; ops += (long long)simt::run_workgroup(seed_lane, &j, 1); }
    return ops;
}

struct RedJob { const float *partial; u32 parts; float *grad; };
static void red_lane(void *arg) { RedJob *j = (RedJob *)arg; azul_a2c_reduce_kernel(j->partial, j->parts, j->grad); }
long long sl_reduce(const float *partial, int parts, float *grad)
{
    RedJob j = {partial, (u32)parts, grad};
    const unsigned blocks = ((unsigned)LG_P_TOTAL + 255u) / 256u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 1;
    for (unsigned blk = 0; blk < blocks; blk++) { simt::g_block_idx = {blk, 0, 0}; ops += (long long)simt::run_workgroup(red_lane, &j, 4, 128u << 10); }
    return ops;
}

}
