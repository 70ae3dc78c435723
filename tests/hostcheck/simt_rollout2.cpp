// simt_rollout2.cpp -- TEST-ONLY: the persistent policy rollout kernel itself (csrc/azul_rollout2.hpp: azul_policy_rollout2_kernel, env
// phases on csrc/azul_env2.hpp, matrix phases on v_mfma_f32_16x16x4_f32 with weights streamed through buffer loads, the sampling head of
// csrc/azul_policy.hpp), UNMODIFIED, as a workgroup of eight emulated wavefronts (simt/simt.hpp: run_workgroup) -- a CPU check of the
// kernel's logic and, under ASan / UBSan, of every LDS and global index it forms.
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
using namespace az;
#include "azul_ops2.hpp"
#include "azul_policy.hpp"
#include "azul_rollout2.hpp"

struct Job { BatchDev b; PolicyWeights W; RolloutArgs a; int lid, opp; };

static void lane_main(void *arg)
{
    Job *j = (Job *)arg;
    if (j->lid) {
        if (j->opp == 2) azul_policy_rollout2_kernel<true, 2>(j->b, j->W, j->a);
        else if (j->opp) azul_policy_rollout2_kernel<true, 1>(j->b, j->W, j->a);
        else azul_policy_rollout2_kernel<true, 0>(j->b, j->W, j->a);
    } else {
        if (j->opp == 2) azul_policy_rollout2_kernel<false, 2>(j->b, j->W, j->a);
        else if (j->opp) azul_policy_rollout2_kernel<false, 1>(j->b, j->W, j->a);
        else azul_policy_rollout2_kernel<false, 0>(j->b, j->W, j->a);
    }
}

static u32 g_move_limit = 0;       // BatchDev::move_limit of the next launches (azul_batch_set_move_limit)

extern "C" {
void sr2_set_move_limit(unsigned m) { g_move_limit = m; }
static long long run_blocks(Job &j, int n_games);

unsigned long long sr2_buffer_oob() { return simt::g_buffer_oob; }

static long long run_blocks(Job &j, int n_games)
{
    const unsigned blocks = ((unsigned)n_games + PF_GAMES - 1u) / PF_GAMES;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 0;
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(lane_main, &j, (int)PR2_WAVES);
    }
    return ops;
}

// the NETWORK-opponent variant (azul_policy_rollout2_kernel<LID, 2>; game_runner.py:27-30): `wo` = the opponent's six weight arrays
long long sr2_rollout_vs(int n_games, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum, int first_player,
                         int tile_pool, unsigned id_base, const float *const *wa, const float *const *wo, int n_steps, float *obs, uint8_t *mask,
                         uint8_t *player, i32 *action, i32 *reward, uint8_t *done, float *value, float *logp, float *entropy, uint8_t *status,
                         float *returns, float gamma, unsigned long long seed, unsigned long long opp_seed, unsigned long long counter,
                         i32 *opp_action, float *opp_logp, uint8_t *opp_replies, int opp_slots)
{
    static double T[T_PAIRS * 2];
    if (!build_sample_pairs(T_ROWS, T)) return -2;
    Job j;
    memset(&j, 0, sizeof(j));
    j.b.state = state; j.b.mt = mt; j.b.mtpos = mtpos; j.b.tab = (const double2 *)T; j.b.episodes = episodes; j.b.stuck = stuck; j.b.stat_sum = stat_sum;
    j.b.n = (u32)n_games; j.b.rules.first_player = (u32)first_player; j.b.rules.tile_pool = (u32)tile_pool; j.b.draw_margin = AZ_DRAW_MARGIN; j.b.move_limit = g_move_limit;
    j.b.id_base = id_base;
    j.W = {wa[0], wa[1], wa[2], wa[3], wa[4], wa[5]};
    j.a.n_steps = n_steps; j.a.obs = obs; j.a.mask = mask; j.a.player = player; j.a.action = action; j.a.reward = reward; j.a.done = done;
    j.a.value = value; j.a.logp = logp; j.a.entropy = entropy; j.a.status = status; j.a.returns = returns; j.a.gamma = gamma;
    j.a.seed = seed; j.a.counter = counter; j.a.counter_dev = nullptr;
    j.a.Wopp = {wo[0], wo[1], wo[2], wo[3], wo[4], wo[5]};
    j.a.opp_seed = opp_seed; j.a.opp_action = opp_action; j.a.opp_logp = opp_logp; j.a.opp_replies = opp_replies; j.a.opp_slots = opp_slots;
    j.lid = tile_pool == POOL_LID; j.opp = 2;
    return run_blocks(j, n_games);
}

// one launch of the kernel over n_games (a multiple of 16 or not: the last workgroup is ragged) for n_steps moves
long long sr2_rollout(int n_games, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum, int first_player,
                      int tile_pool, int opponent_random, unsigned id_base, const float *w1t, const float *b1, const float *w2c,
                      const float *b2c, const float *w2a_t, const float *b2a, int n_steps, float *obs, uint8_t *mask, uint8_t *player,
                      i32 *action, i32 *reward, uint8_t *done, float *value, float *logp, float *entropy, uint8_t *status, float *returns,
                      float gamma, unsigned long long seed, unsigned long long counter)
{
    static double T[T_PAIRS * 2];
    if (!build_sample_pairs(T_ROWS, T)) return -2;
    Job j;
    memset(&j, 0, sizeof(j));
    j.b.state = state; j.b.mt = mt; j.b.mtpos = mtpos; j.b.tab = (const double2 *)T; j.b.episodes = episodes; j.b.stuck = stuck; j.b.stat_sum = stat_sum;
    j.b.n = (u32)n_games; j.b.rules.first_player = (u32)first_player; j.b.rules.tile_pool = (u32)tile_pool; j.b.draw_margin = AZ_DRAW_MARGIN; j.b.move_limit = g_move_limit;
    j.b.id_base = id_base;
    j.W = {w1t, b1, w2c, b2c, w2a_t, b2a};
    j.a.n_steps = n_steps; j.a.obs = obs; j.a.mask = mask; j.a.player = player; j.a.action = action; j.a.reward = reward; j.a.done = done;
    j.a.value = value; j.a.logp = logp; j.a.entropy = entropy; j.a.status = status; j.a.returns = returns; j.a.gamma = gamma;
    j.a.seed = seed; j.a.counter = counter; j.a.counter_dev = nullptr;
    j.lid = tile_pool == POOL_LID; j.opp = opponent_random;
    const unsigned blocks = ((unsigned)n_games + PF_GAMES - 1u) / PF_GAMES;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 0;
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(lane_main, &j, (int)PR2_WAVES);
    }
    return ops;
}

}
