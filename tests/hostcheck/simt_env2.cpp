// simt_env2.cpp -- TEST-ONLY: the ENV SIDE of the persistent policy rollout kernel (csrc/azul_env2.hpp on top of azul_selfplay2.hpp and
// azul_common.hpp, all UNMODIFIED) compiled by g++ and run lane by lane in lockstep (simt/simt.hpp): GameRunner.step,
// GameRunner.reset, GameRunner.get_state and the RandomAgent opponent as azul_policy_rollout2_kernel runs them, two games per wave,
// diffed against the oracle -- and run under UBSan / ASan -- in the build container.
// The function below restates the env half of the kernel's move loop (csrc/azul_rollout2.hpp: load, prime, stream open, `publish`,
// then per move: the env step with the action the head chose, reward / done, `publish`; store, stream close).  The network half is
// replaced by the host: the actions come in as an array, which is what the head hands the env through LDS (actS).
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
#include "azul_selfplay2.hpp"
#include "azul_env2.hpp"

using namespace az;

enum { OBS = 136, OBS_STRIDE = 140 };

struct EnvJob {
    uint8_t *state; u32 *mt; u32 *mtpos; const double2 *T;
    u64 *episodes; u32 *stuck; double *stat_sum;
    u32 n, first_player, move_limit;
    u64 margin;
    int opponent, n_steps;
    const i32 *actions;      // [T][N]
    float *obs;              // [T + 1][N][136]
    uint8_t *mask;           // [T + 1][N][180]
    uint8_t *player;         // [T + 1][N]
    u64 *maskbits;           // [T + 1][N][3]: the packed words the head reads from LDS
    i32 *reward; uint8_t *done;   // [T][N]
    uint8_t *status;         // [N]
    u32 wave_id;
    // "LDS" of the wave
    u32 mt_lds[2][az2::MT_LDS_WORDS];
    double2 tabfs_lds[T_PAIRS];
    float obs_lds[2][OBS_STRIDE];
    u64 mask_lds[2][3];
};

template <bool LID, bool OPP>
static void env_wave(EnvJob *j)
{
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    for (u32 i = lane; i < (u32)T_PAIRS; i += 64u) j->tabfs_lds[i] = j->T[i];
    az2::lds_sync();
    const u32 n = j->n, gi = j->wave_id * 2u + half;
    const bool live = gi < n;
    const u32 gic = live ? gi : n - 1u;                     // a dead half loads a valid game and writes nothing (the kernel's clamp)
    az2::K2 k;
    az2::k2_init(k);
    az2::rng2_set_move_limit(j->mt_lds[half], j->move_limit, l);      // (azul_batch_set_move_limit: the kernels copy BatchDev::move_limit into the game's LDS region)
    az2::Tab2 tab = {j->tabfs_lds};
    az2::G2 g;
    uint8_t *rec = j->state + (size_t)gic * AZUL_RECORD_BYTES;
    az2::g2_load(g, rec, l);
    az2::prime2(g, k);
    az2::Rng2 r;
    u32 *gmt = j->mt + (size_t)gic * 624u;
    az2::rng2_open(r, gmt, j->mt_lds[half], j->mtpos[gic], l);
    az2::Counters2 cnt;
    az2::counters2_open(cnt, j->episodes + gic, j->stuck + gic, j->stat_sum + (size_t)gic * 10, l);
    u32 st_last = ST_OK;
    float *orow = j->obs_lds[half];
    az2::Mask2 m;
    auto publish = [&](u32 slot) {
        az2::legal_mask2(g, k, m);
        const size_t cell = (size_t)slot * n + gi;
        uint8_t *row = j->mask + cell * AZUL_NUM_ACTIONS + l;
        if (l < 30u) {
            for (u32 ww = 0; ww < 6u; ww++) row[30u * ww] = (uint8_t)m.bit[ww];
        }
        if (l == 0u) {
            j->mask_lds[half][0] = (u64)m.m[0] | ((u64)m.m[1] << 30) | ((u64)m.m[2] << 60);
            j->mask_lds[half][1] = ((u64)m.m[2] >> 4) | ((u64)m.m[3] << 26) | ((u64)m.m[4] << 56);
            j->mask_lds[half][2] = ((u64)m.m[4] >> 8) | ((u64)m.m[5] << 22);
            for (int q = 0; q < 3; q++) j->maskbits[cell * 3 + q] = j->mask_lds[half][q];
            j->player[cell] = (uint8_t)g.cur;
        }
        az2::observe2(g, OPP ? 0u : az2::me2(g), orow, j->obs + cell * OBS, l);
    };
    if (live) publish(0u);
    for (int t = 0; t < j->n_steps; t++) {
        if (live) {
            const size_t row_t = (size_t)t * n;
            const i32 av = j->actions[row_t + gi];
            i32 rew = 0;
            u32 dn = 0;
            st_last = OPP ? az2::agent_step2<LID>(g, av, m, j->first_player, r, tab, j->margin, cnt, k, rew, dn)
                          : az2::policy_step2<LID>(g, av, m, j->first_player, r, j->margin, cnt, k, rew, dn);
            if (l == 0u) { j->reward[row_t + gi] = rew; j->done[row_t + gi] = (uint8_t)dn; }
            publish((u32)t + 1u);
        }
    }
    if (live) {
        az2::g2_store(g, rec, l);
        az2::rng2_close(r, gmt, j->mtpos + gi, l);
        az2::counters2_close(cnt, l);
        if (l == 0u) j->status[gi] = (uint8_t)st_last;
    }
}

template <bool LID>
static void env_lane_main(void *arg)
{
    EnvJob *j = (EnvJob *)arg;
    if (j->opponent) env_wave<LID, true>(j); else env_wave<LID, false>(j);
}

static u32 g_move_limit = 0;

extern "C" {
void sh2_set_move_limit(unsigned m) { g_move_limit = m; }


// n_games games advance by n_steps AGENT moves with the given actions (opponent = 0: the policy plays both sides, env_policy_step;
// opponent = 1: GameRunner.step with the RandomAgent opponent + reset at episode end, env_agent_step), two games per wave.
// Streams are [n_steps (+ 1)][N]... like the kernel's trajectory ring.  Returns the number of cross-lane operations executed.
long long sh2_rollout_env(int n_games, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum, int first_player,
                          int tile_pool, unsigned long long margin, int opponent, int n_steps, const i32 *actions, float *obs,
                          uint8_t *mask, uint8_t *player, u64 *maskbits, i32 *reward, uint8_t *done, uint8_t *status)
{
    if (n_games <= 0 || n_steps < 0) return -1;
    static double T[T_PAIRS * 2];
    if (!build_sample_pairs(T_ROWS, T)) return -2;
    long long ops = 0;
    for (u32 w = 0; w < ((u32)n_games + 1u) / 2u; w++) {
        EnvJob *j = (EnvJob *)calloc(1, sizeof(EnvJob));
        j->state = state; j->mt = mt; j->mtpos = mtpos; j->T = (const double2 *)T; j->episodes = episodes; j->stuck = stuck; j->stat_sum = stat_sum;
        j->n = (u32)n_games; j->first_player = (u32)first_player; j->margin = margin ? margin : AZ_DRAW_MARGIN;
        j->move_limit = g_move_limit;
        j->opponent = opponent; j->n_steps = n_steps; j->actions = actions; j->obs = obs; j->mask = mask; j->player = player;
        j->maskbits = maskbits; j->reward = reward; j->done = done; j->status = status; j->wave_id = w;
        ops += (long long)simt::run_wave(tile_pool == POOL_LID ? env_lane_main<true> : env_lane_main<false>, j);
        free(j);
    }
    return ops;
}

}
