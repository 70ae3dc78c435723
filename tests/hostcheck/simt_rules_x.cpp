// simt_rules_x.cpp -- TEST-ONLY: the P-player / D-display rules (csrc/azul_rules_x.hpp on top of azul_common.hpp and
// azul_selfplay2.hpp, all UNMODIFIED) compiled by g++ and run lane by lane in lockstep (simt/simt.hpp): the kernel BODIES the product's
// __global__ wrappers call (azx::op_body_x, azx::selfplay_body_x) run here on host memory, so their logic can be diffed against the
// oracle -- and run under UBSan / ASan -- in the build container, before a GPU sees them.
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
#include "azul_selfplay2.hpp"
#include "azul_rules_x.hpp"

using namespace az;

struct XJob {
    azx::XBatchDev b;
    azx::XOp op;
    azx::XTraj t;
    int players, displays, variant;
    u32 wave;
    u32 mt_lds[2][624];
    u32 mtt_lds[2][624];
    double2 tab_lds[51 * T_STRIDE];
};

template <u32 P, u32 D>
static void lane_op(void *arg)
{
    XJob *j = (XJob *)arg;
    azx::op_body_x<P, D>(j->b, j->op, j->wave, j->mt_lds, j->tab_lds);
}

template <u32 P, u32 D>
static void lane_play(void *arg)
{
    XJob *j = (XJob *)arg;
    switch (j->variant) {
    case 0: azx::selfplay_body_x<P, D, 1, true, true>(j->b, j->t, j->wave, j->mt_lds, j->mtt_lds, j->tab_lds); break;
    case 1: azx::selfplay_body_x<P, D, 1, true, false>(j->b, j->t, j->wave, j->mt_lds, j->mtt_lds, j->tab_lds); break;
    case 3: azx::selfplay_body_x<P, D, 2, false, false>(j->b, j->t, j->wave, j->mt_lds, j->mtt_lds, j->tab_lds); break;
    default: azx::selfplay_body_x<P, D, 0, false, false>(j->b, j->t, j->wave, j->mt_lds, j->mtt_lds, j->tab_lds); break;
    }
}

typedef void (*lane_fn)(void *);
static lane_fn pick_fn(int players, int displays, bool play)
{
#define AZ_CASE(PP, DD) if (players == PP && displays == DD) return play ? lane_play<PP, DD> : lane_op<PP, DD>
    AZ_CASE(2, 5); AZ_CASE(3, 5); AZ_CASE(3, 7); AZ_CASE(4, 5); AZ_CASE(4, 9);
#undef AZ_CASE
    return nullptr;
}

static double *table_for(int displays)
{
    static double tabs[3][51 * T_STRIDE * 2];
    static bool built[3] = {false, false, false};
    const int i = displays == 5 ? 0 : displays == 7 ? 1 : 2;
    if (!built[i]) { if (!build_sample_pairs(5 * (displays + 1) + 1, tabs[i])) return nullptr; built[i] = true; }
    return tabs[i];
}

extern "C" {

// n_games games (wide records [N][256], MT19937 states [N][624] + positions [N], counters) advance by n_steps moves, two per wave.
// variant: 0 = OUT 1 / PAD / BITS, 1 = OUT 1 / PAD, 3 = OUT 2 (run-time subset), 4 = OUT 0.  Returns the number of cross-lane
// operations executed, or a negative number on bad arguments.
long long shx_selfplay(int n_games, int players, int displays, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum,
                       int first_player, int pool, int end_bonus, int short_deal, unsigned long long margin, int n_steps, int variant,
                       uint8_t *mask, int pitch, u64 *maskbits, i32 *action, i32 *reward, uint8_t *done, u32 *packed, uint8_t *rec)
{
    lane_fn fn = pick_fn(players, displays, true);
    double *tab = table_for(displays);
    if (!fn || !tab || n_games <= 0 || n_steps < 0) return -1;
    long long ops = 0;
    for (u32 w = 0; w < ((u32)n_games + 1u) / 2u; w++) {
        XJob *j = (XJob *)calloc(1, sizeof(XJob));
        j->b = {state, mt, mtpos, episodes, stuck, stat_sum, (u32)n_games, margin ? margin : AZ_DRAW_MARGIN,
                {(u32)first_player, (u32)pool, (u32)end_bonus, (u32)short_deal}, (const double2 *)tab, nullptr};
        j->t = {n_steps, mask, maskbits, action, reward, done, rec, packed, (u32)pitch};
        j->players = players; j->displays = displays; j->variant = variant; j->wave = w;
        ops += (long long)simt::run_wave(fn, j);
        free(j);
    }
    return ops;
}

// one rule call on ONE game (the wave's other half stays idle): record [256], stream (624 words + index), results on request
int shx_op(uint8_t *rec, int players, int displays, int first_player, int pool, int end_bonus, int short_deal, unsigned long long margin, int op,
           int action, u32 *mt, u32 *pos, const uint8_t *mask_in, uint8_t *mask_out, float *obs, int persp, int *flags, double *stats10,
           int *action_out, int *player, int *rng_dirty, int *next_action /* NULL: not asked for */, unsigned pos_set /* 0, or 1 + index */)
{
    lane_fn fn = pick_fn(players, displays, false);
    double *tab = table_for(displays);
    if (!fn || !tab) return -1;
    XJob *j = (XJob *)calloc(1, sizeof(XJob));
    u64 episodes = 0; u32 stuck = 0; double stat_sum[10] = {0};
    j->b = {rec, mt, pos, &episodes, &stuck, stat_sum, 1u, margin ? margin : AZ_DRAW_MARGIN,
            {(u32)first_player, (u32)pool, (u32)end_bonus, (u32)short_deal}, (const double2 *)tab, nullptr};
    i32 act_in = action, act_out = 0;
    uint8_t status = 0, fl = 0, pl = 0, rd = 0;
    memset(&j->op, 0, sizeof(j->op));
    j->op.op = op; j->op.actions = &act_in; j->op.mask_in = mask_in; j->op.actions_out = &act_out; j->op.status = &status;
    j->op.mask = mask_out; j->op.obs = obs; j->op.persp = persp; j->op.flags = &fl; j->op.stats = stats10; j->op.player = &pl;
    j->op.rng_dirty = &rd; j->op.first = 0; j->op.count = 1;
    i32 nxt = -2;
    j->op.next_action = next_action ? &nxt : nullptr; j->op.pos_set = pos_set;
    j->players = players; j->displays = displays; j->wave = 0;
    simt::run_wave(fn, j);
    free(j);
    if (flags) *flags = fl;
    if (action_out) *action_out = act_out;
    if (player) *player = pl;
    if (rng_dirty) *rng_dirty = rd;
    if (next_action) *next_action = nxt;
    return status;
}

}
