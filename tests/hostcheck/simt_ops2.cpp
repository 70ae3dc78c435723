// simt_ops2.cpp -- TEST-ONLY: the two-player rule kernel, azul_op_kernel (csrc/azul_selfplay_kernels.hpp on azul_ops2.hpp, azul_env2.hpp and
// azul_selfplay2.hpp, all UNMODIFIED), compiled by g++ and run lane by lane in lockstep (simt/simt.hpp) on host memory: ONE rule call on one
// 128-byte record + MT19937 state, exactly what azul_game_call launches for a single-game facade call.  This is the emulated device behind
// the facade in the CPU suite (tests/hostcheck/hostcheck.py: EmuBackend) and under ASan / UBSan -- the product's own dispatch, not a restatement.
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
using namespace az;
#include "azul_selfplay_kernels.hpp"

static double g_T[T_PAIRS * 2];
static bool g_T_ok = false;
static void table() { if (!g_T_ok) { g_T_ok = build_sample_pairs(T_ROWS, g_T); } }

struct OpJob { BatchDev b; OpArgs a; int lid; };
static void op_main(void *arg)
{
    OpJob *j = (OpJob *)arg;
    if (j->lid) azul_op_kernel<true>(j->b, j->a); else azul_op_kernel<false>(j->b, j->a);
}

static u32 g_move_limit = 0;       // BatchDev::move_limit of the next launches (azul_batch_set_move_limit)

extern "C" {
void sh2_set_move_limit(unsigned m) { g_move_limit = m; }


// one rule call (OP_* of azul_ops2.hpp) on one record; every optional output is written only when its pointer is given.  Returns the status byte.
int sh2_op(uint8_t *rec, int first_player, int tile_pool, unsigned long long margin, int op, int action, u32 *mt, u32 *pos, unsigned pos_set,
           const uint8_t *mask_in, uint8_t *mask_out, float *obs, int persp, int *flags, int *potential, double *stats10, int *reward,
           int *done, int *action_out, int *player, int *rng_dirty, int *next_action, uint8_t *rec_out, unsigned long long *episodes,
           unsigned *stuck, double *stat_sum10)
{
    table();
    u64 ep = episodes ? *episodes : 0; u32 sk = stuck ? *stuck : 0; double ss[10] = {0};
    if (stat_sum10) memcpy(ss, stat_sum10, sizeof(ss));
    OpJob j;
    memset(&j, 0, sizeof(j));
    BatchDev &b = j.b;
    b.state = rec; b.mt = mt; b.mtpos = pos; b.tab = (const double2 *)g_T; b.episodes = &ep; b.stuck = &sk; b.stat_sum = ss; b.n = 1;
    b.rules.first_player = (u32)first_player; b.rules.tile_pool = (u32)tile_pool; b.draw_margin = margin ? margin : AZ_DRAW_MARGIN; b.move_limit = g_move_limit;
    OpArgs &a = j.a;
    i32 act_in = action, act_out = 0, rew = 0, pot = 0, nxt = -2;
    uint8_t status = 0, dn = 0, fl = 0, pl = 0, rd = 0;
    u32 pos_after = 0;
    a.op = op; a.actions = &act_in; a.mask_in = mask_in; a.actions_out = &act_out; a.status = &status;
    a.reward = reward ? &rew : nullptr; a.done = done ? &dn : nullptr; a.mask = mask_out; a.obs = obs; a.persp = persp;
    a.flags = flags ? &fl : nullptr; a.potential = potential ? &pot : nullptr; a.stats = stats10; a.player = &pl; a.rng_dirty = &rd;
    a.rec_out = rec_out; a.pos_out = &pos_after; a.next_action = next_action ? &nxt : nullptr; a.pos_set = pos_set; a.first = 0; a.count = 1;
    j.lid = tile_pool == POOL_LID;
    simt::g_grid_dim = {1, 1, 1};
    simt::g_block_idx = {0, 0, 0};
    simt::run_workgroup(op_main, &j, 1, simt::STACK_BYTES);
    if (flags) *flags = fl;
    if (potential) *potential = pot;
    if (reward) *reward = rew;
    if (done) *done = dn;
    if (action_out) *action_out = act_out;
    if (player) *player = pl;
    if (rng_dirty) *rng_dirty = rd;
    if (next_action) *next_action = nxt;
    if (episodes) *episodes = ep;
    if (stuck) *stuck = sk;
    if (stat_sum10) memcpy(stat_sum10, ss, sizeof(ss));
    return status;
}

// the same kernel over a BATCH of records (two games per wave; an odd count leaves the last wave's upper half without a game):
// rows of actions / status / mask_out / reward / done belong to games 0 .. n - 1
int sh2_op_batch(int n, uint8_t *recs, int first_player, int tile_pool, int op, const i32 *actions, const uint8_t *active, u32 *mt, u32 *pos,
                 uint8_t *status, uint8_t *mask_out, i32 *reward, uint8_t *done, unsigned long long *episodes, unsigned *stuck, double *stat_sum)
{
    table();
    if (n <= 0) return -1;
    OpJob j;
    memset(&j, 0, sizeof(j));
    j.b.state = recs; j.b.mt = mt; j.b.mtpos = pos; j.b.tab = (const double2 *)g_T; j.b.episodes = (u64 *)episodes; j.b.stuck = stuck; j.b.stat_sum = stat_sum; j.b.n = (u32)n;
    j.b.rules.first_player = (u32)first_player; j.b.rules.tile_pool = (u32)tile_pool; j.b.draw_margin = AZ_DRAW_MARGIN; j.b.move_limit = g_move_limit;
    j.a.op = op; j.a.actions = actions; j.a.active = active; j.a.status = status; j.a.mask = mask_out; j.a.reward = reward; j.a.done = done;
    j.a.first = 0; j.a.count = (u32)n;
    j.lid = tile_pool == POOL_LID;
    const unsigned blocks = ((unsigned)n + 1u) / 2u;
    simt::g_grid_dim = {blocks, 1, 1};
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        simt::run_workgroup(op_main, &j, 1, simt::STACK_BYTES);
    }
    return 0;
}

// random.seed(int) as azul_seed_kernel's threads run it
void sh2_seed(unsigned long long seed, u32 *mt) { seed_stream(mt, (u64)seed); }

// the RandomAgent weight table (azul_tables.hpp: CPython's accumulate over 0.01 / 1.0 weights) and the check of its compact form
void sh2_weight_table(double *T /* [31][151] */) { build_weight_table(T); }
int sh2_sample_tab_ok() { double t[T_PAIRS * 2]; return build_sample_pairs(T_ROWS, t) ? 1 : 0; }

}
