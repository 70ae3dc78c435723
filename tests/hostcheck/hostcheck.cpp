// hostcheck.cpp -- TEST-ONLY: compiles the device core (csrc/azul_core.hpp) with g++ against the
// 64-lane host emulation of csrc/azul_wave.hpp (azul_wave_host.hpp, next to this file), so the wave-level game logic can be diffed against the
// oracle in the build container (no GPU).  Not part of the product; never shipped in libazulhip.so.
#include <stdlib.h>
#include <string.h>

#include "azul_hip.h"             // the ABI's constants (record / mask / observation sizes, flags)
#include "azul_wave_host.hpp"      // defines AZ_WAVE_HPP: the device header csrc/azul_wave.hpp is skipped
#include "azul_core.hpp"
#include "azul_tables.hpp"

using namespace az;
#include "azul_ops.hpp"        // the two-player rule kernel's body (op_body), unmodified: hc_op below

static double g_T[T_WORDS];
static double g_fr[T_ROWS * T_BINADES];
static SampleTab g_tab;
static bool g_T_ready = false;
static const SampleTab &table()
{
    if (!g_T_ready) { build_sample_tab(g_T); sample_tab_load(g_tab, g_T, g_fr); g_T_ready = true; }
    return g_tab;
}

struct HostStream {
    uint8_t rec[128];
    u32 mt[624];
    u32 pos;
    u32 lds[624];
    u64 episodes;
    u32 stuck;
    double stat_sum[10];
    Rules rules;
};

template <bool LID>
static int advance(HostStream *h, int n_steps, uint8_t *mask, int32_t *action, int32_t *reward, uint8_t *done, uint8_t *rec_after)
{
    LaneConst k; lane_consts(k);
    Game g; game_load(g, h->rec);
    game_prime<LID>(g, k);
    Rng r; rng_open(r, h->mt, h->lds, h->pos);
    Counters cnt = {&h->episodes, &h->stuck, h->stat_sum};
    int rc = 0;
    // both output forms are exercised: the scalar-pointer form (OUT == 2) into the caller's arrays and, when every
    // stream is requested, the per-lane-pointer form (OUT == 1) -- the caller's arrays are [t][1 game] here
    const bool full = mask && action && reward && done && !rec_after;
    static u64 bits_sink[3 * 4096];
    OutV ov;
    static u32 packed_sink[4096];
    outv_open(ov, 0, 1, mask ? mask : (uint8_t *)bits_sink, bits_sink, action, reward, done, packed_sink);
    OutS os = {mask, nullptr, action, reward, done, rec_after, nullptr};
    for (int t = 0; t < n_steps; t++) {
        u32 f = full ? selfplay_step<LID, 1>(g, h->rules.first_player, k, r, table(), cnt, ov, os)
                     : selfplay_step<LID, 2>(g, h->rules.first_player, k, r, table(), cnt, ov, os);
        if (f & 0x100u) { rc = (int)(f & 0xff); break; }
        if (full) outv_next(ov);
        if (os.mask) os.mask += 180;
        if (os.action) os.action += 1;
        if (os.reward) os.reward += 1;
        if (os.done) os.done += 1;
        if (os.rec) os.rec += 128;
    }
    game_store(g, h->rec);
    rng_close(r, &h->pos);
    return rc;
}


#define BY_POOL(pool, call_true, call_false) ((pool) == POOL_LID ? (call_true) : (call_false))

extern "C" {

void hc_weight_table(double *out /* [31][151] */) { build_weight_table(out); }
int hc_sample_tab_ok() { double t[T_WORDS]; return build_sample_tab(t) ? 1 : 0; }

HostStream *hc_stream_new(unsigned long long seed, int first_player, int tile_pool)
{
    HostStream *h = (HostStream *)calloc(1, sizeof(HostStream));
    h->rules.first_player = (u32)first_player;
    h->rules.tile_pool = (u32)tile_pool;
    seed_stream(h->mt, seed);
    h->pos = 624;
    // GameRunner.__init__ then reset() (without pre-moves), like oz_stream_start
    Game g; memset(&g, 0, sizeof(g));
    Rng r; rng_open(r, h->mt, h->lds, h->pos);
    for (int i = 0; i < 2; i++) BY_POOL(tile_pool, episode_reset<true>(g, h->rules.first_player, r), episode_reset<false>(g, h->rules.first_player, r));
    game_store(g, h->rec);
    rng_close(r, &h->pos);
    return h;
}

void hc_stream_free(HostStream *h) { free(h); }

int hc_stream_advance(HostStream *h, int n_steps, uint8_t *mask, int32_t *action, int32_t *reward, uint8_t *done, uint8_t *rec_after)
{
    return BY_POOL(h->rules.tile_pool, advance<true>(h, n_steps, mask, action, reward, done, rec_after),
                   advance<false>(h, n_steps, mask, action, reward, done, rec_after));
}

void hc_stream_get(HostStream *h, uint8_t *rec, u32 *mt, u32 *pos, u64 *episodes, u32 *stuck, double *stat_sum)
{
    if (rec) memcpy(rec, h->rec, 128);
    if (mt) memcpy(mt, h->mt, sizeof(h->mt));
    if (pos) *pos = h->pos;
    if (episodes) *episodes = h->episodes;
    if (stuck) *stuck = h->stuck;
    if (stat_sum) memcpy(stat_sum, h->stat_sum, sizeof(h->stat_sum));
}

void hc_seed(unsigned long long seed, u32 *mt) { seed_stream(mt, seed); }

// single-record operations (record in/out, explicit MT state)
void hc_mask(const uint8_t *rec, uint8_t *out180)
{
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    Mask m; legal_mask(g, k, m);
    mask_write(m, out180);
}

void hc_observe(const uint8_t *rec, int persp, float *out136)
{
    Game g; game_load(g, rec);
    u32 p = persp == 2 ? me_index(g) : (u32)persp;
    observe(g, p, out136);
}

int hc_potential(const uint8_t *rec, int tile_pool)
{
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    return BY_POOL(tile_pool, potential<true>(g, k), potential<false>(g, k));
}

int hc_flags(const uint8_t *rec)
{
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    game_prime<false>(g, k);
    return (sources_board(g) == 0u ? 1 : 0) | (is_end_of_game(g) ? 2 : 0) | (g.eog ? 4 : 0);
}

void hc_count_score(uint8_t *rec, int tile_pool)
{
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    if (tile_pool == POOL_LID) count_score<true>(g, k); else count_score<false>(g, k);
    game_store(g, rec);
}

void hc_move(uint8_t *rec, int action, int tile_pool)
{
    Game g; game_load(g, rec);
    if (tile_pool == POOL_LID) do_move<true>(g, action_code((u32)action)); else do_move<false>(g, action_code((u32)action));
    game_store(g, rec);
}

int hc_step(uint8_t *rec, int action, int first_player, int tile_pool, u32 *mt, u32 *pos)
{
    static u32 lds[624];
    (void)first_player;
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    if (tile_pool == POOL_LID) game_prime<true>(g, k); else game_prime<false>(g, k);
    Rng r; rng_attach(r, mt, lds, *pos);
    u32 st = BY_POOL(tile_pool, checked_step<true>(g, k, r, action), checked_step<false>(g, k, r, action));
    if (st != ST_ILLEGAL_MOVE && st != ST_GAME_ENDED && st != ST_BAD_ACTION) game_store(g, rec);
    rng_close(r, pos);
    return (int)st;
}

int hc_runner_step(uint8_t *rec, int action, int first_player, int tile_pool, u32 *mt, u32 *pos, int *reward, int *done)
{
    static u32 lds[624];
    (void)first_player;
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    if (tile_pool == POOL_LID) game_prime<true>(g, k); else game_prime<false>(g, k);
    Rng r; rng_attach(r, mt, lds, *pos);
    i32 rew = 0; u32 dn = 0;
    u32 st = BY_POOL(tile_pool, runner_step<true>(g, k, r, table(), action, rew, dn), runner_step<false>(g, k, r, table(), action, rew, dn));
    if (st != ST_ILLEGAL_MOVE && st != ST_GAME_ENDED && st != ST_BAD_ACTION) game_store(g, rec);
    rng_close(r, pos);
    *reward = rew; *done = (int)dn;
    return (int)st;
}

int hc_runner_reset(uint8_t *rec, int first_player, int tile_pool, u32 *mt, u32 *pos, int ctor_only)
{
    static u32 lds[624];
    LaneConst k; lane_consts(k);
    Game g; memset(&g, 0, sizeof(g));
    Rng r; rng_attach(r, mt, lds, *pos);
    u32 st = BY_POOL(tile_pool, episode_reset<true>(g, (u32)first_player, r), episode_reset<false>(g, (u32)first_player, r));
    if (!st && !ctor_only) st = BY_POOL(tile_pool, runner_opponent_loop<true>(g, k, r, table(), true), runner_opponent_loop<false>(g, k, r, table(), true));
    game_store(g, rec);
    rng_close(r, pos);
    return (int)st;
}

int hc_random_action(const uint8_t *rec, u32 *mt, u32 *pos)
{
    static u32 lds[624];
    LaneConst k; lane_consts(k);
    Game g; game_load(g, rec);
    Rng r; rng_attach(r, mt, lds, *pos);
    Mask m; legal_mask(g, k, m);
    u32 code;
    i32 a = random_agent(m, r, table(), k, code);
    rng_close(r, pos);
    return a;
}

int hc_init(uint8_t *rec, int first_player, int tile_pool, u32 *mt, u32 *pos)
{
    static u32 lds[624];
    Game g; game_load(g, rec);
    Rng r; rng_attach(r, mt, lds, *pos);
    if (tile_pool == POOL_LID) game_ctor<true>(g, (u32)first_player, r); else game_ctor<false>(g, (u32)first_player, r);
    game_store(g, rec);
    rng_close(r, pos);
    return 0;
}

int hc_new_round(uint8_t *rec, int tile_pool, u32 *mt, u32 *pos)
{
    static u32 lds[624];
    Game g; game_load(g, rec);
    Rng r; rng_attach(r, mt, lds, *pos);
    u32 st = BY_POOL(tile_pool, new_round<true>(g, r), new_round<false>(g, r));
    game_store(g, rec);
    rng_close(r, pos);
    return (int)st;
}

void hc_next_player(uint8_t *rec)
{
    Game g; game_load(g, rec);
    g.cur = (g.cur < 2u) ? g.cur + 1u : 1u;
    game_store(g, rec);
}

void hc_statistics(const uint8_t *rec, double *out10)
{
    Game g; game_load(g, rec);
    for (u32 q = 0; q < 10u; q++) out10[q] = game_stat(g, q);
}

int hc_sample_mask(const uint8_t *mask180, u32 *mt, u32 *pos)
{
    static u32 lds[624];
    vu32 l = lane();
    Mask m;
    m.b0 = sel(ld_u8(mask180, l, l < 64u) != 0u, splat(1u), splat(0u));
    m.b1 = sel(ld_u8(mask180, l + 64u, l < 64u) != 0u, splat(1u), splat(0u));
    m.b2 = sel(ld_u8(mask180, l + 128u, l < 52u) != 0u, splat(1u), splat(0u));
    m.m0 = ballot(m.b0 != 0u); m.m1 = ballot(m.b1 != 0u); m.m2 = ballot(m.b2 != 0u);
    Rng r; rng_attach(r, mt, lds, *pos);
    LaneConst k; lane_consts(k);
    u32 code;
    i32 a = random_agent(m, r, table(), k, code);
    rng_close(r, pos);
    return a;
}

// ONE rule call exactly as azul_op_kernel performs it (csrc/azul_ops.hpp: op_body) on one game in host memory: record [128], stream (624
// words + index), results on request (NULL: not asked for).  op: the OP_* of azul_ops.hpp.  Returns the status byte.
int hc_op(uint8_t *rec, int first_player, int tile_pool, unsigned long long margin, int op, int action, u32 *mt, u32 *pos, unsigned pos_set,
          const uint8_t *mask_in, uint8_t *mask_out, float *obs, int persp, int *flags, int *potential, double *stats10, int *reward,
          int *done, int *action_out, int *player, int *rng_dirty, int *next_action, uint8_t *rec_out, unsigned long long *episodes,
          unsigned *stuck, double *stat_sum10)
{
    table();
    static u32 mt_lds[624];
    static double fr_lds[T_ROWS * T_BINADES];
    u64 ep = episodes ? *episodes : 0; u32 sk = stuck ? *stuck : 0; double ss[10] = {0};
    if (stat_sum10) memcpy(ss, stat_sum10, sizeof(ss));
    BatchDev b;
    memset(&b, 0, sizeof(b));
    b.state = rec; b.mt = mt; b.mtpos = pos; b.T = g_T; b.episodes = &ep; b.stuck = &sk; b.stat_sum = ss; b.n = 1;
    b.rules.first_player = (u32)first_player; b.rules.tile_pool = (u32)tile_pool; b.draw_margin = margin ? margin : AZ_DRAW_MARGIN;
    OpArgs a;
    memset(&a, 0, sizeof(a));
    i32 act_in = action, act_out = 0, rew = 0, pot = 0, nxt = -2;
    uint8_t status = 0, dn = 0, fl = 0, pl = 0, rd = 0;
    u32 pos_after = 0;
    a.op = op; a.actions = &act_in; a.mask_in = mask_in; a.actions_out = &act_out; a.status = &status;
    a.reward = reward ? &rew : nullptr; a.done = done ? &dn : nullptr; a.mask = mask_out; a.obs = obs; a.persp = persp;
    a.flags = flags ? &fl : nullptr; a.potential = potential ? &pot : nullptr; a.stats = stats10; a.player = &pl; a.rng_dirty = &rd;
    a.rec_out = rec_out; a.pos_out = &pos_after; a.next_action = next_action ? &nxt : nullptr; a.pos_set = pos_set; a.first = 0;
    if (tile_pool == POOL_LID) op_body<true>(b, a, 0u, mt_lds, fr_lds);
    else op_body<false>(b, a, 0u, mt_lds, fr_lds);
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

} // extern "C"
