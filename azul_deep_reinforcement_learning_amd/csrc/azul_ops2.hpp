// azul_ops2.hpp -- the body of the two-player rule kernel (azul_op_kernel: one rule call per game and launch) ON THE TWO-PLAYER RULE BOOK
// of azul_selfplay2.hpp / azul_env2.hpp: TWO GAMES PER WAVEFRONT, the same functions the benchmarked self-play loop and the policy rollout
// run -- a rule is written once.  As a header: azul_kernels.hip wraps op_body2 in the __global__ function, and tests/hostcheck/simt_ops2.cpp
// runs THIS FILE, unmodified, under the lockstep 64-lane emulation -- the facade's emulated device in the CPU suite and under ASan / UBSan
// is the product's own dispatch.
// Reference lines: azulnet/azul.py (every rule method: __init__ :18-61, new_round :64-89, move :118-161, is_legal_move :162-176,
// next_player :177-181, is_end_of_round :182-183, is_end_of_game :184-191, count_score :192-295, step :296-313, get_statistics :314-315),
// azulnet/game_runner.py:23-97 (GameRunner.__init__ / step / get_state / reset, RandomAgent), :113-117 (check_all_valid).
#pragma once

#include "azul_env2.hpp"

struct BatchDev {
    uint8_t *state;      // [N][128]
    u32 *mt;             // [N][624]
    u32 *mtpos;          // [N]
    const double2 *tab;  // the sampler's table as pairs, T_ROWS x T_STRIDE (azul_tables.hpp: build_sample_pairs)
    u64 *episodes;       // [N]
    u32 *stuck;          // [N]
    double *stat_sum;    // [N][10]
    u32 n;
    Rules rules;
    u64 draw_margin;     // AZ_DRAW_MARGIN; tests widen it to force the literal fp64 factory draw
    u64 *prof;           // [SEG_COUNT] segment cycle sums (only written by the -DAZ_PROFILE_SEGMENTS diagnostic build)
    u32 id_base;         // global id of game 0 (azul_batch_set_id_base): keys the policy sampler's Philox stream
    u32 move_limit;      // 0 = none (the reference's behaviour), else an episode is cut at the first end of a round with move_counter >= this (azul_batch_set_move_limit)
};

enum {
    OP_QUERY = 0, OP_INIT, OP_NEW_ROUND, OP_MOVE, OP_NEXT_PLAYER, OP_COUNT_SCORE, OP_STEP,
    OP_RUNNER_INIT, OP_RUNNER_RESET, OP_RUNNER_STEP, OP_RANDOM_ACTION, OP_SAMPLE_MASK, OP_POLICY_STEP, OP_AGENT_STEP,
    OP_NET_BEGIN, OP_NET_REPLY, OP_NET_RESET      // GameRunner.step / reset with an EXTERNAL opponent, cut at opponent_move() (azul_env2.hpp: NET_*)
};

struct OpArgs {
    int op;
    const i32 *actions;      // [count] in  (MOVE / STEP / RUNNER_STEP / POLICY_STEP / AGENT_STEP)
    const uint8_t *active;   // [count] in, optional
    const uint8_t *mask_in;  // [count][180] in (SAMPLE_MASK)
    i32 *actions_out;        // [count] out (RANDOM_ACTION / SAMPLE_MASK)
    uint8_t *status;         // [count] out
    i32 *reward;             // [count] out
    uint8_t *done;           // [count] out
    uint8_t *mask;           // [count][180] out (after the op)
    float *obs;              // [count][136] out (after the op)
    int persp;
    uint8_t *flags;          // [count] out
    i32 *potential;          // [count] out
    double *stats;           // [count][10] out
    uint8_t *player;         // [count] out: current_player after the op
    uint8_t *rng_dirty;      // [count] out: the op regenerated the game's 624 MT19937 words (0 for ops that do not draw)
    uint8_t *rec_out;        // [count][record bytes] out: the game's record after the op
    u32 *pos_out;            // [count] out: index of the game's MT19937 stream after the op
    i32 *next_action;        // [count] out: RandomAgent's choice on the state after the op, drawn at the stream's index after the op WITHOUT moving it
                             //         (-1: nothing legal, -2: not available -- the op failed / did not draw, or the draw would cross a regeneration)
    uint8_t *pending;        // [count] in / out (OP_NET_*): NET_READY / NET_REPLY / NET_OPENING -- does the game owe an opponent_move()?
    uint8_t *replies;        // [count] in / out, optional (OP_NET_*): opponent moves played since the agent's action
    u32 *owing;              // [1] optional (OP_NET_*): += the games of this launch that still owe an opponent_move()
    u32 pos_set;             // 0, or 1 + the stream index to install before the op (single-game calls: the host's index is the authority)
    u32 first, count;        // the launch covers games first .. first + count - 1; row i of the arrays above belongs to game first + i
};

AZ_FN bool op_needs_rng(int op)
{
    return op == OP_INIT || op == OP_NEW_ROUND || op == OP_STEP || op == OP_RUNNER_INIT || op == OP_RUNNER_RESET ||
           op == OP_RUNNER_STEP || op == OP_RANDOM_ACTION || op == OP_SAMPLE_MASK || op == OP_POLICY_STEP || op == OP_AGENT_STEP ||
           op == OP_NET_BEGIN || op == OP_NET_REPLY || op == OP_NET_RESET;
}

constexpr u32 OP2_OBS_STRIDE = 144;      // floats per half of the observation staging row (136 used)

// One rule call on games 2 pair and 2 pair + 1 of the launch (rows of the caller's arrays; games a.first + row of the batch): the op, then
// the queries on the state it leaves.  mt_lds: 624 words of LDS per half for the game's MT19937 state; tabfs_lds: the sampler's table;
// obs_lds: staging row of observe2.
template <bool LID>
AZ_FN void op_body2(const BatchDev &b, const OpArgs &a, u32 pair, u32 (*mt_lds)[az2::MT_LDS_WORDS], double2 *tabfs_lds, float (*obs_lds)[OP2_OBS_STRIDE])
{
    using namespace az2;
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    for (u32 i = lane; i < (u32)T_PAIRS; i += 64u) tabfs_lds[i] = b.tab[i];
    lds_sync();
    const u32 oi = 2u * pair + half;                 // row of the caller's arrays
    if (oi >= a.count) return;                       // odd launch: the last wave serves one game
    const u32 gi = oi + a.first;                     // game of the batch
    const bool act = a.active ? (a.active[oi] != 0) : true;
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES;
    K2 k;
    k2_init(k);
    rng2_set_move_limit(mt_lds[half], b.move_limit, l);
    const Tab2 tab = {tabfs_lds};
    G2 g;
    g2_load(g, rec, l);
    prime2(g, k);
    const u32 fp = b.rules.first_player;
    const u64 margin = b.draw_margin;
    u32 st = ST_OK, rdirty = 0;
    i32 spec = -2;
    if (act && a.op != OP_QUERY) {
        const bool use_rng = op_needs_rng(a.op);
        Rng2 r;
        u32 *gmt = b.mt + (size_t)gi * 624u;
        if (use_rng) rng2_open(r, gmt, mt_lds[half], a.pos_set ? a.pos_set - 1u : b.mtpos[gi], l);      // (move / next_player / count_score never draw: no 2.5 KB staging)
        else { r.lds = mt_lds[half]; r.tlds = nullptr; r.pos = 0; r.dirty = 0; r.wbase = 0; r.wend = 0; r.win = 0; }
        Counters2 cnt;
        const bool net = a.op == OP_NET_BEGIN || a.op == OP_NET_REPLY || a.op == OP_NET_RESET;
        const bool counts = a.op == OP_RUNNER_STEP || a.op == OP_POLICY_STEP || a.op == OP_AGENT_STEP || net;
        if (counts) counters2_open(cnt, b.episodes + gi, b.stuck + gi, b.stat_sum + (size_t)gi * 10, l);
        bool dirty_state = true;
        i32 rew = 0;
        u32 dn = 0;
        switch (a.op) {
        case OP_INIT:
            game_ctor2<LID>(g, fp, r, k);
            break;
        case OP_NEW_ROUND:
            st = new_round2<LID>(g, r, margin, k);
            break;
        case OP_MOVE: {
            const i32 av = a.actions[oi];
            if (av < 0 || av >= 180) { st = ST_BAD_ACTION; dirty_state = false; break; }
            do_move2<LID>(g, action_code2((u32)av, k), g.B, k);
        } break;
        case OP_NEXT_PLAYER:
            g.cur = (g.cur < 2u) ? g.cur + 1u : 1u;
            break;
        case OP_COUNT_SCORE:
            count_score2<LID>(g, k);
            break;
        case OP_STEP: {
            Mask2 m;
            legal_mask2(g, k, m);
            st = checked_step2<LID>(g, a.actions[oi], m, r, margin, k);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
        } break;
        case OP_RUNNER_INIT:
            st = episode_reset2<LID>(g, fp, r, margin, k);
            break;
        case OP_RUNNER_RESET:
            st = reset2<LID>(g, fp, r, margin, k);
            if (!st) st = opponent_loop2<LID>(g, r, tab, margin, k, true);
            break;
        case OP_RUNNER_STEP: {
            Mask2 m;
            legal_mask2(g, k, m);
            st = runner_step2<LID>(g, a.actions[oi], m, r, tab, margin, k, rew, dn);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
            if (!st && dn) episode_stats2(g, cnt, l);
            if (st == ST_STUCK) cnt.stuck_add += 1u;
        } break;
        case OP_POLICY_STEP: {
            Mask2 m;
            legal_mask2(g, k, m);
            st = policy_step2<LID>(g, a.actions[oi], m, fp, r, margin, cnt, k, rew, dn);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_BAD_ACTION);
        } break;
        case OP_AGENT_STEP: {
            Mask2 m;
            legal_mask2(g, k, m);
            st = agent_step2<LID>(g, a.actions[oi], m, fp, r, tab, margin, cnt, k, rew, dn);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_BAD_ACTION);
        } break;
        case OP_NET_BEGIN:
        case OP_NET_REPLY:
        case OP_NET_RESET: {
            // GameRunner.step / reset with an external opponent (game_runner.py:27-30, 37-47, 84-85), one launch per cut: the agent's move,
            // or one opponent_move() of the games that owe one, or the fresh game -- then the loop condition (azul_env2.hpp: net_settle2)
            Mask2 m;
            NetStep ns;
            if (a.op == OP_NET_RESET) {
                net_reset2<LID>(g, m, fp, r, margin, cnt, k, ns);
            } else {
                const bool agent = a.op == OP_NET_BEGIN;
                ns.pending = agent ? (u32)NET_READY : (u32)a.pending[oi];
                ns.replies = a.replies ? a.replies[oi] : 0u; ns.rew = 0; ns.dn = 0; ns.closed = false;
                ns.st = a.status ? a.status[oi] : 0u;                   // the step's status is its FIRST status that was not OK
                dirty_state = agent || ns.pending != NET_READY;
                if (dirty_state) {
                    legal_mask2(g, k, m);
                    net_move2<LID>(g, a.actions[oi], agent, m, fp, r, margin, cnt, k, ns);
                }
            }
            st = ns.st;
            if (l == 0u) {
                a.pending[oi] = (uint8_t)ns.pending;
                if (a.replies) a.replies[oi] = (uint8_t)(ns.replies < 255u ? ns.replies : 255u);
                if (ns.closed && a.reward) a.reward[oi] = ns.rew;       // the agent step's reward / done: written by the launch that closes it
                if (ns.closed && a.done) a.done[oi] = (uint8_t)ns.dn;
                if (ns.pending != NET_READY && a.owing) atomicAdd(a.owing, 1u);
            }
        } break;
        case OP_RANDOM_ACTION: {
            Mask2 m;
            legal_mask2(g, k, m);
            u32 code;
            const i32 av = random_agent2(m, r, tab, k, code) ? (i32)(code >> 17) : -1;
            if (l == 0u) a.actions_out[oi] = av;
            dirty_state = false;
        } break;
        case OP_SAMPLE_MASK: {
            const uint8_t *mi = a.mask_in + (size_t)oi * AZUL_NUM_ACTIONS;
            Mask2 m;
            m.B = g.B;
#pragma unroll
            for (u32 rr = 0; rr < 6u; rr++) {
                m.bit[rr] = (l < 30u && mi[30u * rr + (l < 30u ? l : 0u)] != 0) ? 1u : 0u;
                m.m[rr] = hb(m.bit[rr] != 0u);
            }
            u32 code;
            const i32 av = random_agent2(m, r, tab, k, code) ? (i32)(code >> 17) : -1;
            if (l == 0u) a.actions_out[oi] = av;
            dirty_state = false;
        } break;
        default:
            dirty_state = false;
            break;
        }
        if (a.op == OP_RUNNER_STEP || a.op == OP_POLICY_STEP || a.op == OP_AGENT_STEP) {
            if (a.reward && l == 0u) a.reward[oi] = rew;
            if (a.done && l == 0u) a.done[oi] = (uint8_t)dn;
        }
        if (dirty_state) {
            // the derived fields (sources board, "row accepts colour" boards, what-if cache, wall status) follow the state: the rule methods
            // called one by one (INIT, NEW_ROUND, MOVE, COUNT_SCORE) leave them to whoever looks at the state next
            g.B = hb(g.cs != 0u) & 0x7fffffffu;
            prime2(g, k);
            g2_store(g, rec, l);
        }
        if (a.next_action && use_rng && st == ST_OK && r.pos + 2u <= 624u) {
            // the question a GameRunner loop asks next (nn_runner.py:22-30 with RandomAgent: get_valid_moves -> get_a_output): answered
            // here from the two words the stream would hand out next, index restored -- the caller advances it when it plays the answer
            const u32 keep = r.pos;
            Mask2 m;
            legal_mask2(g, k, m);
            u32 code;
            spec = random_agent2(m, r, tab, k, code) ? (i32)(code >> 17) : -1;
            r.pos = keep;
        }
        if (use_rng) rng2_close(r, gmt, b.mtpos + gi, l);
        rdirty = use_rng ? r.dirty : 0u;
        if (counts) counters2_close(cnt, l);
    }
    if (a.next_action && l == 0u) a.next_action[oi] = spec;
    if (a.rng_dirty && l == 0u) a.rng_dirty[oi] = (uint8_t)rdirty;
    if (a.status && act && l == 0u) a.status[oi] = (uint8_t)st;
    if (a.rec_out) g2_store(g, a.rec_out + (size_t)oi * AZUL_RECORD_BYTES, l);
    if (a.pos_out && l == 0u) a.pos_out[oi] = b.mtpos[gi];      // (written by rng2_close above when the op drew)
    // queries on the post-op state
    if (a.mask) {
        Mask2 m;
        legal_mask2(g, k, m);
        uint8_t *row = a.mask + (size_t)oi * AZUL_NUM_ACTIONS + l;
        if (l < 30u) {
            row[0] = (uint8_t)m.bit[0]; row[30] = (uint8_t)m.bit[1]; row[60] = (uint8_t)m.bit[2];
            row[90] = (uint8_t)m.bit[3]; row[120] = (uint8_t)m.bit[4]; row[150] = (uint8_t)m.bit[5];
        }
    }
    if (a.obs) {
        const u32 p = (a.persp == AZUL_PERSP_CURRENT) ? me2(g) : (u32)a.persp;
        observe2(g, p, obs_lds[half], a.obs + (size_t)oi * AZUL_OBS_SIZE, l);
    }
    if (a.flags) {
        const u32 f = (g.B == 0u ? AZUL_FLAG_END_OF_ROUND : 0) | (g.over ? AZUL_FLAG_END_OF_GAME : 0) | (g.eog ? AZUL_FLAG_ENDED_FLAG : 0);
        if (l == 0u) a.flags[oi] = (uint8_t)f;
    }
    if (a.potential && l == 0u) a.potential[oi] = g.wi0 - g.wi1;       // game_runner.py:48-50 (the what-if cache is current)
    if (a.stats && l == 0u) {
        for (u32 q = 0; q < 10u; q++) a.stats[(size_t)oi * 10 + q] = game_stat2(g, q);
    }
    if (a.player && l == 0u) a.player[oi] = (uint8_t)g.cur;
}
