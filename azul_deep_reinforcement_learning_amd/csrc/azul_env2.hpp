// azul_env2.hpp -- GameRunner on the two-player rule book: GameRunner.step / GameRunner.reset / GameRunner.get_state /
// RandomAgent.get_a_output / Azul.step with its legality test, for TWO GAMES PER WAVEFRONT on top of azul_selfplay2.hpp's half-uniform
// rules (state in VGPRs, one game per 32-lane half).  Two users: the env side of the persistent policy rollout (azul_rollout2.hpp calls
// these between its matrix phases) and the two-player rule entries of the C ABI (azul_ops2.hpp).  tests/hostcheck/simt_env2.cpp and
// simt_ops2.cpp run THIS FILE, unmodified, lane by lane on the CPU against the oracle.
// Reference lines: azulnet/azul.py:296-313 (step), azulnet/game_runner.py:43-55 (GameRunner.step), :56-72 (get_state), :76-85 (reset),
// :87-97 (RandomAgent); azulnet/nn_runner.py:17-47.
#pragma once
#include "azul_selfplay2.hpp"

namespace az2 {

// ---- observation: game_runner.py:56-72, 136 values; value j of my game for j = lane + 32 i ------------------------------------------
// (layout: cells 0..30 | pattern_lines[order[0]] | pattern_lines[order[1]] | walls[order[0]] | walls[order[1]] |
// floors | scores | next first player, seen from player `persp`)
template <int I>
AZ_FN u32 observe_val2(const G2 &g, u32 o0 /* half-uniform: 0 / 1 */, u32 l)
{
    const u32 cpa = o0 ? g.cp1 : g.cp0, cpb = o0 ? g.cp0 : g.cp1;
    const u32 wall_a = o0 ? g.wall1 : g.wall0, wall_b = o0 ? g.wall0 : g.wall1;
    if (I == 0) {
        const u32 pa0 = hread(cpa, 0u);
        return l < 31u ? g.cs : pa0;                                   // j = 31: order[0]'s cell 0
    }
    if (I == 1) {                                                       // j = 32 + l: order[0] cells 1..24 (l < 24), order[1] cells 0..7
        const u32 va = hread(cpa, l + 1u), vb = hread(cpb, l - 24u);
        return l < 24u ? va : vb;
    }
    if (I == 2) {                                                       // j = 64 + l: order[1] cells 8..24 (l < 17), walls[order[0]] bits 0..14
        const u32 vb = hread(cpb, l + 8u);
        return l < 17u ? vb : (wall_a >> ((l - 17u) & 31u)) & 1u;
    }
    if (I == 3)                                                         // j = 96 + l: walls[order[0]] bits 15..24 (l < 10), walls[order[1]] bits 0..21
        return l < 10u ? (wall_a >> (l + 15u)) & 1u : (wall_b >> ((l - 10u) & 31u)) & 1u;
    // j = 128 + l (l < 8): walls[order[1]] bits 22..24, floors, scores, next first player
    const u32 floor_a = o0 ? g.floor1 : g.floor0, floor_b = o0 ? g.floor0 : g.floor1;
    const i32 score_a = o0 ? g.score1 : g.score0, score_b = o0 ? g.score0 : g.score1;
    const u32 pnfp = g.nfp > 0u ? (((g.nfp - 1u - o0) & 1u) + 1u) : 0u;     // game_runner.py:58-61
    u32 v = (wall_b >> ((l + 22u) & 31u)) & 1u;
    v = l == 3u ? floor_a : v;
    v = l == 4u ? floor_b : v;
    v = l == 5u ? (u32)score_a : v;
    v = l == 6u ? (u32)score_b : v;
    v = l == 7u ? pnfp : v;
    return v;
}

// writes the 136 floats of my game to `lds_row` (the network's A operand) and, when given, to `glob` (trajectory slot)
AZ_FN void observe2(const G2 &g, u32 persp, float *lds_row, float *glob, u32 l)
{
    const u32 o0 = persp & 1u;
    float v0 = (float)(i32)observe_val2<0>(g, o0, l), v1 = (float)(i32)observe_val2<1>(g, o0, l), v2 = (float)(i32)observe_val2<2>(g, o0, l),
          v3 = (float)(i32)observe_val2<3>(g, o0, l), v4 = (float)(i32)observe_val2<4>(g, o0, l);
    lds_row[l] = v0; lds_row[l + 32u] = v1; lds_row[l + 64u] = v2; lds_row[l + 96u] = v3;
    if (l < 8u) lds_row[l + 128u] = v4;
    if (glob) {          // (the rollout kernel passes NULL: its trajectory slot is filled from the LDS rows by waves that idle during the head phase)
        glob[l] = v0; glob[l + 32u] = v1; glob[l + 64u] = v2; glob[l + 96u] = v3;
        if (l < 8u) glob[l + 128u] = v4;
    }
}

// ---- RandomAgent.get_a_output (game_runner.py:87-97): selfplay_step2's decision as a function --------------------------------------
// Returns false when nothing is legal (ValueError in the reference, raised BEFORE random() is called: no words consumed).
AZ_FN bool random_agent2(const Mask2 &m, Rng2 &r, const Tab2 &T, const K2 &k, u32 &code)
{
    const u32 l = k.l;
    const u32 c0 = __popc(m.m[0]), c1 = __popc(m.m[1]), c2 = __popc(m.m[2]), c3 = __popc(m.m[3]), c4 = __popc(m.m[4]), c5 = __popc(m.m[5]);
    const u32 J = c0;
    const u32 p1 = c0, p2 = p1 + c1, p3 = p2 + c2, p4 = p3 + c3, p5 = p4 + c4, L = p5 + c5;
    code = 0;
    if (L == 0u) return false;
    // (floor-only masks, M == 0, take the table's ninth pair {fl(100 S[J]), 0} with the ordinal counted from 0: selfplay_step2, azul_tables.hpp)
    const u32 M = L - J, Mc = M ? M : 256u, Jc = J < 31u ? J : 30u;
    const u32 kbase = M ? J : 0u;
    const double2 fs = T.fs[9u * Jc + 31u - (u32)__builtin_clz(Mc)];
    const double sJ = fs.y;
    const double total = ((double)M + fs.x) + 0.0;
    const double u = rng2_random(r, l);
    const double x = u * total;
    const double d = x - sJ;
    const u32 fl = (u32)d;
    const double fr = d - (double)fl;
    u32 kg = kbase + fl + 1u;
    const bool edge = !(__builtin_fabs(fr - 0.5) < 0.5 - 1e-9) | (kg > L);
    if (AZ_UNLIKELY(edge)) {
        const double sT = M ? sJ : T.fs[9u * Jc].y;      // the boundary search works on CPython's own x = random() * (S[J] + 0.0)
        kg = sample_slow2(T, M ? x : u * (sT + 0.0), sT, J, M, L);
    }
    const u32 want = kg - 1u;
    const bool g1 = want >= p1, g2 = want >= p2, g3 = want >= p3, g4 = want >= p4, g5 = want >= p5;
    // (the selects as a tree of depth three -- the g's are monotone, g5 => g4 => .. => g1 -- instead of a chain of five: -0.45 % on the headline kernel)
    const u32 w01 = g1 ? m.m[1] : m.m[0], w23 = g3 ? m.m[3] : m.m[2], w45 = g5 ? m.m[5] : m.m[4];
    const u32 b01 = g1 ? p1 : 0u, b23 = g3 ? p3 : p2, b45 = g5 ? p5 : p4;
    const u32 mword = g4 ? w45 : (g2 ? w23 : w01);
    const u32 base = g4 ? b45 : (g2 ? b23 : b01);
    const u32 prow_ = (u32)g1 + (u32)g2 + (u32)g3 + (u32)g4 + (u32)g5;
    // my bit is set AND its rank among the word's set bits is the wanted one, as ONE compare (a ballot of an and of two compares goes
    // through a 0 / 1 register): 2 (rank - wanted) + bit == 1
    const bool hit = (((u32)__popc(mword & ((1u << l) - 1u)) - (want - base)) << 1) + ((mword >> l) & 1u) == 1u;
    const u32 ln = (u32)__builtin_ctz(hb(hit) | 0x80000000u);
    code = hbcast(k.lcode, ln) | (prow_ << 13) | ((30u * prow_ + ln) << 17);
    return true;
}

// the same packing for an action given by number
AZ_FN u32 action_code2(u32 a, const K2 &k)
{
    const u32 prow_ = a / 30u, ln = a - 30u * prow_;
    return hbcast(k.lcode, ln) | (prow_ << 13) | (a << 17);
}

// ---- Azul.step's body for a legal move (azul.py:304-313: move, then end of round -> scoring -> end of game / next round), what-if caches kept
template <bool LID>
AZ_FN u32 apply_step2(G2 &g, u32 code, Rng2 &r, u64 margin, const K2 &k)
{
    const u32 me = me2(g);
    const bool filled = do_move2<LID>(g, code, g.B, k);               // azul.py:304
    g.B = hb(g.cs != 0u) & 0x7fffffffu;
    const bool eor = g.B == 0u;                                        // :306 (then count_score2 prices the lines for real and clears the cache)
    i32 wc = me ? g.wc1 : g.wc0;
    if (wave_any(filled & !eor)) {
        const i32 fresh = wall_points2(me ? g.wall1 : g.wall0, full_lines2(me ? g.cp1 : g.cp0, k), k);
        wc = filled ? fresh : wc;
    }
    const i32 wi = clamp0((me ? g.score1 : g.score0) + floor_penalty(me ? g.floor1 : g.floor0) + wc);
    g.wc0 = me ? g.wc0 : wc; g.wc1 = me ? wc : g.wc1;
    g.wi0 = me ? g.wi0 : wi; g.wi1 = me ? wi : g.wi1;
    g.cur = eor ? g.cur : (g.cur & 1u) + 1u;              // :313
    u32 st = ST_OK;
    // (rare events are tested per wave and kept out of line: two waves per SIMD cannot hide a taken branch's instruction refetch)
    if (AZ_UNLIKELY(wave_any(eor))) {
        if (eor) {
            count_score2<LID>(g, k);                                   // :307
            if (g.over) g.eog = 1;                                     // :308-309
            else if (g.moves + 1u >= rng2_move_limit(r)) st = ST_TRUNCATED;      // move limit (beyond the reference, off by default; the caller counts this move next)
            else st = new_round2<LID>(g, r, margin, k);                // :311
        }
    }
    return st;
}

// Azul.step with the legality test of azul.py:298-302; `m` is the mask of the current state
template <bool LID>
AZ_FN u32 checked_step2(G2 &g, i32 a, const Mask2 &m, Rng2 &r, u64 margin, const K2 &k)
{
    if (g.eog) return ST_GAME_ENDED;
    if (a < 0 || a >= 180) return ST_BAD_ACTION;
    const u32 row = (u32)a / 30u, bit = (u32)a - 30u * row;
    const u32 word = row == 0u ? m.m[0] : row == 1u ? m.m[1] : row == 2u ? m.m[2] : row == 3u ? m.m[3] : row == 4u ? m.m[4] : m.m[5];
    if (((word >> bit) & 1u) == 0u) return ST_ILLEGAL_MOVE;             // state untouched
    return apply_step2<LID>(g, action_code2((u32)a, k), r, margin, k);
}

AZ_FN u32 mask_count2(const Mask2 &m) { return __popc(m.m[0]) + __popc(m.m[1]) + __popc(m.m[2]) + __popc(m.m[3]) + __popc(m.m[4]) + __popc(m.m[5]); }

AZ_FN void episode_stats2(const G2 &g, Counters2 &cnt, u32 l)
{
    const double f0 = (double)(g.fps & 0xffffu), f1 = (double)(g.fps >> 16);
    counters2_episode(cnt, stat_lane(l, g.score0, g.score1, g.turn, f0 / (f0 + f1) * 100, g.fp0, g.mc0, g.cl0));
}

template <bool LID>
AZ_FN u32 reset2(G2 &g, u32 first_player, Rng2 &r, u64 margin, const K2 &k)
{
    u32 st = episode_reset2<LID>(g, first_player, r, margin, k);
    prime2(g, k);
    return st;
}

// One env move of policy-driven self-play on a register-resident game: Azul.step (azul.py:296-313) for the current player, the shaped
// reward of game_runner.py:48-52 per move, done, statistics and the auto-reset of game_runner.py:76-82 (ONE reset site)
template <bool LID>
AZ_FN u32 policy_step2(G2 &g, i32 av, const Mask2 &m /* of the current state: the one that was published */, u32 first_player, Rng2 &r, u64 margin,
                        Counters2 &cnt, const K2 &k, i32 &rew, u32 &dn)
{
    rew = 0; dn = 0;
    // "no action" is legitimate only when nothing is legal (hazard H3)
    bool stuck = false;
    if (AZ_UNLIKELY(wave_any(av < 0))) stuck = (av < 0) & (g.eog == 0u) & (mask_count2(m) == 0u);
    u32 st = ST_OK;
    bool restart = false;
    if (!stuck) {
        st = checked_step2<LID>(g, av, m, r, margin, k);
        const bool dirty = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
        if (dirty) {
            g.moves += 1u;
            const i32 phi = g.wi0 - g.wi1;                             // game_runner.py:48-50 (the what-if caches are current)
            rew = phi - g.pscore;
            g.pscore = phi;
            dn = g.over ? 1u : (st == ST_TRUNCATED ? 3u : 0u);         // is_end_of_game(): the walls, not the record's flag; 3: cut by the move limit
            restart = dn && (st == ST_OK || st == ST_TRUNCATED);
        } else if (st == ST_GAME_ENDED) {                               // a finished game handed in: restart the slot, report done
            dn = 1u;
            restart = true;
        }
    }
    if (AZ_UNLIKELY(wave_any(stuck | restart))) {
        if (stuck | restart) {
            if (stuck) { cnt.stuck_add += 1u; dn = 2u; }
            else if (st == ST_TRUNCATED) cnt.stuck_add += 1u;          // (a cut episode is no finished game)
            else if (st == ST_OK) episode_stats2(g, cnt, k.l);         // (a game handed in finished is not counted: st == ST_GAME_ENDED)
            const bool cut = st == ST_TRUNCATED;
            u32 st0 = reset2<LID>(g, first_player, r, margin, k);
            st = stuck ? (st0 ? st0 : (u32)ST_STUCK) : (cut ? (st0 ? st0 : (u32)ST_TRUNCATED) : st0);
        }
    }
    return st;
}

// GameRunner's opponent loop (game_runner.py:46-47 / :84)
template <bool LID>
AZ_FN u32 opponent_loop2(G2 &g, Rng2 &r, const Tab2 &T, u64 margin, const K2 &k, bool until_player1_only)
{
#pragma unroll 1
    for (u32 guard = 0; guard < 4096u; guard++) {
        Mask2 m;
        legal_mask2(g, k, m);
        const bool keep = until_player1_only ? (g.cur != 1u) : ((g.cur != 1u || mask_count2(m) < 2u) && !g.over);
        if (!keep) break;
        u32 code;
        if (!random_agent2(m, r, T, k, code)) return ST_STUCK;
        if (g.eog) return ST_GAME_ENDED;
        u32 st = apply_step2<LID>(g, code, r, margin, k);
        if (st) return st;
        g.moves += 1u;
    }
    return ST_OK;
}

// GameRunner.step (game_runner.py:43-55): the agent's move, the opponent's RandomAgent replies, the shaped reward, done.  No reset.
template <bool LID>
AZ_FN u32 runner_step2(G2 &g, i32 av, const Mask2 &m /* of the current state */, Rng2 &r, const Tab2 &T, u64 margin, const K2 &k, i32 &rew, u32 &dn)
{
    rew = 0;
    dn = g.over ? 1u : 0u;
    u32 st = checked_step2<LID>(g, av, m, r, margin, k);                // game_runner.py:44
    if (!st) {
        g.moves += 1u;                                                  // :45
        st = opponent_loop2<LID>(g, r, T, margin, k, false);            // :46-47
        if (!st) {
            const i32 phi = g.wi0 - g.wi1;                             // :48-50
            rew = phi - g.pscore;                                      // :51
            g.pscore = phi;                                            // :52
            dn = g.over ? 1u : 0u;                                     // :55
        }
    }
    return st;
}

// One AGENT step of NNRunner.run_episode (nn_runner.py:24-29): GameRunner.step and, when the episode ends, the GameRunner.reset() that
// opens the next run_episode (nn_runner.py:20 -> game_runner.py:76-82, incl. the opponent's opening moves), so the observation / mask
// taken afterwards are the next decision's.
template <bool LID>
AZ_FN u32 agent_step2(G2 &g, i32 av, const Mask2 &m /* of the current state */, u32 first_player, Rng2 &r, const Tab2 &T, u64 margin,
                       Counters2 &cnt, const K2 &k, i32 &rew, u32 &dn)
{
    u32 st = runner_step2<LID>(g, av, m, r, T, margin, k, rew, dn);
    const bool dirty = !(st == ST_ILLEGAL_MOVE || st == ST_BAD_ACTION);
    if (st == ST_STUCK) { cnt.stuck_add += 1u; dn = 2u; rew = 0; }   // hazard H3: nobody can move
    else if (st == ST_TRUNCATED) { cnt.stuck_add += 1u; dn = 3u; rew = 0; }       // move limit: the episode was cut at the end of a round
    else if (st == ST_GAME_ENDED) dn = 1u;
    else if (st == ST_OK && dn) episode_stats2(g, cnt, k.l);
    if (dirty && dn) {
        u32 st2 = reset2<LID>(g, first_player, r, margin, k);
        if (!st2) st2 = opponent_loop2<LID>(g, r, T, margin, k, true);
        if (st == ST_OK || (st == ST_TRUNCATED && st2)) st = st2;
    }
    return st;
}

// ---- GameRunner with an EXTERNAL opponent (game_runner.py:27-30: GameRunner(opponent=Agent(...)); scripts/run_batch.py:6-10) --------------
// The reference's GameRunner.step / reset call opponent.get_a_output(...) in a data-dependent loop (game_runner.py:46-47, :84-85).  When the
// opponent is a network the answer comes from OUTSIDE the rule code (the matrix phases of the rollout kernel, or a separate launch), so the
// two methods are cut at their opponent_move() calls into a three-state protocol per game:
//     NET_REPLY    inside GameRunner.step's loop (:46): the opponent moves while (current_player != 1 or player 1 has fewer than two legal
//                  moves) and the game is not over -- player 1's FORCED moves are the opponent's too
//     NET_OPENING  inside GameRunner.reset's loop (:84): the opponent moves while current_player != 1
//     NET_READY    nothing owed: the agent's next decision
// net_move2 plays the agent's action (:44-45) or one opponent_move() (:37-42) with the action somebody chose on the mask /
// perspective-rotated observation net_settle2 left; net_settle2 runs the loop conditions, closes the step (:48-55: shaped reward, done,
// statistics) and opens the next episode (nn_runner.py:20 -> game_runner.py:76-82) -- agent_step2's bookkeeping, statement for statement,
// with the RandomAgent's draw replaced by "owed".  `m` is always the legal mask of the state left behind.
enum : u32 { NET_READY = 0, NET_REPLY = 1, NET_OPENING = 2, NET_RESET = 3 /* internal: the next episode's game is due (never stored) */ };

struct NetStep {
    u32 pending;     // NET_*
    u32 replies;     // opponent moves played since the agent's action (opening moves of the next episode included)
    i32 rew;         // valid once `closed`
    u32 dn;
    bool closed;     // the agent step's reward / done are final
    u32 st;          // first status that was not ST_OK
};

// The loop conditions.  Leaves ns.pending = NET_REPLY / NET_OPENING (an opponent_move() is owed on the state and mask left behind) or
// NET_READY.  ONE site each for the episode reset and the legal mask: the rollout kernel inlines this function once.
template <bool LID>
AZ_FN void net_settle2(G2 &g, NetStep &ns, Mask2 &m, u32 first_player, Rng2 &r, u64 margin, Counters2 &cnt, const K2 &k)
{
#pragma unroll 1
    for (u32 pass = 0; pass < 4u; pass++) {
        if (ns.pending == NET_RESET) {                                          // the next run_episode: nn_runner.py:20 -> game_runner.py:76-82
            const u32 st2 = reset2<LID>(g, first_player, r, margin, k);
            if (!ns.st) ns.st = st2;
            ns.pending = st2 ? (u32)NET_READY : (u32)NET_OPENING;
            continue;
        }
        legal_mask2(g, k, m);
        if (ns.pending == NET_READY) break;
        const u32 legal = mask_count2(m);
        if (ns.pending == NET_REPLY) {
            const bool keep = (g.cur != 1u || legal < 2u) && !g.over;          // game_runner.py:46
            if (keep && legal) break;                                           // :47 -- an opponent_move() is owed
            ns.closed = true;
            if (keep) {                                                         // nobody can move (hazard H3; an Agent raises IllegalMask, model.py:33-34)
                cnt.stuck_add += 1u; ns.dn = 2u; ns.rew = 0;
                if (!ns.st) ns.st = ST_STUCK;
            } else {
                const i32 phi = g.wi0 - g.wi1;                                 // :48-50 (the what-if caches are current)
                ns.rew = phi - g.pscore;                                       // :51
                g.pscore = phi;                                                // :52
                ns.dn = g.over ? 1u : 0u;                                      // :55
                if (ns.dn) episode_stats2(g, cnt, k.l);                        // :53-54
            }
            ns.pending = ns.dn ? (u32)NET_RESET : (u32)NET_READY;
            if (!ns.dn) break;                                                  // (`m` is the mask of this state: the agent's next decision)
        } else {                                                                // NET_OPENING
            if (g.cur != 1u && legal) break;                                    // game_runner.py:84-85 -- an opponent_move() is owed
            if (g.cur != 1u && !ns.st) ns.st = ST_STUCK;
            ns.pending = NET_READY;
            break;
        }
    }
}

// One move of the protocol: the AGENT's action of GameRunner.step (game_runner.py:44-45; `agent`) or one opponent_move() (:37-42) of a game
// that owes one, then the loop conditions.  `m`: in, the legal mask of the current state; out, of the state left behind.
// An opponent's answer that is not a legal action leaves the game and the debt as they are (the reference lets IllegalMove escape from
// GameRunner.step); refused / failed agent moves follow agent_step2's bookkeeping.
template <bool LID>
AZ_FN void net_move2(G2 &g, i32 av, bool agent, Mask2 &m, u32 first_player, Rng2 &r, u64 margin, Counters2 &cnt, const K2 &k, NetStep &ns)
{
    if (agent) { ns.pending = NET_READY; ns.replies = 0; ns.rew = 0; ns.dn = g.over ? 1u : 0u; ns.closed = false; ns.st = ST_OK; }
    const u32 st = checked_step2<LID>(g, av, m, r, margin, k);                 // :44 / :41
    if (st == ST_OK) {
        g.moves += 1u;                                                          // :45 / :42
        if (agent) ns.pending = NET_REPLY; else ns.replies += 1u;
    } else {
        if (!ns.st) ns.st = st;
        if (st == ST_ILLEGAL_MOVE || st == ST_BAD_ACTION) {                     // state untouched
            if (agent) ns.closed = true;
            return;
        }
        const bool in_step = agent || ns.pending == NET_REPLY;                  // (runner_step2 returning a status: reward 0, done only for GAME_ENDED)
        ns.pending = NET_READY;
        if (in_step) {
            ns.closed = true; ns.rew = 0;
            ns.dn = st == ST_GAME_ENDED ? 1u : (st == ST_TRUNCATED ? 3u : (agent ? ns.dn : 0u));
            if (st == ST_TRUNCATED) {                                           // move limit: the episode was cut at the end of a round
                cnt.stuck_add += 1u;
                if (!agent) ns.replies += 1u;                                   // (the opponent's move WAS played: the round it ended is scored)
            }
            if (ns.dn) ns.pending = NET_RESET;
        }
    }
    net_settle2<LID>(g, ns, m, first_player, r, margin, cnt, k);
}

// GameRunner.reset() with an external opponent (game_runner.py:76-85): the fresh game, then the opening loop's condition
template <bool LID>
AZ_FN void net_reset2(G2 &g, Mask2 &m, u32 first_player, Rng2 &r, u64 margin, Counters2 &cnt, const K2 &k, NetStep &ns)
{
    ns.pending = NET_RESET; ns.replies = 0; ns.rew = 0; ns.dn = 0; ns.closed = false; ns.st = ST_OK;
    net_settle2<LID>(g, ns, m, first_player, r, margin, cnt, k);
}

} // namespace az2
