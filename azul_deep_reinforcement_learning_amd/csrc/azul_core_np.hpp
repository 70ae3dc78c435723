// azul_core_np.hpp -- the Azul rules for THREE and FOUR players (row N4 of SURVEY.md 8f, first slice): exactly what the
// reference does for Azul(players=3|4) -- per-player pattern lines / walls / floors / scores / statistics for P players on the
// reference's FIVE factory displays (azulnet/azul.py:19, TODO at tests/test_azul.py:14), turn order 1..P (azul.py:177-181),
// the first-player draw random.choice(range(1, P+1)) (azul.py:37).  Seven / nine displays, end-of-game bonuses and a finite
// bag are beyond the reference and are not built.
//
// Same mapping as azul_core.hpp -- ONE GAME PER 64-LANE WAVEFRONT -- with a layout that scales in P:
//   cs      (VGPR)  lane 5d+c = displays[d][c], lane 25+c = center[c], lane 30 = token        (as in the two-player core)
//   cp[p]   (VGPR)  lane 5r+c = pattern_lines[p][r][c]                                        (one register per player)
//   pv      (VGPR)  lane 8q+p = per-player scalar q of player p: wall board, floor, score, first_player_stats,
//                   floor_penalty, max_combo, completed_lines (3 bytes)
//   cur, nfp, eog, turn, box, lid   uniform scalars
// The factory draw, the mask tail, the wall pricing, the RandomAgent sampler and the MT19937 stream are the two-player core's.
//
// Reference lines: __init__ azul.py:18-61, new_round :64-89, move :118-161, is_legal_move :162-176, next_player :177-181,
// is_end_of_round :182-183, is_end_of_game :184-191, count_score :192-295, step :296-313, get_statistics :314-315.
//
// Record: 256 bytes per game (include/azul_hip.h "wide record").
#pragma once
#include "azul_core.hpp"

namespace az {

enum { NP_RECORD_BYTES = 256 };
enum { PQ_WALL = 0, PQ_FLOOR = 1, PQ_SCORE = 2, PQ_FPS = 3, PQ_FPEN = 4, PQ_MAXC = 5, PQ_COMPL = 6, PQ_COUNT = 7 };

template <u32 P>
struct GameN {
    vu32 cs;
    vu32 cp[P];
    vu32 pv;
    u32 cur, nfp, eog, turn;
    u64 box, lid;
};

template <u32 P> AZ_FN u32 pget(const GameN<P> &g, u32 q, u32 p) { return readlane(g.pv, 8u * q + p); }
template <u32 P> AZ_FN void pset(GameN<P> &g, u32 q, u32 p, u32 v) { g.pv = writelane(g.pv, v, 8u * q + p); }
template <u32 P> AZ_FN u32 me_np(const GameN<P> &g) { return g.cur == 0u ? P - 1u : g.cur - 1u; }   // numpy [-1] before the first round

template <u32 P>
AZ_FN vu32 cp_of(const GameN<P> &g, u32 p)
{
    vu32 v = g.cp[0];
    for (u32 i = 1; i < P; i++) v = selu(p == i, g.cp[i], v);
    return v;
}

// byte offset / size of per-player scalar q of player p inside the wide record
AZ_FN void pv_layout(vu32 &off, vu32 &size, vbool &valid, vbool &is_signed, u32 players)
{
    vu32 l = lane();
    vu32 q = l >> 3, p = l & 7u;
    valid = (p < players) & (q < (u32)PQ_COUNT);
    off = sel(q == 0u, 136u + 4u * p, sel(q == 1u, 132u + p, sel(q == 2u, 152u + 2u * p, sel(q == 3u, 172u + 2u * p,
          sel(q == 4u, 180u + 2u * p, sel(q == 5u, 188u + p, 192u + 3u * p))))));
    size = sel(q == 0u, splat(4u), sel((q == 1u) | (q == 5u), splat(1u), sel(q == 6u, splat(3u), splat(2u))));
    is_signed = (q == 2u) | (q == 4u);
}

template <u32 P>
AZ_FN void gamen_load(GameN<P> &g, const uint8_t *rec)
{
    vu32 l = lane();
    vu32 a = ld_u8(rec, l, l < 32u);
    u32 flags = readlane(a, 31);
    g.cur = flags & 7u; g.nfp = (flags >> 3) & 7u; g.eog = (flags >> 6) & 1u;
    g.cs = sel(l < 31u, a, splat(0u));
    for (u32 p = 0; p < P; p++) g.cp[p] = ld_u8(rec + 32u + 25u * p, l, l < 25u);
    vu32 off, size;
    vbool valid, sgn;
    pv_layout(off, size, valid, sgn, P);
    vu32 v = ld_u8(rec, off, valid) | (ld_u8(rec, off + 1u, valid & (size > 1u)) << 8) |
             (ld_u8(rec, off + 2u, valid & (size > 2u)) << 16) | (ld_u8(rec, off + 3u, valid & (size > 3u)) << 24);
    v = sel(sgn, (v ^ 0x8000u) - 0x8000u, v);
    g.pv = sel(valid, v, splat(0u));
    vu32 t = ld_u8(rec + 160, l, l < 12u);                // box[5], lid[5], turn_counter (u16)
    g.box = 0; g.lid = 0;
    for (u32 c = 0; c < 5u; c++) { g.box |= (u64)readlane(t, c) << (8u * c); g.lid |= (u64)readlane(t, 5u + c) << (8u * c); }
    g.turn = readlane(t, 10) | (readlane(t, 11) << 8);
}

template <u32 P>
AZ_FN void gamen_store(const GameN<P> &g, uint8_t *rec)
{
    vu32 l = lane();
    u32 flags = (g.cur & 7u) | ((g.nfp & 7u) << 3) | ((g.eog & 1u) << 6);
    st_u8(rec, l, writelane(g.cs, flags, 31), l < 32u);
    for (u32 p = 0; p < P; p++) st_u8(rec + 32u + 25u * p, l, g.cp[p], l < 25u);
    vu32 off, size;
    vbool valid, sgn;
    pv_layout(off, size, valid, sgn, P);
    st_u8(rec, off, g.pv & 0xffu, valid);
    st_u8(rec, off + 1u, (g.pv >> 8) & 0xffu, valid & (size > 1u));
    st_u8(rec, off + 2u, (g.pv >> 16) & 0xffu, valid & (size > 2u));
    st_u8(rec, off + 3u, (g.pv >> 24) & 0xffu, valid & (size > 3u));
    vu32 t = splat(0u);
    for (u32 c = 0; c < 5u; c++) {
        t = writelane(t, (u32)(g.box >> (8u * c)) & 0xffu, c);
        t = writelane(t, (u32)(g.lid >> (8u * c)) & 0xffu, 5u + c);
    }
    t = writelane(t, g.turn & 0xffu, 10);
    t = writelane(t, (g.turn >> 8) & 0xffu, 11);
    st_u8(rec + 160, l, t, l < 12u);
    st_u8(rec + 204, l, splat(P), l < 1u);
}

// ---- legal-move mask: azul.py:162-176 for the player to move ----
template <u32 P>
AZ_FN u32 sources_board_np(const GameN<P> &g) { return (u32)ballot(g.cs != 0u) & 0x7fffffffu; }

template <u32 P>
AZ_FN void legal_mask_np(const GameN<P> &g, const LaneConst &k, Mask &out)
{
    u32 me = me_np(g);
    vu32 mine = cp_of(g, me);
    u32 pme = (u32)ballot((mine != 0u) & (lane() < 25u)) & 0x1ffffffu;
    mask_from_boards(sources_board_np(g), pme, pget(g, PQ_WALL, me), k, out);
}

// ---- move: azul.py:118-161 ----
template <bool LID, u32 P>
AZ_FN void do_move_np(GameN<P> &g, u32 code)
{
    u32 me = me_np(g);
    vu32 l = lane();
    const u32 src = code & 31u, db = (code >> 5) & 31u, c = (code >> 10) & 7u, row = (code >> 13) & 7u;
    const bool from_display = ((code >> 16) & 1u) != 0u;
    u32 n = readlane(g.cs, src);                                       // :127 / :136
    bool token = (!from_display) & (readlane(g.cs, 30) == 1u);         // :140
    vu32 moved = bperm(g.cs, l - 25u + db);                            // :131 the rest of the display slides into the centre
    vbool centre = (l >= 25u) & (l < 30u) & (l != 25u + c) & from_display;
    vbool gone = ((l >= db) & (l < db + 5u) & from_display) | (l == src) | ((l == 30u) & token);   // :129,:133,:138,:141
    g.cs = sel(gone, splat(0u), sel(centre, g.cs + moved, g.cs));
    g.nfp = token ? g.cur : g.nfp;                                     // :142
    u32 fl = pget(g, PQ_FLOOR, me) + (token ? 1u : 0u);                // :143
    fl = fl < 7u ? fl : 7u;
    u32 cell = 5u * ((row ? row : 1u) - 1u) + c;
    vu32 mine = cp_of(g, me);
    u32 old = readlane(mine, cell);
    i32 overflow = row ? (i32)row - (i32)old - (i32)n : -(i32)n;       // :147 ; floor move: everything "overflows"
    u32 spill = overflow < 0 ? (u32)(-overflow) : 0u;
    u32 newv = overflow < 0 ? row : old + n;                           // :150 / :152
    mine = sel((l == cell) & (row != 0u), splat(newv), mine);
    for (u32 i = 0; i < P; i++) g.cp[i] = selu(me == i, mine, g.cp[i]);
    fl += spill;                                                       // :154 / :159
    fl = fl < 7u ? fl : 7u;
    pset(g, PQ_FLOOR, me, fl);
    if (LID) byte_add(g.lid, c, spill);                                // :156-157 / :160-161
}

// ---- is_end_of_game: azul.py:184-191 ----
template <u32 P>
AZ_FN bool walls_end_game_np(const GameN<P> &g)
{
    bool over = false;
    for (u32 p = 0; p < P; p++) over = over | any_row_full(pget(g, PQ_WALL, p));
    return over;
}

// ---- count_score: azul.py:291-295, player by player ----
template <bool LID, u32 P>
AZ_FN void count_score_np(GameN<P> &g, const LaneConst &k)
{
    vu32 l = lane();
    for (u32 p = 0; p < P; p++) {
        u32 wall = pget(g, PQ_WALL, p);
        u32 F = (u32)ballot((g.cp[p] == k.rowp1) & (l < 25u)) & 0x1ffffffu;     // azul.py:216: the full lines
        ScoreVec sv;
        score_boards(splat(wall) | (splat(F) & k.pbelow), k, sv);                // :219: placements in ascending (row, colour) order
        i32 cnt = 0;
        u32 mc = pget(g, PQ_MAXC, p);
        u32 rest = F;
        while (rest) {
            u32 i = ctz32(rest);
            rest &= rest - 1u;
            cnt += (i32)readlane(sv.val, i);                                     // :289
            u32 ps = readlane(sv.pos, i);
            if (ps > mc) mc = ps;                                                // :264
            if (LID) { u32 r = (i * 205u) >> 10; byte_add(g.lid, i - 5u * r, r); }   // :220-222
        }
        pset(g, PQ_WALL, p, wall | F);                                           // :219
        pset(g, PQ_MAXC, p, mc);
        u32 cl = pget(g, PQ_COMPL, p);
        cl += popc64(sv.rowdone & (u64)F) + (popc64(sv.colordone & (u64)F) << 8) + (popc64(sv.coldone & (u64)F) << 16);   // :270,:278,:286
        pset(g, PQ_COMPL, p, cl);
        i32 pen = floor_penalty(pget(g, PQ_FLOOR, p));                           // count_floor, :200-210
        pset(g, PQ_FPEN, p, (u32)((i32)pget(g, PQ_FPEN, p) + pen));              // :208
        pset(g, PQ_FLOOR, p, 0u);                                                // :209
        pset(g, PQ_SCORE, p, (u32)clamp0((i32)pget(g, PQ_SCORE, p) + pen + cnt));   // :292-295
        g.cp[p] = sel((g.cp[p] == k.rowp1) & (l < 25u), splat(0u), g.cp[p]);     // :218
    }
}

// ---- new_round: azul.py:64-89 ----
template <bool LID, u32 P>
AZ_FN u32 new_round_np(GameN<P> &g, Rng &r)
{
    g.cur = g.nfp;
    u32 who = g.nfp == 0u ? P - 1u : g.nfp - 1u;                       // :67 (numpy [-1] == the last player when nfp == 0)
    pset(g, PQ_FPS, who, pget(g, PQ_FPS, who) + 1u);
    g.turn += 1u;
    g.nfp = 0;
    return deal_factories<LID>(g.cs, g.box, g.lid, r);
}

// ---- Azul.__init__: azul.py:18-61 ----
template <bool LID, u32 P>
AZ_FN void game_ctor_np(GameN<P> &g, u32 first_player, Rng &r)
{
    g.cs = splat(0u);
    for (u32 p = 0; p < P; p++) g.cp[p] = splat(0u);
    g.pv = splat(0u);
    g.cur = 0; g.eog = 0; g.turn = 0;
    // random.choice(list(range(1, P+1))) (:37) == _randbelow(P): rejection on getrandbits(P.bit_length())
    if (first_player == 0u) g.nfp = 1u + rng_below(r, P, P == 4u ? 3u : 2u);
    else g.nfp = first_player;
    if (LID) { g.box = 0x1414141414ull; g.lid = 0; }                      // :51-52
    else { g.box = 0; g.lid = 0; }
}

// ---- step: azul.py:296-313 ----
template <bool LID, u32 P>
AZ_FN u32 checked_step_np(GameN<P> &g, const LaneConst &k, Rng &r, i32 a)
{
    if (g.eog) return ST_GAME_ENDED;                     // :298-299
    if (a < 0 || a >= 180) return ST_BAD_ACTION;
    Mask m;
    legal_mask_np(g, k, m);
    if (!mask_test(m, (u32)a)) return ST_ILLEGAL_MOVE;   // :301-302, state untouched
    do_move_np<LID>(g, action_code((u32)a));             // :304
    if (sources_board_np(g) == 0u) {                     // :306 (the token counts)
        count_score_np<LID>(g, k);                       // :307
        if (walls_end_game_np(g)) { g.eog = 1; return ST_OK; }   // :308-309
        return new_round_np<LID>(g, r);                  // :311
    }
    g.cur = (g.cur < P) ? g.cur + 1u : 1u;               // :313 next_player (:177-181)
    return ST_OK;
}

// ---- get_statistics: azul.py:314-315 (players 0 and 1, whatever P is) ----
template <u32 P>
AZ_FN double game_stat_np(const GameN<P> &g, u32 q)
{
    switch (q) {
    case 0: return (double)(i32)pget(g, PQ_SCORE, 0);
    case 1: return (double)(i32)pget(g, PQ_SCORE, 1);
    case 2: return (double)g.turn;
    case 3: {
        double sum = 0.0;
        for (u32 p = 0; p < P; p++) sum += (double)pget(g, PQ_FPS, p);   // first_player_stats.sum(): left to right
        return (double)pget(g, PQ_FPS, 0) / sum * 100;
    }
    case 4: return -(double)(i32)pget(g, PQ_FPEN, 0);
    case 5: return (double)pget(g, PQ_MAXC, 0);
    case 6: return (double)(pget(g, PQ_COMPL, 0) & 0xffu);
    case 7: return (double)((pget(g, PQ_COMPL, 0) >> 16) & 0xffu);
    case 8: return (double)((pget(g, PQ_COMPL, 0) >> 8) & 0xffu);
    default: return (i32)pget(g, PQ_SCORE, 0) > (i32)pget(g, PQ_SCORE, 1) ? 1.0 : 0.0;
    }
}

// ---- flat random-agent self-play for P players (row N4): the loop  mask -> RandomAgent -> Azul.step  on a register-resident game
// (game_runner.py:87-97 works on any mask, azul.py:296-313 is P-generic), a fresh Azul(players=P, rules) + new_round() whenever a
// game ends or nobody can move.  GameRunner's shaped reward is two-player (game_runner.py:50): the reward stream carries zeros.
// Same output conventions as selfplay_step (OUT 0 / 1 / 2); `rec` snapshots are 256-byte wide records.
template <bool LID, u32 P>
AZ_FN u32 restart_np(GameN<P> &g, u32 first_player, Rng &r)
{
    game_ctor_np<LID>(g, first_player, r);
    return new_round_np<LID>(g, r);
}

template <bool LID, u32 P, int OUT>
AZ_FN void selfplay_outputs_np(const GameN<P> &g, const OutV &ov, const OutS &os, i32 a, u32 dn)
{
    if (OUT == 1) {
        vu32 l = lane();
        vst_u32(ov.p32, sel(l == 1u, splat(0u), sel(l == 2u, splat(pack_move(a, dn, 0)), splat((u32)(a >= 0 ? a : -1)))));
        vst_u8(ov.p8, splat(dn));
    } else if (OUT == 2) {
        if (os.action) stu_i32(os.action, a >= 0 ? a : -1);
        if (os.reward) stu_i32(os.reward, 0);
        if (os.done) stu_u8(os.done, dn);
        if (os.packed) stu_i32((i32 *)os.packed, (i32)pack_move(a, dn, 0));
        if (os.rec) gamen_store(g, os.rec);
    }
}

template <bool LID, u32 P, int OUT>
AZ_FN u32 selfplay_step_np(GameN<P> &g, u32 first_player, const LaneConst &k, Rng &r, const SampleTab &T, const Counters &cnt,
                           const OutV &ov, const OutS &os)
{
    Mask m;
    legal_mask_np(g, k, m);
    if (OUT == 1) {
        vu32 l = lane();
        vst_u8(ov.pm, m.b0);
        vst_u8_at(ov.pm, 64, m.b1);
        vst_u8(ov.pm2, sel(l < 52u, m.b2, splat((u32)(m.m2 >> 51) & 1u)));
        vu32 lo = sel(l == 0u, splat((u32)m.m0), sel(l == 1u, splat((u32)m.m1), splat((u32)m.m2)));
        vu32 hi = sel(l == 0u, splat((u32)(m.m0 >> 32)), sel(l == 1u, splat((u32)(m.m1 >> 32)), splat((u32)(m.m2 >> 32))));
        vst_u64(ov.p64, lo, hi);
    } else if (OUT == 2) {
        if (os.mask) mask_write(m, os.mask);
        if (os.maskbits) mask_write_bits(m, os.maskbits);
    }
    u32 code = 0;
    i32 a = g.eog ? -2 : random_agent(m, r, T, k, code);
    if (AZ_UNLIKELY(a < 0)) {
        // nothing legal (hazard H3) or a finished game handed in: report, restart the slot
        AZ_LANE0(*cnt.stuck += 1u);
        selfplay_outputs_np<LID, P, OUT>(g, ov, os, -1, 2u);
        u32 st0 = restart_np<LID>(g, first_player, r);
        return st0 ? (0x100u | st0) : 2u;
    }
    do_move_np<LID>(g, code);                            // azul.py:304
    u32 st = ST_OK;
    if (sources_board_np(g) == 0u) {                     // :306 (the token counts)
        count_score_np<LID>(g, k);                       // :307
        if (walls_end_game_np(g)) g.eog = 1;             // :308-309
        else st = new_round_np<LID>(g, r);               // :311
    } else {
        g.cur = (g.cur < P) ? g.cur + 1u : 1u;           // :313 next_player
    }
    const u32 dn = g.eog ? 1u : 0u;
    selfplay_outputs_np<LID, P, OUT>(g, ov, os, a, dn);
    if (AZ_UNLIKELY(st != ST_OK)) return 0x100u | st;
    if (AZ_UNLIKELY(dn != 0u)) {
        for (u32 q = 0; q < 10u; q++) { double sv = game_stat_np(g, q); AZ_LANE0(cnt.stat_sum[q] += sv); }
        AZ_LANE0(*cnt.episodes += 1ull);
        st = restart_np<LID>(g, first_player, r);
        if (st) return 0x100u | st;
    }
    return dn;
}

} // namespace az
