// azul_rules_x.hpp -- the Azul rules for P = 2, 3, 4 players on D = 5 or 2 P + 1 factory displays, TWO GAMES PER 64-LANE WAVEFRONT
// (lanes 0..31 / 32..63, the mapping of azul_selfplay2.hpp), with the rule switches of row N4 of SURVEY.md 8f.  Included by
// azul_kernels.hip after azul_selfplay2.hpp, whose building blocks it uses (half ballots, LDS-crossbar gathers, DPP reductions, the
// MT19937 stream with its tempered copy, the wall pricing, the parallel factory draw).
//
// What is what:
//   * D = 5, every switch off: exactly what the reference does for Azul(players = P) -- per-player lines / walls / floors / scores for P
//     players on FIVE displays (azulnet/azul.py:19; TODO at tests/test_azul.py:14), turn order 1..P (azul.py:177-181), first-player draw
//     random.choice(range(1, P + 1)) (azul.py:37).  Pinned to the reference through tests/golden/traj_players*.npz.
//   * beyond the reference ("parity unpinned": no reference behaviour exists; the test infrastructure restates the published rulebook
//     twice, independently -- a plain-C restatement and a plain-Python model -- and the kernels are compared with the former):
//       D = 2 P + 1      the rulebook's 5 / 7 / 9 displays; action a = source + (D + 1) colour + 5 (D + 1) row, (D + 1) * 30 actions
//       end_bonus        +2 / +7 / +10 per complete row / column / colour paid ONCE at the end of the game (the reference pays them in the
//                        round the tile lands, under that round's clamp: azul.py:266-295)
//       short_deal       bag and lid both empty at a draw: the round starts with what could be dealt (the reference raises: azul.py:86-87)
//       pool BAG         tile_pool "Random" drawing WITHOUT replacement from a 100-tile bag (the TODO at azul.py:72): _randbelow(tiles left)
//
// Layout inside a half (l = lane & 31):
//   cs0       lane 5 d + c = displays[d][c] for d < 5, lane 25 + c = center[c], lane 30 = token          (az2's cs)
//   cs1       lane 5 (d - 5) + c = displays[d][c] for d >= 5                                              (only when D > 5)
//   cpk       lane 5 r + c = pattern_lines[p][r][c] of ALL players, one byte per player (byte p): the mover's cell is a bit field
//             at 8 * me -- no per-player registers to select between on the move's path
//   floors    half-uniform, byte p = floors[p] (the record's dword); okv lane p = player p's "row accepts colour" board
//   everything else half-uniform; the remaining per-player values are arrays indexed by compile-time constants (scoring walks the
//   players) -- no run-time indexed memory.  (Round 4's first version kept cp / floor / ok as per-player registers behind
//   v_cndmask chains on the mover's index: measured 10-15 % of the launch for three / four players.)
// Legal mask: per pattern row r, bit q = source + (D + 1) colour (< 5 (D + 1)) of a 30- / 40- / 50-bit row word; lane l owns bit l of the
// low 32 and (D > 5) bit 32 + l of the rest.
//
// Reference lines: __init__ azul.py:18-61, new_round :64-89, move :118-161, is_legal_move :162-176, next_player :177-181,
// is_end_of_round :182-183, is_end_of_game :184-191, count_score :192-295, step :296-313, get_statistics :314-315;
// RandomAgent game_runner.py:87-97, check_all_valid :113-117, get_state :56-72.
#pragma once
#include "azul_selfplay2.hpp"

namespace azx {
using namespace az;
using namespace az2;

enum { XPOOL_RANDOM = 0, XPOOL_LID = 1, XPOOL_BAG = 2 };

struct RulesX {
    u32 first_player;   // 0 = "Random", 1..P fixed
    u32 pool;           // XPOOL_*
    u32 end_bonus;      // line bonuses at the end of the game instead of per round
    u32 short_deal;     // partial deal instead of ST_BOX_EMPTY
};

template <u32 D>
struct Dim {
    static constexpr u32 S = D + 1;                 // sources: the centre + D displays
    static constexpr u32 Q = 5 * S;                 // actions per pattern row
    static constexpr u32 NA = 6 * Q;                // actions
    static constexpr u32 NW = Q > 32 ? 2 : 1;       // 32-bit mask words per pattern row
    static constexpr u32 NL = (NA + 63) / 64;       // 64-bit words of the bit-packed mask
    static constexpr u32 TROWS = Q + 1;             // rows of the sampling table (J = 0 .. Q legal floor moves)
    static constexpr bool WIDE = D > 5;
    static constexpr u32 XCELLS = 5 * (D - 5);      // cells of cs1
};
template <u32 P, u32 D> constexpr u32 obs_size() { return 5 * D + 6 + 52 * P + 1; }          // game_runner.py:65-72

// ---- per-lane constants --------------------------------------------------------------------------------------------------------
template <u32 D>
struct KX {
    K2 k;                            // pattern-cell constants (rowp1, prow, pcol, pbcol, pbelow, pcolboard), l, h4
    u32 spos[Dim<D>::NW];            // word w, bit l: lane (in cs0 or cs1) of that action's source cell; 31 = no action (bit 31 of a source board is 0)
    u32 sreg[Dim<D>::NW];            // ... and whether it lives in cs1
    u32 col[Dim<D>::NW];             // ... and its colour
};

template <u32 D>
AZ_FN void kx_init(KX<D> &K)
{
    k2_init(K.k);
    const u32 l = K.k.l;
#pragma unroll
    for (u32 w = 0; w < Dim<D>::NW; w++) {
        const u32 q = 32u * w + l;
        const bool valid = q < Dim<D>::Q;
        const u32 s = valid ? q % Dim<D>::S : 0u, c = valid ? q / Dim<D>::S : 0u;
        const u32 d = s ? s - 1u : 0u;
        const u32 cell = s == 0u ? 25u + c : (d < 5u ? 5u * d + c : 5u * (d - 5u) + c);
        K.spos[w] = valid ? cell : 31u;
        K.sreg[w] = (valid && s > 5u) ? 1u : 0u;
        K.col[w] = c;
    }
}

// ---- game state ----------------------------------------------------------------------------------------------------------------
template <u32 P, u32 D>
struct GX {
    u32 cs0, cs1;
    u32 cpk;                    // lane 5 r + c: byte p = pattern_lines[p][r][c]
    u32 floors;                 // half-uniform: byte p = floors[p]
    u32 wall[P];
    i32 score[P];
    u32 okv;                    // derived: lane p = player p's "row r accepts colour c" board (az2::ok_board2)
    u32 fps[P];                 // first_player_stats
    i32 fpen[P];                // floor_penalty (16 bits stored)
    u32 mc[P], cl[P];           // max_combo (8 bits), completed_lines (3 x 8 bits)
    u32 cur, nfp, eog, turn;
    u64 box, lid;
    u32 lidp;                   // per lane: tiles of colour l scoring has returned to the lid since the last fold (az2::lid_fold2)
    u32 over;                   // derived: some wall row is complete
    u32 B0, B1;                 // derived: sources holding tiles, hb(cs0 != 0) & 0x7fffffff / hb(cs1 != 0)
};

// a[i] for a half-uniform i, as a chain of v_cndmask.  (The elements pass through an empty asm first: a select between two LOADS of a
// struct that is reached through a reference is folded into one load from a selected ADDRESS before the struct is split into registers,
// which pins the whole game state in scratch memory -- see azul_selfplay2.hpp.)
template <u32 P, typename T>
AZ_FN T pick(const T (&a)[P], u32 i)
{
    T t[P];
#pragma unroll
    for (u32 p = 0; p < P; p++) { t[p] = a[p]; asm volatile("" : "+v"(t[p])); }
    T v = t[0];
#pragma unroll
    for (u32 p = 1; p < P; p++) v = i == p ? t[p] : v;
    return v;
}
template <u32 P, typename T>
AZ_FN void put(T (&a)[P], u32 i, T v)
{
#pragma unroll
    for (u32 p = 0; p < P; p++) a[p] = i == p ? v : a[p];
}

template <u32 P, u32 D> AZ_FN u32 mex(const GX<P, D> &g) { return g.cur == 0u ? P - 1u : g.cur - 1u; }     // numpy [-1] before the first round

template <u32 P, u32 D>
AZ_FN void sources_x(GX<P, D> &g)
{
    g.B0 = hb(g.cs0 != 0u) & 0x7fffffffu;
    g.B1 = Dim<D>::WIDE ? hb(g.cs1 != 0u) & ((1u << Dim<D>::XCELLS) - 1u) : 0u;
}

template <u32 P, u32 D>
AZ_FN void prime_x(GX<P, D> &g, const KX<D> &K)
{
    bool over = false;
#pragma unroll
    for (u32 p = 0; p < P; p++) {
        over = over | any_row_full(g.wall[p]);
        const u32 o = ok_board2((g.cpk >> (8u * p)) & 0xffu, g.wall[p], K.k);
        g.okv = K.k.l == p ? o : g.okv;
    }
    g.over = over ? 1u : 0u;
    sources_x(g);
}

// ---- 256-byte wide record <-> registers (layout: include/azul_hip.h) ----------------------------------------------------------------
template <u32 P, u32 D>
AZ_FN void gx_load(GX<P, D> &g, const uint8_t *rec, u32 l)
{
    const u32 a = rec[l];
    const u32 flags = hread(a, 31);
    g.cur = flags & 7u; g.nfp = (flags >> 3) & 7u; g.eog = (flags >> 6) & 1u;
    g.cs0 = l < 31u ? a : 0u;
    g.cs1 = 0u;
    if (Dim<D>::WIDE) g.cs1 = l < Dim<D>::XCELLS ? (u32)rec[208u + l] : 0u;
    g.cpk = 0;
#pragma unroll
    for (u32 p = 0; p < P; p++) g.cpk |= (l < 25u ? (u32)rec[32u + 25u * p + l] : 0u) << (8u * p);
    const u32 t = l < 19u ? ((const u32 *)(rec + 132))[l] : 0u;          // bytes 132 .. 207
    const u32 fl = hread(t, 0), sa = hread(t, 5), sb = hread(t, 6), b7 = hread(t, 7), b8 = hread(t, 8), b9 = hread(t, 9);
    const u32 fa = hread(t, 10), fb = hread(t, 11), pa = hread(t, 12), pb = hread(t, 13), mc = hread(t, 14);
    const u32 c0 = hread(t, 15), c1 = hread(t, 16), c2 = hread(t, 17);
    const u32 wl[4] = {hread(t, 1), hread(t, 2), hread(t, 3), hread(t, 4)};
    const u32 sc[4] = {sa & 0xffffu, sa >> 16, sb & 0xffffu, sb >> 16};
    const u32 fs[4] = {fa & 0xffffu, fa >> 16, fb & 0xffffu, fb >> 16};
    const u32 pn[4] = {pa & 0xffffu, pa >> 16, pb & 0xffffu, pb >> 16};
    const u32 cl[4] = {c0 & 0xffffffu, (c0 >> 24) | ((c1 & 0xffffu) << 8), (c1 >> 16) | ((c2 & 0xffu) << 16), c2 >> 8};
#pragma unroll
    for (u32 p = 0; p < P; p++) {
        g.wall[p] = wl[p];
        g.score[p] = (i32)(int16_t)sc[p];
        g.fps[p] = fs[p];
        g.fpen[p] = (i32)(int16_t)pn[p];
        g.mc[p] = (mc >> (8u * p)) & 0xffu;
        g.cl[p] = cl[p];
    }
    g.floors = P == 4u ? fl : fl & ((1u << (8u * (P & 3u))) - 1u);
    g.box = (u64)b7 | ((u64)(b8 & 0xffu) << 32);
    g.lid = (u64)(b8 >> 8) | ((u64)(b9 & 0xffffu) << 24);
    g.turn = b9 >> 16;
    g.lidp = 0;
    g.over = 0; g.B0 = g.B1 = 0;
    g.okv = 0;
}

template <u32 P, u32 D>
AZ_FN void gx_store(const GX<P, D> &g, uint8_t *rec, u32 l)
{
    const u32 flags = (g.cur & 7u) | ((g.nfp & 7u) << 3) | ((g.eog & 1u) << 6);
    rec[l] = (uint8_t)(l == 31u ? flags : g.cs0);
    if (Dim<D>::WIDE) { if (l < Dim<D>::XCELLS) rec[208u + l] = (uint8_t)g.cs1; }
#pragma unroll
    for (u32 p = 0; p < P; p++) { if (l < 25u) rec[32u + 25u * p + l] = (uint8_t)(g.cpk >> (8u * p)); }
    const u64 lid = g.lid + lid_fold2(g.lidp, l);
    const u32 fl = g.floors;
    u32 mc = 0;
    u32 sc[4] = {0, 0, 0, 0}, fs[4] = {0, 0, 0, 0}, pn[4] = {0, 0, 0, 0}, cl[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
    for (u32 p = 0; p < P; p++) {
        mc |= (g.mc[p] & 0xffu) << (8u * p);
        wl[p] = g.wall[p];
        sc[p] = (u32)g.score[p] & 0xffffu;
        fs[p] = g.fps[p] & 0xffffu;
        pn[p] = (u32)g.fpen[p] & 0xffffu;
        cl[p] = g.cl[p] & 0xffffffu;
    }
    u32 t = 0;
    t = l == 0u ? fl : t;
    t = l == 1u ? wl[0] : t; t = l == 2u ? wl[1] : t; t = l == 3u ? wl[2] : t; t = l == 4u ? wl[3] : t;
    t = l == 5u ? (sc[0] | (sc[1] << 16)) : t;
    t = l == 6u ? (sc[2] | (sc[3] << 16)) : t;
    t = l == 7u ? (u32)g.box : t;
    t = l == 8u ? ((u32)((g.box >> 32) & 0xffu) | ((u32)lid << 8)) : t;
    t = l == 9u ? ((u32)((lid >> 24) & 0xffffu) | (g.turn << 16)) : t;
    t = l == 10u ? (fs[0] | (fs[1] << 16)) : t;
    t = l == 11u ? (fs[2] | (fs[3] << 16)) : t;
    t = l == 12u ? (pn[0] | (pn[1] << 16)) : t;
    t = l == 13u ? (pn[2] | (pn[3] << 16)) : t;
    t = l == 14u ? mc : t;
    t = l == 15u ? (cl[0] | (cl[1] << 24)) : t;
    t = l == 16u ? ((cl[1] >> 8) | (cl[2] << 16)) : t;
    t = l == 17u ? ((cl[2] >> 16) | (cl[3] << 8)) : t;
    t = l == 18u ? (P | (D != 5u ? D << 8 : 0u)) : t;
    if (l < 19u) ((u32 *)(rec + 132))[l] = t;
}

// ---- legal-move mask: azul.py:162-176 over all (D + 1) * 30 actions (game_runner.py:113-117) -------------------------------------
template <u32 D>
struct MaskX {
    u32 m[6][Dim<D>::NW];       // half-uniform: bit l of word w of row r = action Q r + 32 w + l is legal
    u32 bit[6][Dim<D>::NW];     // the same bit for MY action of each word
};

template <u32 P, u32 D>
AZ_FN void legal_mask_x(const GX<P, D> &g, const KX<D> &K, MaskX<D> &out)
{
    const u32 okm = hread4(g.okv, mex(g), K.k.h4);
#pragma unroll
    for (u32 w = 0; w < Dim<D>::NW; w++) {
        const u32 B = (Dim<D>::WIDE && K.sreg[w]) ? g.B1 : g.B0;
        const u32 has = (B >> K.spos[w]) & 1u;                       // my source holds tiles of my colour (azul.py:164-169)
        const u32 okc = okm >> K.col[w];                             // bit 5 (r - 1): pattern row r accepts my colour (:171-175)
        out.bit[0][w] = has;                                         // the floor "row" accepts everything
        out.m[0][w] = hb(has != 0u);
#pragma unroll
        for (u32 r = 1; r < 6u; r++) {
            out.bit[r][w] = has & (okc >> (5u * (r - 1u)));
            out.m[r][w] = hb((out.bit[r][w] & 1u) != 0u);
        }
    }
}

template <u32 D>
AZ_FN u32 row_count(const MaskX<D> &m, u32 r)
{
    u32 c = (u32)__popc(m.m[r][0]);
    if (Dim<D>::NW > 1) c += (u32)__popc(m.m[r][Dim<D>::NW - 1]);
    return c;
}

// the bit-packed mask (bit a & 63 of word a >> 6): the six Q-bit row words back to back
template <u32 D>
AZ_FN void mask_limbs(const MaskX<D> &m, u64 (&limb)[Dim<D>::NL + 1])
{
#pragma unroll
    for (u32 i = 0; i <= Dim<D>::NL; i++) limb[i] = 0;
#pragma unroll
    for (u32 r = 0; r < 6u; r++) {
        u64 R = m.m[r][0];
        if (Dim<D>::NW > 1) R |= (u64)m.m[r][Dim<D>::NW - 1] << 32;
        const u32 pos = r * Dim<D>::Q, i = pos >> 6, sh = pos & 63u;
        limb[i] |= R << sh;
        if (sh + Dim<D>::Q > 64u) limb[i + 1] |= R >> (64u - sh);
    }
}

// ---- RandomAgent (game_runner.py:87-97 + random.choices): azul_selfplay2.hpp's table (Tab2: rows of nine pairs at 9 J + b, here for
// J = 0 .. Q legal floor moves) and its boundary search (sample_slow2), unchanged ------------------------------------------------------
// one decision: counts of the legal actions, the ordinal from random(), the chosen action (row, source, colour)
template <u32 D>
struct Choice { i32 a; u32 row, s, c; };

template <u32 D>
AZ_FN void pick_action_x(const MaskX<D> &m, const u32 (&pre)[7], u32 kg, const K2 &k, Choice<D> &ch)
{
    // kg-th legal action: its pattern row from the prefix counts, then (D > 5) its word, then ONE rank test per lane
    const u32 l = k.l;
    const u32 want = kg - 1u;
    u32 row = 0;
#pragma unroll
    for (u32 r = 1; r < 6u; r++) row += want >= pre[r] ? 1u : 0u;
    // (the words and prefix counts pass through an empty asm first: a select chain over elements of a struct reached through a reference is
    // folded into ONE load from a selected address, which pins the struct in scratch memory)
    u32 lo[6], hi[6], pr[6];
#pragma unroll
    for (u32 r = 0; r < 6u; r++) {
        lo[r] = m.m[r][0]; hi[r] = m.m[r][Dim<D>::NW - 1]; pr[r] = pre[r];
        asm volatile("" : "+v"(lo[r]), "+v"(hi[r]), "+v"(pr[r]));
    }
    u32 w0 = lo[0], w1 = hi[0], base = 0;
#pragma unroll
    for (u32 r = 1; r < 6u; r++) {
        const bool ge = want >= pr[r];
        w0 = ge ? lo[r] : w0;
        if (Dim<D>::NW > 1) w1 = ge ? hi[r] : w1;
        base = ge ? pr[r] : base;
    }
    u32 rank = want - base, word = w0, hiw = 0;
    if (Dim<D>::NW > 1) {
        const u32 c0 = (u32)__popc(w0);
        hiw = rank >= c0 ? 1u : 0u;
        word = hiw ? w1 : w0;
        rank = hiw ? rank - c0 : rank;
    }
    const bool hit = (((u32)__popc(word & ((1u << l) - 1u)) - rank) << 1) + ((word >> l) & 1u) == 1u;
    const u32 who = hb(hit);
    const u32 q = (u32)__builtin_ctz(who | 0x80000000u) + 32u * hiw;
    ch.row = row;
    ch.c = q / Dim<D>::S;
    ch.s = q - ch.c * Dim<D>::S;
    ch.a = (i32)(Dim<D>::Q * row + q);
}

// ---- move: azul.py:118-161 ---------------------------------------------------------------------------------------------------------
template <u32 P, u32 D>
AZ_FN void do_move_x(GX<P, D> &g, u32 s, u32 c, u32 row, bool tracked, const KX<D> &K)
{
    const u32 l = K.k.l, h4 = K.k.h4;
    const u32 me = mex(g);
    const bool from_display = s != 0u;
    const u32 d = from_display ? s - 1u : 0u;
    const bool hi = Dim<D>::WIDE && d >= 5u;                           // the display lives in cs1
    const u32 db = 5u * (hi ? d - 5u : d);
    const u32 src = from_display ? db + c : 25u + c;
    const u32 dreg = hi ? g.cs1 : g.cs0;
    const u32 cell = 5u * ((row ? row : 1u) - 1u) + c;
    const u32 sh = 8u * me;                                            // the mover's byte of cpk / floors
    const u32 n = hread4(dreg, src, h4);                               // :127 / :136
    const u32 moved = hread4(dreg, l - 25u + db, h4);                  // :131 the rest of the display slides into the centre
    const u32 old = hread4((g.cpk >> sh) & 0xffu, cell, h4);
    u32 okm = hread4(g.okv, me, h4);
    const bool token = (!from_display) & (((g.B0 >> 30) & 1u) != 0u);  // :140
    const bool centre = (l >= 25u) & (l < 30u) & (l != 25u + c) & from_display;
    const bool disp = (l >= db) & (l < db + 5u) & from_display;
    const bool gone0 = (disp & !hi) | ((l == src) & !from_display) | ((l == 30u) & token);     // :129,:133,:138,:141
    const u32 grown = g.cs0 + (centre ? moved : 0u);
    g.cs0 = gone0 ? 0u : grown;
    if (Dim<D>::WIDE) g.cs1 = (disp & hi) ? 0u : g.cs1;
    g.nfp = token ? g.cur : g.nfp;                                     // :142
    u32 fl = ((g.floors >> sh) & 0xffu) + (token ? 1u : 0u);           // :143 (the cap of :120-123 is applied once, below: it is monotone)
    const i32 overflow = row ? (i32)row - (i32)old - (i32)n : -(i32)n; // :147
    const u32 spill = overflow < 0 ? (u32)(-overflow) : 0u;
    const u32 newv = overflow < 0 ? row : old + n;                     // :150 / :152
    g.cpk = ((l == cell) & (row != 0u)) ? (g.cpk & ~(0xffu << sh)) | (newv << sh) : g.cpk;
    {   // the row now holds colour c only (the move was legal: the row was empty or held c, the wall cell is free)
        const u32 rs = 5u * ((row ? row : 1u) - 1u);
        okm = row ? ((okm & ~(31u << rs)) | (1u << (rs + c))) : okm;
        g.okv = l == me ? okm : g.okv;
    }
    fl += spill;                                                       // :154 / :159
    fl = fl < 7u ? fl : 7u;
    g.floors = (g.floors & ~(0xffu << sh)) | (fl << sh);
    g.lid += tracked ? (u64)spill << (8u * c) : 0ull;                  // :156-157 / :160-161
}

// ---- count_score: azul.py:291-295, player by player (az2::count_player2 with the bonus switch) ------------------------------------
template <u32 P, u32 D>
AZ_FN void count_score_x(GX<P, D> &g, bool tracked, bool end_bonus, const KX<D> &K)
{
    const K2 &k = K.k;
    bool over = false;
    // (the two rule switches as per-lane masks behind an empty asm: as uniform conditions they become scalar branches between the
    // players' blocks, and the P independent chains are only interleaved by the scheduler inside ONE basic block)
    u32 tmask = tracked ? ~0u : 0u, bmask = end_bonus ? 0u : ~0u;
    asm volatile("" : "+v"(tmask), "+v"(bmask));
#pragma unroll
    for (u32 p = 0; p < P; p++) {
        const u32 cells = (g.cpk >> (8u * p)) & 0xffu;
        const u32 F = full_lines2(cells, k);                                               // :216
        Score2 s;
        score2(g.wall[p] | (F & k.pbelow), k, s);                                          // :219: placements in ascending (row, colour) order
        const bool on = ((F >> k.l) & 1u) != 0u;
        // beyond the reference (end_bonus): the +2 / +10 / +7 of :266-288 are not part of the round's count
        const i32 cnt = (i32)hsum(on ? s.pos + ((s.val - s.pos) & bmask) : 0u);            // :289
        g.mc[p] = umax(g.mc[p], hmax(on ? s.pos : 0u));                                    // :264
        g.cl[p] += (u32)__popc(s.rowdone & F) + ((u32)__popc(s.colordone & F) << 8) + ((u32)__popc(s.coldone & F) << 16);   // :270,:278,:286
        g.lidp += lid_tally2(F, k.l) & tmask;                                              // :220-222
        g.wall[p] |= F;
        const u32 left = (cells == k.rowp1) ? 0u : cells;                                  // :218
        g.cpk = (g.cpk & ~(0xffu << (8u * p))) | (left << (8u * p));
        const i32 pen = floor_penalty((g.floors >> (8u * p)) & 0xffu);
        g.fpen[p] += pen;                                                                  // :208
        g.score[p] = clamp0(g.score[p] + pen + cnt);                                       // :292-295
        over = over | any_row_full(g.wall[p]);
        const u32 o = ok_board2(left, g.wall[p], k);
        g.okv = k.l == p ? o : g.okv;
    }
    g.floors = 0;                                                                          // :209
    g.over = over ? 1u : 0u;
}

// beyond the reference: the rulebook's end-of-game bonuses from the final walls -- 2 per complete row, 7 per complete board column,
// 10 per colour with all five tiles placed -- added once, after the last round's clamp
AZ_FN i32 wall_bonus(u32 w)
{
    const u32 rows = (u32)__popc(w & (w >> 1) & (w >> 2) & (w >> 3) & (w >> 4) & 0x108421u);
    const u32 colours = (u32)__popc(w & (w >> 5) & (w >> 10) & (w >> 15) & (w >> 20) & 31u);
    u32 cols = 0;
    cols += (w & column_board_c(0)) == column_board_c(0) ? 1u : 0u;
    cols += (w & column_board_c(1)) == column_board_c(1) ? 1u : 0u;
    cols += (w & column_board_c(2)) == column_board_c(2) ? 1u : 0u;
    cols += (w & column_board_c(3)) == column_board_c(3) ? 1u : 0u;
    cols += (w & column_board_c(4)) == column_board_c(4) ? 1u : 0u;
    return (i32)(2u * rows + 7u * cols + 10u * colours);
}

template <u32 P, u32 D>
AZ_FN void end_game_bonus_x(GX<P, D> &g)
{
#pragma unroll
    for (u32 p = 0; p < P; p++) g.score[p] += wall_bonus(g.wall[p]);
}

// ---- new_round: azul.py:64-89 --------------------------------------------------------------------------------------------------------
// one tile onto display d (cells of d < 5 in cs0, the others in cs1)
template <u32 D>
AZ_FN void add_tile(u32 &cs0, u32 &cs1, u32 d, u32 colour, u32 l)
{
    const bool hi = Dim<D>::WIDE && d >= 5u;
    const u32 cell = 5u * (hi ? d - 5u : d) + colour;
    cs0 += ((l == cell) & !hi) ? 1u : 0u;
    if (Dim<D>::WIDE) cs1 += ((l == cell) & hi) ? 1u : 0u;
}

// the draws of a round one after the other, for the pools whose draws consume a data-dependent number of words: "Random"
// (_randbelow(5) per tile, azul.py:78) and -- beyond the reference -- the finite bag (_randbelow(tiles left) per tile)
template <u32 D>
AZ_FN u32 deal_serial_x(u32 &cs0, u32 &cs1, u64 &box, u64 &lid, u32 &lidp, u32 pool, bool short_deal, Rng2 &r, u32 l)
{
    if (pool == (u32)XPOOL_RANDOM) {
#pragma unroll 1
        for (u32 t = 0; t < 4u * D; t++) add_tile<D>(cs0, cs1, t >> 2, rng2_below(r, 5u, 3u, l), l);        // :78 randrange(0,5,1)
        return ST_OK;
    }
    u64 Pp = ((box & 0xffffffffffull) * 0x0101010101ull) & 0xffffffffffull;     // byte c = box_0 + .. + box_c
#pragma unroll 1
    for (u32 t = 0; t < 4u * D; t++) {
        u32 total = (u32)(Pp >> 32) & 0xffu;
        if (total == 0u) {                                                           // the bag is empty: refill it from the lid (like :81-83)
            box = lid + lid_fold2(lidp, l); lid = 0; lidp = 0;
            Pp = ((box & 0xffffffffffull) * 0x0101010101ull) & 0xffffffffffull;
            total = (u32)(Pp >> 32) & 0xffu;
            if (total == 0u) return short_deal ? (u32)ST_OK : (u32)ST_BOX_EMPTY;     // beyond the reference: the short deal
        }
        // the nth of the `total` tiles left, tiles ordered by colour; random.randrange(total) = _randbelow(total)
        const u32 nth = rng2_below(r, total, 32u - (u32)__builtin_clz(total), l);
        const u32 pc = ((u32)Pp >> ((l & 3u) * 8u)) & 0xffu;
        const u32 color = (u32)__popc(hb((pc <= nth) & (l < 4u)));
        box -= 1ull << (8u * color);
        Pp -= (0x0101010101ull << (8u * color)) & 0xffffffffffull;
        add_tile<D>(cs0, cs1, t >> 2, color, l);
    }
    return ST_OK;
}

template <u32 P, u32 D>
AZ_FN u32 new_round_x(GX<P, D> &g, const RulesX &rules, Rng2 &r, u64 margin, const KX<D> &K)
{
    const u32 l = K.k.l;
    g.cur = g.nfp;
    {
        const u32 who = g.nfp == 0u ? P - 1u : g.nfp - 1u;            // :67 (numpy [-1] == the last player when nfp == 0)
        put<P>(g.fps, who, pick<P>(g.fps, who) + 1u);
    }
    g.turn += 1u;
    g.nfp = 0;
    g.cs0 = l == 30u ? 1u : 0u;                                       // :71,:73
    g.cs1 = 0u;
    u32 st = ST_OK;
    if (rules.pool == (u32)XPOOL_LID) {
        st = deal_lid2<4u * D, Dim<D>::XCELLS>(g.cs0, g.cs1, g.box, g.lid, g.lidp, r, margin, rules.short_deal != 0u, K.k);
    } else {
        st = deal_serial_x<D>(g.cs0, g.cs1, g.box, g.lid, g.lidp, rules.pool, rules.short_deal != 0u, r, l);
    }
    sources_x(g);
    return st;
}

// ---- Azul.__init__: azul.py:18-61 ------------------------------------------------------------------------------------------------------
template <u32 P, u32 D>
AZ_FN void game_ctor_x(GX<P, D> &g, const RulesX &rules, Rng2 &r, const KX<D> &K)
{
    g.cs0 = 0; g.cs1 = 0;
#pragma unroll
    for (u32 p = 0; p < P; p++) { g.wall[p] = 0; g.score[p] = 0; g.fps[p] = 0; g.fpen[p] = 0; g.mc[p] = 0; g.cl[p] = 0; }
    g.cpk = 0; g.floors = 0;
    g.okv = 0x1ffffffu;                                               // empty lines, empty walls: every row accepts every colour
    g.cur = 0; g.eog = 0; g.turn = 0; g.over = 0; g.B0 = g.B1 = 0;
    // random.choice(list(range(1, P + 1))) (:37) == _randbelow(P): rejection on getrandbits(P.bit_length())
    if (rules.first_player == 0u) g.nfp = 1u + rng2_below(r, P, P == 4u ? 3u : 2u, K.k.l);
    else g.nfp = rules.first_player;
    g.box = rules.pool != (u32)XPOOL_RANDOM ? 0x1414141414ull : 0ull;   // :51-52 (and the finite bag's 100 tiles)
    g.lid = 0; g.lidp = 0;
}

template <u32 P, u32 D>
AZ_FN u32 restart_x(GX<P, D> &g, const RulesX &rules, Rng2 &r, u64 margin, const KX<D> &K)
{
    game_ctor_x(g, rules, r, K);
    return new_round_x(g, rules, r, margin, K);
}

// what follows a move (azul.py:305-313): end of round -> scoring -> end of game (+ the final bonus) or the next round; else the next player
template <u32 P, u32 D>
AZ_FN u32 after_move_x(GX<P, D> &g, const RulesX &rules, Rng2 &r, u64 margin, const KX<D> &K)
{
    u32 st = ST_OK;
    if ((g.B0 | g.B1) == 0u) {                            // :306 (the token counts)
        count_score_x(g, rules.pool != (u32)XPOOL_RANDOM, rules.end_bonus != 0u, K);      // :307
        if (g.over) {                                     // :308-309
            g.eog = 1;
            if (rules.end_bonus) end_game_bonus_x(g);
        } else {
            st = new_round_x(g, rules, r, margin, K);     // :311
        }
    } else {
        g.cur = (g.cur < P) ? g.cur + 1u : 1u;            // :313 next_player (:177-181)
    }
    return st;
}

template <u32 D>
AZ_FN bool mask_test_x(const MaskX<D> &m, u32 a, u32 l)
{
    // half-uniform action a: bit q of row r, fetched from the lane that owns it
    const u32 r = a / Dim<D>::Q, q = a - r * Dim<D>::Q;
    u32 w = 0;
#pragma unroll
    for (u32 rr = 0; rr < 6u; rr++) {
        u32 lo = m.m[rr][0], hi = m.m[rr][Dim<D>::NW - 1];
        asm volatile("" : "+v"(lo), "+v"(hi));
        const u32 word = (Dim<D>::NW > 1 && q >= 32u) ? hi : lo;
        w = r == rr ? word : w;
    }
    (void)l;
    return ((w >> (q & 31u)) & 1u) != 0u;
}

// ---- step: azul.py:296-313 ---------------------------------------------------------------------------------------------------------------
template <u32 P, u32 D>
AZ_FN u32 checked_step_x(GX<P, D> &g, const RulesX &rules, const KX<D> &K, Rng2 &r, u64 margin, i32 a)
{
    if (g.eog) return ST_GAME_ENDED;                     // :298-299
    if (a < 0 || a >= (i32)Dim<D>::NA) return ST_BAD_ACTION;
    MaskX<D> m;
    legal_mask_x(g, K, m);
    if (!mask_test_x<D>(m, (u32)a, K.k.l)) return ST_ILLEGAL_MOVE;   // :301-302, state untouched
    const u32 row = (u32)a / Dim<D>::Q, q = (u32)a - row * Dim<D>::Q, c = q / Dim<D>::S, s = q - c * Dim<D>::S;
    do_move_x(g, s, c, row, rules.pool != (u32)XPOOL_RANDOM, K);     // :304
    sources_x(g);
    return after_move_x(g, rules, r, margin, K);
}

// ---- get_statistics: azul.py:314-315 (players 0 and 1, whatever P is) ----------------------------------------------------------------------
template <u32 P, u32 D>
AZ_FN double game_stat_x(const GX<P, D> &g, u32 q)
{
    switch (q) {
    case 0: return (double)g.score[0];
    case 1: return (double)g.score[1];
    case 2: return (double)g.turn;
    case 3: {
        double sum = 0.0;
        for (u32 p = 0; p < P; p++) sum += (double)g.fps[p];         // first_player_stats.sum(): left to right
        return (double)g.fps[0] / sum * 100;
    }
    case 4: return -(double)(i32)(int16_t)((u32)g.fpen[0] & 0xffffu);
    case 5: return (double)(g.mc[0] & 0xffu);
    case 6: return (double)(g.cl[0] & 0xffu);
    case 7: return (double)((g.cl[0] >> 16) & 0xffu);
    case 8: return (double)((g.cl[0] >> 8) & 0xffu);
    default: return g.score[0] > g.score[1] ? 1.0 : 0.0;
    }
}

// ---- observation: game_runner.py:56-72 for P players on D displays (5 D + 6 + 52 P + 1 integers, written as f32) ---------------------------
template <u32 P, u32 D>
AZ_FN void observe_x(const GX<P, D> &g, u32 persp, float *out, u32 l)
{
    // order = [perspective] + the other players ascending (:57); perspective-relative next first player (:58-61)
    const u32 pnfp = g.nfp > 0u ? ((g.nfp - 1u + P - persp) % P) + 1u : 0u;
    constexpr u32 N = obs_size<P, D>(), A = 5u * D, Bc = A + 6u, Cp = Bc + 25u * P, Dw = Cp + 25u * P, Ef = Dw + P, Fs = Ef + P;
#pragma unroll 1
    for (u32 base = 0; base < N; base += 32u) {
        const u32 j = base + l;
        // every gather is executed by all lanes (clamped indices); the index class selects the value afterwards
        const u32 cell = j < A ? j : 0u;
        const u32 d0 = hread(g.cs0, cell < 25u ? cell : 0u);
        u32 d1 = 0;
        if (Dim<D>::WIDE) d1 = hread(g.cs1, cell >= 25u ? cell - 25u : 0u);
        const u32 v_disp = cell < 25u ? d0 : d1;
        const u32 v_cen = hread(g.cs0, 25u + (j >= A && j < Bc ? j - A : 0u));
        const u32 pi = j >= Bc && j < Cp ? j - Bc : 0u;                                      // pattern: order index * 25 + cell
        const u32 wi = j >= Cp && j < Dw ? j - Cp : 0u;
        const u32 po = pi / 25u, pc = pi - 25u * po, wo = wi / 25u, wc = wi - 25u * wo;
        const u32 fo = j >= Dw && j < Ef ? j - Dw : (j >= Ef && j < Fs ? j - Ef : 0u);
        // order[i] = persp for i == 0, else the (i - 1)-th of the other players
        const u32 pp = po == 0u ? persp : (po - 1u < persp ? po - 1u : po);
        const u32 wp = wo == 0u ? persp : (wo - 1u < persp ? wo - 1u : wo);
        const u32 fp = fo == 0u ? persp : (fo - 1u < persp ? fo - 1u : fo);
        const u32 v_pat = (hread(g.cpk, pc) >> (8u * pp)) & 0xffu;
        const u32 v_wall = (pick<P>(g.wall, wp) >> wc) & 1u, v_floor = (g.floors >> (8u * fp)) & 0xffu, v_score = (u32)pick<P>(g.score, fp);
        u32 v = j < A ? v_disp : j < Bc ? v_cen : j < Cp ? v_pat : j < Dw ? v_wall : j < Ef ? v_floor : j < Fs ? v_score : pnfp;
        if (j < N) out[j] = (float)(i32)v;
    }
}

// ---- flat random-agent self-play: mask -> RandomAgent -> Azul.step, a fresh Azul + new_round() when a game ends or nobody can move ------------
// (game_runner.py:87-97 works on any mask, azul.py:296-313 is P-generic; GameRunner's shaped reward is two-player, game_runner.py:50: the
// reward stream carries zeros.)  OUT as in az2: 0 = no streams, 1 = mask + action + reward + done + packed (+ BITS: maskbits), 2 = any subset.
template <u32 D, bool PAD>
AZ_FN void store_mask_x(const Out2 &o, const MaskX<D> &m, u32 l)
{
    if (!Dim<D>::WIDE) {
        store_mask_row2<PAD>(o, m.m[0][0], m.m[1][0], m.m[2][0], m.m[3][0], m.m[4][0], m.m[5][0],
                             m.bit[0][0], m.bit[1][0], m.bit[2][0], m.bit[3][0], m.bit[4][0], m.bit[5][0], l);
    } else if (PAD) {
        // rows of >= NA + 4 bytes, 8-byte aligned: the NA bits concatenated, lane j takes bits 8 j .. 8 j + 7 of the first 256 (and
        // lane j < NA / 8 - 32 bits 256 + 8 j ..), spreads them into eight 0 / 1 bytes and writes them with ONE 8-byte store
        u64 limb[Dim<D>::NL + 1];
        mask_limbs<D>(m, limb);
        constexpr u32 CH = (Dim<D>::NA + 7u) / 8u;                        // 8-byte chunks of a row: 30 (seven displays) / 38 (nine)
        // lane j's eight bits are byte j & 7 of limb j >> 3 (lanes beyond the row repeat its last chunk: same address, same data)
        const u32 j = l < CH ? l : CH - 1u;
        const u32 s8 = 8u * (j & 7u), qd = j >> 3;
        u32 b0 = (u32)(limb[0] >> s8), b1 = (u32)(limb[1] >> s8), b2 = (u32)(limb[2] >> s8), b3 = (u32)(limb[3] >> s8);
        asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));      // (plain v_cndmask on lane constants below, no branches around the shifts)
        const u32 by = (qd == 0u ? b0 : qd == 1u ? b1 : qd == 2u ? b2 : b3) & 0xffu;
        const u32 lo = ((by & 15u) * 0x00204081u) & 0x01010101u, hi = ((by >> 4) * 0x00204081u) & 0x01010101u;
        *(u64 *)(o.mask + (o.e * o.pitch + 8u * j)) = (u64)lo | ((u64)hi << 32);
        if (CH > 32u) {
            const u32 j2 = l < CH - 32u ? l : CH - 33u;
            const u32 by2 = (u32)(limb[4] >> (8u * j2)) & 0xffu;
            const u32 lo2 = ((by2 & 15u) * 0x00204081u) & 0x01010101u, hi2 = ((by2 >> 4) * 0x00204081u) & 0x01010101u;
            *(u64 *)(o.mask + (o.e * o.pitch + 256u + 8u * j2)) = (u64)lo2 | ((u64)hi2 << 32);
        }
    } else {
        uint8_t *row = o.mask + (o.e * o.pitch + l);
#pragma unroll
        for (u32 r = 0; r < 6u; r++) {
            row[Dim<D>::Q * r] = (uint8_t)m.bit[r][0];
            if (Dim<D>::NW > 1) { if (l < Dim<D>::Q - 32u) row[Dim<D>::Q * r + 32u] = (uint8_t)m.bit[r][Dim<D>::NW - 1]; }
        }
    }
}

template <u32 D>
AZ_FN void store_maskbits_x(const Out2 &o, const MaskX<D> &m, u32 l)
{
    u64 limb[Dim<D>::NL + 1];
    mask_limbs<D>(m, limb);
    // every lane stores (lanes NL.. repeat the last word's address and data): no exec masking
    const u32 q = l < Dim<D>::NL ? l : Dim<D>::NL - 1u;
    u64 v = limb[0];
#pragma unroll
    for (u32 i = 1; i < Dim<D>::NL; i++) v = q == i ? limb[i] : v;
    o.maskbits[o.e * Dim<D>::NL + q] = v;
}

template <u32 P, u32 D, int OUT>
AZ_FN void outputs_x(const GX<P, D> &g, const Out2 &o, i32 a, u32 dn, u32 l)
{
    if (OUT == 0) return;
    const i32 av = a >= 0 ? a : -1;
    // compact record: action | done << 8; actions beyond 254 do not fit the byte (nine displays: 300 actions) -> 16 bits of action in
    // the record's reward half, which is free here (the shaped reward is GameRunner's, two players): action | done << 8 for a < 255,
    // 0xff | done << 8 | action << 16 otherwise
    const u32 pk = (a >= 0 && a < 255) ? pack_move(a, dn, 0) : (0xffu | (dn << 8) | ((u32)(a >= 0 ? a : 0xffff) << 16));
    if (OUT == 1) {
        o.dw3[o.e] = l == 0u ? (u32)av : (l == 1u ? 0u : pk);
        o.done[o.e] = (uint8_t)dn;
    } else {
        if (l == 0u) {
            if (o.action) o.action[o.e] = av;
            if (o.reward) o.reward[o.e] = 0;
            if (o.packed) o.packed[o.e] = pk;
            if (o.done) o.done[o.e] = (uint8_t)dn;
        }
        if (o.rec) gx_store(g, o.rec + o.e * (u32)AZUL_RECORD_BYTES_WIDE, l);
    }
}

// returns 0 = move played, 1 = game ended with this move, 2 = stuck, 0x100 | status on a rule error
template <u32 P, u32 D, int OUT, bool PAD, bool BITS>
AZ_FN u32 selfplay_step_x(GX<P, D> &g, const RulesX &rules, const KX<D> &K, Rng2 &r, const Tab2 &T, u64 margin, Counters2 &cnt,
                          const Out2 &o, bool &dead, SegProf *prof_ = nullptr)
{
    (void)prof_;
    AZ_STAMP(SEG_LOOP);
    const K2 &k = K.k;
    const u32 l = k.l;
    // the two MT19937 words of this move's random(), fetched speculatively from the tempered copy (az2::selfplay_step2)
    const bool hard = r.pos + 2u > 624u;
    u32 wa, wb;
    {
        const u32 i = hard ? 622u : r.pos;
        wa = r.tlds[i]; wb = r.tlds[i + 1u];
    }
    MaskX<D> m;
    legal_mask_x(g, K, m);
    u32 pre[7];
    pre[0] = 0;
#pragma unroll
    for (u32 rr = 0; rr < 6u; rr++) pre[rr + 1] = pre[rr] + row_count<D>(m, rr);
    const u32 J = pre[1], L = pre[6];                     // legal floor moves (row 0, weight 0.01) / all legal moves
    const bool nomove = (L == 0u) | (g.eog != 0u);        // ValueError in the reference (raised before random()) / a finished game handed in
    // (floor-only masks, M == 0, take the table's ninth pair {fl(100 S[J]), 0} with the ordinal counted from 0: selfplay_step2, azul_tables.hpp)
    const u32 M = L - J, Mc = M ? M : 256u, Jc = J < Dim<D>::TROWS ? J : Dim<D>::TROWS - 1u;
    const u32 kbase = M ? J : 0u;
    const double u01 = ((double)(wa >> 5) * 67108864.0 + (double)(wb >> 6)) * (1.0 / 9007199254740992.0);      // random()
    const double2 fs = T.fs[9u * Jc + 31u - (u32)__builtin_clz(Mc)];  // {Fr[J][ilog2 M], S[J]} | {fl(100 S[J]), 0}
    if (OUT == 1 || (OUT == 2 && o.mask)) store_mask_x<D, (PAD && OUT == 1)>(o, m, l);
    if ((OUT == 1 && BITS) || (OUT == 2 && o.maskbits)) store_maskbits_x<D>(o, m, l);
    AZ_STAMP(SEG_MASK);
    const double sJ = fs.y;
    const double total = ((double)M + fs.x) + 0.0;
    double u = u01;
    double x = u * total;
    double d = x - sJ;
    u32 fl = (u32)d;
    double fr = d - (double)fl;
    u32 kg = kbase + fl + 1u;
    bool edge = !(__builtin_fabs(fr - 0.5) < 0.5 - 1e-9) | (kg > L);
    r.pos += 2u;
    bool any_nomove = false;
    if (AZ_UNLIKELY(wave_any(hard | edge | nomove))) {
        r.pos -= (hard | nomove) ? 2u : 0u;
        if (hard & !nomove) {
            // CPython's index is 623 (the first word is the last of this state) or 624: regenerate, then read the tempered copy
            const bool one = r.pos == 623u;
            const u32 last = r.tlds[623];
            lds_sync();
            rng2_twist(r, l);
            const u32 t0 = r.tlds[0], t1 = r.tlds[1];
            wa = one ? last : t0; wb = one ? t0 : t1;
            r.pos = one ? 1u : 2u;
            u = ((double)(wa >> 5) * 67108864.0 + (double)(wb >> 6)) * (1.0 / 9007199254740992.0);
            x = u * total;
            d = x - sJ; fl = (u32)d; fr = d - (double)fl;
            kg = kbase + fl + 1u;
            edge = !(__builtin_fabs(fr - 0.5) < 0.5 - 1e-9) | (kg > L);
        }
        if (edge & !nomove) {
            const double sT = M ? sJ : T.fs[9u * Jc].y;      // the boundary search works on CPython's own x = random() * (S[J] + 0.0)
            kg = sample_slow2(T, M ? x : u * (sT + 0.0), sT, J, M, L);
        }
        any_nomove = wave_any(nomove);
    }
    Choice<D> ch;
    pick_action_x<D>(m, pre, kg, k, ch);
    AZ_STAMP(SEG_SAMPLE);

    u32 ret = 0;
    if (AZ_UNLIKELY(any_nomove)) {
        if (nomove) {
            // stuck (hazard H3), or handed an already finished game: report, restart the slot
            cnt.stuck_add += 1u;
            {
                MaskX<D> none;                           // ... an empty mask row, like a stuck slot's (nothing is legal)
#pragma unroll
                for (u32 rr = 0; rr < 6u; rr++)
#pragma unroll
                    for (u32 ww = 0; ww < Dim<D>::NW; ww++) { none.m[rr][ww] = 0u; none.bit[rr][ww] = 0u; }
                if (OUT == 1 || (OUT == 2 && o.mask)) store_mask_x<D, (PAD && OUT == 1)>(o, none, l);
                if ((OUT == 1 && BITS) || (OUT == 2 && o.maskbits)) store_maskbits_x<D>(o, none, l);
            }
            outputs_x<P, D, OUT>(g, o, -1, 2u, l);
            const u32 st0 = restart_x(g, rules, r, margin, K);
            ret = st0 ? (0x100u | st0) : 2u;
        }
        dead |= (ret & 0x100u) != 0u;
        AZ_STAMP(SEG_RESET);
    }
    if (!nomove) {
        do_move_x(g, ch.s, ch.c, ch.row, rules.pool != (u32)XPOOL_RANDOM, K);        // azul.py:304
        sources_x(g);
        AZ_STAMP(SEG_MOVE);
        const bool eor = (g.B0 | g.B1) == 0u;                 // :306 is_end_of_round (the token counts)
        g.cur = eor ? g.cur : (g.cur < P ? g.cur + 1u : 1u);  // :313 next_player
        u32 st = ST_OK;
        bool any_done = false;
        AZ_STAMP(SEG_AFTERMOVE);
        if (AZ_UNLIKELY(wave_any(eor))) {
            // (two divergent regions with wave-uniform stamps between them, like az2::after_move2: the diagnostic build's lane 0 sees
            // the round ends of BOTH games of the wave)
            if (eor) count_score_x(g, rules.pool != (u32)XPOOL_RANDOM, rules.end_bonus != 0u, K);         // :307
            AZ_STAMP(SEG_SCORE);
            if (eor) {
                if (g.over) {                                 // :308-309
                    g.eog = 1;
                    if (rules.end_bonus) end_game_bonus_x(g);
                } else {
                    st = new_round_x(g, rules, r, margin, K); // :311
                }
            }
            AZ_STAMP(SEG_NEWROUND);
            any_done = wave_any(eor & (g.over != 0u));
            dead |= st != ST_OK;
        }
        const u32 dn = g.eog ? 1u : 0u;
        outputs_x<P, D, OUT>(g, o, ch.a, dn, l);
        AZ_STAMP(SEG_TAIL);
        ret = st != ST_OK ? (0x100u | st) : dn;
        if (AZ_UNLIKELY(any_done)) {
            if (dn != 0u) {
                {
                    double fsum = 0.0;
#pragma unroll
                    for (u32 p = 0; p < P; p++) fsum += (double)g.fps[p];          // first_player_stats.sum(): left to right
                    counters2_episode(cnt, stat_lane(l, g.score[0], g.score[1], g.turn, (double)g.fps[0] / fsum * 100, g.fpen[0], g.mc[0], g.cl[0]));
                }
                const u32 st2 = restart_x(g, rules, r, margin, K);      // a fresh Azul(players = P, rules) + new_round()
                if (st2) ret = 0x100u | st2;
            }
            dead |= (ret & 0x100u) != 0u;
            AZ_STAMP(SEG_RESET);
        }
    }
    return ret;
}

// RandomAgent on a mask (the game's own, or a caller's): one random.choices draw from the game's stream; -1 when nothing is legal
// (no word consumed).  The single-call form (op kernel): words through the stream's window, no speculation.
template <u32 D>
AZ_FN i32 random_agent_x(const MaskX<D> &m, Rng2 &r, const Tab2 &T, const K2 &k)
{
    u32 pre[7];
    pre[0] = 0;
#pragma unroll
    for (u32 rr = 0; rr < 6u; rr++) pre[rr + 1] = pre[rr] + row_count<D>(m, rr);
    const u32 J = pre[1], L = pre[6];
    i32 a = -1;
    if (L != 0u) {
        const u32 M = L - J;
        const double sJ = T.fs[9u * J].y;
        const double total = (M ? tpat2(T, J, M) : sJ) + 0.0;
        const double x = rng2_random(r, k.l) * total;
        const u32 kg = sample_slow2(T, x, sJ, J, M, L);
        Choice<D> ch;
        pick_action_x<D>(m, pre, kg, k, ch);
        a = ch.a;
    }
    return a;
}

// ---- kernel bodies (azul_kernels.hip wraps them in __global__ functions that own the LDS; tests/hostcheck/simt_rules_x.cpp runs the
// very same bodies on the lockstep 64-lane emulation) -------------------------------------------------------------------------------------
struct XBatchDev {
    uint8_t *state;      // [N][256]
    u32 *mt;             // [N][624]
    u32 *mtpos;          // [N]
    u64 *episodes;       // [N]
    u32 *stuck;          // [N]
    double *stat_sum;    // [N][10]
    u32 n;
    u64 draw_margin;
    RulesX rules;
    const double2 *tab;  // {Fr[J][b], S[J]} pairs, Dim<D>::TROWS x 8
    u64 *prof;           // [SEG_COUNT] segment cycle sums (only written by the -DAZ_PROFILE_SEGMENTS diagnostic build)
};

enum { XOP_QUERY = 0, XOP_INIT, XOP_NEW_ROUND, XOP_MOVE, XOP_NEXT_PLAYER, XOP_COUNT_SCORE, XOP_STEP, XOP_RANDOM_ACTION, XOP_SAMPLE_MASK };

struct XOp {
    int op;
    const i32 *actions;      // [count] in  (MOVE / STEP)
    const uint8_t *active;   // [count] in, optional
    const uint8_t *mask_in;  // [count][NA] in (SAMPLE_MASK)
    i32 *actions_out;        // [count] out (RANDOM_ACTION / SAMPLE_MASK)
    uint8_t *status;         // [count] out
    uint8_t *mask;           // [count][NA] out (after the op)
    float *obs;              // [count][obs_size] out (after the op)
    int persp;               // 0 .. P-1, or >= P: the player to move
    uint8_t *flags;          // [count] out
    double *stats;           // [count][10] out
    uint8_t *player;         // [count] out
    uint8_t *rng_dirty;      // [count] out: the op regenerated the game's MT19937 words
    uint8_t *rec_out;        // [count][256] out: the game's record after the op
    u32 *pos_out;            // [count] out: index of the game's MT19937 stream after the op
    i32 *next_action;        // [count] out: RandomAgent's choice on the state after the op, drawn at the stream's index after the op WITHOUT moving
                             //         it (-1: nothing legal, -2: not available: the op failed / did not draw, or the draw would cross a regeneration)
    u32 pos_set;             // 0, or 1 + the stream index to install before the op (single-game calls: the host's index is the authority)
    u32 first, count;        // the launch covers games first .. first + count - 1; row i of the arrays belongs to game first + i
};

struct XTraj {
    int n_steps;
    uint8_t *mask; u64 *maskbits; i32 *action; i32 *reward; uint8_t *done; uint8_t *rec; u32 *packed;
    u32 mask_stride;
};

AZ_FN bool xop_draws(int op) { return op == XOP_INIT || op == XOP_NEW_ROUND || op == XOP_STEP || op == XOP_RANDOM_ACTION || op == XOP_SAMPLE_MASK; }

template <u32 D>
AZ_FN void stage_tab_x(const double2 *tab, double2 *tab_lds, u32 lane)
{
    for (u32 i = lane; i < Dim<D>::TROWS * (u32)T_STRIDE; i += 64u) tab_lds[i] = tab[i];
    lds_sync();
}

// one rule call per game (two games per wave): Azul.__init__ / new_round / move / next_player / count_score / step, the RandomAgent sampler
// on the game's own or a caller's mask, and the queries (mask, observation, flags, statistics, player to move) on the state after the call
template <u32 P, u32 D>
AZ_FN void op_body_x(const XBatchDev &b, const XOp &a, u32 pair /* games 2 pair, 2 pair + 1 of the launch */, u32 (*mt_lds)[624], double2 *tab_lds)
{
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    stage_tab_x<D>(b.tab, tab_lds, lane);
    const u32 oi = 2u * pair + half;
    if (oi >= a.count) return;
    const u32 gi = oi + a.first;
    const bool act = a.active ? (a.active[oi] != 0) : true;
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES_WIDE;
    KX<D> K;
    kx_init(K);
    const Tab2 tab = {tab_lds};
    GX<P, D> g;
    gx_load(g, rec, l);
    prime_x(g, K);
    const bool tracked = b.rules.pool != (u32)XPOOL_RANDOM;
    u32 st = ST_OK, rdirty = 0;
    i32 spec = -2;
    if (act && a.op != XOP_QUERY) {
        const bool use_rng = xop_draws(a.op);
        Rng2 r;
        u32 *gmt = b.mt + (size_t)gi * 624u;
        rng2_open(r, gmt, mt_lds[half], use_rng ? (a.pos_set ? a.pos_set - 1u : b.mtpos[gi]) : 0u, l);
        bool dirty_state = true;
        switch (a.op) {
        case XOP_INIT:
            game_ctor_x(g, b.rules, r, K);
            break;
        case XOP_NEW_ROUND:
            st = new_round_x(g, b.rules, r, b.draw_margin, K);
            break;
        case XOP_MOVE: {
            const i32 av = a.actions[oi];
            if (av < 0 || av >= (i32)Dim<D>::NA) { st = ST_BAD_ACTION; dirty_state = false; break; }
            const u32 row = (u32)av / Dim<D>::Q, q = (u32)av - row * Dim<D>::Q, c = q / Dim<D>::S, s = q - c * Dim<D>::S;
            do_move_x(g, s, c, row, tracked, K);
            sources_x(g);
        } break;
        case XOP_NEXT_PLAYER:
            g.cur = (g.cur < P) ? g.cur + 1u : 1u;
            break;
        case XOP_COUNT_SCORE:
            count_score_x(g, tracked, b.rules.end_bonus != 0u, K);
            break;
        case XOP_STEP:
            st = checked_step_x(g, b.rules, K, r, b.draw_margin, a.actions[oi]);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
            break;
        case XOP_RANDOM_ACTION: {
            MaskX<D> m;
            legal_mask_x(g, K, m);
            const i32 av = random_agent_x<D>(m, r, tab, K.k);
            if (l == 0u) a.actions_out[oi] = av;
            dirty_state = false;
        } break;
        case XOP_SAMPLE_MASK: {
            const uint8_t *mi = a.mask_in + (size_t)oi * Dim<D>::NA;
            MaskX<D> m;
#pragma unroll
            for (u32 rr = 0; rr < 6u; rr++) {
#pragma unroll
                for (u32 w = 0; w < Dim<D>::NW; w++) {
                    const u32 q = 32u * w + l;
                    const u32 bit = (q < Dim<D>::Q && mi[Dim<D>::Q * rr + (q < Dim<D>::Q ? q : 0u)] != 0) ? 1u : 0u;
                    m.bit[rr][w] = bit;
                    m.m[rr][w] = hb(bit != 0u);
                }
            }
            const i32 av = random_agent_x<D>(m, r, tab, K.k);
            if (l == 0u) a.actions_out[oi] = av;
            dirty_state = false;
        } break;
        default:
            dirty_state = false;
            break;
        }
        if (dirty_state) gx_store(g, rec, l);
        if (a.next_action && use_rng && st == ST_OK && r.pos + 2u <= 624u) {
            // the question a mask -> RandomAgent -> step loop asks next, answered from the two words the stream would hand out next;
            // the index is restored: the caller advances it when it plays the answer (azul_game_call: AZUL_WANT_NEXT_ACTION / _POS_IN)
            const u32 keep = r.pos;
            MaskX<D> m;
            legal_mask_x(g, K, m);
            spec = random_agent_x<D>(m, r, tab, K.k);
            r.pos = keep;
        }
        if (use_rng) rng2_close(r, gmt, b.mtpos + gi, l);
        rdirty = r.dirty;
    }
    if (a.next_action && l == 0u) a.next_action[oi] = spec;
    if (a.rng_dirty && l == 0u) a.rng_dirty[oi] = (uint8_t)rdirty;
    if (a.status && act && l == 0u) a.status[oi] = (uint8_t)st;
    if (a.rec_out) gx_store(g, a.rec_out + (size_t)oi * AZUL_RECORD_BYTES_WIDE, l);
    if (a.pos_out && l == 0u) a.pos_out[oi] = b.mtpos[gi];       // (written by rng2_close above when the op drew)
    // queries on the post-op state
    if (a.mask) {
        MaskX<D> m;
        legal_mask_x(g, K, m);
        uint8_t *row = a.mask + (size_t)oi * Dim<D>::NA + l;
#pragma unroll
        for (u32 rr = 0; rr < 6u; rr++) {
            if (l < (Dim<D>::Q < 32u ? Dim<D>::Q : 32u)) row[Dim<D>::Q * rr] = (uint8_t)m.bit[rr][0];
            if (Dim<D>::NW > 1) { if (l < Dim<D>::Q - 32u) row[Dim<D>::Q * rr + 32u] = (uint8_t)m.bit[rr][Dim<D>::NW - 1]; }
        }
    }
    if (a.obs) observe_x(g, (u32)a.persp < P ? (u32)a.persp : mex(g), a.obs + (size_t)oi * obs_size<P, D>(), l);
    if (a.flags) {
        const u32 f = ((g.B0 | g.B1) == 0u ? AZUL_FLAG_END_OF_ROUND : 0) | (g.over ? AZUL_FLAG_END_OF_GAME : 0) | (g.eog ? AZUL_FLAG_ENDED_FLAG : 0);
        if (l == 0u) a.flags[oi] = (uint8_t)f;
    }
    if (a.stats) {
        if (l == 0u) for (u32 q = 0; q < 10u; q++) a.stats[(size_t)oi * 10 + q] = game_stat_x(g, q);
    }
    if (a.player && l == 0u) a.player[oi] = (uint8_t)g.cur;
}

// the persistent self-play loop of one wave (two games): t.n_steps moves per game, state and streams resident
template <u32 P, u32 D, int OUT, bool PAD, bool BITS>
AZ_FN void selfplay_body_x(const XBatchDev &b, const XTraj &t, u32 wave_id, u32 (*mt_lds)[624], u32 (*mtt_lds)[624], double2 *tab_lds)
{
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    stage_tab_x<D>(b.tab, tab_lds, lane);
    const u32 gi = wave_id * 2u + half;
    if (gi >= b.n) return;                               // odd batch: the last wave plays one game
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES_WIDE;
    KX<D> K;
    kx_init(K);
    const Tab2 tab = {tab_lds};
    GX<P, D> g;
    gx_load(g, rec, l);
    prime_x(g, K);
    Rng2 r;
    u32 *gmt = b.mt + (size_t)gi * 624u;
    rng2_open(r, gmt, mt_lds[half], b.mtpos[gi], l);
    rng2_attach_tempered(r, mtt_lds[half], l);
    Counters2 cnt;
    counters2_open(cnt, b.episodes + gi, b.stuck + gi, b.stat_sum + (size_t)gi * 10, l);
    Out2 o = {t.mask, t.maskbits, t.action, t.reward, t.done, t.packed, t.rec, t.mask_stride, gi,
              l == 0u ? (u32 *)t.action : (l == 1u ? (u32 *)t.reward : t.packed)};
    bool dead = false;               // a game stopped by a rule error (bag and lid empty without the short-deal rule) stays as it is (see the loop)
#if defined(AZ_PROFILE_SEGMENTS)
    SegProf prof;
    for (int q = 0; q < SEG_COUNT; q++) prof.acc[q] = 0;
    prof.last = __builtin_amdgcn_s_memtime();
    SegProf *pp = &prof;
#else
    SegProf *pp = nullptr;
#endif
#pragma unroll 1
    for (int s = 0; s < t.n_steps; s++) {
        if (!dead) selfplay_step_x<P, D, OUT, PAD, BITS>(g, b.rules, K, r, tab, b.draw_margin, cnt, o, dead, pp);
        else {
            // the game was stopped by a rule error (bag and lid empty without the short-deal rule; the reference raises, azul.py:86-87): no
            // move is played in this slot -- it is marked like a stuck slot (action -1, done 2) and counted with them, so that whoever
            // counts env moves as slots minus `stuck` stays right and the trajectory carries no stale data
            cnt.stuck_add += 1u;
            {
                MaskX<D> none;                           // ... an empty mask row, like a stuck slot's (nothing is legal)
#pragma unroll
                for (u32 rr = 0; rr < 6u; rr++)
#pragma unroll
                    for (u32 ww = 0; ww < Dim<D>::NW; ww++) { none.m[rr][ww] = 0u; none.bit[rr][ww] = 0u; }
                if (OUT == 1 || (OUT == 2 && o.mask)) store_mask_x<D, (PAD && OUT == 1)>(o, none, l);
                if ((OUT == 1 && BITS) || (OUT == 2 && o.maskbits)) store_maskbits_x<D>(o, none, l);
            }
            outputs_x<P, D, OUT>(g, o, -1, 2u, l);
        }
        o.e += b.n;
    }
#if defined(AZ_PROFILE_SEGMENTS)
    if (lane == 0u) for (int q = 0; q < SEG_COUNT; q++) atomicAdd((unsigned long long *)(b.prof + q), (unsigned long long)prof.acc[q]);
#endif
    gx_store(g, rec, l);
    rng2_close(r, gmt, b.mtpos + gi, l);
    counters2_close(cnt, l);
}

} // namespace azx
