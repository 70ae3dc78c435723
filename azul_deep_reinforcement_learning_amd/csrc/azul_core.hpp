// azul_core.hpp -- Azul rules, legal-move mask, scoring, RandomAgent and the CPython-exact random
// stream, written for ONE GAME PER 64-LANE WAVEFRONT (see azul_wave.hpp for the model).
//
// Data layout inside a wave (P = 2 players):
//   cs  (VGPR, "source cells")  lane 5d+c = displays[d][c] (0..24), lane 25+c = center[c], lane 30 = token
//   cp  (VGPR, "pattern cells") lane 25p+5r+c = pattern_lines[p][r][c] (0..49)
//   walls[p]                    uniform 25-bit bitboards, bit 5r+c  (colour-indexed like the reference)
//   everything else             uniform scalars
// so that  ballot(cs != 0)  IS the "which sources hold tiles" bitboard (end-of-round test and the
// legal-move mask both fall out of it) and  ballot(cp == row+1)  IS the set of full pattern lines.
//
// Behaviour restated from the reference (paths relative to /root/reference):
//   new_round        azulnet/azul.py:64-89        move            azulnet/azul.py:118-161
//   is_legal_move    azulnet/azul.py:162-176      next_player     azulnet/azul.py:177-181
//   is_end_of_round  azulnet/azul.py:182-183      is_end_of_game  azulnet/azul.py:184-191
//   count_score      azulnet/azul.py:192-295      step            azulnet/azul.py:296-313
//   GameRunner.step / reset / get_state           azulnet/game_runner.py:43-55, 76-85, 56-72
//   RandomAgent      azulnet/game_runner.py:87-97 check_all_valid azulnet/game_runner.py:113-117
//   random.seed / random() / getrandbits / _randbelow / choices   CPython 3.10 (_randommodule.c, random.py)
#pragma once
#include "azul_wave.hpp"

namespace az {
using namespace wv;

enum { ST_OK = 0, ST_ILLEGAL_MOVE = 1, ST_GAME_ENDED = 2, ST_STUCK = 3, ST_BAD_ACTION = 4, ST_BOX_EMPTY = 5 };
enum { POOL_RANDOM = 0, POOL_LID = 1 };
#ifndef AZ_DRAW_MARGIN
#define AZ_DRAW_MARGIN 8192ull      // see new_round(): > 21.1 * 255, the largest possible fp64 disagreement window
#endif
enum { T_ROWS = 31, T_BINADES = 8, T_WORDS = T_ROWS * T_BINADES + T_ROWS };   // RandomAgent weight table, see azul_tables.hpp

// ---- optional in-kernel segment stamps (diagnostic build only: -DAZ_PROFILE_SEGMENTS; cdna_hip_programming.md 7) ----
// s_memtime deltas are accumulated per segment and added to a global buffer when the wave ends.  The real kernel
// contains no stamp; never quote the diagnostic build's run time, only its SHARES (tools/segment_profile.py).
enum { SEG_MASK = 0, SEG_SAMPLE, SEG_MOVE, SEG_AFTERMOVE, SEG_TAIL, SEG_NEWROUND, SEG_SCORE, SEG_RESET, SEG_LOOP, SEG_COUNT };
#if defined(AZ_PROFILE_SEGMENTS)
struct SegProf { u64 last; u64 acc[SEG_COUNT]; };
// (null-safe: callers outside the self-play kernels pass no SegProf -- an unguarded store would be undefined behaviour, which the
// compiler is free to "optimise" into dropping the code behind the stamp: such a build ran the policy rollout 26 % faster, and wrong)
#define AZ_STAMP(seg) do { if (prof_) { u64 now_ = __builtin_amdgcn_s_memtime(); prof_->acc[seg] += now_ - prof_->last; prof_->last = now_; } } while (0)
#else
struct SegProf { int unused; };
#define AZ_STAMP(seg) do { } while (0)
#endif

struct Rules {
    u32 first_player;   // 0 = "Random", 1..2 = fixed
    u32 tile_pool;      // POOL_RANDOM / POOL_LID
};

// ------------------------------------------------------------------------------------------------
// CPython MT19937 stream of one game.  The 624-word state lives in global memory (mt[624], row of
// a [N][624] array, so a wave's accesses are contiguous); 64-word chunks are staged into a
// per-wave LDS window on first use and the regeneration ("twist") runs in LDS across all 64 lanes.
// ------------------------------------------------------------------------------------------------
struct Rng {
    u32 *gmt;       // this game's 624 words in global memory
    u32 *lds;       // this wave's 624-word LDS window (staged whole when the stream is opened)
    u32 pos;        // CPython's `index` (0..624)
    u32 dirty;      // bit 0: LDS state differs from global memory (a twist happened); bit 1: state not staged into LDS yet
    u32 wbase;      // first word held by `win`
    u32 wend;       // one past the last word `win` can serve (0: no window loaded)
    u64 margin;     // factory-draw disagreement window (new_round); AZ_DRAW_MARGIN unless a test widens it
    vu32 win;       // lane i: TEMPERED output word wbase + i  (one readlane per random word)
};

AZ_FN vu32 temper_v(vu32 y)
{
    y = y ^ (y >> 11);
    y = y ^ ((y << 7) & 0x9d2c5680u);
    y = y ^ ((y << 15) & 0xefc60000u);
    y = y ^ (y >> 18);
    return y;
}

AZ_FN void rng_stage(Rng &r)
{
    // ten coalesced 256-byte loads, all in flight before the first LDS write (one memory latency, not ten)
    const u32 *gmt = r.gmt;
    u32 *lds = r.lds;
    vu32 l = lane();
    vu32 w0 = ld_u32(gmt, l, l < 64u), w1 = ld_u32(gmt, l + 64u, l < 64u), w2 = ld_u32(gmt, l + 128u, l < 64u),
         w3 = ld_u32(gmt, l + 192u, l < 64u), w4 = ld_u32(gmt, l + 256u, l < 64u), w5 = ld_u32(gmt, l + 320u, l < 64u),
         w6 = ld_u32(gmt, l + 384u, l < 64u), w7 = ld_u32(gmt, l + 448u, l < 64u), w8 = ld_u32(gmt, l + 512u, l < 64u),
         w9 = ld_u32(gmt, l + 576u, l < 48u);
    lds_st(lds, l, w0, l < 64u); lds_st(lds, l + 64u, w1, l < 64u); lds_st(lds, l + 128u, w2, l < 64u);
    lds_st(lds, l + 192u, w3, l < 64u); lds_st(lds, l + 256u, w4, l < 64u); lds_st(lds, l + 320u, w5, l < 64u);
    lds_st(lds, l + 384u, w6, l < 64u); lds_st(lds, l + 448u, w7, l < 64u); lds_st(lds, l + 512u, w8, l < 64u);
    lds_st(lds, l + 576u, w9, l < 48u);
    lds_fence();
    r.dirty &= ~2u;
}

AZ_FN void rng_open(Rng &r, u32 *gmt, u32 *lds, u32 pos)
{
    r.gmt = gmt; r.lds = lds; r.pos = pos; r.dirty = 0; r.wbase = 0; r.wend = 0; r.win = splat(0u); r.margin = AZ_DRAW_MARGIN;
    rng_stage(r);
}

// Like rng_open, but the 2.5 KB state is only fetched when the first random word is asked for: most single env steps
// (everything but a round rollover, an episode reset or an opponent's move) never draw.
AZ_FN void rng_attach(Rng &r, u32 *gmt, u32 *lds, u32 pos)
{
    r.gmt = gmt; r.lds = lds; r.pos = pos; r.dirty = 2u; r.wbase = 0; r.wend = 0; r.win = splat(0u); r.margin = AZ_DRAW_MARGIN;
}

AZ_FN void rng_twist(Rng &r)
{
    // genrand_uint32's regeneration loop; ascending 64-wide chunks are legal because element i needs
    // OLD mt[i], mt[i+1] and (i<227: OLD mt[i+397] | i>=227: NEW mt[i-227]); see DESIGN.md.
#pragma unroll 1
    for (u32 k = 0; k < 10; k++) {
        vu32 i = lane() + k * 64u;
        vbool act = i < 624u;
        vu32 i1 = sel(i == 623u, splat(0u), i + 1u);
        vu32 i2 = sel(i < 227u, i + 397u, i - 227u);
        vu32 a = lds_ld(r.lds, i, act), b = lds_ld(r.lds, i1, act), c = lds_ld(r.lds, i2, act);
        vu32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
        vu32 v = c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        lds_fence();
        lds_st(r.lds, i, v, act);
        lds_fence();
    }
    r.dirty = 1;
    r.pos = 0;
    r.wend = 0;
}

AZ_FN void rng_refill(Rng &r)
{
    if (AZ_UNLIKELY(r.dirty & 2u)) rng_stage(r);
    if (r.pos >= 624u) rng_twist(r);
    r.wbase = r.pos;
    r.wend = r.pos + 64u < 624u ? r.pos + 64u : 624u;
    vu32 i = lane() + r.pos;
    r.win = temper_v(lds_ld(r.lds, i, i < 624u));
}

AZ_FN u32 rng_u32(Rng &r)
{
    if (AZ_UNLIKELY(r.pos >= r.wend)) rng_refill(r);    // window exhausted, state exhausted (pos == 624) or none yet
    u32 y = readlane(r.win, r.pos - r.wbase);
    r.pos += 1;
    return y;
}

AZ_FN double rng_random(Rng &r)
{
    // random_random(): two consecutive words; one window check serves both when they sit in the same window
    u32 a, b;
    if (!AZ_UNLIKELY(r.pos + 2u > r.wend)) {
        u32 off = r.pos - r.wbase;
        a = readlane(r.win, off);
        b = readlane(r.win, off + 1u);
        r.pos += 2u;
    } else {
        a = rng_u32(r);
        b = rng_u32(r);
    }
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

AZ_FN u32 rng_below(Rng &r, u32 n, u32 bits)
{
    // _randbelow_with_getrandbits: k = n.bit_length(); rejection on getrandbits(k)
    u32 v = rng_u32(r) >> (32u - bits);
    while (v >= n) v = rng_u32(r) >> (32u - bits);
    return v;
}

AZ_FN void rng_close(Rng &r, u32 *pos_out)
{
    if (r.dirty & 1u) {
#pragma unroll 1
        for (u32 k = 0; k < 10; k++) {
            vu32 i = lane() + k * 64u;
            vu32 w = lds_ld(r.lds, i, i < 624u);
            st_u32(r.gmt, i, w, i < 624u);
        }
    }
    AZ_LANE0(*pos_out = r.pos);
}

// ------------------------------------------------------------------------------------------------
// game state (all uniform fields are named scalars: no runtime-indexed arrays -> no scratch memory)
// ------------------------------------------------------------------------------------------------
struct Game {
    vu32 cs, cp;
    u32 wall0, wall1;
    i32 score0, score1;
    u32 floor0, floor1;
    u32 cur, nfp, eog;
    u64 box, lid;           // byte c = tiles of colour c
    u32 turn;
    u32 fps;                // lo16 = first_player_stats[0], hi16 = [1]
    u32 fpen;               // two i16: floor_penalty[0], [1]
    u32 maxc;               // byte0 = max_combo[0], byte1 = [1]
    u64 compl_;             // byte 3p+k = completed_lines[p][k]
    i32 pscore;             // GameRunner.player_score
    u32 moves;              // GameRunner.move_counter
    // derived, never stored: what-if scoring cache (game_runner.py:48-50) and wall status
    i32 wc0, wc1;           // points the players' currently FULL pattern lines would earn (count_wall)
    i32 wi0, wi1;           // what-if scores: max(0, score + floor penalty + wc)
    u32 over;               // is_end_of_game(): some wall row is complete (walls only change at scoring)
};

constexpr u32 column_board_c(int col)
{
    u32 m = 0;
    for (int j = 0; j < 5; j++) m |= 1u << (5 * j + ((col - j + 5) % 5));
    return m;
}
static_assert(column_board_c(0) == 0x222201u && column_board_c(4) == 0x111110u, "column boards");

struct LaneConst {
    vu32 spos[3];    // action a=64w+lane: cs lane of its source cell (31 = none)
    vu32 okpos[3];   // action a: bit of the "row accepts colour" board (31 = floor move, always ok)
    vu32 acode[3];   // action a=64w+lane decoded once: src cell | display base << 5 | colour << 10 | row << 13 | from_display << 16 | a << 17
    vu32 rowp1;      // cp lane l<50: row+1, else 0xff
    // per pattern cell (lane l<50: player l/25, cell i=l%25, row r=i/5, colour c=i%5):
    vu32 prow, pcol_, pbcol; // r, c, board column (c+r)%5
    vu32 pbelow;             // bits 0..i of a 25-bit board ("this placement and everything scored before it")
    vu32 pcolboard;          // wall cells lying in board column (c+r)%5
};

AZ_FN void lane_consts(LaneConst &k)
{
    for (u32 w = 0; w < 3; w++) {
        vu32 a = lane() + w * 64u;
        vu32 d = a % 6u, c = (a / 6u) % 5u, r = a / 30u;
        vu32 sp = sel(d == 0u, c + 25u, (d - 1u) * 5u + c);
        k.spos[w] = sel(a < 180u, sp, splat(31u));
        k.okpos[w] = sel(r == 0u, splat(31u), (r - 1u) * 5u + c);
        vu32 db = sel(d == 0u, splat(0u), (d - 1u) * 5u);
        k.acode[w] = sp | (db << 5) | (c << 10) | (r << 13) | (sel(d == 0u, splat(0u), splat(1u)) << 16) | (a << 17);
    }
    vu32 l = lane();
    k.rowp1 = sel(l < 50u, ((l % 25u) / 5u) + 1u, splat(0xffu));
    vu32 i = l % 25u;
    k.prow = i / 5u;
    k.pcol_ = i % 5u;
    vu32 bc = k.prow + k.pcol_;
    k.pbcol = sel(bc >= 5u, bc - 5u, bc);
    k.pbelow = (2u << i) - 1u;
    k.pcolboard = sel(k.pbcol == 0u, splat(column_board_c(0)), sel(k.pbcol == 1u, splat(column_board_c(1)),
                  sel(k.pbcol == 2u, splat(column_board_c(2)), sel(k.pbcol == 3u, splat(column_board_c(3)), splat(column_board_c(4))))));
}

AZ_FN u32 me_index(const Game &g) { return g.cur == 0u ? 1u : g.cur - 1u; }   // numpy [-1] before the first round

// ---- 128-byte record <-> registers (layout: include/azul_hip.h) ----
AZ_FN void game_load(Game &g, const uint8_t *rec)
{
    vu32 l = lane();
    vu32 a = ld_u8(rec, l, l < 32u);
    vu32 b = ld_u8(rec + 32, l, l < 52u);
    vu32 t = ld_u32((const u32 *)(rec + 84), l, l < 11u);
    u32 flags = readlane(a, 31);
    g.cur = flags & 7u; g.nfp = (flags >> 3) & 7u; g.eog = (flags >> 6) & 1u;
    g.cs = sel(l < 31u, a, splat(0u));
    g.cp = sel(l < 50u, b, splat(0u));
    g.floor0 = readlane(b, 50); g.floor1 = readlane(b, 51);
    g.wall0 = readlane(t, 0); g.wall1 = readlane(t, 1);
    u32 sc = readlane(t, 2);
    g.score0 = (i32)(int16_t)(sc & 0xffffu); g.score1 = (i32)(int16_t)(sc >> 16);
    u32 w3 = readlane(t, 3), w4 = readlane(t, 4), w5 = readlane(t, 5);
    g.box = (u64)w3 | ((u64)(w4 & 0xffu) << 32);
    g.lid = (u64)(w4 >> 8) | ((u64)(w5 & 0xffffu) << 24);
    g.turn = w5 >> 16;
    g.fps = readlane(t, 6);
    g.fpen = readlane(t, 7);
    u32 w8 = readlane(t, 8), w9 = readlane(t, 9), w10 = readlane(t, 10);
    g.maxc = w8 & 0xffffu;
    g.compl_ = (u64)(w8 >> 16) | ((u64)w9 << 16);
    g.pscore = (i32)(int16_t)(w10 & 0xffffu);
    g.moves = w10 >> 16;
    g.wc0 = g.wc1 = 0; g.wi0 = g.wi1 = 0; g.over = 0;
}

AZ_FN void game_store(const Game &g, uint8_t *rec)
{
    vu32 l = lane();
    u32 flags = (g.cur & 7u) | ((g.nfp & 7u) << 3) | ((g.eog & 1u) << 6);
    vu32 a = writelane(g.cs, flags, 31);
    st_u8(rec, l, a, l < 32u);
    vu32 b = writelane(writelane(g.cp, g.floor0, 50), g.floor1, 51);
    st_u8(rec + 32, l, b, l < 52u);
    vu32 t = splat(0u);
    t = writelane(t, g.wall0, 0);
    t = writelane(t, g.wall1, 1);
    t = writelane(t, ((u32)g.score0 & 0xffffu) | ((u32)g.score1 << 16), 2);
    t = writelane(t, (u32)g.box, 3);
    t = writelane(t, (u32)((g.box >> 32) & 0xffu) | ((u32)g.lid << 8), 4);
    t = writelane(t, (u32)((g.lid >> 24) & 0xffffu) | (g.turn << 16), 5);
    t = writelane(t, g.fps, 6);
    t = writelane(t, g.fpen, 7);
    t = writelane(t, (g.maxc & 0xffffu) | ((u32)(g.compl_ & 0xffffu) << 16), 8);
    t = writelane(t, (u32)(g.compl_ >> 16), 9);
    t = writelane(t, ((u32)g.pscore & 0xffffu) | (g.moves << 16), 10);
    st_u32((u32 *)(rec + 84), l, t, l < 11u);
}

// ---- legal-move mask: azul.py:162-176 over all 180 actions (game_runner.py:113-117) ----
struct Mask {
    u64 m0, m1, m2;      // action a: bit (a & 63) of word (a >> 6)
    vu32 b0, b1, b2;     // the same bits as 0/1 per lane (lane = a & 63): what the byte mask stores
};

AZ_FN u32 sources_board(const Game &g) { return (u32)ballot(g.cs != 0u) & 0x7fffffffu; }

// shared tail of the mask: B = sources holding tiles (31 bits), pme = the mover's non-empty pattern cells (25 bits),
// wl = the mover's wall
AZ_FN void mask_from_boards(u32 B, u32 pme, u32 wl, const LaneConst &k, Mask &out)
{
    // "row r accepts colour c": no OTHER colour lies on the row (azul.py:172) and the wall cell is free (:174);
    // one lane per (row, colour), ballot -> 25-bit board; bit 31: the floor "row" accepts everything
    vu32 l = lane();
    vu32 rb = (pme >> (k.prow * 5u)) & 31u;            // colours already lying on my row
    vu32 mine = (rb >> k.pcol_) & 1u;
    vbool alone = (rb == 0u) | (((rb & (rb - 1u)) == 0u) & (mine != 0u));
    vbool free_ = ((wl >> (l & 31u)) & 1u) == 0u;
    u32 ok = ((u32)ballot(alone & free_ & (l < 25u)) & 0x1ffffffu) | 0x80000000u;
    out.b0 = ((B >> (k.spos[0] & 31u)) & (ok >> (k.okpos[0] & 31u))) & 1u;
    out.b1 = ((B >> (k.spos[1] & 31u)) & (ok >> (k.okpos[1] & 31u))) & 1u;
    out.b2 = ((B >> (k.spos[2] & 31u)) & (ok >> (k.okpos[2] & 31u))) & 1u;
    out.m0 = ballot(out.b0 != 0u);
    out.m1 = ballot(out.b1 != 0u);
    out.m2 = ballot(out.b2 != 0u);
}

AZ_FN void legal_mask(const Game &g, const LaneConst &k, Mask &out)
{
    u32 B = sources_board(g);
    u64 PL = ballot(g.cp != 0u);
    u32 me = me_index(g);
    u32 pme = (u32)(PL >> (25u * me)) & 0x1ffffffu;
    u32 wl = me ? g.wall1 : g.wall0;
    mask_from_boards(B, pme, wl, k, out);
}

AZ_FN u64 mask_word(const Mask &m, u32 w) { return w == 0u ? m.m0 : (w == 1u ? m.m1 : m.m2); }
AZ_FN u32 mask_test(const Mask &m, u32 a) { return (u32)(mask_word(m, a >> 6) >> (a & 63u)) & 1u; }
AZ_FN u32 mask_count(const Mask &m) { return popc64(m.m0) + popc64(m.m1) + popc64(m.m2); }

AZ_FN void mask_write(const Mask &m, uint8_t *out)
{
    vu32 l = lane();
    st_u8(out, l, m.b0, l < 64u);
    st_u8(out, l + 64u, m.b1, l < 64u);
    st_u8(out, vmin(l, 51u) + 128u, sel(l < 52u, m.b2, splat((u32)(m.m2 >> 51) & 1u)), l < 64u);   // lanes 52.. repeat lane 51
}

AZ_FN void mask_write_bits(const Mask &m, u64 *out)
{
    // bit-packed form (24 bytes per game): what the multi-GPU trajectory all-gather ships
    stu_u64(out, m.m0); stu_u64(out + 1, m.m1); stu_u64(out + 2, m.m2);
}

// ---- RandomAgent: game_runner.py:87-97 + random.choices (random.py:506-541) ----
// The cumulative weight CPython's accumulate() reaches after J weights of 0.01 (legal floor moves, a < 30)
// and m weights of 1.0 is  T(J, 0) = S[J]  and, for m >= 1,  T(J, m) = m + Fr[J][floor(log2 m)]  EXACTLY:
// adding 1.0 only rounds when the sum enters a new binade, so per J there are 8 distinct fractional parts
// (azul_tables.hpp builds both tables with the very additions CPython performs; tests check all 31*151 sums).
// The 248 + 31 doubles live in ten VGPRs per wave: no memory access on the sampling path.
struct SampleTab {
    const double *fr;          // LDS: Fr[J][b] at 8J + b (248 doubles, staged once per wave)
    vf64 s;                    // S[J] in lane J
};

AZ_FN void sample_tab_load(SampleTab &t, const double *tab /* T_WORDS doubles: Fr[31][8] then S[31] */, double *lds_fr)
{
    vu32 l = lane();
    for (u32 q = 0; q < 4u; q++) {
        vu32 i = l + q * 64u;
        lds_st_f64(lds_fr, i, ld_f64(tab, i, i < 248u), i < 248u);
    }
    lds_fence();
    t.fr = lds_fr;
    t.s = ld_f64(tab + T_ROWS * T_BINADES, vmin(l, 30u), l < 64u);
}

// cumulative weight after m >= 1 pattern moves on top of J floor moves
AZ_FN double tpat(const SampleTab &t, u32 J, u32 m) { return (double)m + lds_ldu_f64(t.fr, 8u * J + 31u - clz32(m)); }

// cumulative weight after the k-th legal action (k = 0 -> 0.0)
AZ_FN double tseq(const SampleTab &t, u32 J, u32 k)
{
    if (k <= J) return readlane_d(t.s, k);
    return tpat(t, J, k - J);
}

AZ_FN i32 random_agent(const Mask &m, Rng &r, const SampleTab &T, const LaneConst &k, u32 &code)
{
    code = 0;
    u32 c0 = popc64(m.m0), c1 = popc64(m.m1), c2 = popc64(m.m2);
    u32 J = popc64(m.m0 & 0x3fffffffull);             // legal floor moves (a < 30, weight 0.01)
    u32 L = c0 + c1 + c2;
    if (AZ_UNLIKELY(L == 0u)) return -1;               // ValueError in the reference, raised before random()
    u32 M = L - J;
    double sJ = readlane_d(T.s, J);
    double total = (M ? tpat(T, J, M) : sJ) + 0.0;
    double x = rng_random(r) * total;
    // bisect_right over the cumulative weights == smallest ordinal k with cum(k) > x
    u32 kg;
    if (AZ_UNLIKELY(x < sJ)) {
        // inside the 0.01-weight floor moves (rare unless nothing else is legal): generic search
        kg = (u32)(x * 100.0) + 1u;
        kg = kg > J ? J : kg;
        for (u32 it = 0; it < 64u; it++) {
            bool below = x < tseq(T, J, kg - 1u), inside = x < tseq(T, J, kg);
            if (below && kg > 1u) kg -= 1u;
            else if (!inside && kg < L) kg += 1u;
            else break;
        }
    } else {
        // pattern moves: cum(J + m) = m + Fr[J][ilog2 m] with |Fr - S[J]| < 2^-44 (a handful of half-ulp roundings
        // below 256), and d = x - S[J] carries an fp64 error below 2^-45: whenever d is further than 1e-9 from an
        // integer, floor(d) + 1 IS the ordinal; otherwise the exact table values decide.
        double d = x - sJ;
        u32 fl = (u32)d;
        double fr = d - (double)fl;
        u32 mg = fl + 1u;
        if (AZ_UNLIKELY(!(fr > 1e-9 && fr < 1.0 - 1e-9) || mg > M)) {
            mg = mg > M ? M : mg;
            for (u32 it = 0; it < 256u; it++) {
                u32 ml = mg - 1u;
                double lo = ml ? tpat(T, J, ml) : sJ;
                double hi = tpat(T, J, mg);
                if (x < lo && mg > 1u) mg -= 1u;
                else if (!(x < hi) && mg < M) mg += 1u;
                else break;
            }
        }
        kg = J + mg;
    }
    // kg-th legal action: every lane ranks its own three actions (prefix popcounts), the one with rank kg answers
    u32 want = kg - 1u;
    vbool p0 = (m.b0 != 0u) & (mbcnt(m.m0) == want);
    vbool p1 = (m.b1 != 0u) & (mbcnt(m.m1) + c0 == want);
    vbool p2 = (m.b2 != 0u) & (mbcnt(m.m2) + (c0 + c1) == want);
    u32 ln = ctz64(ballot(p0 | p1 | p2));
    code = readlane(sel(p0, k.acode[0], sel(p1, k.acode[1], k.acode[2])), ln);
    return (i32)(code >> 17);
}

// ---- move: azul.py:118-161 ----
AZ_FN void byte_add(u64 &v, u32 idx, u32 n) { v += (u64)n << (8u * idx); }

// returns true when the targeted pattern line is full after the move (its wall pricing must be (re)computed).
// Written with selects instead of branches: a taken branch costs a wave far more than the few extra lane ops.
AZ_FN u32 action_code(u32 a)
{
    // the same packing as LaneConst::acode, for an action given by number (game_runner.py:107-111)
    u32 d = a % 6u, c = (a / 6u) % 5u, row = a / 30u;
    u32 db = d ? 5u * (d - 1u) : 0u;
    return (d ? db + c : 25u + c) | (db << 5) | (c << 10) | (row << 13) | ((d ? 1u : 0u) << 16) | (a << 17);
}

template <bool LID>
AZ_FN bool do_move(Game &g, u32 code)
{
    u32 me = me_index(g);
    vu32 l = lane();
    const u32 src = code & 31u, db = (code >> 5) & 31u, c = (code >> 10) & 7u, row = (code >> 13) & 7u;   // the source cell, ...
    const bool from_display = ((code >> 16) & 1u) != 0u;
    u32 n = readlane(g.cs, src);                                       // :127 / :136
    bool token = (!from_display) & (readlane(g.cs, 30) == 1u);         // :140 (no short-circuit: a branch costs more)
    // display: every other colour of that display slides into the centre (:131), the display empties (:129,:133)
    vu32 moved = bperm(g.cs, l - 25u + db);
    vbool centre = (l >= 25u) & (l < 30u) & (l != 25u + c) & from_display;
    vbool gone = ((l >= db) & (l < db + 5u) & from_display) | (l == src) | ((l == 30u) & token);   // :138, :141
    g.cs = sel(gone, splat(0u), sel(centre, g.cs + moved, g.cs));
    g.nfp = token ? g.cur : g.nfp;                                     // :142
    u32 fl = (me ? g.floor1 : g.floor0) + (token ? 1u : 0u);           // :143 (a token never overflows the cap alone)
    fl = fl < 7u ? fl : 7u;
    // pattern line or floor
    u32 cell = 25u * me + 5u * ((row ? row : 1u) - 1u) + c;
    u32 old = readlane(g.cp, cell);
    i32 overflow = row ? (i32)row - (i32)old - (i32)n : -(i32)n;       // :147 ; floor move: everything "overflows"
    u32 spill = overflow < 0 ? (u32)(-overflow) : 0u;
    u32 newv = overflow < 0 ? row : old + n;                           // :150 / :152
    g.cp = sel((l == cell) & (row != 0u), splat(newv), g.cp);
    fl += spill;                                                       // :154 / :159
    fl = fl < 7u ? fl : 7u;
    g.floor0 = me ? g.floor0 : fl;
    g.floor1 = me ? fl : g.floor1;
    if (LID) byte_add(g.lid, c, spill);                                // :156-157 / :160-161
    return (row != 0u) & (overflow <= 0);
}

// ---- scoring: azul.py:192-295 on 25-bit wall bitboards ----
AZ_FN u64 full_lines(const Game &g, const LaneConst &k) { return ballot(g.cp == k.rowp1); }              // azul.py:216

// Vector half of count_wall (azul.py:211-290): EVERY pattern cell (lane = player, row, colour) prices "a tile
// placed here now" against its player's wall plus the full lines that are scored before it (ascending row,
// colour -- `pbelow`), so the sequential dependency of the reference loop is reproduced without a loop.
struct ScoreVec {
    vu32 val;        // pos_count + bonus_count of that placement
    vu32 pos;        // pos_count alone (max_combo)
    u64 rowdone, colordone, coldone;   // lanes whose placement completes a row / colour / column
};

AZ_FN vu32 run_length_v(vu32 bits, vu32 pos)
{
    vu32 up = vctz(~(bits >> pos));
    vu32 below = ~bits & ((1u << pos) - 1u);
    vu32 down = sel(below != 0u, pos + vclz(below | 1u) - 32u, pos);
    return up + down;
}

// w: per lane, the wall its placement is priced against (the player's wall + the full lines scored before it)
AZ_FN void score_boards(vu32 w, const LaneConst &k, ScoreVec &s)
{
    vu32 rowbits = (w >> (k.prow * 5u)) & 31u;
    vu32 h = ((rowbits << k.prow) | (rowbits >> (5u - k.prow))) & 31u;       // the row in board-column order
    vu32 hr = run_length_v(h, k.pbcol);                                      // :230-242
    vu32 wc = w & k.pcolboard;
    vu32 t = wc | (wc >> 1) | (wc >> 2) | (wc >> 3) | (wc >> 4);
    vu32 v = (((t & 0x108421u) * 0x111110u) >> 20) & 31u;                     // the board column, bit j = row j
    vu32 vr = run_length_v(v, k.prow);                                       // :244-257
    vu32 both = hr + vr;
    s.pos = sel((hr == 1u) & (vr == 1u), splat(1u), sel((hr > 1u) & (vr > 1u), both, both - 1u));    // :258-263
    vbool rd = rowbits == 31u;                                               // :266-272
    vbool cd = ((w >> k.pcol_) & 0x108421u) == 0x108421u;                     // :274-280
    vbool kd = v == 31u;                                                     // :282-288
    s.val = s.pos + sel(rd, splat(2u), splat(0u)) + sel(cd, splat(10u), splat(0u)) + sel(kd, splat(7u), splat(0u));
    s.rowdone = ballot(rd); s.colordone = ballot(cd); s.coldone = ballot(kd);
}

AZ_FN void score_vec(const Game &g, const LaneConst &k, u64 F, ScoreVec &s)
{
    vbool p1 = lane() >= 25u;
    u32 f0 = (u32)F & 0x1ffffffu, f1 = (u32)(F >> 25) & 0x1ffffffu;
    vu32 w = sel(p1, splat(g.wall1), splat(g.wall0)) | (sel(p1, splat(f1), splat(f0)) & k.pbelow);   // :219
    score_boards(w, k, s);
}

AZ_FN i32 floor_penalty(u32 floor_tiles)
{
    u32 f = floor_tiles > 7u ? 7u : floor_tiles;         // count_floor, azul.py:200-210: 0,-1,-2,-4,-6,-8,-11,-14
    return -(i32)((0x0e0b080604020100ull >> (8u * f)) & 0xffu);
}

AZ_FN i32 clamp0(i32 s) { return s < 0 ? 0 : s; }       // azul.py:294-295

// scalar half: the sum over this player's full lines (+ statistics / wall / lid commits when REAL)
template <bool REAL, bool LID, u32 P>
AZ_FN i32 wall_points(Game &g, u64 F, const ScoreVec &sv)
{
    u32 fp = (u32)(F >> (25u * P)) & 0x1ffffffu;
    i32 cnt = 0;
    u32 mc = (g.maxc >> (8u * P)) & 0xffu;
    u32 rest = fp;
    while (rest) {
        u32 i = ctz32(rest);
        rest &= rest - 1u;
        cnt += (i32)readlane(sv.val, i + 25u * P);                           // :289
        if (REAL) {
            u32 ps = readlane(sv.pos, i + 25u * P);
            if (ps > mc) mc = ps;                                            // :264
            if (LID) { u32 r = (i * 205u) >> 10; byte_add(g.lid, i - 5u * r, r); }   // :220-222
        }
    }
    if (REAL) {
        if (P) g.wall1 |= fp; else g.wall0 |= fp;                            // :219
        g.maxc = (g.maxc & ~(0xffu << (8u * P))) | (mc << (8u * P));
        u64 mine = (u64)fp << (25u * P);
        g.compl_ += ((u64)popc64(sv.rowdone & mine) << (8u * (3u * P + 0u)))   // :270
                  + ((u64)popc64(sv.colordone & mine) << (8u * (3u * P + 1u))) // :278
                  + ((u64)popc64(sv.coldone & mine) << (8u * (3u * P + 2u)));  // :286
    }
    return cnt;
}

AZ_FN bool any_row_full(u32 w) { return ((w & (w >> 1) & (w >> 2) & (w >> 3) & (w >> 4)) & 0x108421u) != 0u; }
AZ_FN bool walls_end_game(const Game &g) { return any_row_full(g.wall0) || any_row_full(g.wall1); }   // azul.py:184-191
AZ_FN bool is_end_of_game(const Game &g) { return g.over != 0u; }

template <bool LID>
AZ_FN void count_score(Game &g, const LaneConst &k)
{
    // azul.py:291-295 for both players, then the full lines are emptied (:218)
    u64 F = full_lines(g, k);
    ScoreVec sv;
    score_vec(g, k, F, sv);
    i32 p0 = floor_penalty(g.floor0), p1 = floor_penalty(g.floor1);
    g.score0 = clamp0(g.score0 + p0 + wall_points<true, LID, 0>(g, F, sv));
    g.score1 = clamp0(g.score1 + p1 + wall_points<true, LID, 1>(g, F, sv));
    g.fpen = (((u32)((i32)(int16_t)(g.fpen & 0xffffu) + p0)) & 0xffffu) |
             ((((u32)((i32)(int16_t)(g.fpen >> 16) + p1)) & 0xffffu) << 16);   // :208
    g.floor0 = g.floor1 = 0;                             // :209
    g.cp = sel(g.cp == k.rowp1, splat(0u), g.cp);        // :218
    g.wc0 = g.wc1 = 0; g.wi0 = g.score0; g.wi1 = g.score1;
    g.over = walls_end_game(g) ? 1u : 0u;
}

// What-if scores (game_runner.py:48-50: deepcopy + count_score).  A move only changes the MOVER's lines and
// floor; the wall pricing of a player's full lines (wc) only changes when one more of his lines becomes full.
AZ_FN i32 wall_points_of(const ScoreVec &sv, u64 F, u32 p)
{
    // sum of the placement values of player p's full lines (no commits): one readlane per full line
    u32 rest = (u32)(F >> (25u * p)) & 0x1ffffffu;
    i32 cnt = 0;
    while (rest) {
        u32 i = ctz32(rest);
        rest &= rest - 1u;
        cnt += (i32)readlane(sv.val, i + 25u * p);
    }
    return cnt;
}

template <bool LID>
AZ_FN void whatif_refresh(Game &g, const LaneConst &k, u32 which /* 0, 1, or 2 = both */)
{
    u64 F = full_lines(g, k);
    ScoreVec sv;
    score_vec(g, k, F, sv);
    if (which == 2u) {
        g.wc0 = wall_points_of(sv, F, 0);
        g.wc1 = wall_points_of(sv, F, 1);
    } else {
        i32 w = wall_points_of(sv, F, which);
        g.wc0 = which ? g.wc0 : w;
        g.wc1 = which ? w : g.wc1;
    }
    g.wi0 = clamp0(g.score0 + floor_penalty(g.floor0) + g.wc0);
    g.wi1 = clamp0(g.score1 + floor_penalty(g.floor1) + g.wc1);
}

// derived fields after a record was loaded
template <bool LID>
AZ_FN void game_prime(Game &g, const LaneConst &k)
{
    g.over = walls_end_game(g) ? 1u : 0u;
    whatif_refresh<LID>(g, k, 2u);
}

template <bool LID>
AZ_FN i32 potential(Game &g, const LaneConst &k)
{
    whatif_refresh<LID>(g, k, 2u);
    return g.wi0 - g.wi1;
}

// ---- new_round: azul.py:64-89 ----
AZ_FN u32 byte_sum5(u64 v) { return (u32)(((v & 0xffffffffffull) * 0x0101010101ull) >> 32) & 0xffu; }

// The factory draw itself (azul.py:71-89): the centre receives the first-player token, five displays receive four tiles each.
// Shared by the two-player core below and by the 3/4-player core (azul_core_np.hpp): the reference deals 5 displays whatever
// the number of players (azul.py:19, TODO at tests/test_azul.py:14).
template <bool LID>
AZ_FN u32 deal_factories(vu32 &cs, u64 &box, u64 &lid, Rng &r)
{
    vu32 l = lane();
    cs = sel(l == 30u, splat(1u), splat(0u));            // :71,:73
    if (!LID) {
#pragma unroll 1
        for (u32 t = 0; t < 20u; t++) {
            u32 color = rng_below(r, 5u, 3u);            // :78 randrange(0,5,1)
            cs = cs + sel(l == (t >> 2) * 5u + color, splat(1u), splat(0u));   // :88
        }
        return ST_OK;
    }
    // "Lid" pool (azul.py:79-89): every draw is one random.choices over weights box_c / total -> exactly one
    // random() = two MT words, so the 20 draws consume words pos..pos+39.  When they do not straddle a
    // regeneration, all 40 are tempered by the lanes up front (lane t: the 53-bit integer K of draw t,
    // random() == K / 2^53); otherwise word by word.
    //
    // Deciding a draw.  CPython returns  #{c < 4 : cum_c <= x}  with cum_c the left-to-right fp64 sum of
    // fl(box_j / total) and x = fl(random() * cum_4).  In exact arithmetic that is  P_c / T <= K / 2^53, i.e.
    // P_c * 2^53 <= K * T  (P_c = box_0 + .. + box_c, T = total).  All fp64 roundings together move cum_c and x by
    // less than 21.1 * 2^-53 (five quotients, four sums, one product, all <= 1 + 2^-50), so the fp64 decision can
    // differ from the exact one only if |K*T - P_c*2^53| <= 21.1 * T <= 5381.  P_c * 2^53 is a multiple of 2^32: when
    // no multiple of 2^32 lies within AZ_DRAW_MARGIN = 8192 of K*T the integer comparison IS CPython's answer;
    // otherwise (about 4 draws in a million) the draw is decided by the literal fp64 computation below.
    if (AZ_UNLIKELY(r.dirty & 2u)) rng_stage(r);                 // lazily attached stream: the words are read from LDS below
    const bool batched = r.pos + 40u <= 624u;
    vu32 klo = splat(0u), khi = splat(0u);
    if (batched) {
        vu32 wa = temper_v(lds_ld(r.lds, r.pos + 2u * l, l < 20u)) >> 5, wb = temper_v(lds_ld(r.lds, r.pos + 2u * l + 1u, l < 20u)) >> 6;
        klo = (wa << 26) | wb;
        khi = wa >> 6;
    }
    u64 P = ((box & 0xffffffffffull) * 0x0101010101ull) & 0xffffffffffull;     // byte c = box_0 + .. + box_c
    // When the box holds at least 20 tiles no refill can happen: draw t sees total T0 - t, so K*T and the margin
    // test of all 20 draws are evaluated by the lanes before the sequential loop.
    const u32 T0 = (u32)(P >> 32) & 0xffu;
    const bool pre = batched && T0 >= 20u;
    vu32 kthi = splat(0u);
    u64 risky = 0;
    if (pre) {
        vu32 tt = T0 - l;                                 // valid for l < 20
        vu32 lo = klo * tt, hi = khi * tt + vmulhi(klo, tt);
        u32 mg = (u32)r.margin;                           // margin < 2^31
        // a multiple of 2^32 within `margin` of K*T  <=>  lo < margin  or  lo >= 2^32 - margin
        risky = ballot(((lo < mg) | (lo >= 0u - mg)) & (l < 20u));
        kthi = hi;
    }
    if (pre && !AZ_UNLIKELY(risky != 0ull)) {
        // the common round: twenty branch-free integer draws, one display (four draws) per loop trip
        u32 plo = (u32)P;                                  // prefix sums of colours 0..3 (the only ones compared)
        vu32 sh = (l & 3u) * 8u;
#pragma unroll 1
        for (u32 d = 0; d < 5u; d++) {
#pragma unroll
            for (u32 j = 0; j < 4u; j++) {
                vu32 pc = (plo >> sh) & 0xffu;
                u32 color = popc64(ballot(((pc << 21) <= readlane(kthi, d * 4u + j)) & (l < 4u)));
                box -= 1ull << (8u * color);               // :89
                plo -= (u32)(0x0101010101ull << (8u * color));   // colours >= `color` lose one tile from their prefix (colour 4: no-op)
                cs = cs + sel(l == d * 5u + color, splat(1u), splat(0u));   // :88
            }
        }
        r.pos += 40u;
        return ST_OK;
    }
#pragma unroll 1
    for (u32 t = 0; t < 20u; t++) {
        u32 total = (u32)(P >> 32) & 0xffu;
        if (AZ_UNLIKELY(total == 0u)) {                                              // :81-83, :85
            box = lid; lid = 0;
            P = ((box & 0xffffffffffull) * 0x0101010101ull) & 0xffffffffffull;
            total = (u32)(P >> 32) & 0xffu;
            if (total == 0u) return ST_BOX_EMPTY;
        }
        u32 Klo, Khi;
        if (batched) { Klo = readlane(klo, t); Khi = readlane(khi, t); r.pos += 2u; }
        else { u32 a27 = rng_u32(r) >> 5, b26 = rng_u32(r) >> 6; Klo = (a27 << 26) | b26; Khi = a27 >> 6; }
        u64 KT = (u64)Klo * total + (((u64)Khi * total) << 32);
        u32 color;
        if (!AZ_UNLIKELY(((KT - r.margin) >> 32) != ((KT + r.margin) >> 32))) {
            vu32 pc = ((u32)P >> ((l & 3u) * 8u)) & 0xffu;
            color = popc64(ballot(((pc << 21) <= (u32)(KT >> 32)) & (l < 4u)));
        } else {
            // weights = box_c / total (fp64), cumulative left-to-right, x = random() * cum[-1]   (:87, choices)
            u32 blo = (u32)box, bhi = (u32)(box >> 32);
            vu32 mine = sel(l < 4u, (blo >> ((l & 3u) * 8u)) & 0xffu, splat(bhi & 0xffu));
            vf64 wq = divlanes(mine, (double)total);
            double c0 = readlane_d(wq, 0);
            double c1 = c0 + readlane_d(wq, 1);
            double c2 = c1 + readlane_d(wq, 2);
            double c3 = c2 + readlane_d(wq, 3);
            double c4 = c3 + readlane_d(wq, 4);
            double u = ((double)Khi * 4294967296.0 + (double)Klo) * (1.0 / 9007199254740992.0);
            double x = u * (c4 + 0.0);
            color = (u32)!(x < c0) + (u32)!(x < c1) + (u32)!(x < c2) + (u32)!(x < c3);    // bisect_right(cum, x, 0, 4)
        }
        box -= 1ull << (8u * color);                   // :89
        P -= (0x0101010101ull << (8u * color)) & 0xffffffffffull;
        cs = cs + sel(l == (t >> 2) * 5u + color, splat(1u), splat(0u));   // :88
    }
    return ST_OK;
}

template <bool LID>
AZ_FN u32 new_round(Game &g, Rng &r)
{
    g.cur = g.nfp;
    g.fps += (g.nfp == 1u) ? 1u : 0x10000u;              // :67 (numpy [-1] == player 2 when nfp == 0)
    g.turn += 1u;
    g.nfp = 0;
    return deal_factories<LID>(g.cs, g.box, g.lid, r);
}

// ---- Azul.__init__ + GameRunner reset bookkeeping: azul.py:18-61, game_runner.py:76-82 ----
template <bool LID>
AZ_FN void game_ctor(Game &g, u32 first_player, Rng &r)
{
    g.cs = splat(0u); g.cp = splat(0u);
    g.wall0 = g.wall1 = 0; g.score0 = g.score1 = 0; g.floor0 = g.floor1 = 0;
    g.cur = 0; g.eog = 0; g.turn = 0;
    g.fps = 0; g.fpen = 0; g.maxc = 0; g.compl_ = 0;
    g.wc0 = g.wc1 = 0; g.wi0 = g.wi1 = 0; g.over = 0;
    if (first_player == 0u) g.nfp = 1u + rng_below(r, 2u, 2u);           // random.choice([1,2]) (:37)
    else g.nfp = first_player;
    if (LID) { g.box = 0x1414141414ull; g.lid = 0; }                      // :51-52
    else { g.box = 0; g.lid = 0; }
}

template <bool LID>
AZ_FN u32 episode_reset(Game &g, u32 first_player, Rng &r)
{
    game_ctor<LID>(g, first_player, r);
    u32 st = new_round<LID>(g, r);
    g.pscore = 0;
    g.moves = 0;
    return st;
}

// ---- step: azul.py:296-313 (legality is checked by the caller against the mask) ----
// move + end-of-round bookkeeping; returns true when a new round has to be dealt (azul.py:304-313)
template <bool LID>
AZ_FN bool move_and_score(Game &g, const LaneConst &k, u32 code, SegProf *prof_ = nullptr)
{
    (void)prof_;
    u32 me = me_index(g);
    bool filled = do_move<LID>(g, code);                 // :304
    AZ_STAMP(SEG_MOVE);
    if (sources_board(g) == 0u) {                        // :306 (the token counts)
        count_score<LID>(g, k);                          // :307
        AZ_STAMP(SEG_SCORE);
        if (is_end_of_game(g)) { g.eog = 1; return false; }   // :308-309
        return true;                                     // :311
    }
    if (filled) whatif_refresh<LID>(g, k, me);
    else {
        i32 w = clamp0((me ? g.score1 : g.score0) + floor_penalty(me ? g.floor1 : g.floor0) + (me ? g.wc1 : g.wc0));
        g.wi0 = me ? g.wi0 : w;
        g.wi1 = me ? w : g.wi1;
    }
    g.cur = (g.cur < 2u) ? g.cur + 1u : 1u;              // :313 next_player
    return false;
}

template <bool LID>
AZ_FN u32 apply_step(Game &g, const LaneConst &k, Rng &r, u32 code)
{
    if (move_and_score<LID>(g, k, code)) return new_round<LID>(g, r);
    return ST_OK;
}

template <bool LID>
AZ_FN u32 checked_step(Game &g, const LaneConst &k, Rng &r, i32 a)
{
    if (g.eog) return ST_GAME_ENDED;                     // :298-299
    if (a < 0 || a >= 180) return ST_BAD_ACTION;
    Mask m;
    legal_mask(g, k, m);
    if (!mask_test(m, (u32)a)) return ST_ILLEGAL_MOVE;   // :301-302, state untouched
    return apply_step<LID>(g, k, r, action_code((u32)a));
}

// ---- GameRunner.step with the default RandomAgent opponent: game_runner.py:43-55 ----
template <bool LID>
AZ_FN u32 runner_opponent_loop(Game &g, const LaneConst &k, Rng &r, const SampleTab &T, bool until_player1_only)
{
#pragma unroll 1
    for (u32 guard = 0; guard < 4096u; guard++) {
        Mask m;
        legal_mask(g, k, m);
        bool keep = until_player1_only ? (g.cur != 1u)
                                       : ((g.cur != 1u || mask_count(m) < 2u) && !is_end_of_game(g));   // :46 / :84
        if (!keep) break;
        u32 code;
        i32 a = random_agent(m, r, T, k, code);          // opponent_move, :37-42
        if (a < 0) return ST_STUCK;
        if (g.eog) return ST_GAME_ENDED;
        u32 st = apply_step<LID>(g, k, r, code);
        if (st) return st;
        g.moves += 1u;
    }
    return ST_OK;
}

template <bool LID>
AZ_FN u32 runner_step(Game &g, const LaneConst &k, Rng &r, const SampleTab &T, i32 a, i32 &reward, u32 &done)
{
    reward = 0;
    done = is_end_of_game(g) ? 1u : 0u;
    u32 st = checked_step<LID>(g, k, r, a);              // :44
    if (st) return st;
    g.moves += 1u;                                       // :45
    st = runner_opponent_loop<LID>(g, k, r, T, false);   // :46-47
    if (st) return st;
    i32 phi = potential<LID>(g, k);                      // :48-50
    reward = phi - g.pscore;                             // :51
    g.pscore = phi;                                      // :52
    done = is_end_of_game(g) ? 1u : 0u;                  // :55
    return ST_OK;
}

// ---- observation: game_runner.py:56-72 (136 integers; written as f32 for the policy net) ----
AZ_FN void observe(const Game &g, u32 persp, float *out)
{
    vu32 l = lane();
    u32 o0 = persp & 1u, o1 = o0 ^ 1u;
    u32 pnfp = g.nfp > 0u ? (((g.nfp - 1u - o0) & 1u) + 1u) : 0u;     // :58-61
    u32 wall_a = o0 ? g.wall1 : g.wall0, wall_b = o0 ? g.wall0 : g.wall1;
    u32 floor_a = o0 ? g.floor1 : g.floor0, floor_b = o0 ? g.floor0 : g.floor1;
    i32 score_a = o0 ? g.score1 : g.score0, score_b = o0 ? g.score0 : g.score1;
    // j = 0..63: displays+centre (cs lanes 0..30), then pattern_lines[order]
    {
        vu32 j = l;
        vu32 k = j - 31u;                                              // pattern cell index 0..32
        vu32 src = sel(k < 25u, k + 25u * o0, k - 25u + 25u * o1);
        vu32 v = sel(j < 31u, g.cs, bperm(g.cp, src));
        st_f32(out, j, v, j < 64u);
    }
    // j = 64..127: rest of pattern_lines[order[1]] (cells 8..24), then walls[order]
    {
        vu32 j = l + 64u;
        vu32 pc = j - 56u;                                             // cell of order[1] for j <= 80
        vu32 pv = bperm(g.cp, pc + 25u * o1);
        vu32 wb = j - 81u;                                             // wall bit index 0..46 (two boards of 25)
        vu32 wsel = sel(wb < 25u, splat(wall_a), splat(wall_b));
        vu32 wv_ = (wsel >> (sel(wb < 25u, wb, wb - 25u) & 31u)) & 1u;
        st_f32(out, j, sel(j <= 80u, pv, wv_), j < 128u);
    }
    // j = 128..135: last three wall bits of order[1], floors, scores, next first player
    {
        vu32 j = l + 128u;
        vu32 v = (wall_b >> ((l + 22u) & 31u)) & 1u;
        v = sel(l == 3u, splat(floor_a), v);
        v = sel(l == 4u, splat(floor_b), v);
        v = sel(l == 5u, splat((u32)score_a), v);
        v = sel(l == 6u, splat((u32)score_b), v);
        v = sel(l == 7u, splat(pnfp), v);
        st_f32(out, j, v, l < 8u);
    }
}

// ---- get_statistics: azul.py:314-315 (ten values per finished game, key order of game_runner.py:12) ----
AZ_FN double game_stat(const Game &g, u32 q)
{
    double f0 = (double)(g.fps & 0xffffu), f1 = (double)(g.fps >> 16);
    switch (q) {
    case 0: return (double)g.score0;
    case 1: return (double)g.score1;
    case 2: return (double)g.turn;
    case 3: return f0 / (f0 + f1) * 100;
    case 4: return -(double)(i32)(int16_t)(g.fpen & 0xffffu);
    case 5: return (double)(g.maxc & 0xffu);
    case 6: return (double)(g.compl_ & 0xffu);
    case 7: return (double)((g.compl_ >> 16) & 0xffu);
    case 8: return (double)((g.compl_ >> 8) & 0xffu);
    default: return g.score0 > g.score1 ? 1.0 : 0.0;
    }
}

// ---- random.seed(int): CPython init_by_array over the 32-bit words of the seed (one stream per THREAD) ----
AZ_FN void seed_stream(u32 *mt, u64 seed)
{
    u32 key0 = (u32)(seed & 0xffffffffu), key1 = (u32)(seed >> 32);
    u32 len = key1 ? 2u : 1u;
    u32 prev = 19650218u;
    mt[0] = prev;
    for (u32 i = 1; i < 624u; i++) { prev = 1812433253u * (prev ^ (prev >> 30)) + i; mt[i] = prev; }   // init_genrand
    u32 i = 1, j = 0;
    prev = mt[0];
    for (u32 k = 624u; k; k--) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + (j ? key1 : key0) + j;
        mt[i] = prev;
        i++; j++;
        if (i >= 624u) { mt[0] = prev; i = 1; }
        if (j >= len) j = 0;
    }
    for (u32 k = 623u; k; k--) {
        prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - i;
        mt[i] = prev;
        i++;
        if (i >= 624u) { mt[0] = prev; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// ---- one env move of flat random-agent self-play (the benchmarked step) ----
struct Counters {
    u64 *episodes;      // [1] finished games of this slot
    u32 *stuck;         // [1] hazard-H3 resets of this slot
    double *stat_sum;   // [10] get_statistics() summed over finished games
};

// returns 0 = move played, 1 = game ended with this move, 2 = stuck, 0x100|status on a rule error.
// Trajectory streams of the self-play kernel.  OUT == 1 (every stream present, the benchmarked form) keeps them as
// per-lane pointers; OUT == 2 (any subset, run-time checked, plus the test-only record stream) as scalar pointers.
struct OutV {
    vptr p32;    // lane 1: reward[t][g], lane 2: packed[t][g], every other lane: action[t][g]
    vptr p8;     // done[t][g]
    vptr p64;    // maskbits[t][g][min(lane, 2)]
    vptr pm;     // mask[t][g][lane]  (bytes 0..63; bytes 64..127 through the +64 immediate)
    vptr pm2;    // mask[t][g][128 + min(lane, 51)]
    u32 s32, s8, s64, sm;   // byte strides between consecutive moves
};

AZ_FN void outv_open(OutV &o, u32 gi, u32 n, uint8_t *mask, u64 *maskbits, i32 *action, i32 *reward, uint8_t *done, u32 *packed)
{
    vu32 l = lane();
    o.p32 = vptr_sel(l == 1u, vptr_splat(reward + gi), vptr_sel(l == 2u, vptr_splat(packed + gi), vptr_splat(action + gi)));
    o.p8 = vptr_splat(done + gi);
    o.p64 = vptr_off(vptr_splat(maskbits + (size_t)gi * 3), vmin(l, 2u) * 8u);
    o.pm = vptr_off(vptr_splat(mask + (size_t)gi * 180), l);
    o.pm2 = vptr_off(vptr_splat(mask + (size_t)gi * 180 + 128), vmin(l, 51u));
    o.s32 = n * 4u; o.s8 = n; o.s64 = n * 24u; o.sm = n * 180u;
}

AZ_FN void outv_next(OutV &o)
{
    o.p32 = vptr_add(o.p32, o.s32); o.p8 = vptr_add(o.p8, o.s8); o.p64 = vptr_add(o.p64, o.s64);
    o.pm = vptr_add(o.pm, o.sm); o.pm2 = vptr_add(o.pm2, o.sm);
}

struct OutS { uint8_t *mask; u64 *maskbits; i32 *action; i32 *reward; uint8_t *done; uint8_t *rec; u32 *packed; };

// compact trajectory record of one move (what the multi-GPU all-gather ships): action (0xff = none) | done << 8 | reward << 16
AZ_FN u32 pack_move(i32 a, u32 dn, i32 reward) { return ((u32)(a >= 0 ? a : 0xff) & 0xffu) | ((dn & 0xffu) << 8) | (((u32)reward & 0xffffu) << 16); }

// per-move trajectory outputs (action, reward, done, compact record; record snapshot in the test form)
template <int OUT>
AZ_FN void selfplay_outputs(const Game &g, const OutV &ov, const OutS &os, i32 a, i32 reward, u32 dn)
{
    if (OUT == 1) {
        vu32 l = lane();
        vst_u32(ov.p32, sel(l == 1u, splat((u32)reward), sel(l == 2u, splat(pack_move(a, dn, reward)), splat((u32)(a >= 0 ? a : -1)))));
        vst_u8(ov.p8, splat(dn));
    } else if (OUT == 2) {
        if (os.action) stu_i32(os.action, a >= 0 ? a : -1);
        if (os.reward) stu_i32(os.reward, reward);
        if (os.done) stu_u8(os.done, dn);
        if (os.packed) stu_i32((i32 *)os.packed, (i32)pack_move(a, dn, reward));
        if (os.rec) game_store(g, os.rec);
    }
}

// OUT: 0 = no trajectory outputs, 1 = all five streams through OutV, 2 = any subset through OutS (run-time checks)
// returns 0 = move played, 1 = game ended with this move, 2 = stuck, 0x100|status on a rule error.
template <bool LID, int OUT>
AZ_FN u32 selfplay_step(Game &g, u32 first_player, const LaneConst &k, Rng &r, const SampleTab &T, const Counters &cnt,
                        const OutV &ov, const OutS &os, SegProf *prof_ = nullptr)
{
    (void)prof_;
    AZ_STAMP(SEG_LOOP);
    Mask m;
    legal_mask(g, k, m);
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
    AZ_STAMP(SEG_MASK);
    u32 code = 0;
    i32 a = g.eog ? -2 : random_agent(m, r, T, k, code);
    AZ_STAMP(SEG_SAMPLE);
    // Straight-line common path (a move, sometimes followed by a new round); the two rare exits (stuck slot,
    // finished game) restart the episode out of line.
    if (AZ_UNLIKELY(a < 0)) {
        // stuck (or handed an already finished game): report, restart the slot
        AZ_LANE0(*cnt.stuck += 1u);
        selfplay_outputs<OUT>(g, ov, os, -1, 0, 2u);
        game_ctor<LID>(g, first_player, r);               // GameRunner.reset(): Azul(rules) ... new_round()
        g.pscore = 0;
        g.moves = 0;
        u32 st0 = new_round<LID>(g, r);
        return st0 ? (0x100u | st0) : 2u;
    }
    bool deal = move_and_score<LID>(g, k, code, prof_);
    g.moves += 1u;
    AZ_STAMP(SEG_AFTERMOVE);
    u32 st = ST_OK;
    if (deal) { st = new_round<LID>(g, r); AZ_STAMP(SEG_NEWROUND); }
    i32 phi = g.wi0 - g.wi1;
    i32 reward = phi - g.pscore;
    g.pscore = phi;
    u32 dn = is_end_of_game(g) ? 1u : 0u;
    selfplay_outputs<OUT>(g, ov, os, a, reward, dn);
    AZ_STAMP(SEG_TAIL);
    if (AZ_UNLIKELY(st != ST_OK)) return 0x100u | st;
    if (AZ_UNLIKELY(dn != 0u)) {
        for (u32 q = 0; q < 10u; q++) { double sv = game_stat(g, q); AZ_LANE0(cnt.stat_sum[q] += sv); }
        AZ_LANE0(*cnt.episodes += 1ull);
        game_ctor<LID>(g, first_player, r);               // GameRunner.reset(): Azul(rules) ... new_round()
        g.pscore = 0;
        g.moves = 0;
        st = new_round<LID>(g, r);
        AZ_STAMP(SEG_RESET);
        if (st) return 0x100u | st;
    }
    return dn;
}

} // namespace az
