// azul_selfplay2.hpp -- THE TWO-PLAYER RULE BOOK of libazulhip.so and the flat random-agent self-play step built on it (BASELINE
// configs[1], the benchmarked hot path), for TWO GAMES PER 64-LANE WAVEFRONT: lanes 0..31 play one game, lanes 32..63 another.
// Included by azul_kernels.hip; azul_env2.hpp (GameRunner's step / reset / observation), azul_ops2.hpp (every two-player rule entry of
// the C ABI) and azul_rules_x.hpp (3 / 4 players, extended rules) build on the functions below.
//
// Why two games per wave: with one game per wave (rounds 1-4 kept such a core; LABNOTES.md) every per-game quantity is wave-uniform,
// the compiler keeps it in SGPRs and the move is bound by the SCALAR pipe: one scalar ALU per CU issues ~1 instruction every 4.3
// cycles per SIMD, the vector ALUs one every ~2.4 (tools/issue_model.hip, profiles/round2_issue_model.txt).
// Here every per-game quantity lives in a VGPR, replicated across the game's 32 lanes ("half-uniform"), so the rules run
// on the vector pipe, one instruction serves two games, and the scalar pipe only carries the loop and the exec masks of
// the rare paths (factory draw, scoring, episode reset), which are ordinary divergent branches between the two halves.
//
// Layout inside a half (l = lane & 31):
//   cs        lane 5d+c = displays[d][c] (0..24), lane 25+c = center[c], lane 30 = token          (31 cells)
//   cp0, cp1  lane 5r+c = pattern_lines[p][r][c]                                                   (25 cells each)
//   everything else: half-uniform values (walls as 25-bit boards, floors, scores, turn data, box / lid byte vectors ...)
// Cross-lane traffic stays inside a half: hb() = my half of a wave ballot (the half's cell array as a bitboard), hread() =
// ds_bpermute with the half's base lane (cell gather / broadcast), hsum() / hmax() = DPP reductions over 32 lanes.
// Legal-move mask: ONE WORD PER PATTERN ROW -- action a = d + 6 c + 30 r is bit l = d + 6 c (< 30) of word r, i.e. lane l of a half
// owns (display d, colour c) for all six rows: its source cell is the same in every word and a chosen action decodes into (row =
// word, lane's constants) without a table.
// MT19937: each game's 624 words in LDS (two regions per wave); the regeneration runs 32 lanes wide.
//
// Exactness arguments: DESIGN.md 4; the trajectories are byte-identical to the oracle (tests/test_gpu_selfplay.py,
// tests/test_full_size_configs.py).
// Reference lines: new_round azulnet/azul.py:64-89, move :118-161, is_legal_move :162-176, next_player / is_end_of_round /
// is_end_of_game :177-191, count_score :192-295, step :296-313; GameRunner.step / reset azulnet/game_runner.py:43-55, 76-85,
// RandomAgent :87-97, check_all_valid :113-117; random.seed / random() / getrandbits / _randbelow / choices: CPython 3.10
// (_randommodule.c, random.py).
#pragma once
#include "azul_common.hpp"

namespace az2 {
using namespace az;

AZ_FN u32 wlane() { return wv::lane(); }
AZ_FN bool upper() { return (wlane() & 32u) != 0u; }
// my half's word of a wave ballot.  The select becomes ONE v_lshrrev_b64 by (lane & 32): a quarter-rate instruction (~8 cycles of the pipe),
// but the kernel is bound by the number of instructions a wave issues, not by pipe cycles -- the two full-rate instructions of
// (lo & mlo) | (hi & mhi) with per-lane masks measured 4.6 % SLOWER (round 3; LABNOTES.md)
AZ_FN u32 hsel(u64 b) { return upper() ? (u32)(b >> 32) : (u32)b; }
// the half's 32 lanes as a bitboard
AZ_FN u32 hb(bool p) { return hsel(__builtin_amdgcn_ballot_w64(p)); }
// value of lane `idx` of MY half, idx per lane (a gather through the LDS crossbar; idx in 0..31)
AZ_FN u32 hread(u32 v, u32 idx) { return (u32)__builtin_amdgcn_ds_bpermute((int)((idx << 2) | ((wlane() & 32u) << 2)), (int)v); }
// the same for a HALF-UNIFORM idx.  (Measured: the LDS crossbar beats a v_readlane pair per half + select -- two instructions and one
// wait against seven instructions with SGPR hazards: 1.125 vs 1.195 ms per 512-move launch.)
AZ_FN u32 hbcast(u32 v, u32 idx) { return hread(v, idx); }
// the same with the half's byte offset handed in (K2::h4): one address instruction instead of two
AZ_FN u32 hread4(u32 v, u32 idx, u32 h4) { return (u32)__builtin_amdgcn_ds_bpermute((int)((idx << 2) | h4), (int)v); }
template <u32 IDX>
AZ_FN u32 hbcast_c(u32 v)
{
    u32 a = (u32)__builtin_amdgcn_readlane((int)v, (int)IDX), b = (u32)__builtin_amdgcn_readlane((int)v, (int)(IDX + 32u));
    return upper() ? b : a;
}
// is the predicate true in ANY lane of the wave?  (a scalar branch: no exec-mask bookkeeping for the rare paths)
AZ_FN bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
AZ_FN double hread_d(double v, u32 idx)
{
    u64 b = (u64)__double_as_longlong(v);
    u32 lo = hread((u32)b, idx), hi = hread((u32)(b >> 32), idx);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

template <int CTRL, int ROWMASK>
AZ_FN u32 dpp0(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true); }   // lanes without a source read 0
// sum / maximum over the 32 lanes of my half (every lane receives it): an all-reduce inside each 16-lane row with four DPP steps
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then gfx950's v_permlane16_swap exchanges the two rows of each half
// (odd rows of the first operand with even rows of the second): six instructions, no trip through the scalar registers
// (the round-2 form -- row_shr chain, row_bcast:15, two v_readlane, two moves, a select -- needed ten)
AZ_FN u32 hsum(u32 v)
{
    v += dpp0<0xB1, 0xf>(v); v += dpp0<0x4E, 0xf>(v); v += dpp0<0x141, 0xf>(v); v += dpp0<0x140, 0xf>(v);
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (u32)r[0] + (u32)r[1];
}
AZ_FN u32 umax(u32 a, u32 b) { return a > b ? a : b; }
AZ_FN u32 hmax(u32 v)
{
    v = umax(v, dpp0<0xB1, 0xf>(v)); v = umax(v, dpp0<0x4E, 0xf>(v)); v = umax(v, dpp0<0x141, 0xf>(v)); v = umax(v, dpp0<0x140, 0xf>(v));
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return umax((u32)r[0], (u32)r[1]);
}

// ---- per-lane constants -------------------------------------------------------------------------------------------------
struct K2 {
    u32 l;               // lane & 31
    u32 spos;            // lane l < 30 owns (display d = l % 6, colour c = l / 6): cs lane of that source cell (31 = none: bit 31 of the
                         // source board is always 0)
    u32 col6;            // c = l / 6 (0 for l >= 30)
    u32 lcode;           // (d, c) decoded once, LaneConst::acode's packing without row and action number
    u32 rowp1;           // l < 25: row + 1, else 0xff
    u32 prow, pcol, pbcol, pbelow, pcolboard;      // pattern cell l < 25: row, colour, board column, bits 0..l, cells of that board column
    u32 h4;              // byte offset of my half in a ds_bpermute address: (lane & 32) << 2
};

AZ_FN void k2_init(K2 &k)
{
    const u32 l = wlane() & 31u;
    k.l = l;
    k.h4 = (wlane() & 32u) << 2;
    asm volatile("" : "+v"(k.h4));     // opaque: (idx << 2) | h4 stays ONE v_lshl_or_b32 (else it is re-associated into or + shift)
    {
        const u32 d = l % 6u, c = l / 6u;
        const u32 sp = d == 0u ? c + 25u : (d - 1u) * 5u + c;
        const u32 db = d == 0u ? 0u : (d - 1u) * 5u;
        k.spos = l < 30u ? sp : 31u;
        k.col6 = l < 30u ? c : 0u;
        k.lcode = sp | (db << 5) | (c << 10) | ((d == 0u ? 0u : 1u) << 16);
    }
    k.rowp1 = l < 25u ? l / 5u + 1u : 0xffu;
    u32 i = l < 25u ? l : 0u;
    k.prow = i / 5u;
    k.pcol = i % 5u;
    u32 bc = k.prow + k.pcol;
    k.pbcol = bc >= 5u ? bc - 5u : bc;
    k.pbelow = (2u << i) - 1u;
    k.pcolboard = k.pbcol == 0u ? column_board_c(0) : k.pbcol == 1u ? column_board_c(1) : k.pbcol == 2u ? column_board_c(2)
                : k.pbcol == 3u ? column_board_c(3) : column_board_c(4);
}

// ---- game state: every field is a per-lane value; all but cs / cp0 / cp1 are half-uniform -------------------------------
struct G2 {
    u32 cs, cp0, cp1;
    u32 wall0, wall1, floor0, floor1;
    i32 score0, score1;
    u32 cur, nfp, eog, turn, fps;
    // statistics, one register per player (packed into the record's bytes by g2_store only: scoring adds to them as they are)
    i32 fp0, fp1;               // floor_penalty[p], low 16 bits stored
    u32 mc0, mc1;               // max_combo[p], low byte stored
    u32 cl0, cl1;               // completed_lines[p][0..2], one byte each (24 bits stored)
    u64 box, lid;
    u32 lidp;                   // per lane: tiles of colour l (< 5) that scoring has returned to the lid since the last fold (lid_fold2)
    i32 pscore;
    u32 moves;
    // what-if cache (game_runner.py:48-50: deepcopy + count_score after every move).  A move only changes the MOVER's lines and floor, and
    // the wall pricing of a player's full lines only changes when one more of his lines becomes full: wc = points the player's currently
    // FULL pattern lines would earn (count_wall), wi = what-if score max(0, score + floor penalty + wc).  Derived, never stored.
    i32 wc0, wc1, wi0, wi1;
    u32 over;
    u32 B;                      // derived: sources holding tiles, hb(cs != 0) & 0x7fffffff (refreshed whenever cs changes)
    u32 ok0, ok1;               // derived: the players' "row r accepts colour c" boards (bit 5r + c), see ok_board2
};

AZ_FN u32 me2(const G2 &g) { return g.cur == 0u ? 1u : g.cur - 1u; }

AZ_FN void g2_load(G2 &g, const uint8_t *rec, u32 l)
{
    u32 a = rec[l];
    u32 b0 = l < 25u ? (u32)rec[32u + l] : 0u;
    u32 b1 = l < 27u ? (u32)rec[57u + l] : 0u;                     // pattern_lines[1] (25 cells), floors[2]
    u32 t = l < 11u ? ((const u32 *)(rec + 84))[l] : 0u;
    u32 flags = hread(a, 31);
    g.cur = flags & 7u; g.nfp = (flags >> 3) & 7u; g.eog = (flags >> 6) & 1u;
    g.cs = l < 31u ? a : 0u;
    g.cp0 = b0;
    g.cp1 = l < 25u ? b1 : 0u;
    g.floor0 = hread(b1, 25); g.floor1 = hread(b1, 26);
    g.wall0 = hread(t, 0); g.wall1 = hread(t, 1);
    u32 sc = hread(t, 2);
    g.score0 = (i32)(int16_t)(sc & 0xffffu); g.score1 = (i32)(int16_t)(sc >> 16);
    u32 w3 = hread(t, 3), w4 = hread(t, 4), w5 = hread(t, 5);
    g.box = (u64)w3 | ((u64)(w4 & 0xffu) << 32);
    g.lid = (u64)(w4 >> 8) | ((u64)(w5 & 0xffffu) << 24);
    g.turn = w5 >> 16;
    g.fps = hread(t, 6);
    { const u32 fpen = hread(t, 7); g.fp0 = (i32)(int16_t)(fpen & 0xffffu); g.fp1 = (i32)(int16_t)(fpen >> 16); }
    u32 w8 = hread(t, 8), w9 = hread(t, 9), w10 = hread(t, 10);
    g.mc0 = w8 & 0xffu; g.mc1 = (w8 >> 8) & 0xffu;
    g.cl0 = (w8 >> 16) | ((w9 & 0xffu) << 16); g.cl1 = w9 >> 8;
    g.pscore = (i32)(int16_t)(w10 & 0xffffu);
    g.moves = w10 >> 16;
    g.wc0 = g.wc1 = 0; g.wi0 = g.wi1 = 0; g.over = 0;
    g.lidp = 0;
    g.B = hb(g.cs != 0u) & 0x7fffffffu;
}

AZ_FN u64 lid_fold2(u32 lidp, u32 l);

AZ_FN void g2_store(const G2 &g, uint8_t *rec, u32 l)
{
    u32 flags = (g.cur & 7u) | ((g.nfp & 7u) << 3) | ((g.eog & 1u) << 6);
    rec[l] = (uint8_t)(l == 31u ? flags : g.cs);
    if (l < 25u) rec[32u + l] = (uint8_t)g.cp0;
    // (three separate stores: a nested select between struct fields would be folded into a select of ADDRESSES and keep the
    // whole game state in scratch memory)
    if (l < 25u) rec[57u + l] = (uint8_t)g.cp1;
    if (l == 25u) rec[82] = (uint8_t)g.floor0;
    if (l == 26u) rec[83] = (uint8_t)g.floor1;
    const u64 lid = g.lid + lid_fold2(g.lidp, l);        // (the tally scoring keeps per lane, folded into the record's bytes)
    u32 t = 0;
    t = l == 0u ? g.wall0 : t;
    t = l == 1u ? g.wall1 : t;
    t = l == 2u ? (((u32)g.score0 & 0xffffu) | ((u32)g.score1 << 16)) : t;
    t = l == 3u ? (u32)g.box : t;
    t = l == 4u ? ((u32)((g.box >> 32) & 0xffu) | ((u32)lid << 8)) : t;
    t = l == 5u ? ((u32)((lid >> 24) & 0xffffu) | (g.turn << 16)) : t;
    t = l == 6u ? g.fps : t;
    t = l == 7u ? (((u32)g.fp0 & 0xffffu) | ((u32)g.fp1 << 16)) : t;
    t = l == 8u ? ((g.mc0 & 0xffu) | ((g.mc1 & 0xffu) << 8) | (g.cl0 << 16)) : t;
    t = l == 9u ? (((g.cl0 >> 16) & 0xffu) | (g.cl1 << 8)) : t;
    t = l == 10u ? (((u32)g.pscore & 0xffffu) | (g.moves << 16)) : t;
    if (l < 11u) ((u32 *)(rec + 84))[l] = t;
}

// ---- CPython MT19937 stream of one game ---------------------------------------------------------------------------------------
// The 624-word state lives in global memory (row of a [N][624] array: a half's accesses are contiguous); it is staged into the game's LDS
// region when the stream is opened, the regeneration ("twist") runs there 32 lanes wide, and it is written back on close if it changed.
constexpr u32 MT_LDS_WORDS = 626;   // LDS words per game: the 624 MT19937 words + the batch's move limit (word 624; ~0u = none) + a pad word.  The limit
                                    // is only ever read in the RARE end-of-round blocks: kept in LDS it occupies no register in a move loop
struct Rng2 {
    u32 *lds;        // my game's 624 words in LDS (+ the move limit at word 624: rng2_set_move_limit)
    u32 *tlds;       // optional: the same 624 words TEMPERED (kept current by the regeneration), what genrand_uint32 returns for index i;
                     // the self-play loop reads a move's two words from here with one LDS read and no arithmetic
    u32 pos;         // CPython's `index`
    u32 dirty;       // a regeneration happened: LDS differs from global memory
    u32 wbase, wend; // the window `win` serves words wbase .. wend-1 (wend == 0: none loaded)
    u32 win;         // lane l: TEMPERED word wbase + l
};

// the batch's move limit (BatchDev::move_limit: 0 = none) into / out of the game's LDS region; written once per kernel by lane 0 of the half
AZ_FN void rng2_set_move_limit(u32 *lds, u32 limit, u32 l) { if (l == 0u) lds[624] = limit ? limit : ~0u; }
AZ_FN u32 rng2_move_limit(const Rng2 &r) { return r.lds[624]; }

AZ_FN u32 temper2(u32 y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

AZ_FN void lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

AZ_FN void rng2_open(Rng2 &r, const u32 *gmt, u32 *lds, u32 pos, u32 l)
{
    r.lds = lds; r.tlds = nullptr; r.pos = pos; r.dirty = 0; r.wbase = 0; r.wend = 0; r.win = 0;
    u32 w[20];
#pragma unroll
    for (u32 q = 0; q < 20u; q++) { u32 i = l + 32u * q; w[q] = i < 624u ? gmt[i] : 0u; }     // all loads in flight, then the LDS writes
#pragma unroll
    for (u32 q = 0; q < 20u; q++) { u32 i = l + 32u * q; if (i < 624u) lds[i] = w[q]; }
    lds_sync();
}

// attach the tempered copy (624 more words of LDS per game) and fill it from the words rng2_open staged
AZ_FN void rng2_attach_tempered(Rng2 &r, u32 *tlds, u32 l)
{
    r.tlds = tlds;
#pragma unroll 1
    for (u32 q = 0; q < 20u; q++) { u32 i = l + 32u * q; if (i < 624u) tlds[i] = temper2(r.lds[i]); }
    lds_sync();
}

// genrand_uint32's regeneration, ascending 32-wide chunks: element i needs OLD mt[i], OLD mt[i+1] (NEW mt[0] for i == 623) and
// (i < 227: OLD mt[i+397], 12.4 chunks ahead | i >= 227: NEW mt[i-227], 7.09 chunks back) -- so the sources of a chunk are never
// written inside its own GROUP of seven chunks: a group's 21 LDS reads are issued together and its writes follow in one go
// (three groups, six fences, instead of twenty dependent read -> write round trips; the words are the same).
// One rolled loop over the groups: the code is inlined at every site that can run into the end of the state.
AZ_FN void rng2_twist(Rng2 &r, u32 l)
{
#pragma unroll 1
    for (u32 q0 = 0; q0 < 21u; q0 += 7u) {
        u32 v[7];
#pragma unroll
        for (u32 q = 0; q < 7u; q++) {
            const u32 i = l + (q0 + q) * 32u;
            const u32 ic = i < 624u ? i : 623u;              // (chunk 19 is half empty, "chunk 20" empty: those lanes repeat element 623 and write nothing)
            const u32 i1 = ic == 623u ? 0u : ic + 1u;
            const u32 i2 = ic < 227u ? ic + 397u : ic - 227u;
            const u32 a = r.lds[ic], b = r.lds[i1], c = r.lds[i2];
            const u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
            v[q] = c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        lds_sync();
#pragma unroll
        for (u32 q = 0; q < 7u; q++) {
            const u32 i = l + (q0 + q) * 32u;
            if (i < 624u) { r.lds[i] = v[q]; if (r.tlds) r.tlds[i] = temper2(v[q]); }
        }
        lds_sync();
    }
    r.dirty = 1;
    r.pos = 0;
    r.wend = 0;
}

AZ_FN void rng2_refill(Rng2 &r, u32 l)
{
    if (r.pos >= 624u) rng2_twist(r, l);
    r.wbase = r.pos;
    r.wend = r.pos + 32u < 624u ? r.pos + 32u : 624u;
    u32 i = l + r.pos;
    r.win = temper2(i < 624u ? r.lds[i] : 0u);
}

AZ_FN u32 rng2_u32(Rng2 &r, u32 l)
{
    if (r.pos >= r.wend) rng2_refill(r, l);
    u32 y = hbcast(r.win, r.pos - r.wbase);
    r.pos += 1u;
    return y;
}

// random_random(): (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53, exact
AZ_FN double rng2_random(Rng2 &r, u32 l)
{
    u32 a = 0, b = 0;
    if (r.pos + 2u <= r.wend) {
        u32 off = r.pos - r.wbase;
        a = hbcast(r.win, off);
        b = hbcast(r.win, off + 1u);
        r.pos += 2u;
    } else {
        // (ONE call site for both words: the regeneration is inlined wherever a word is fetched)
#pragma unroll 1
        for (u32 w = 0; w < 2u; w++) { a = b; b = rng2_u32(r, l); }
    }
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

AZ_FN u32 rng2_below(Rng2 &r, u32 n, u32 bits, u32 l)
{
    u32 v;
    do { v = rng2_u32(r, l) >> (32u - bits); } while (v >= n);
    return v;
}

AZ_FN void rng2_close(Rng2 &r, u32 *gmt, u32 *pos_out, u32 l)
{
    if (r.dirty) {
#pragma unroll 1
        for (u32 q = 0; q < 20u; q++) { u32 i = l + 32u * q; if (i < 624u) gmt[i] = r.lds[i]; }
    }
    if (l == 0u) *pos_out = r.pos;
}

// ---- legal-move mask: six 32-bit words per game ---------------------------------------------------------------------------
struct Mask2 {
    u32 m[6];        // half-uniform: bit l (< 30) of word r = action 30 r + l is legal
    u32 bit[6];      // the same bit for MY action of each word (what the byte mask stores)
    u32 B;           // sources holding tiles (31 bits; bit 30 = the token)
};

// "row r accepts colour c" (azul.py:171-175) for one player as a 25-bit board: no OTHER colour lies on the row and the wall cell is
// free.  Walls only change at scoring and a move changes ONE row of the mover (afterwards that row holds the moved colour only), so
// the boards are kept in the game state: rebuilt here after loading / scoring / a reset, patched arithmetically by do_move2.
AZ_FN u32 ok_board2(u32 cp, u32 wall, const K2 &k)
{
    // cell (r, c) accepts colour c when row r holds no OTHER colour and the wall cell is free -- written as ONE compare: a ballot of
    // an and / or of compares goes through a 0 / 1 register and a second compare (5 instructions instead of 3)
    const u32 l = k.l;
    const u32 pme = hb(cp != 0u) & 0x1ffffffu;
    const u32 rb = (pme >> (k.prow * 5u)) & 31u;
    const u32 bad = (rb & ~(1u << k.pcol)) | ((wall >> l) & 1u);
    return hb(bad == 0u) & 0x1ffffffu;
}

AZ_FN void legal_mask2(const G2 &g, const K2 &k, Mask2 &out)
{
    const u32 B = g.B;
    const u32 has = (B >> k.spos) & 1u;                          // my source holds tiles of my colour (azul.py:164-169)
    const u32 okc = (me2(g) ? g.ok1 : g.ok0) >> k.col6;          // bit 5 (r - 1): pattern row r accepts my colour (:171-175)
    out.B = B;
    out.bit[0] = has;                                            // the floor "row" accepts everything
    out.m[0] = hb(has != 0u);
#pragma unroll
    for (u32 w = 1; w < 6u; w++) {
        out.bit[w] = has & (okc >> (5u * (w - 1u)));
        out.m[w] = hb((out.bit[w] & 1u) != 0u);
    }
}

// ---- RandomAgent: game_runner.py:87-97 + random.choices (random.py:506-541) ---------------------------------------------------
// The cumulative weight CPython's accumulate() reaches after J weights of 0.01 (legal floor moves, a < 30) and m weights of 1.0 is
// T(J, 0) = S[J]  and, for m >= 1,  T(J, m) = m + Fr[J][floor(log2 m)]  EXACTLY: adding 1.0 only rounds when the sum enters a new binade,
// so per J there are 8 distinct fractional parts (azul_tables.hpp builds both tables with the very additions CPython performs; the host
// checks all 31 * 151 sums when a batch is created).  bisect_right over the cumulative weights == the smallest ordinal k with cum(k) > x.
// Pattern moves: |Fr - S[J]| < 2^-44 (a handful of half-ulp roundings below 256) and d = x - S[J] carries an fp64 error below 2^-45, so
// whenever d is further than 1e-9 from an integer, floor(d) + 1 IS the ordinal; otherwise the exact table values decide (sample_slow2).
// Floor-only masks (m == 0): the ordinal is floor(random() * fl(100 S[J])) + 1 on the same margin (azul_tables.hpp: the ninth pair of a row).
struct Tab2 { const double2 *fs; };      // LDS: the pairs {Fr[J][b], S[J]} at 9 J + b, b < 8, and the floor-only pair at 9 J + 8 (azul_tables.hpp: build_sample_pairs): both table values of a decision in one 16-byte read

AZ_FN double tpat2(const Tab2 &t, u32 J, u32 m) { return (double)m + t.fs[9u * J + 31u - (u32)__builtin_clz(m)].x; }      // T(J, m), m >= 1
AZ_FN double tseq2(const Tab2 &t, u32 J, u32 kk) { return kk <= J ? t.fs[9u * kk].y : tpat2(t, J, kk - J); }              // cumulative weight after the kk-th legal action

// bisect_right over the cumulative weights == smallest ordinal kk with cum(kk) > x, for the draws the one-compare fast path leaves (x inside
// the 0.01-weight floor moves, a guess too close to an integer boundary, the clamp at the last weight).  A first guess -- floor(100 x) + 1
// inside the floor moves, J + floor(x - S[J]) + 1 above them -- and, unless it is known to be exact, a walk along the cumulative weights
// until cum(kg - 1) <= x < cum(kg) (the weights are positive: the answer is unique; clamped to L like bisect's hi = n - 1).  Exact without
// the walk: x < S[J] and 100 x further than 1e-9 from an integer -- S[k] is k additions of 0.01, |100 S[k] - k| < 1e-13 for k <= 60, and
// x * 100 rounds by < 1e-13.  (Floor-ONLY masks, M == 0, do not come here any more unless their draw sits at a boundary: the callers decide
// them in the one-compare path from the table's ninth pair -- a game whose every move is a floor move, hazard H9, used to take this
// function on every decision and its wave ran ~10 % slower than the others; a launch lasts as long as its slowest wave.)
AZ_FN u32 sample_slow2(const Tab2 &T, double x, double sJ, u32 J, u32 M, u32 L)
{
    const bool floors = x < sJ;
    const double y = floors ? x * 100.0 : x - sJ;
    const u32 fy = (u32)y;
    const u32 cap = floors ? J : M;
    const u32 g = fy + 1u > cap ? cap : fy + 1u;
    u32 kg = floors ? g : J + g;
    const bool exact = floors & (__builtin_fabs((y - (double)fy) - 0.5) < 0.5 - 1e-9);
    for (u32 it = exact ? 256u : 0u; it < 256u; it++) {
        if (kg > 1u && x < tseq2(T, J, kg - 1u)) kg -= 1u;
        else if (kg < L && !(x < tseq2(T, J, kg))) kg += 1u;
        else break;
    }
    return kg;
}

// ---- move: azul.py:118-161.  Returns whether the targeted pattern line is full afterwards ------------------------------------
template <bool LID>
AZ_FN bool do_move2f(G2 &g, u32 src, u32 db, u32 c, u32 row, bool from_display, u32 B /* sources before the move */, const K2 &k)
{
    const u32 l = k.l;
    const u32 me = me2(g);
    // the three cell gathers of a move depend on the chosen action only: requested together (ONE LDS round trip on the move's chain)
    const u32 cell = 5u * ((row ? row : 1u) - 1u) + c;
    u32 mine = me ? g.cp1 : g.cp0;
    const u32 n = hread4(g.cs, src, k.h4);                             // :127 / :136
    const u32 moved = hread4(g.cs, l - 25u + db, k.h4);                // :131
    const u32 old = hread4(mine, cell, k.h4);
    bool token = (!from_display) & (((B >> 30) & 1u) != 0u);           // :140
    bool centre = (l >= 25u) & (l < 30u) & (l != 25u + c) & from_display;
    bool gone = ((l >= db) & (l < db + 5u) & from_display) | (l == src) | ((l == 30u) & token);   // :129,:133,:138,:141
    const u32 grown = g.cs + (centre ? moved : 0u);                    // (a plain sum: no branch around the gathered value)
    g.cs = gone ? 0u : grown;
    g.nfp = token ? g.cur : g.nfp;                                     // :142
    u32 fl = (me ? g.floor1 : g.floor0) + (token ? 1u : 0u);           // :143 (the cap of :120-123 is applied once, below: it is monotone)
    i32 overflow = row ? (i32)row - (i32)old - (i32)n : -(i32)n;       // :147
    u32 spill = overflow < 0 ? (u32)(-overflow) : 0u;
    u32 newv = overflow < 0 ? row : old + n;                           // :150 / :152
    mine = ((l == cell) & (row != 0u)) ? newv : mine;
    g.cp0 = me ? g.cp0 : mine;
    g.cp1 = me ? mine : g.cp1;
    {   // the row now holds colour c only (the move was legal: the row was empty or held c, the wall cell is free)
        const u32 sh = 5u * ((row ? row : 1u) - 1u);
        u32 okm = me ? g.ok1 : g.ok0;
        okm = row ? ((okm & ~(31u << sh)) | (1u << (sh + c))) : okm;
        g.ok0 = me ? g.ok0 : okm;
        g.ok1 = me ? okm : g.ok1;
    }
    fl += spill;                                                       // :154 / :159
    fl = fl < 7u ? fl : 7u;
    g.floor0 = me ? g.floor0 : fl;
    g.floor1 = me ? fl : g.floor1;
    if (LID) g.lid += (u64)spill << (8u * c);                          // :156-157 / :160-161
    return (row != 0u) & (overflow <= 0);
}

template <bool LID>
AZ_FN bool do_move2(G2 &g, u32 code, u32 B /* sources before the move */, const K2 &k)
{
    return do_move2f<LID>(g, code & 31u, (code >> 5) & 31u, (code >> 10) & 7u, (code >> 13) & 7u, ((code >> 16) & 1u) != 0u, B, k);
}

// ---- wall pricing (count_wall, azul.py:211-290) for one player in lanes 0..24 -----------------------------------------------------
// EVERY pattern cell (lane = row, colour) prices "a tile placed here now" against the player's wall plus the full lines that are scored
// before it (ascending row, colour: K2::pbelow), so the sequential dependency of the reference's loop is reproduced without a loop.
struct Score2 { u32 val, pos; u32 rowdone, colordone, coldone; };

AZ_FN u32 run_length2(u32 bits, u32 pos)
{
    // ones at and above pos, plus the ones directly below it: the highest ZERO below pos decides; a sentinel bit under the shifted
    // zero map makes "no zero below" the same formula (no compare + select around count-leading-zeros of 0)
    const u32 up = (u32)__builtin_ctz(~(bits >> pos));
    const u32 below = ~bits & ((1u << pos) - 1u);
    const u32 down = pos + (u32)__builtin_clz((below << 1) | 1u) - 31u;
    return up + down;
}

AZ_FN void score2(u32 w /* per lane: the wall the placement is priced against */, const K2 &k, Score2 &s)
{
    u32 rowbits = (w >> (k.prow * 5u)) & 31u;
    u32 h = ((rowbits << k.prow) | (rowbits >> (5u - k.prow))) & 31u;
    u32 hr = run_length2(h, k.pbcol);                                  // azul.py:230-242
    u32 wc = w & k.pcolboard;
    u32 t = wc | (wc >> 1) | (wc >> 2) | (wc >> 3) | (wc >> 4);
    u32 v = (((t & 0x108421u) * 0x111110u) >> 20) & 31u;
    u32 vr = run_length2(v, k.prow);                                   // :244-257
    u32 both = hr + vr;
    s.pos = both - ((hr < vr ? hr : vr) > 1u ? 0u : 1u);                                  // :258-263 (1 + 1 - 1 is the lone tile's point)
    bool rd = rowbits == 31u, cd = ((w >> k.pcol) & 0x108421u) == 0x108421u, kd = v == 31u;
    s.val = s.pos + (rd ? 2u : 0u) + (cd ? 10u : 0u) + (kd ? 7u : 0u);                    // :266-288
    s.rowdone = hb(rd); s.colordone = hb(cd); s.coldone = hb(kd);
}

// sum of the placement values of one player's full lines F (25 bits); no commits (the what-if of game_runner.py:48-50)
AZ_FN i32 wall_points2(u32 wall, u32 F, const K2 &k)
{
    Score2 s;
    score2(wall | (F & k.pbelow), k, s);
    return (i32)hsum(((F >> k.l) & 1u) ? s.val : 0u);
}

AZ_FN u32 full_lines2(u32 cp, const K2 &k) { return hb(cp == k.rowp1) & 0x1ffffffu; }    // azul.py:216

AZ_FN void whatif_scores2(G2 &g)
{
    g.wi0 = clamp0(g.score0 + floor_penalty(g.floor0) + g.wc0);
    g.wi1 = clamp0(g.score1 + floor_penalty(g.floor1) + g.wc1);
}

AZ_FN void prime2(G2 &g, const K2 &k)
{
    g.over = (any_row_full(g.wall0) | any_row_full(g.wall1)) ? 1u : 0u;
    g.wc0 = wall_points2(g.wall0, full_lines2(g.cp0, k), k);
    g.wc1 = wall_points2(g.wall1, full_lines2(g.cp1, k), k);
    whatif_scores2(g);
    g.ok0 = ok_board2(g.cp0, g.wall0, k);
    g.ok1 = ok_board2(g.cp1, g.wall1, k);
}

// Σ_r r * [line (r, c) is full] for my colour c = l (< 5): the tiles the lid receives (azul.py:220-222; rows 1, 3 weigh one bit, rows 2, 3
// two, row 4 four).  Scoring only ADDS this to a per-lane tally (G2::lidp); the byte vector the record holds is formed when the lid is
// looked at -- a round that starts with an empty box, a record store -- by lid_fold2: lanes 0..3 are packed into the low word with two
// DPP steps, lane 4 is the fifth byte (two cross-lane broadcasts that a round end no longer pays for).
AZ_FN u32 lid_tally2(u32 F, u32 l)
{
    const u32 t = (F >> (l < 5u ? l : 0u)) & 0x108421u;                      // bits 5 r of colour l
    return (u32)__popc(t & 0x8020u) + 2u * (u32)__popc(t & 0x8400u) + 4u * (u32)__popc(t & 0x100000u);
}
AZ_FN u64 lid_fold2(u32 lidp, u32 l)
{
    u32 v = l < 4u ? lidp << (8u * l) : 0u;
    v += dpp0<0x111, 0xf>(v);                                               // row_shr:1
    v += dpp0<0x112, 0xf>(v);                                               // row_shr:2  -> lane 3 holds bytes 0..3 (sums below 256 each)
    return (u64)hbcast_c<3>(v) | ((u64)hbcast_c<4>(lidp) << 32);
}

// count_wall + count_floor for one player (azul.py:200-290), committed
template <bool LID>
AZ_FN void count_player2(u32 &wall, u32 &cp, u32 &floor_, i32 &score, u32 &maxc8, u32 &compl24, i32 &fpen16, u32 &lidp, const K2 &k)
{
    // (no "any full line?" branch: with F == 0 every term below is zero, and without the branch the two players' chains sit in ONE
    // basic block, where the scheduler interleaves them -- the kernel is bound by dependent-issue latency)
    const u32 F = full_lines2(cp, k);
    Score2 s;
    score2(wall | (F & k.pbelow), k, s);
    const bool on = ((F >> k.l) & 1u) != 0u;
    const i32 cnt = (i32)hsum(on ? s.val : 0u);                                        // :289
    maxc8 = umax(maxc8, hmax(on ? s.pos : 0u));                                        // :264
    compl24 += (u32)__popc(s.rowdone & F) + ((u32)__popc(s.colordone & F) << 8) + ((u32)__popc(s.coldone & F) << 16);   // :270,:278,:286
    if (LID) lidp += lid_tally2(F, k.l);                                                    // :220-222
    wall |= F;                                                                         // :219
    cp = (cp == k.rowp1) ? 0u : cp;                                                    // :218
    i32 pen = floor_penalty(floor_);
    fpen16 += pen;                                                                     // :208
    floor_ = 0;                                                                        // :209
    score = clamp0(score + pen + cnt);                                                 // :292-295
}

template <bool LID>
AZ_FN void count_score2(G2 &g, const K2 &k)
{
    count_player2<LID>(g.wall0, g.cp0, g.floor0, g.score0, g.mc0, g.cl0, g.fp0, g.lidp, k);
    count_player2<LID>(g.wall1, g.cp1, g.floor1, g.score1, g.mc1, g.cl1, g.fp1, g.lidp, k);
    // (byte-wise counters: no carries between them in the reference's floats either -- each stays below 256 for real games)
    g.wc0 = g.wc1 = 0; g.wi0 = g.score0; g.wi1 = g.score1;
    g.over = (any_row_full(g.wall0) | any_row_full(g.wall1)) ? 1u : 0u;
    g.ok0 = ok_board2(g.cp0, g.wall0, k);                // walls and lines changed
    g.ok1 = ok_board2(g.cp1, g.wall1, k);
}

// ---- new_round: azul.py:64-89 ------------------------------------------------------------------------------------------------------
// "Lid" pool (azul.py:79-89): every draw is one random.choices over weights box_c / total -> exactly one random() = two MT words.
// Deciding a draw.  CPython returns  #{c < 4 : cum_c <= x}  with cum_c the left-to-right fp64 sum of fl(box_j / total) and
// x = fl(random() * cum_4).  In exact arithmetic that is  P_c / T <= K / 2^53, i.e.  P_c * 2^53 <= K * T  (P_c = box_0 + .. + box_c,
// T = total, K = the 53-bit integer of random()).  All fp64 roundings together move cum_c and x by less than 21.1 * 2^-53 (five quotients,
// four sums, one product, all <= 1 + 2^-50), so the fp64 decision can differ from the exact one only if |K*T - P_c*2^53| <= 21.1 * T <=
// 5381.  P_c * 2^53 is a multiple of 2^32: when no multiple of 2^32 lies within AZ_DRAW_MARGIN = 8192 of K*T the integer comparison IS
// CPython's answer; otherwise (about 4 draws in a million) the draw is decided by the literal fp64 computation.
template <bool LID>
AZ_FN u32 deal2(G2 &g, Rng2 &r, u64 margin, const K2 &k);

template <bool LID>
AZ_FN u32 new_round2(G2 &g, Rng2 &r, u64 margin, const K2 &k)
{
    u32 st = deal2<LID>(g, r, margin, k);
    g.B = hb(g.cs != 0u) & 0x7fffffffu;
    return st;
}

// n <= 32 consecutive "Lid" draws of a round at once, when the box holds at least n tiles (no refill can happen inside): az2::deal_tiles2's parallel fixed point for any number of draws -- lane t owns draw t,
//     colour_t = #{c < 4 : (P_c - n_c(t)) * 2^53 <= K_t * (T0 - t)},   n_c(t) = #{s < t : colour_s <= c},
// iterated from a first guess until nothing changes (the unique fixed point is the sequential result) -- with the exactness
// argument above (a draw whose K * T lies within `margin` of a multiple of 2^32 sends the n draws through the literal fp64 code, one after
// the other, on the words already fetched) and the same handling of a regeneration inside the 2 n words.
// XC: cells of the second cell register (displays 5 ..: azul_rules_x.hpp; 0 = the reference's five displays, cs1 unused).
template <u32 XC>
AZ_FN void deal_batch2(u32 &cs0, u32 &cs1, u64 &box, Rng2 &r, u64 margin, u32 t0 /* first draw of the batch within the round */, u32 n, const K2 &k)
{
    const u32 l = k.l;
    const u32 nmask = n >= 32u ? 0xffffffffu : (1u << n) - 1u;
    const u32 blo = (u32)box;
    const u32 p0 = blo & 0xffu, p1 = p0 + ((blo >> 8) & 0xffu), p2 = p1 + ((blo >> 16) & 0xffu), p3 = p2 + (blo >> 24);
    const u32 T0 = p3 + ((u32)(box >> 32) & 0xffu);
    const bool room = r.pos + 2u * n <= 624u;
    const u32 lc = l < n ? l : n - 1u;
    const u32 i0 = r.pos + 2u * lc, i1 = i0 + 1u;
    const u32 j0 = i0 < 624u ? i0 : i0 - 624u, j1 = i1 < 624u ? i1 : i1 - 624u;
    const u32 *words = r.tlds ? r.tlds : r.lds;
    u32 wa = words[j0], wb = words[j1];
    if (!room) {
        const u32 p = r.pos;
        lds_sync();
        rng2_twist(r, l);
        const u32 na = words[j0], nb = words[j1];
        wa = i0 < 624u ? wa : na; wb = i1 < 624u ? wb : nb;
        r.pos = p - 624u;                                // (wraps; the 2 n words bring it to p + 2 n - 624)
    }
    if (!r.tlds) { wa = temper2(wa); wb = temper2(wb); }
    wa >>= 5; wb >>= 6;
    const u32 klo = (wa << 26) | wb, khi = wa >> 6;      // K_t, the 53-bit integer of random() (random() == K / 2^53)
    const u32 tt = T0 - l;                               // draw t sees T0 - t tiles (valid for l < n <= T0)
    const u32 lo = klo * tt, kthi = khi * tt + __umulhi(klo, tt);
    const u32 mg = (u32)margin;
    const u32 risky = hb(lo + mg < 2u * mg) & nmask;
    u32 col = 0;
    if (risky == 0u) {
        const u32 below = (1u << l) - 1u;
        const u32 kt0hi = khi * T0 + __umulhi(klo, T0);  // first guess: the colour drawn from the undepleted box
        col = (u32)((p0 << 21) <= kt0hi) + (u32)((p1 << 21) <= kt0hi) + (u32)((p2 << 21) <= kt0hi) + (u32)((p3 << 21) <= kt0hi);
#pragma unroll 1
        for (u32 it = 0; it < 34u; it++) {
            const u32 b0 = hb(col == 0u) & nmask, b1 = hb(col <= 1u) & nmask, b2 = hb(col <= 2u) & nmask, b3 = hb(col <= 3u) & nmask;
            const u32 n0 = (u32)__popc(b0 & below), n1 = (u32)__popc(b1 & below), n2 = (u32)__popc(b2 & below), n3 = (u32)__popc(b3 & below);
            const u32 nc = (u32)(((p0 - n0) << 21) <= kthi) + (u32)(((p1 - n1) << 21) <= kthi) + (u32)(((p2 - n2) << 21) <= kthi) +
                           (u32)(((p3 - n3) << 21) <= kthi);
            const u32 changed = hb(nc != col) & nmask;
            col = nc;
            if (changed == 0u) break;
        }
    } else {
        // the same draws one after the other on the fetched words, each decided by the integer comparison or, inside the margin, by the literal fp64 code
        u64 Pp = ((box & 0xffffffffffull) * 0x0101010101ull) & 0xffffffffffull;
        u64 bx = box;
#pragma unroll 1
        for (u32 t = 0; t < n; t++) {
            const u32 total = T0 - t;
            const u32 Klo = hread(klo, t), Khi = hread(khi, t);
            const u64 KT = (u64)Klo * total + (((u64)Khi * total) << 32);
            u32 color;
            if (((KT - margin) >> 32) == ((KT + margin) >> 32)) {
                const u32 pc = ((u32)Pp >> ((l & 3u) * 8u)) & 0xffu;
                color = (u32)__popc(hb(((pc << 21) <= (u32)(KT >> 32)) & (l < 4u)));
            } else {
                const double tot = (double)total;
                const double q0 = (double)((u32)bx & 0xffu) / tot, q1 = (double)((u32)(bx >> 8) & 0xffu) / tot,
                             q2 = (double)((u32)(bx >> 16) & 0xffu) / tot, q3 = (double)((u32)(bx >> 24) & 0xffu) / tot,
                             q4 = (double)((u32)(bx >> 32) & 0xffu) / tot;
                const double c0 = q0, c1 = c0 + q1, c2 = c1 + q2, c3 = c2 + q3, c4 = c3 + q4;
                const double u = ((double)Khi * 4294967296.0 + (double)Klo) * (1.0 / 9007199254740992.0);
                const double x = u * (c4 + 0.0);
                color = (u32)!(x < c0) + (u32)!(x < c1) + (u32)!(x < c2) + (u32)!(x < c3);
            }
            bx -= 1ull << (8u * color);
            Pp -= (0x0101010101ull << (8u * color)) & 0xffffffffffull;
            col = l == t ? color : col;
        }
    }
    // commit: the draws of each colour as boards, the box, the cells of displays d0 .. d0 + n / 4 - 1
    u32 e0 = hb(col == 0u) & nmask, e1 = hb(col == 1u) & nmask, e2 = hb(col == 2u) & nmask, e3 = hb(col == 3u) & nmask, e4 = hb(col == 4u) & nmask;
    asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4));
    const u32 ec = k.pcol == 0u ? e0 : k.pcol == 1u ? e1 : k.pcol == 2u ? e2 : k.pcol == 3u ? e3 : e4;
    // display d receives the round's draws 4 d .. 4 d + 3, i.e. bits 4 d - t0 .. of this batch (none when they lie outside it)
    {
        const u32 mine = (u32)(((u64)0xfu << (4u * k.prow)) >> t0) & nmask;                 // cs0: lane 5 d + c, display d = prow
        cs0 += l < 25u ? (u32)__popc(ec & mine) : 0u;
    }
    if (XC > 0u) {
        const u32 mine = (u32)(((u64)0xfu << (4u * (5u + k.prow))) >> t0) & nmask;          // cs1: lane 5 (d - 5) + c
        cs1 += l < XC ? (u32)__popc(ec & mine) : 0u;
    }
    const u32 take_lo = (u32)__popc(e0) | ((u32)__popc(e1) << 8) | ((u32)__popc(e2) << 16) | ((u32)__popc(e3) << 24);
    box -= (u64)take_lo | ((u64)(u32)__popc(e4) << 32);  // :89 (no borrows: a colour is only drawn while the box holds it)
    r.pos += 2u * n;
}

// The "Lid" pool's draws of a round (azul.py:79-89; NDRAWS = 4 per display) in batches of up to 32 -- as many as the box holds; an empty
// box is refilled from the lid (:81-83) and the dealing continues; box and lid both empty: ST_BOX_EMPTY (where the reference raises,
// :85-87) or -- short_deal, beyond the reference (row N4) -- the round starts with what could be dealt.  (Round 3 dealt a round whose
// box held fewer than 20 tiles one draw at a time: one round in four or five.)
template <u32 NDRAWS, u32 XC>
AZ_FN u32 deal_lid2(u32 &cs0, u32 &cs1, u64 &box, u64 &lid, u32 &lidp, Rng2 &r, u64 margin, bool short_deal, const K2 &k)
{
    u32 done = 0;
#pragma unroll 1
    while (done < NDRAWS) {
        u32 T = byte_sum5(box);
        if (T == 0u) {
            box = lid + lid_fold2(lidp, k.l); lid = 0; lidp = 0;
            T = byte_sum5(box);
            if (T == 0u) return short_deal ? (u32)ST_OK : (u32)ST_BOX_EMPTY;
        }
        u32 n = NDRAWS - done;
        n = n < 32u ? n : 32u;
        n = n < T ? n : T;
        deal_batch2<XC>(cs0, cs1, box, r, margin, done, n, k);
        done += n;
    }
    return ST_OK;
}

// The factory draw itself (azul.py:71-89) on the cell register of five displays + centre + token.
template <bool LID>
AZ_FN u32 deal_tiles2(u32 &cs, u64 &box, u64 &lid, u32 &lidp, Rng2 &r, u64 margin, const K2 &k)
{
    const u32 l = k.l;
    cs = l == 30u ? 1u : 0u;                             // :71,:73
    if (!LID) {
#pragma unroll 1
        for (u32 t = 0; t < 20u; t++) {
            u32 color = rng2_below(r, 5u, 3u, l);        // :78 randrange(0,5,1)
            cs += (l == (t >> 2) * 5u + color) ? 1u : 0u;   // :88
        }
        return ST_OK;
    }
    u32 none = 0;
    return deal_lid2<20u, 0u>(cs, none, box, lid, lidp, r, margin, false, k);
}

template <bool LID>
AZ_FN u32 deal2(G2 &g, Rng2 &r, u64 margin, const K2 &k)
{
    g.cur = g.nfp;
    g.fps += (g.nfp == 1u) ? 1u : 0x10000u;              // :67 (numpy [-1] == player 2 when nfp == 0)
    g.turn += 1u;
    g.nfp = 0;
    return deal_tiles2<LID>(g.cs, g.box, g.lid, g.lidp, r, margin, k);
}

// Azul.__init__ (azul.py:18-61): an empty board, the first player drawn or fixed, the box filled
template <bool LID>
AZ_FN void game_ctor2(G2 &g, u32 first_player, Rng2 &r, const K2 &k)
{
    g.cs = 0; g.cp0 = 0; g.cp1 = 0;
    g.wall0 = g.wall1 = 0; g.score0 = g.score1 = 0; g.floor0 = g.floor1 = 0;
    g.cur = 0; g.eog = 0; g.turn = 0;
    g.fps = 0; g.fp0 = g.fp1 = 0; g.mc0 = g.mc1 = 0; g.cl0 = g.cl1 = 0;
    g.wc0 = g.wc1 = 0; g.wi0 = g.wi1 = 0; g.over = 0;
    g.ok0 = g.ok1 = 0x1ffffffu;                          // empty lines, empty walls: every row accepts every colour
    if (first_player == 0u) g.nfp = 1u + rng2_below(r, 2u, 2u, k.l);     // random.choice([1, 2]) (:37)
    else g.nfp = first_player;
    if (LID) { g.box = 0x1414141414ull; g.lid = 0; }
    else { g.box = 0; g.lid = 0; }
    g.lidp = 0;
}

// Azul.__init__ + GameRunner's reset bookkeeping (game_runner.py:76-82), then the first round
template <bool LID>
AZ_FN u32 episode_reset2(G2 &g, u32 first_player, Rng2 &r, u64 margin, const K2 &k)
{
    game_ctor2<LID>(g, first_player, r, k);
    g.pscore = 0;
    g.moves = 0;
    return new_round2<LID>(g, r, margin, k);
}

AZ_FN double game_stat2(const G2 &g, u32 q)
{
    double f0 = (double)(g.fps & 0xffffu), f1 = (double)(g.fps >> 16);
    switch (q) {
    case 0: return (double)g.score0;
    case 1: return (double)g.score1;
    case 2: return (double)g.turn;
    case 3: return f0 / (f0 + f1) * 100;
    case 4: return -(double)(i32)(int16_t)((u32)g.fp0 & 0xffffu);
    case 5: return (double)(g.mc0 & 0xffu);
    case 6: return (double)(g.cl0 & 0xffu);
    case 7: return (double)((g.cl0 >> 16) & 0xffu);
    case 8: return (double)((g.cl0 >> 8) & 0xffu);
    default: return g.score0 > g.score1 ? 1.0 : 0.0;
    }
}

// ---- trajectory streams ------------------------------------------------------------------------------------------------------
// Base pointers of the time-major streams (wave-uniform: they stay in SGPRs) and ONE per-lane element index e = t * N + game,
// advanced by N per move: every address is base + e * element size (32-bit offsets: the host refuses streams of 4 GiB or more).
struct Out2 {
    uint8_t *mask;       // [T][N][pitch]: lane l writes bytes l, 32 + l, ... of its game's row
    u64 *maskbits;       // [T][N][3]
    i32 *action, *reward;
    uint8_t *done;
    u32 *packed;
    uint8_t *rec;        // test stream: the record after the move
    u32 pitch;           // bytes between the mask rows of consecutive games
    u32 e;
    u32 *dw3;            // OUT == 1: lane 0 -> action, lane 1 -> reward, lanes 2.. -> packed: the three 4-byte records of a move leave in ONE store
};

template <int OUT>
AZ_FN void outputs2(const G2 &g, const Out2 &o, i32 a, i32 reward, u32 dn, u32 l)
{
    if (OUT == 0) return;
    const i32 av = a >= 0 ? a : -1;
    if (OUT == 1) {
        // A store instruction costs this kernel ~45 cycles of a wave's time (ten times a vector instruction: measured by leaving
        // stores out), so the three 4-byte records of a move leave in ONE instruction -- lane 0 of the half writes the action, lane 1
        // the reward, the other lanes the compact record (same address, same data: no exec masking) -- and `done` in a second one.
        const u32 pk = pack_move(a, dn, reward);
        o.dw3[o.e] = l == 0u ? (u32)av : (l == 1u ? (u32)reward : pk);
        o.done[o.e] = (uint8_t)dn;
    } else {
        if (l == 0u) {
            if (o.action) o.action[o.e] = av;
            if (o.reward) o.reward[o.e] = reward;
            if (o.packed) o.packed[o.e] = pack_move(a, dn, reward);
            if (o.done) o.done[o.e] = (uint8_t)dn;
        }
        if (o.rec) g2_store(g, o.rec + o.e * (u32)AZUL_RECORD_BYTES, l);
    }
}

// The legal mask of a decision as the byte row the policy net consumes (byte a = 30 r + l of the game's row = bit l of word r).
// PAD (rows of >= 192 bytes, 8-byte aligned): the 180 bits are concatenated (the six 30-bit row words back to back), lane j < 23 of
// the half takes bits 8 j .. 8 j + 7, spreads them into eight 0 / 1 bytes (two 24-bit multiplies) and ONE 8-byte store per lane
// writes bytes 0 .. 183 of the row (180 .. 183 are padding; lanes 23.. repeat lane 22: no exec masking).  Otherwise: six stores
// of 30 bytes, byte 30 r + l from lane l.
template <bool PAD>
AZ_FN void store_mask_row2(const Out2 &o, u32 m0, u32 m1, u32 m2, u32 m3, u32 m4, u32 m5, u32 b0, u32 b1, u32 b2, u32 b3, u32 b4, u32 b5, u32 l)
{
    if (PAD) {
        // the 180 bits as six dwords (the 30-bit row words back to back); lane j's eight bits are byte j & 3 of dword j >> 2 -- a
        // byte-aligned field, picked with lane-constant masks (no lane-dependent control flow) and one bit-field extract
        u32 D0 = m0 | (m1 << 30), D1 = (m1 >> 2) | (m2 << 28), D2 = (m2 >> 4) | (m3 << 26), D3 = (m3 >> 6) | (m4 << 24),
            D4 = (m4 >> 8) | (m5 << 22), D5 = m5 >> 10;
        const u32 j = l < 22u ? l : 22u;
        // five v_cndmask on lane-constant conditions (the dwords pass through an empty asm: otherwise each is computed inside a branch of
        // the select, which turns into nested divergent branches)
        asm volatile("" : "+v"(D0), "+v"(D1), "+v"(D2), "+v"(D3), "+v"(D4), "+v"(D5));
        const u32 D = j < 4u ? D0 : j < 8u ? D1 : j < 12u ? D2 : j < 16u ? D3 : j < 20u ? D4 : D5;
        const u32 by = (D >> (8u * (j & 3u))) & 0xffu;
        const u32 lo = ((by & 15u) * 0x00204081u) & 0x01010101u, hi = ((by >> 4) * 0x00204081u) & 0x01010101u;
        *(u64 *)(o.mask + (o.e * o.pitch + 8u * j)) = (u64)lo | ((u64)hi << 32);
    } else {
        uint8_t *row = o.mask + (o.e * o.pitch + l);
        if (l < 30u) {
            row[0] = (uint8_t)b0; row[30] = (uint8_t)b1; row[60] = (uint8_t)b2; row[90] = (uint8_t)b3; row[120] = (uint8_t)b4; row[150] = (uint8_t)b5;
        }
    }
}

// Per-game counters of a launch (azul_batch_counters): finished episodes, stuck resets, and the sums of get_statistics() over finished
// games.  They live in REGISTERS while a kernel runs -- lane q < 10 of the half carries stat_sum[q], loaded when the kernel opens the
// game and written back when it closes it; the additions happen in the order the games end, so the sums are bit for bit what ten
// read-modify-writes per finished game gave (round 3), without their memory round trips on the episode-end path.
struct Counters2 {
    u64 *episodes; u32 *stuck; double *stat_sum;     // the game's slots in the batch's arrays
    double acc;                                      // lane q < 10: stat_sum[q] so far
    u32 ep_add, stuck_add;                           // half-uniform: this launch's increments
};


AZ_FN void counters2_open(Counters2 &c, u64 *episodes, u32 *stuck, double *stat_sum, u32 l)
{
    c.episodes = episodes; c.stuck = stuck; c.stat_sum = stat_sum;
    c.acc = l < 10u ? stat_sum[l] : 0.0;
    c.ep_add = 0; c.stuck_add = 0;
}

AZ_FN void counters2_close(const Counters2 &c, u32 l)
{
    if (l < 10u) c.stat_sum[l] = c.acc;
    if (l == 0u) { *c.episodes += (u64)c.ep_add; *c.stuck += c.stuck_add; }
}

// get_statistics() of a finished game, value q in lane q (azul.py:314-315, key order of game_runner.py:12): the integer statistics are
// picked with lane-constant selects and converted once; the percentage is the one true double
AZ_FN double stat_lane(u32 l, i32 score0, i32 score1, u32 turn, double pct_first, i32 fpen0, u32 mc0, u32 cl0)
{
    i32 v = score0;
    v = l == 1u ? score1 : v;
    v = l == 2u ? (i32)turn : v;
    v = l == 4u ? -(i32)(int16_t)((u32)fpen0 & 0xffffu) : v;
    v = l == 5u ? (i32)(mc0 & 0xffu) : v;
    v = l == 6u ? (i32)(cl0 & 0xffu) : v;
    v = l == 7u ? (i32)((cl0 >> 16) & 0xffu) : v;
    v = l == 8u ? (i32)((cl0 >> 8) & 0xffu) : v;
    v = l == 9u ? (score0 > score1 ? 1 : 0) : v;
    const double d = (double)v;
    return l == 3u ? pct_first : (l < 10u ? d : 0.0);
}

AZ_FN void counters2_episode(Counters2 &c, double stat_of_my_lane)
{
    c.acc = c.acc + stat_of_my_lane;
    c.ep_add += 1u;
}

// Everything of a move that follows do_move2 (g.B already holds the sources after the move): what-if score of the mover, next
// player or end of round (scoring, end of game, next round), shaped reward, outputs, episode statistics and reset.
// LIM: the batch has a move limit (azul_batch_set_move_limit).  It is a compile-time switch of the self-play kernel because ANY addition to
// this function's rare blocks measurably moves the common path (code placement, spill choices: +0.5 .. 1.5 % on the headline kernel,
// profiles/round6_headline_ab.txt): without a limit the kernel is, instruction for instruction, the one that has no such code.
template <bool LID, int OUT, bool LIM>
AZ_FN u32 after_move2(G2 &g, u32 first_player, const K2 &k, Rng2 &r, u64 margin, Counters2 &cnt, const Out2 &o, u32 me, bool filled,
                      i32 a, SegProf *prof_, bool &dead /* per game, only ever set: a rule error stopped it (set in the rare blocks: nothing on the common path) */)
{
    (void)prof_;
    const u32 l = k.l;
    // a move only changes the MOVER's lines and floor; the pricing of his full lines only when one more of them filled
    // (not when the move ends the round: count_score2 below prices the lines for real and clears the cache)
    const bool eor = g.B == 0u;                              // :306 is_end_of_round (the token counts)
    i32 wc = me ? g.wc1 : g.wc0;
    if (wave_any(filled & !eor)) {
        i32 fresh = wall_points2(me ? g.wall1 : g.wall0, full_lines2(me ? g.cp1 : g.cp0, k), k);
        wc = filled ? fresh : wc;
    }
    const i32 wi = clamp0((me ? g.score1 : g.score0) + floor_penalty(me ? g.floor1 : g.floor0) + wc);
    g.wc0 = me ? g.wc0 : wc; g.wc1 = me ? wc : g.wc1;
    g.wi0 = me ? g.wi0 : wi; g.wi1 = me ? wi : g.wi1;
    g.cur = eor ? g.cur : (g.cur & 1u) + 1u;    // :313 next_player
    u32 st = ST_OK;
    AZ_STAMP(SEG_AFTERMOVE);
    // (ONE wave-uniform test for the end of a round and for a game that is over -- a test costs ~33 cycles even when it falls through;
    // the episode-end block further down branches on a flag that is already scalar.  `over` without the end of a round: a state
    // handed in with a complete wall row and the flag clear)
    bool any_done = false;
    if (AZ_UNLIKELY(wave_any(eor | (g.over != 0u)))) {
        if (eor) {
            count_score2<LID>(g, k);                         // :307 (also resets the what-if cache)
            if (g.over) g.eog = 1;                           // :308-309
        }
        AZ_STAMP(SEG_SCORE);
        // MOVE LIMIT (beyond the reference, off unless azul_batch_set_move_limit: k.move_limit == ~0u): a round ended, the game did not, and
        // the episode has played its limit -- under the reference's rules a game can reach a state from which it NEVER ends (every tile of a
        // colour locked in lines that cannot be completed: no wall row can fill, azul.py:184-191 stays false): it is cut here, no round dealt
        // (the cut rides on `over` -- 0 / 1 from the walls, 3 here; the episode reset below clears it)
        if (LIM) { if (eor & !g.over) { if (g.moves >= rng2_move_limit(r)) g.over = 3u; } }
        if (eor & !g.over) st = new_round2<LID>(g, r, margin, k);      // :311
        AZ_STAMP(SEG_NEWROUND);
        any_done = wave_any((g.over != 0u) & (st == ST_OK));
        dead |= st != ST_OK;
    }
    const i32 phi = g.wi0 - g.wi1;
    const i32 reward = phi - g.pscore;
    g.pscore = phi;
    const u32 dn = LIM ? g.over : (g.over ? 1u : 0u);        // 0, 1 (a wall row is complete), 3 (cut by the move limit)
    outputs2<OUT>(g, o, a, reward, dn, l);
    AZ_STAMP(SEG_TAIL);
    u32 ret = st != ST_OK ? (0x100u | st) : dn;
    if (AZ_UNLIKELY(any_done)) {
        if ((dn != 0u) & (st == ST_OK)) {
            if (!LIM || dn == 1u) {
                const double f0 = (double)(g.fps & 0xffffu), f1 = (double)(g.fps >> 16);
                counters2_episode(cnt, stat_lane(l, g.score0, g.score1, g.turn, f0 / (f0 + f1) * 100, g.fp0, g.mc0, g.cl0));
            } else cnt.stuck_add += 1u;                      // (a cut episode is no finished game: counted with the restarted slots)
            u32 st2 = episode_reset2<LID>(g, first_player, r, margin, k);     // GameRunner.reset(): Azul(rules) ... new_round()
            if (st2) ret = 0x100u | st2;
        }
        dead |= (ret & 0x100u) != 0u;
        AZ_STAMP(SEG_RESET);
    }
    return ret;
}

// One env move of flat random-agent self-play for the two games of a wave.
// Returns 0 = move played, 1 = game ended with this move, 2 = stuck, 0x100 | status on a rule error.
//
// Control flow: two waves per SIMD cannot hide a taken branch's instruction refetch, so the common move is ONE fall-through path;
// every rare event (window refill across a regeneration, stuck slot, sampler boundary case, end of round, end of game) is tested
// for the whole wave with one scalar branch (wave_any, hinted unlikely -> placed out of line) and handled per half inside.
template <bool LID, int OUT, bool PAD, bool BITS, bool LIM = false>
AZ_FN u32 selfplay_step2(G2 &g, u32 first_player, const K2 &k, Rng2 &r, const Tab2 &T, u64 margin, Counters2 &cnt, const Out2 &o,
                        SegProf *prof_, bool &dead /* per game, only ever set: a rule error stopped it in this move */)
{
    (void)prof_;
    AZ_STAMP(SEG_LOOP);
    const u32 l = k.l;
    // -- the two MT19937 words of this move's random(), fetched speculatively (independent of the mask: overlaps with it): one
    //    8-byte read of the tempered copy; words that straddle / follow a regeneration are left to the rare path below
    const bool hard = r.pos + 2u > 624u;
    u32 wa, wb;
    {
        const u32 i = hard ? 622u : r.pos;
        wa = r.tlds[i]; wb = r.tlds[i + 1u];
    }

    Mask2 m;
    legal_mask2(g, k, m);
    // -- RandomAgent (game_runner.py:87-97): counts of the legal actions first -- they address the weight table, and that LDS read
    //    is on the move's chain: what does not need its answer (the mask row's bytes and stores) is placed between the request and
    //    its use (the scheduler does not know an LDS round trip costs ~60-100 cycles)
    const u32 c0 = __popc(m.m[0]), c1 = __popc(m.m[1]), c2 = __popc(m.m[2]), c3 = __popc(m.m[3]), c4 = __popc(m.m[4]), c5 = __popc(m.m[5]);
    const u32 J = c0;                                    // legal floor moves (row 0: a < 30, weight 0.01)
    const u32 p1 = c0, p2 = p1 + c1, p3 = p2 + c2, p4 = p3 + c3, p5 = p4 + c4, L = p5 + c5;
    const bool nomove = (L == 0u) | (g.eog != 0u);        // ValueError in the reference (raised before random()) / a finished game handed in
    const u32 M = L - J;
    // Floor-only masks (M == 0) take the same one-compare form from their own pair {fl(100 S[J]), 0} -- "binade 8": M = 0 counts as 256 --
    // with the ordinal counted from 0 instead of J: selects on counts that are known before the table answers, nothing added to the move's
    // chain (azul_tables.hpp).  (The index as  M ? 9 J + ilog2 M : 9 J + 8  -- a select between two sums -- ran 1.4 % slower.)
    const u32 Jc = J < 31u ? J : 30u;
    const u32 Mc = M ? M : 256u;
    const u32 kbase = M ? J : 0u;
    // (random()'s conversion sits BEFORE the request: the two words were asked for ~40 instructions ago, and a use of them after the
    // request would make the compiler wait for both reads there)
    const double u01 = ((double)(wa >> 5) * 67108864.0 + (double)(wb >> 6)) * (1.0 / 9007199254740992.0);      // random()
    __builtin_amdgcn_sched_barrier(0);
    const double2 fs = T.fs[9u * Jc + 31u - (u32)__builtin_clz(Mc)];      // {Fr[J][ilog2 M], S[J]} | {fl(100 S[J]), 0}
    __builtin_amdgcn_sched_barrier(0);
    if (OUT == 1 || (OUT == 2 && o.mask))
        store_mask_row2<(PAD && OUT == 1)>(o, m.m[0], m.m[1], m.m[2], m.m[3], m.m[4], m.m[5], m.bit[0], m.bit[1], m.bit[2], m.bit[3], m.bit[4], m.bit[5], l);
    if ((OUT == 1 && BITS) || (OUT == 2 && o.maskbits)) {
        // the same 180 bits packed (bit a & 63 of word a >> 6): the six 30-bit row words concatenated
        const u64 q0 = (u64)m.m[0] | ((u64)m.m[1] << 30) | ((u64)m.m[2] << 60);
        const u64 q1 = ((u64)m.m[2] >> 4) | ((u64)m.m[3] << 26) | ((u64)m.m[4] << 56);
        const u64 q2 = ((u64)m.m[4] >> 8) | ((u64)m.m[5] << 22);
        // every lane stores (lanes 3.. repeat lane 2's address and data): no exec masking
        const u32 q = l < 2u ? l : 2u;
        o.maskbits[o.e * 3u + q] = q == 0u ? q0 : (q == 1u ? q1 : q2);
    }
    __builtin_amdgcn_sched_barrier(0);
    AZ_STAMP(SEG_MASK);
    const double sJ = fs.y;
    // cum(J + M) = M + Fr[J][ilog2 M] (M >= 1); floor-only: 0 + fl(100 S[J]), x in hundredths
    const double total = ((double)M + fs.x) + 0.0;
    // pattern moves: cum(J + mm) = mm + Fr[J][ilog2 mm]; away from integer boundaries floor(x - S[J]) + 1 IS the ordinal
    double u = u01;
    double x = u * total;
    double d = x - sJ;
    u32 fl = (u32)d;
    double fr = d - (double)fl;
    u32 kg = kbase + fl + 1u;
    // "x is not safely inside a pattern move's unit interval": fr within 1e-9 of 0 or 1 as ONE compare, |fr - 0.5| >= 0.5 - 1e-9 (the
    // subtraction's rounding, 2^-54, is far inside the margin; the slow path gives the fast path's answer wherever both apply).  x < S[J]
    // (inside the floor moves) needs no test of its own: then -1 < d < 0, fl == 0 and fr == d < 0.
    bool edge = !(__builtin_fabs(fr - 0.5) < 0.5 - 1e-9) | (kg > L);
    r.pos += 2u;                                             // (taken back in the block below when no random() was consumed here)
    // ONE test for everything unusual about this decision (a random() that crosses a regeneration, a draw at a boundary of the
    // cumulative weights, nothing legal); the stuck slot itself is restarted further down
    // (a wave-uniform test -- vector compare, VCC, scalar branch -- costs ~33 cycles even when it falls through, tools/pattern_cost.hip:
    // "nothing legal" rides on this one, and the stuck-slot block below branches on a flag that is already scalar)
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
            // the boundary search works on CPython's own quantities: for a floor-only mask x = random() * (S[J] + 0.0)
            const double sT = M ? sJ : T.fs[9u * Jc].y;
            kg = sample_slow2(T, M ? x : u * (sT + 0.0), sT, J, M, L);
        }
        any_nomove = wave_any(nomove);
    }
    // kg-th legal action: its pattern row (= mask word) from the prefix counts (half-uniform compares), then ONE rank test per lane;
    // the lane that answers holds (display, colour) as a constant, the row is the word index
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
    const u32 who = hb(hit);
    const u32 ln = (u32)__builtin_ctz(who | 0x80000000u);
    // (display, colour) of the answering lane: ln = d + 6 c decoded arithmetically (K2::lcode holds the same as lane constants, but
    // fetching it from lane ln is an LDS round trip on the move's chain: ~100 cycles against ~10 instructions)
    const u32 ac = (ln * 43u) >> 8, ad = ln - 6u * ac;               // ln / 6, ln % 6 for ln < 32
    const bool a_disp = ad != 0u;
    const u32 a_db = a_disp ? 5u * ad - 5u : 0u;
    const u32 a_src = a_disp ? a_db + ac : 25u + ac;
    const i32 a = (i32)(30u * prow_ + ln);
    AZ_STAMP(SEG_SAMPLE);

    u32 ret = 0;
    if (AZ_UNLIKELY(any_nomove)) {
        if (nomove) {
            // stuck (hazard H3), or handed an already finished game: report, restart the slot
            cnt.stuck_add += 1u;
            outputs2<OUT>(g, o, -1, 0, 2u, l);
            u32 st0 = episode_reset2<LID>(g, first_player, r, margin, k);
            ret = st0 ? (0x100u | st0) : 2u;
        }
        dead |= (ret & 0x100u) != 0u;
        AZ_STAMP(SEG_RESET);
    }
    if (!nomove) {
        const u32 me = me2(g);
        const bool filled = do_move2f<LID>(g, a_src, a_db, ac, prow_, a_disp, m.B, k);    // azul.py:304
        g.moves += 1u;
        AZ_STAMP(SEG_MOVE);
        g.B = hb(g.cs != 0u) & 0x7fffffffu;                      // the sources after the move (next move's mask reads it)
        ret = after_move2<LID, OUT, LIM>(g, first_player, k, r, margin, cnt, o, me, filled, a, prof_, dead);
    }
    return ret;
}

// A slot of a game that a rule error stopped earlier in this launch (crafted states only: box and lid empty when a round has to be dealt;
// the reference raises, azul.py:86-87): no move is played -- the slot is marked like a stuck slot (no legal action, action -1, done 2) and
// counted with them, so that whoever counts env moves as slots minus `stuck` stays right and the trajectory carries no stale bytes
// (azul_rules_x.hpp does the same for three / four players).
// Called AFTER the move loop, which only skips a stopped game and counts the skipped slots: they are the launch's LAST `skipped` slots.
template <int OUT, bool PAD, bool BITS>
AZ_FN void dead_slots2(const G2 &g, Out2 o, Counters2 &cnt, u32 n_games, u32 end_e, u32 skipped, u32 l)
{
#pragma unroll 1
    for (o.e = end_e - skipped * n_games; o.e < end_e; o.e += n_games) {
        cnt.stuck_add += 1u;
        if (OUT == 1 || (OUT == 2 && o.mask)) store_mask_row2<(PAD && OUT == 1)>(o, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, l);
        if ((OUT == 1 && BITS) || (OUT == 2 && o.maskbits)) o.maskbits[o.e * 3u + (l < 2u ? l : 2u)] = 0ull;
        outputs2<OUT>(g, o, -1, 0, 2u, l);
    }
}

} // namespace az2
