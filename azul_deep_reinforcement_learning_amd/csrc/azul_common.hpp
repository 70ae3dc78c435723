// azul_common.hpp -- what every device header of libazulhip.so shares: integer types, status / pool codes, the sizes of the RandomAgent
// weight table, the optional segment stamps, and the few scalar rule helpers that do not depend on how a game is laid out in a wavefront
// (floor penalty, score clamp, complete wall row, column boards, the compact trajectory record, CPython's seeding).
//
// The rules themselves live in ONE place per game shape:
//   azul_selfplay2.hpp + azul_env2.hpp   two players under the reference's rules (GameRunner's shaped reward, opponent loop, reset): the
//                                        benchmarked self-play loop, every two-player rule entry of the C ABI (azul_ops2.hpp) and the env
//                                        side of the policy rollout -- two games per 64-lane wavefront, state in VGPRs
//   azul_rules_x.hpp                     P = 2..4 players, D = 5 or 2 P + 1 displays, the extended rule switches (row N4), built from the
//                                        same wave primitives (half ballots, LDS-crossbar gathers, wall pricing, parallel factory draw)
// This is gfx950 device code only: there is no CPU execution path in the product (tests/hostcheck/simt emulates the 64 lanes in lockstep to
// run these headers, unmodified, in the build container).
// Reference lines: azulnet/azul.py:184-191 (is_end_of_game), :200-210 (count_floor), :294-295 (score clamp); CPython 3.10 _randommodule.c
// (init_by_array) for random.seed(int) as used at tests/test_azul.py:36, tests/test_game_runner.py:27.
#pragma once
#include <stdint.h>

typedef uint32_t u32;
typedef int32_t  i32;
typedef uint64_t u64;
typedef int64_t  i64;

#if !defined(__HIPCC__)
#error "libazulhip's headers are gfx950 device code: compile with hipcc --offload-arch=gfx950"
#endif
#include <hip/hip_runtime.h>
#define AZ_FN __device__ __forceinline__
#define AZ_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace wv {
AZ_FN u32 lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
} // namespace wv

namespace az {

enum { ST_OK = 0, ST_ILLEGAL_MOVE = 1, ST_GAME_ENDED = 2, ST_STUCK = 3, ST_BAD_ACTION = 4, ST_BOX_EMPTY = 5,
       ST_TRUNCATED = 6 /* move limit (beyond the reference, off by default): the episode was cut at the end of a round */ };
enum { POOL_RANDOM = 0, POOL_LID = 1 };
#ifndef AZ_DRAW_MARGIN
#define AZ_DRAW_MARGIN 8192ull      // factory draw: > 21.1 * 255, the largest possible fp64 disagreement window
#endif
enum { T_ROWS = 31, T_BINADES = 8, T_STRIDE = 9 /* pairs per row: eight binades + the floor-only pair */, T_PAIRS = T_ROWS * T_STRIDE };   // RandomAgent weight table of the 180-action game: 31 rows (J = 0..30 legal floor moves), see azul_tables.hpp

// ---- optional in-kernel segment stamps (diagnostic build only: -DAZ_PROFILE_SEGMENTS; cdna_hip_programming.md 7) ----
// s_memtime deltas are accumulated per segment and added to a global buffer when the wave ends.  The real kernel
// contains no stamp; never quote the diagnostic build's run time, only its SHARES (tools/segment_profile.py).
enum { SEG_MASK = 0, SEG_SAMPLE, SEG_MOVE, SEG_AFTERMOVE, SEG_TAIL, SEG_NEWROUND, SEG_SCORE, SEG_RESET, SEG_LOOP, SEG_COUNT };
constexpr int AZ_PROF_SLOTS = 48;   // u64 slots of BatchDev::prof: the self-play kernel's SEG_COUNT segments, the rollout kernel's phases (0..8) and matrix sub-phases (16..31)
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

// wall cells lying in board column `col` (walls are colour-indexed like the reference's: bit 5 r + c sits in board column (c + r) % 5)
constexpr u32 column_board_c(int col)
{
    u32 m = 0;
    for (int j = 0; j < 5; j++) m |= 1u << (5 * j + ((col - j + 5) % 5));
    return m;
}
static_assert(column_board_c(0) == 0x222201u && column_board_c(4) == 0x111110u, "column boards");

AZ_FN i32 floor_penalty(u32 floor_tiles)
{
    u32 f = floor_tiles > 7u ? 7u : floor_tiles;         // count_floor, azul.py:200-210: 0,-1,-2,-4,-6,-8,-11,-14
    return -(i32)((0x0e0b080604020100ull >> (8u * f)) & 0xffu);
}

AZ_FN i32 clamp0(i32 s) { return s < 0 ? 0 : s; }       // azul.py:294-295

AZ_FN bool any_row_full(u32 w) { return ((w & (w >> 1) & (w >> 2) & (w >> 3) & (w >> 4)) & 0x108421u) != 0u; }     // azul.py:184-191

AZ_FN u32 byte_sum5(u64 v) { return (u32)(((v & 0xffffffffffull) * 0x0101010101ull) >> 32) & 0xffu; }

// compact trajectory record of one move (what the multi-GPU all-gather ships): action (0xff = none) | done << 8 | reward << 16
AZ_FN u32 pack_move(i32 a, u32 dn, i32 reward) { return ((u32)(a >= 0 ? a : 0xff) & 0xffu) | ((dn & 0xffu) << 8) | (((u32)reward & 0xffffu) << 16); }

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

} // namespace az
