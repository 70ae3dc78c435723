/*
 * azul_oracle.c -- TEST INFRASTRUCTURE ONLY (see azul_oracle.h).
 *
 * A deliberately literal, scalar restatement: loops follow the reference line by
 * line so that a reader can diff behaviour, not performance.  Every function cites
 * the reference lines it follows (paths relative to /root/reference).
 *
 * Third-party arithmetic on the path (not vendored in the reference): CPython 3.10
 * `random` -- Modules/_randommodule.c (MT19937: init_genrand, init_by_array,
 * genrand_uint32, random_random, getrandbits) and Lib/random.py
 * (_randbelow_with_getrandbits :239-249, choice :375-378, randrange :292-,
 * choices :506-541).  Pinned by the reference's own seeded tests
 * (tests/test_azul.py:36-39,100-106; tests/test_game_runner.py:43-51) and by
 * tests/golden/ vectors produced from the real interpreter.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math  (IEEE fp64, no FMA contraction).
 */
#include "azul_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* CPython random                                                            */
/* ------------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397

static void init_genrand(oz_rng *r, uint32_t s)
{
    /* _randommodule.c init_genrand */
    uint32_t *mt = r->mt;
    mt[0] = s;
    for (int i = 1; i < MT_N; i++)
        mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = MT_N;
}

static void init_by_array(oz_rng *r, const uint32_t *key, int len)
{
    /* _randommodule.c init_by_array */
    uint32_t *mt = r->mt;
    init_genrand(r, 19650218U);
    int i = 1, j = 0;
    int k = (MT_N > len) ? MT_N : len;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000U;
}

void oz_rng_seed(oz_rng *r, uint64_t seed)
{
    /* random.seed(int): random_seed() splits abs(n) into 32-bit little-endian words; 0 -> [0]. */
    uint32_t key[2];
    int len = 1;
    key[0] = (uint32_t)(seed & 0xffffffffu);
    key[1] = (uint32_t)(seed >> 32);
    if (key[1] != 0) len = 2;
    init_by_array(r, key, len);
    r->words = 0;
}

void oz_rng_set(oz_rng *r, const uint32_t mt[624], int32_t idx)
{
    memcpy(r->mt, mt, sizeof(r->mt));
    r->idx = idx;
    r->words = 0;
}

uint32_t oz_rng_u32(oz_rng *r)
{
    /* _randommodule.c genrand_uint32 */
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    uint32_t *mt = r->mt;
    uint32_t y;
    if (r->idx >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 0x1U];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 0x1U];
        }
        y = (mt[MT_N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 0x1U];
        r->idx = 0;
    }
    y = mt[r->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    r->words++;
    return y;
}

double oz_rng_random(oz_rng *r)
{
    /* _randommodule.c random_random */
    uint32_t a = oz_rng_u32(r) >> 5, b = oz_rng_u32(r) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

uint32_t oz_rng_getrandbits(oz_rng *r, int k)
{
    /* _randommodule.c random_getrandbits, k <= 32 fast path */
    return oz_rng_u32(r) >> (32 - k);
}

uint32_t oz_rng_randbelow(oz_rng *r, uint32_t n)
{
    /* random.py:239-249 */
    if (!n) return 0;
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;          /* n.bit_length() */
    uint32_t v = oz_rng_getrandbits(r, k);
    while (v >= n) v = oz_rng_getrandbits(r, k);
    return v;
}

int oz_rng_choices(oz_rng *r, const double *w, int n)
{
    /* random.py:506-541 with k=1: cum = list(accumulate(w)); total = cum[-1] + 0.0;
     * bisect_right(cum, random()*total, 0, n-1).  Error paths are checked BEFORE random(). */
    double cum[180];
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        acc = (i == 0) ? w[0] : acc + w[i];
        cum[i] = acc;
    }
    double total = cum[n - 1] + 0.0;
    if (total <= 0.0) return -1;
    if (!isfinite(total)) return -2;
    double x = oz_rng_random(r) * total;
    int lo = 0, hi = n - 1;                         /* bisect_right(cum, x, 0, n-1) */
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* ------------------------------------------------------------------------- */
/* azul.py                                                                   */
/* ------------------------------------------------------------------------- */
static int pidx(const oz_game *g)
{
    /* numpy negative indexing: current_player == 0 (before the first new_round) selects the LAST player. */
    int p = g->current_player - 1;
    if (p < 0) p += g->players;
    return p;
}

/* ---- beyond the reference: D displays, the finite bag (see the OZ_EXT_* flags in azul_oracle.h) ---- */
static int nd(const oz_game *g) { return g->n_displays > 0 ? g->n_displays : 5; }          /* azul.py:19 unless OZ_EXT_DISPLAYS_2P1 */
static int64_t *drow(oz_game *g, int d) { return d < 5 ? g->displays[d] : g->xdisplays[d - 5]; }
static const int64_t *drow_c(const oz_game *g, int d) { return d < 5 ? g->displays[d] : g->xdisplays[d - 5]; }
/* tiles are tracked (box / lid) under the reference's "Lid" pool (azul.py:48-52) and under the finite bag */
static int tracks_tiles(const oz_game *g) { return g->tile_pool == OZ_POOL_LID || (g->ext & OZ_EXT_FINITE_BAG); }

int oz_init_ext(oz_game *g, int players, int first_player, int tile_pool, int ext, oz_rng *r)
{
    /* azul.py:18-61 */
    memset(g, 0, sizeof(*g));
    g->players = players;
    g->ext = ext;
    /* rulebook, "Setup": "In a 2-player game, place 5 Factory displays ... 3-player game: 7 ... 4-player game: 9" */
    g->n_displays = (ext & OZ_EXT_DISPLAYS_2P1) ? 2 * players + 1 : 5;
    if (first_player == 0) {                                   /* "Random", azul.py:36-37: random.choice([1..players]) */
        g->next_first_player = 1 + (int)oz_rng_randbelow(r, (uint32_t)players);
    } else if (first_player > 0) {                             /* azul.py:38-41 */
        if (first_player > players) return OZ_ILLEGAL_RULE;
        g->next_first_player = first_player;
    } else {                                                   /* azul.py:42-43 */
        g->next_first_player = 1;
    }
    g->tile_pool = tile_pool;
    if (tile_pool == OZ_POOL_LID) {                            /* azul.py:48-52 */
        if (ext & OZ_EXT_FINITE_BAG) return OZ_ILLEGAL_RULE;   /* the "Lid" pool already is a finite bag */
        for (int c = 0; c < 5; c++) { g->box[c] = 20; g->lid[c] = 0; }
    } else if (tile_pool != OZ_POOL_RANDOM) {
        return OZ_ILLEGAL_RULE;
    } else if (ext & OZ_EXT_FINITE_BAG) {
        /* rulebook, "Setup": "Fill the bag with the 100 tiles (20 of each color)" */
        for (int c = 0; c < 5; c++) { g->box[c] = 20; g->lid[c] = 0; }
    }
    return OZ_OK;
}

int oz_init(oz_game *g, int players, int first_player, int tile_pool, oz_rng *r)
{
    return oz_init_ext(g, players, first_player, tile_pool, 0, r);
}

int oz_new_round(oz_game *g, oz_rng *r)
{
    /* azul.py:64-89 */
    g->current_player = g->next_first_player;
    {
        int i = g->next_first_player - 1;
        if (i < 0) i += g->players;
        g->first_player_stats[i] += 1;
    }
    g->turn_counter += 1;
    g->next_first_player = 0;
    for (int c = 0; c < 5; c++) g->center[c] = 0;
    g->center[5] = 1;
    memset(g->displays, 0, sizeof(g->displays));
    memset(g->xdisplays, 0, sizeof(g->xdisplays));
    for (int i = 0; i < nd(g); i++) {                           /* azul.py:75: range(5) */
        for (int j = 0; j < 4; j++) {
            if (g->tile_pool == OZ_POOL_RANDOM && !(g->ext & OZ_EXT_FINITE_BAG)) {
                /* azul.py:78 random.randrange(0,5,1) -> _randbelow(5) */
                drow(g, i)[oz_rng_randbelow(r, 5)] += 1;
            }
            if (tracks_tiles(g)) {
                int64_t total = 0;
                for (int c = 0; c < 5; c++) total += g->box[c];
                if (total == 0) {                               /* azul.py:81-83 */
                    /* rulebook, "Preparing the next round": "If the bag is empty, refill it with all the tiles that you have placed
                     * in the lid of the game box and then continue filling the remaining Factory displays." */
                    for (int c = 0; c < 5; c++) { g->box[c] = g->lid[c]; g->lid[c] = 0; }
                }
                total = 0;
                for (int c = 0; c < 5; c++) total += g->box[c];      /* azul.py:85 */
                if (total == 0) {
                    /* rulebook: "In the rare case that you run out of tiles again while there are none left in the lid, start the
                     * new round as usual even though not all Factory displays are properly filled." */
                    if (g->ext & OZ_EXT_SHORT_DEAL) return OZ_OK;
                    return OZ_BOX_EMPTY;                        /* reference: ValueError out of random.choices (TODO azul.py:86) */
                }
                int color;
                if (g->tile_pool == OZ_POOL_LID) {
                    double w[5];
                    for (int c = 0; c < 5; c++) w[c] = (double)g->box[c] / (double)total;   /* azul.py:87 */
                    color = oz_rng_choices(r, w, 5);
                } else {
                    /* finite bag for the "Random" pool (the TODO at azul.py:72): one of the `total` tiles in the bag, uniformly --
                     * random.randrange(total) = _randbelow(total) -- tiles ordered by colour; pure integer arithmetic */
                    int64_t k = (int64_t)oz_rng_randbelow(r, (uint32_t)total), cum = 0;
                    color = 0;
                    for (int c = 0; c < 5; c++) { cum += g->box[c]; if (k < cum) { color = c; break; } }
                }
                drow(g, i)[color] += 1;                         /* azul.py:88 */
                g->box[color] -= 1;                             /* azul.py:89 */
            }
        }
    }
    return OZ_OK;
}

static void add_to_floor(oz_game *g, int64_t nr)
{
    /* azul.py:119-123 */
    int p = pidx(g);
    if (g->floors[p] + nr < 7) g->floors[p] += nr;
    else g->floors[p] = 7;
}

void oz_move(oz_game *g, int display, int color, int pattern)
{
    /* azul.py:118-161 */
    int p = pidx(g);
    int64_t nr_tiles;
    if (display != 0) {
        int64_t *dsp = drow(g, display - 1);
        nr_tiles = dsp[color];                                  /* :127 */
        dsp[color] = 0;                                         /* :129 */
        for (int c = 0; c < 5; c++) g->center[c] += dsp[c];     /* :131 */
        for (int c = 0; c < 5; c++) dsp[c] = 0;                 /* :133 */
    } else {
        nr_tiles = g->center[color];                            /* :136 */
        g->center[color] = 0;                                   /* :138 */
        if (g->center[5] == 1) {                                /* :140-143 */
            g->center[5] = 0;
            g->next_first_player = g->current_player;
            add_to_floor(g, 1);
        }
    }
    if (pattern != 0) {
        int64_t overflow = pattern - g->pattern_lines[p][pattern - 1][color] - nr_tiles;   /* :147 */
        if (overflow >= 0) {
            g->pattern_lines[p][pattern - 1][color] += nr_tiles;                           /* :150 */
        } else {
            g->pattern_lines[p][pattern - 1][color] = pattern;                             /* :152 */
            add_to_floor(g, -overflow);                                                    /* :154 */
            if (tracks_tiles(g)) g->lid[color] += -overflow;                               /* :156-157 */
        }
    } else {
        add_to_floor(g, nr_tiles);                                                         /* :159 */
        if (tracks_tiles(g)) g->lid[color] += nr_tiles;                                    /* :160-161 */
    }
}

int oz_is_legal_move(const oz_game *g, int display, int color, int pattern)
{
    /* azul.py:162-176 */
    int p = pidx(g);
    if (display > 0) {
        if (drow_c(g, display - 1)[color] < 1) return 0;
    } else {
        if (g->center[color] < 1) return 0;
    }
    if (pattern != 0) {
        int others = 0;
        for (int c = 0; c < 5; c++)
            if (c != color && g->pattern_lines[p][pattern - 1][c] != 0) others++;          /* :172 */
        if (others > 0) return 0;
        if (g->walls[p][pattern - 1][color]) return 0;                                     /* :174 */
    }
    return 1;
}

void oz_next_player(oz_game *g)
{
    /* azul.py:177-181 */
    if (g->current_player < g->players) g->current_player += 1;
    else g->current_player = 1;
}

int oz_is_end_of_round(const oz_game *g)
{
    /* azul.py:182-183 (the first-player token counts) */
    int nz = 0;
    for (int d = 0; d < nd(g); d++) for (int c = 0; c < 5; c++) nz += (drow_c(g, d)[c] != 0);
    for (int c = 0; c < 6; c++) nz += (g->center[c] != 0);
    return nz < 1;
}

int oz_is_end_of_game(const oz_game *g)
{
    /* azul.py:184-191 */
    for (int p = 0; p < g->players; p++)
        for (int i = 0; i < 5; i++) {
            int n = 0;
            for (int c = 0; c < 5; c++) n += (g->walls[p][i][c] != 0);
            if (n == 5) return 1;
        }
    return 0;
}

static int mod5(int x) { int m = x % 5; return m < 0 ? m + 5 : m; }       /* Python % */
static int to_wall_position(int color, int pattern) { return mod5(color + pattern); }     /* azul.py:194-196 */
static int from_wall_position(int color, int pattern) { return mod5(color - pattern); }   /* azul.py:197-199 */

static int64_t count_floor(oz_game *g, int player)
{
    /* azul.py:200-210 */
    int64_t count;
    if (g->floors[player] <= 2) count = -g->floors[player];
    else if (g->floors[player] <= 5) count = -2 - (g->floors[player] - 2) * 2;
    else count = -8 - (g->floors[player] - 5) * 3;
    g->floor_penalty[player] += (double)count;
    g->floors[player] = 0;
    return count;
}

static int64_t count_wall(oz_game *g, int player)
{
    /* azul.py:211-290 */
    int64_t count = 0;
    for (int pattern = 0; pattern < 5; pattern++) {
        for (int color = 0; color < 5; color++) {
            if (g->pattern_lines[player][pattern][color] == pattern + 1) {                 /* :216 */
                g->pattern_lines[player][pattern][color] = 0;                              /* :218 */
                g->walls[player][pattern][color] = 1;                                      /* :219 */
                if (tracks_tiles(g)) g->lid[color] += pattern;                             /* :220-222 */
                int64_t pos_count = 0, bonus_count = 0;
                int only_row = 1, only_col = 1;
                for (int i = to_wall_position(color, pattern) + 1; i < 5; i++) {           /* :230-236 */
                    if (g->walls[player][pattern][from_wall_position(i, pattern)]) { pos_count++; only_row = 0; }
                    else break;
                }
                for (int i = to_wall_position(color, pattern) - 1; i > -1; i--) {          /* :237-242 */
                    if (g->walls[player][pattern][from_wall_position(i, pattern)]) { pos_count++; only_row = 0; }
                    else break;
                }
                for (int j = pattern + 1; j < 5; j++) {                                    /* :244-250 */
                    if (g->walls[player][j][to_wall_position(color, pattern - j)]) { pos_count++; only_col = 0; }
                    else break;
                }
                for (int j = pattern - 1; j > -1; j--) {                                   /* :251-257 */
                    if (g->walls[player][j][to_wall_position(color, pattern - j)]) { pos_count++; only_col = 0; }
                    else break;
                }
                if (only_row && only_col) pos_count = 1;                                   /* :258-263 */
                else if (!(only_row || only_col)) pos_count += 2;
                else pos_count += 1;
                if ((double)pos_count > g->max_combo[player]) g->max_combo[player] = (double)pos_count;   /* :264 */
                for (int i = 0; i < 5; i++) {                                              /* :266-272 */
                    if (g->walls[player][pattern][i]) {
                        if (i == 4) { bonus_count += 2; g->completed_lines[player][0] += 1; }
                    } else break;
                }
                for (int j = 0; j < 5; j++) {                                              /* :274-280 */
                    if (g->walls[player][j][color]) {
                        if (j == 4) { bonus_count += 10; g->completed_lines[player][1] += 1; }
                    } else break;
                }
                for (int k = 0; k < 5; k++) {                                              /* :282-288 */
                    if (g->walls[player][k][from_wall_position(to_wall_position(color, pattern), k)]) {
                        if (k == 4) { bonus_count += 7; g->completed_lines[player][2] += 1; }
                    } else break;
                }
                /* OZ_EXT_END_BONUS: the line bonuses are paid once, when the game has ended (oz_end_game_bonus); the
                 * completed_lines statistics are still counted as the lines complete */
                if (g->ext & OZ_EXT_END_BONUS) bonus_count = 0;
                count += pos_count + bonus_count;                                          /* :289 */
            }
        }
    }
    return count;
}

void oz_count_score(oz_game *g)
{
    /* azul.py:291-295 */
    for (int player = 0; player < g->players; player++) {
        int64_t f = count_floor(g, player);
        int64_t w = count_wall(g, player);
        g->score[player] += f + w;
        if (g->score[player] < 0) g->score[player] = 0;
    }
}

void oz_end_game_bonus(oz_game *g)
{
    /* rulebook, "End of the game": "Once the game has ended, score additional points if you have achieved the following goals:
     * Gain 2 points for each complete horizontal line of 5 consecutive tiles on your wall.  Gain 7 points for each complete
     * vertical line of 5 consecutive tiles on your wall.  Gain 10 points for each color of which you have placed all 5 tiles on
     * your wall."  Paid once, after the last round's scoring and its clamp (azul.py:294-295). */
    for (int p = 0; p < g->players; p++) {
        int64_t bonus = 0;
        for (int row = 0; row < 5; row++) {                     /* horizontal lines */
            int n = 0;
            for (int c = 0; c < 5; c++) n += (g->walls[p][row][c] != 0);
            if (n == 5) bonus += 2;
        }
        for (int col = 0; col < 5; col++) {                     /* vertical lines: board column of (row, colour) is (colour + row) % 5 */
            int n = 0;
            for (int row = 0; row < 5; row++) n += (g->walls[p][row][from_wall_position(col, row)] != 0);
            if (n == 5) bonus += 7;
        }
        for (int c = 0; c < 5; c++) {                           /* colours */
            int n = 0;
            for (int row = 0; row < 5; row++) n += (g->walls[p][row][c] != 0);
            if (n == 5) bonus += 10;
        }
        g->score[p] += bonus;
    }
}

int oz_step(oz_game *g, int display, int color, int pattern, oz_rng *r)
{
    /* azul.py:296-313 */
    if (g->end_of_game) return OZ_GAME_ENDED;
    if (!oz_is_legal_move(g, display, color, pattern)) return OZ_ILLEGAL_MOVE;
    oz_move(g, display, color, pattern);
    if (oz_is_end_of_round(g)) {
        oz_count_score(g);
        if (oz_is_end_of_game(g)) {
            g->end_of_game = 1;
            if (g->ext & OZ_EXT_END_BONUS) oz_end_game_bonus(g);
        }
        else return oz_new_round(g, r);
    } else {
        oz_next_player(g);
    }
    return OZ_OK;
}

/* Azul.step under the MOVE LIMIT -- beyond the reference ("parity unpinned"; include/azul_hip.h: azul_batch_set_move_limit): when the move
 * ends a round without ending the game and the episode has then played `limit` moves or more, the round is scored and NO new round is
 * dealt: OZ_TRUNCATED.  limit <= 0: exactly oz_step. */
int oz_step_limited(oz_game *g, int display, int color, int pattern, oz_rng *r, int64_t moves_after, int64_t limit)
{
    if (g->end_of_game) return OZ_GAME_ENDED;
    if (!oz_is_legal_move(g, display, color, pattern)) return OZ_ILLEGAL_MOVE;
    oz_move(g, display, color, pattern);
    if (oz_is_end_of_round(g)) {
        oz_count_score(g);
        if (oz_is_end_of_game(g)) {
            g->end_of_game = 1;
            if (g->ext & OZ_EXT_END_BONUS) oz_end_game_bonus(g);
        }
        else if (limit > 0 && moves_after >= limit) return OZ_TRUNCATED;
        else return oz_new_round(g, r);
    } else {
        oz_next_player(g);
    }
    return OZ_OK;
}

void oz_get_statistics(const oz_game *g, double out[10])
{
    /* azul.py:314-315; key order of game_runner.py:12 */
    double fsum = 0;
    for (int p = 0; p < g->players; p++) fsum += g->first_player_stats[p];
    out[0] = (double)g->score[0];
    out[1] = (double)g->score[1];
    out[2] = (double)g->turn_counter;
    out[3] = g->first_player_stats[0] / fsum * 100;
    out[4] = -g->floor_penalty[0];
    out[5] = g->max_combo[0];
    out[6] = g->completed_lines[0][0];
    out[7] = g->completed_lines[0][2];
    out[8] = g->completed_lines[0][1];
    out[9] = (g->score[0] > g->score[1]) ? 1.0 : 0.0;
}

int oz_equal(const oz_game *a, const oz_game *b)
{
    /* azul.py:62-63 (box/lid/stats/rules are NOT compared) */
    if (a->players != b->players) return 0;
    if (memcmp(a->displays, b->displays, sizeof(a->displays))) return 0;
    if (memcmp(a->center, b->center, sizeof(a->center))) return 0;
    for (int p = 0; p < a->players; p++) {
        if (memcmp(a->pattern_lines[p], b->pattern_lines[p], sizeof(a->pattern_lines[p]))) return 0;
        if (memcmp(a->walls[p], b->walls[p], sizeof(a->walls[p]))) return 0;
        if (a->floors[p] != b->floors[p] || a->score[p] != b->score[p]) return 0;
    }
    return a->current_player == b->current_player && a->next_first_player == b->next_first_player &&
           a->end_of_game == b->end_of_game && a->turn_counter == b->turn_counter;
}

/* ------------------------------------------------------------------------- */
/* game_runner.py                                                            */
/* ------------------------------------------------------------------------- */
int oz_serialize(int display, int color, int pattern) { return display + color * 6 + pattern * 5 * 6; }   /* :102-103 */

void oz_deserialize(int a, int *display, int *color, int *pattern)
{
    /* :107-111 */
    *display = a % 6;
    *color = (a / 6) % 5;
    *pattern = a / 30;
}

void oz_check_all_valid(const oz_game *g, uint8_t out[180])
{
    /* :113-117 */
    for (int i = 0; i < 180; i++) {
        int d, c, p;
        oz_deserialize(i, &d, &c, &p);
        out[i] = (uint8_t)oz_is_legal_move(g, d, c, p);
    }
}

int oz_random_agent(const uint8_t mask[180], oz_rng *r)
{
    /* :87-97: weight_table = ones(180) with 0.01 for pattern==0 (a<30), times the mask. */
    double w[180];
    for (int a = 0; a < 180; a++) {
        double base = 1.0;
        int d, c, p;
        oz_deserialize(a, &d, &c, &p);
        if (p == 0) base = 0.01;
        w[a] = base * (mask[a] ? 1.0 : 0.0);
    }
    return oz_rng_choices(r, w, 180);
}

void oz_get_state(const oz_game *g, int perspective, int64_t out[136])
{
    /* :56-72, two players: order = [perspective, other] */
    int order[OZ_MAXP];
    int n = 0;
    order[n++] = perspective;
    for (int p = 0; p < g->players; p++) if (p != perspective) order[n++] = p;
    int64_t pnfp = 0;
    if (g->next_first_player > 0) {
        int m = (g->next_first_player - 1 - perspective) % g->players;
        if (m < 0) m += g->players;
        pnfp = m + 1;
    }
    int k = 0;
    for (int d = 0; d < 5; d++) for (int c = 0; c < 5; c++) out[k++] = g->displays[d][c];
    for (int c = 0; c < 6; c++) out[k++] = g->center[c];
    for (int i = 0; i < g->players; i++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) out[k++] = g->pattern_lines[order[i]][r][c];
    for (int i = 0; i < g->players; i++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) out[k++] = g->walls[order[i]][r][c];
    for (int i = 0; i < g->players; i++) out[k++] = g->floors[order[i]];
    for (int i = 0; i < g->players; i++) out[k++] = g->score[order[i]];
    out[k++] = pnfp;
}

/* ---- the same wrapper functions for D displays / P players (beyond the reference: game_runner.py hard-codes 6, 180 and two players) ---- */
int oz_num_actions(const oz_game *g) { return (nd(g) + 1) * 5 * 6; }                         /* game_runner.py:115: 6*5*6 */
int oz_obs_size(const oz_game *g) { return nd(g) * 5 + 6 + g->players * 52 + 1; }           /* game_runner.py:65-72 */

void oz_deserialize_x(const oz_game *g, int a, int *display, int *color, int *pattern)
{
    /* game_runner.py:107-111 with 6 -> D + 1 */
    const int S = nd(g) + 1;
    *display = a % S;
    *color = (a / S) % 5;
    *pattern = a / (5 * S);
}

void oz_check_all_valid_x(const oz_game *g, uint8_t *out)
{
    /* game_runner.py:113-117 */
    const int n = oz_num_actions(g);
    for (int i = 0; i < n; i++) {
        int d, c, p;
        oz_deserialize_x(g, i, &d, &c, &p);
        out[i] = (uint8_t)oz_is_legal_move(g, d, c, p);
    }
}

int oz_random_agent_x(const uint8_t *mask, int n_actions, oz_rng *r)
{
    /* game_runner.py:87-97: weight 0.01 for pattern == 0 (the first n_actions / 6 actions), 1.0 otherwise, times the mask;
     * random.choices(range(n_actions), weights) */
    double cum[OZ_MAX_ACTIONS];
    double acc = 0.0;
    const int floor_moves = n_actions / 6;
    for (int a = 0; a < n_actions; a++) {
        double w = (a < floor_moves ? 0.01 : 1.0) * (mask[a] ? 1.0 : 0.0);
        acc = (a == 0) ? w : acc + w;
        cum[a] = acc;
    }
    double total = cum[n_actions - 1] + 0.0;
    if (total <= 0.0) return -1;
    double x = oz_rng_random(r) * total;
    int lo = 0, hi = n_actions - 1;                 /* bisect_right(cum, x, 0, n-1) */
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

void oz_get_state_x(const oz_game *g, int perspective, int64_t *out)
{
    /* game_runner.py:56-72: order = [perspective] + the other players ascending (a set of small ints iterates in order) */
    int order[OZ_MAXP];
    int n = 0;
    order[n++] = perspective;
    for (int p = 0; p < g->players; p++) if (p != perspective) order[n++] = p;
    int64_t pnfp = 0;
    if (g->next_first_player > 0) {
        int m = (g->next_first_player - 1 - perspective) % g->players;
        if (m < 0) m += g->players;
        pnfp = m + 1;
    }
    int k = 0;
    for (int d = 0; d < nd(g); d++) for (int c = 0; c < 5; c++) out[k++] = drow_c(g, d)[c];
    for (int c = 0; c < 6; c++) out[k++] = g->center[c];
    for (int i = 0; i < g->players; i++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) out[k++] = g->pattern_lines[order[i]][r][c];
    for (int i = 0; i < g->players; i++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) out[k++] = g->walls[order[i]][r][c];
    for (int i = 0; i < g->players; i++) out[k++] = g->floors[order[i]];
    for (int i = 0; i < g->players; i++) out[k++] = g->score[order[i]];
    out[k++] = pnfp;
}

int oz_runner_init(oz_runner *q, int first_player, int tile_pool, oz_rng *r)
{
    /* :23-36 */
    q->first_player = first_player;
    q->tile_pool = tile_pool;
    int st = oz_init(&q->game, 2, first_player, tile_pool, r);
    if (st) return st;
    st = oz_new_round(&q->game, r);
    q->player_score = 0;
    q->move_counter = 0;
    return st;
}

static int runner_reset_noplay(oz_runner *q, oz_rng *r)
{
    /* :79-82 */
    int st = oz_init(&q->game, 2, q->first_player, q->tile_pool, r);
    if (st) return st;
    st = oz_new_round(&q->game, r);
    q->player_score = 0;
    q->move_counter = 0;
    return st;
}

int oz_runner_opponent_move(oz_runner *q, oz_rng *r)
{
    /* :37-42 with the default RandomAgent opponent */
    uint8_t mask[180];
    oz_check_all_valid(&q->game, mask);
    int a = oz_random_agent(mask, r);
    if (a < 0) return OZ_STUCK;                        /* reference: ValueError from random.choices */
    int d, c, p;
    oz_deserialize(a, &d, &c, &p);
    int st = oz_step(&q->game, d, c, p, r);
    if (st) return st;
    q->move_counter += 1;
    return OZ_OK;
}

int oz_runner_reset(oz_runner *q, oz_rng *r)
{
    /* :76-85 */
    int st = runner_reset_noplay(q, r);
    if (st) return st;
    while (q->game.current_player != 1) {
        st = oz_runner_opponent_move(q, r);
        if (st) return st;
    }
    return OZ_OK;
}

int64_t oz_potential(const oz_game *g)
{
    /* :48-50 */
    oz_game copy = *g;
    oz_count_score(&copy);
    return copy.score[0] - copy.score[1];
}

int oz_runner_step(oz_runner *q, int action, oz_rng *r, int64_t *reward, int *done)
{
    /* :43-55 */
    int d, c, p;
    oz_deserialize(action, &d, &c, &p);
    int st = oz_step(&q->game, d, c, p, r);
    if (st) return st;
    q->move_counter += 1;
    for (;;) {
        uint8_t mask[180];
        oz_check_all_valid(&q->game, mask);
        int nvalid = 0;
        for (int i = 0; i < 180; i++) nvalid += mask[i];
        if (!((q->game.current_player != 1 || nvalid < 2) && !oz_is_end_of_game(&q->game))) break;
        st = oz_runner_opponent_move(q, r);
        if (st) return st;
    }
    int64_t nps = oz_potential(&q->game);
    *reward = nps - q->player_score;
    q->player_score = nps;
    *done = oz_is_end_of_game(&q->game);
    return OZ_OK;
}

/* ---- GameRunner with ANY opponent: game_runner.py:27-30 stores whatever it is given (scripts/run_batch.py:6-10 and
 * tests/test_nn_runner.py:63-67, 84-90 pass a second Agent); opponent_move hands it the observation from the mover's
 * perspective and the legal mask and plays what it answers (:37-42).  `opp` stands for opponent.get_a_output; a negative
 * answer stands for the exception an Agent raises on an all-false mask (model.py:33-34 IllegalMask). ---- */
int oz_runner_opponent_move_with(oz_runner *q, oz_rng *r, oz_opponent_fn opp, void *ctx)
{
    int64_t state[136];
    uint8_t mask[180];
    oz_get_state(&q->game, q->game.current_player - 1, state);       /* :38 */
    oz_check_all_valid(&q->game, mask);                              /* :39 */
    int a = opp(ctx, state, mask);                                   /* :40 */
    if (a < 0) return OZ_STUCK;
    int d, c, p;
    oz_deserialize(a, &d, &c, &p);
    int st = oz_step(&q->game, d, c, p, r);                          /* :41 */
    if (st) return st;
    q->move_counter += 1;                                            /* :42 */
    return OZ_OK;
}

int oz_runner_reset_with(oz_runner *q, oz_rng *r, oz_opponent_fn opp, void *ctx)
{
    /* :76-85 */
    int st = runner_reset_noplay(q, r);
    if (st) return st;
    while (q->game.current_player != 1) {                            /* :84 */
        st = oz_runner_opponent_move_with(q, r, opp, ctx);           /* :85 */
        if (st) return st;
    }
    return OZ_OK;
}

int oz_runner_step_with(oz_runner *q, int action, oz_rng *r, oz_opponent_fn opp, void *ctx, int64_t *reward, int *done)
{
    /* :43-55 */
    int d, c, p;
    oz_deserialize(action, &d, &c, &p);
    int st = oz_step(&q->game, d, c, p, r);                          /* :44 */
    if (st) return st;
    q->move_counter += 1;                                            /* :45 */
    for (;;) {
        uint8_t mask[180];
        oz_check_all_valid(&q->game, mask);
        int nvalid = 0;
        for (int i = 0; i < 180; i++) nvalid += mask[i];
        if (!((q->game.current_player != 1 || nvalid < 2) && !oz_is_end_of_game(&q->game))) break;   /* :46 */
        st = oz_runner_opponent_move_with(q, r, opp, ctx);           /* :47 */
        if (st) return st;
    }
    int64_t nps = oz_potential(&q->game);                            /* :48-50 */
    *reward = nps - q->player_score;                                 /* :51 */
    q->player_score = nps;                                           /* :52 */
    *done = oz_is_end_of_game(&q->game);                             /* :55 */
    return OZ_OK;
}

/* GameRunner.step with any opponent (opp == NULL: the default RandomAgent drawing from r) under the MOVE LIMIT: a move of either side
 * that ends a round without ending the game when the episode has played `limit` moves cuts the episode -- reward 0, *done = 3 (the
 * caller resets, as after done = 1).  limit <= 0: oz_runner_step_with / oz_runner_step. */
int oz_runner_step_limited(oz_runner *q, int action, oz_rng *r, oz_opponent_fn opp, void *ctx, int64_t limit, int64_t *reward, int *done)
{
    int d, c, p;
    oz_deserialize(action, &d, &c, &p);
    *reward = 0; *done = 0;
    int st = oz_step_limited(&q->game, d, c, p, r, q->move_counter + 1, limit);      /* :44 */
    if (st == OZ_TRUNCATED) { *done = 3; return OZ_OK; }
    if (st) return st;
    q->move_counter += 1;                                                             /* :45 */
    for (;;) {
        uint8_t mask[180];
        oz_check_all_valid(&q->game, mask);
        int nvalid = 0;
        for (int i = 0; i < 180; i++) nvalid += mask[i];
        if (!((q->game.current_player != 1 || nvalid < 2) && !oz_is_end_of_game(&q->game))) break;   /* :46 */
        int64_t state[136];
        oz_get_state(&q->game, q->game.current_player - 1, state);                   /* :38 */
        int a = opp ? opp(ctx, state, mask) : oz_random_agent(mask, r);              /* :40 */
        if (a < 0) return OZ_STUCK;
        oz_deserialize(a, &d, &c, &p);
        st = oz_step_limited(&q->game, d, c, p, r, q->move_counter + 1, limit);      /* :41 */
        if (st == OZ_TRUNCATED) { *done = 3; return OZ_OK; }
        if (st) return st;
        q->move_counter += 1;                                                         /* :42 */
    }
    int64_t nps = oz_potential(&q->game);
    *reward = nps - q->player_score;
    q->player_score = nps;
    *done = oz_is_end_of_game(&q->game);
    return OZ_OK;
}

/* ------------------------------------------------------------------------- */
/* canonical 128-byte record (layout documented in include/azul_hip.h)       */
/* ------------------------------------------------------------------------- */
static void put16(uint8_t *p, int v) { p[0] = (uint8_t)(v & 0xff); p[1] = (uint8_t)((v >> 8) & 0xff); }
static int  get16s(const uint8_t *p) { return (int16_t)(p[0] | (p[1] << 8)); }
static int  get16u(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int oz_pack(const oz_runner *q, uint8_t rec[128])
{
    const oz_game *g = &q->game;
    int bad = 0;
    memset(rec, 0, 128);
    if (g->players != 2) return -1;
#define CHK(v, lo, hi) do { if ((v) < (lo) || (v) > (hi)) bad = 1; } while (0)
    for (int d = 0; d < 5; d++) for (int c = 0; c < 5; c++) { CHK(g->displays[d][c], 0, 15); rec[d * 5 + c] = (uint8_t)g->displays[d][c]; }
    for (int c = 0; c < 6; c++) { CHK(g->center[c], 0, c == 5 ? 1 : 15); rec[25 + c] = (uint8_t)g->center[c]; }
    CHK(g->current_player, 0, 2); CHK(g->next_first_player, 0, 2);
    rec[31] = (uint8_t)((g->current_player & 7) | ((g->next_first_player & 7) << 3) | ((g->end_of_game ? 1 : 0) << 6));
    for (int p = 0; p < 2; p++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) {
        CHK(g->pattern_lines[p][r][c], 0, 15);
        rec[32 + p * 25 + r * 5 + c] = (uint8_t)g->pattern_lines[p][r][c];
    }
    for (int p = 0; p < 2; p++) { CHK(g->floors[p], 0, 7); rec[82 + p] = (uint8_t)g->floors[p]; }
    for (int p = 0; p < 2; p++) {
        uint32_t w = 0;
        for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) if (g->walls[p][r][c]) w |= 1u << (r * 5 + c);
        rec[84 + 4 * p] = (uint8_t)(w & 0xff); rec[85 + 4 * p] = (uint8_t)((w >> 8) & 0xff);
        rec[86 + 4 * p] = (uint8_t)((w >> 16) & 0xff); rec[87 + 4 * p] = (uint8_t)((w >> 24) & 0xff);
    }
    for (int p = 0; p < 2; p++) { CHK(g->score[p], -32768, 32767); put16(rec + 92 + 2 * p, (int)g->score[p]); }
    for (int c = 0; c < 5; c++) { CHK(g->box[c], 0, 255); rec[96 + c] = (uint8_t)g->box[c]; }
    for (int c = 0; c < 5; c++) { CHK(g->lid[c], 0, 255); rec[101 + c] = (uint8_t)g->lid[c]; }
    CHK(g->turn_counter, 0, 65535); put16(rec + 106, g->turn_counter);
    for (int p = 0; p < 2; p++) { CHK(g->first_player_stats[p], 0, 65535); put16(rec + 108 + 2 * p, (int)g->first_player_stats[p]); }
    for (int p = 0; p < 2; p++) { CHK(g->floor_penalty[p], -32768, 32767); put16(rec + 112 + 2 * p, (int)g->floor_penalty[p]); }
    for (int p = 0; p < 2; p++) { CHK(g->max_combo[p], 0, 255); rec[116 + p] = (uint8_t)g->max_combo[p]; }
    for (int p = 0; p < 2; p++) for (int k = 0; k < 3; k++) { CHK(g->completed_lines[p][k], 0, 255); rec[118 + p * 3 + k] = (uint8_t)g->completed_lines[p][k]; }
    CHK(q->player_score, -32768, 32767); put16(rec + 124, (int)q->player_score);
    CHK(q->move_counter, 0, 65535); put16(rec + 126, (int)q->move_counter);
#undef CHK
    return bad ? -1 : 0;
}

void oz_unpack(oz_runner *q, const uint8_t rec[128], int tile_pool, int first_player)
{
    oz_game *g = &q->game;
    memset(q, 0, sizeof(*q));
    q->tile_pool = tile_pool;
    q->first_player = first_player;
    g->players = 2;
    g->tile_pool = tile_pool;
    for (int d = 0; d < 5; d++) for (int c = 0; c < 5; c++) g->displays[d][c] = rec[d * 5 + c];
    for (int c = 0; c < 6; c++) g->center[c] = rec[25 + c];
    g->current_player = rec[31] & 7;
    g->next_first_player = (rec[31] >> 3) & 7;
    g->end_of_game = (rec[31] >> 6) & 1;
    for (int p = 0; p < 2; p++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++)
        g->pattern_lines[p][r][c] = rec[32 + p * 25 + r * 5 + c];
    for (int p = 0; p < 2; p++) g->floors[p] = rec[82 + p];
    for (int p = 0; p < 2; p++) {
        uint32_t w = (uint32_t)rec[84 + 4 * p] | ((uint32_t)rec[85 + 4 * p] << 8) | ((uint32_t)rec[86 + 4 * p] << 16) | ((uint32_t)rec[87 + 4 * p] << 24);
        for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) g->walls[p][r][c] = (uint8_t)((w >> (r * 5 + c)) & 1);
    }
    for (int p = 0; p < 2; p++) g->score[p] = get16s(rec + 92 + 2 * p);
    for (int c = 0; c < 5; c++) g->box[c] = rec[96 + c];
    for (int c = 0; c < 5; c++) g->lid[c] = rec[101 + c];
    g->turn_counter = get16u(rec + 106);
    for (int p = 0; p < 2; p++) g->first_player_stats[p] = get16u(rec + 108 + 2 * p);
    for (int p = 0; p < 2; p++) g->floor_penalty[p] = get16s(rec + 112 + 2 * p);
    for (int p = 0; p < 2; p++) g->max_combo[p] = rec[116 + p];
    for (int p = 0; p < 2; p++) for (int k = 0; k < 3; k++) g->completed_lines[p][k] = rec[118 + p * 3 + k];
    q->player_score = get16s(rec + 124);
    q->move_counter = get16u(rec + 126);
}

/* ------------------------------------------------------------------------- */
/* wide record: 256 bytes, 2..4 players (layout: include/azul_hip.h, row N4)  */
/* ------------------------------------------------------------------------- */
int oz_pack_np(const oz_game *g, uint8_t rec[256])
{
    int bad = 0;
    const int P = g->players;
    memset(rec, 0, 256);
    if (P < 2 || P > OZ_MAXP) return -1;
#define CHK(v, lo, hi) do { if ((v) < (lo) || (v) > (hi)) bad = 1; } while (0)
    for (int d = 0; d < 5; d++) for (int c = 0; c < 5; c++) { CHK(g->displays[d][c], 0, 255); rec[d * 5 + c] = (uint8_t)g->displays[d][c]; }
    for (int c = 0; c < 6; c++) { CHK(g->center[c], 0, 255); rec[25 + c] = (uint8_t)g->center[c]; }
    CHK(g->current_player, 0, P); CHK(g->next_first_player, 0, P);
    rec[31] = (uint8_t)((g->current_player & 7) | ((g->next_first_player & 7) << 3) | ((g->end_of_game ? 1 : 0) << 6));
    for (int p = 0; p < P; p++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) {
        CHK(g->pattern_lines[p][r][c], 0, 255);
        rec[32 + p * 25 + r * 5 + c] = (uint8_t)g->pattern_lines[p][r][c];
    }
    for (int p = 0; p < P; p++) { CHK(g->floors[p], 0, 7); rec[132 + p] = (uint8_t)g->floors[p]; }
    for (int p = 0; p < P; p++) {
        uint32_t w = 0;
        for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) if (g->walls[p][r][c]) w |= 1u << (r * 5 + c);
        for (int k = 0; k < 4; k++) rec[136 + 4 * p + k] = (uint8_t)((w >> (8 * k)) & 0xff);
    }
    for (int p = 0; p < P; p++) { CHK(g->score[p], -32768, 32767); put16(rec + 152 + 2 * p, (int)g->score[p]); }
    for (int c = 0; c < 5; c++) { CHK(g->box[c], 0, 255); rec[160 + c] = (uint8_t)g->box[c]; }
    for (int c = 0; c < 5; c++) { CHK(g->lid[c], 0, 255); rec[165 + c] = (uint8_t)g->lid[c]; }
    CHK(g->turn_counter, 0, 65535); put16(rec + 170, g->turn_counter);
    for (int p = 0; p < P; p++) { CHK(g->first_player_stats[p], 0, 65535); put16(rec + 172 + 2 * p, (int)g->first_player_stats[p]); }
    for (int p = 0; p < P; p++) { CHK(g->floor_penalty[p], -32768, 32767); put16(rec + 180 + 2 * p, (int)g->floor_penalty[p]); }
    for (int p = 0; p < P; p++) { CHK(g->max_combo[p], 0, 255); rec[188 + p] = (uint8_t)g->max_combo[p]; }
    for (int p = 0; p < P; p++) for (int k = 0; k < 3; k++) { CHK(g->completed_lines[p][k], 0, 255); rec[192 + p * 3 + k] = (uint8_t)g->completed_lines[p][k]; }
    rec[204] = (uint8_t)P;
    /* beyond the reference: factory displays 5 .. 8 live in bytes 208 .. 227 and byte 205 holds their total number when it is not
     * the reference's five -- a five-display record is byte for byte what it was */
    if (nd(g) != 5) {
        rec[205] = (uint8_t)nd(g);
        for (int d = 5; d < nd(g); d++) for (int c = 0; c < 5; c++) { CHK(g->xdisplays[d - 5][c], 0, 255); rec[208 + (d - 5) * 5 + c] = (uint8_t)g->xdisplays[d - 5][c]; }
    }
#undef CHK
    return bad ? -1 : 0;
}

void oz_unpack_np(oz_game *g, const uint8_t rec[256], int tile_pool)
{
    memset(g, 0, sizeof(*g));
    const int P = rec[204];
    g->players = P;
    g->tile_pool = tile_pool;
    for (int d = 0; d < 5; d++) for (int c = 0; c < 5; c++) g->displays[d][c] = rec[d * 5 + c];
    for (int c = 0; c < 6; c++) g->center[c] = rec[25 + c];
    g->current_player = rec[31] & 7;
    g->next_first_player = (rec[31] >> 3) & 7;
    g->end_of_game = (rec[31] >> 6) & 1;
    for (int p = 0; p < P; p++) for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++)
        g->pattern_lines[p][r][c] = rec[32 + p * 25 + r * 5 + c];
    for (int p = 0; p < P; p++) g->floors[p] = rec[132 + p];
    for (int p = 0; p < P; p++) {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= (uint32_t)rec[136 + 4 * p + k] << (8 * k);
        for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) g->walls[p][r][c] = (uint8_t)((w >> (r * 5 + c)) & 1);
    }
    for (int p = 0; p < P; p++) g->score[p] = get16s(rec + 152 + 2 * p);
    for (int c = 0; c < 5; c++) g->box[c] = rec[160 + c];
    for (int c = 0; c < 5; c++) g->lid[c] = rec[165 + c];
    g->turn_counter = get16u(rec + 170);
    for (int p = 0; p < P; p++) g->first_player_stats[p] = get16u(rec + 172 + 2 * p);
    for (int p = 0; p < P; p++) g->floor_penalty[p] = get16s(rec + 180 + 2 * p);
    for (int p = 0; p < P; p++) g->max_combo[p] = rec[188 + p];
    for (int p = 0; p < P; p++) for (int k = 0; k < 3; k++) g->completed_lines[p][k] = rec[192 + p * 3 + k];
    g->n_displays = rec[205] ? rec[205] : 5;
    for (int d = 5; d < g->n_displays && d < OZ_MAX_DISPLAYS; d++) for (int c = 0; c < 5; c++) g->xdisplays[d - 5][c] = rec[208 + (d - 5) * 5 + c];
}

/* ------------------------------------------------------------------------- */
/* batched drivers                                                           */
/* ------------------------------------------------------------------------- */
int oz_stream_start(oz_runner *q, oz_rng *r, uint64_t seed, int first_player, int tile_pool)
{
    /* random.seed(seed); runner = GameRunner(rules=...); then the first reset() (without the
     * opponent pre-moves: in flat self-play those are ordinary env moves). */
    oz_rng_seed(r, seed);
    int st = oz_runner_init(q, first_player, tile_pool, r);
    if (st) return st;
    return runner_reset_noplay(q, r);
}

int oz_stream_advance(oz_runner *q, oz_rng *r, int n_steps,
                      uint8_t *mask, int32_t *action, int32_t *reward, uint8_t *done,
                      uint8_t *rec_after, uint64_t *stuck_count, uint64_t *episodes, double *stats_sum)
{
    return oz_stream_advance_limited(q, r, n_steps, 0, mask, action, reward, done, rec_after, stuck_count, episodes, stats_sum);
}

/* the same flat loop under the move limit (limit <= 0: none): a cut episode reports done = 3, counts as a restarted slot, not as an episode */
int oz_stream_advance_limited(oz_runner *q, oz_rng *r, int n_steps, int64_t limit,
                              uint8_t *mask, int32_t *action, int32_t *reward, uint8_t *done,
                              uint8_t *rec_after, uint64_t *stuck_count, uint64_t *episodes, double *stats_sum)
{
    for (int t = 0; t < n_steps; t++) {
        uint8_t m[180];
        oz_check_all_valid(&q->game, m);
        if (mask) memcpy(mask + (size_t)t * 180, m, 180);
        int a = oz_random_agent(m, r);
        if (a < 0) {
            /* hazard H3: no legal move although the round has not ended (only the token is left) */
            if (stuck_count) (*stuck_count)++;
            if (action) action[t] = -1;
            if (reward) reward[t] = 0;
            if (done) done[t] = 2;
            if (rec_after) oz_pack(q, rec_after + (size_t)t * 128);
            int st = runner_reset_noplay(q, r);
            if (st) return st;
            continue;
        }
        int d, c, p;
        oz_deserialize(a, &d, &c, &p);
        int st = oz_step_limited(&q->game, d, c, p, r, q->move_counter + 1, limit);
        const int cut = st == OZ_TRUNCATED;
        if (st && !cut) return st;
        q->move_counter += 1;
        int64_t phi = oz_potential(&q->game);
        int64_t rew = phi - q->player_score;
        q->player_score = phi;
        int dn = cut ? 3 : oz_is_end_of_game(&q->game);
        if (action) action[t] = a;
        if (reward) reward[t] = (int32_t)rew;
        if (done) done[t] = (uint8_t)dn;
        if (rec_after) oz_pack(q, rec_after + (size_t)t * 128);
        if (dn) {
            if (cut) { if (stuck_count) (*stuck_count)++; }
            else {
                if (stats_sum) {
                    double s[10];
                    oz_get_statistics(&q->game, s);
                    for (int i = 0; i < 10; i++) stats_sum[i] += s[i];
                }
                if (episodes) (*episodes)++;
            }
            st = runner_reset_noplay(q, r);
            if (st) return st;
        }
    }
    return OZ_OK;
}

/* The flat random-agent loop for `players` players (row N4; what azul_batch_selfplay plays for 3- and 4-player batches):
 *     random.seed(seed); g = Azul(players=P, rules=rules); g.new_round()
 *     repeat: a = RandomAgent().get_a_output(None, mask of g)   (game_runner.py:87-97: any mask)
 *             g.step(*nn_deserialize(a))                         (azul.py:296-313: P-generic)
 *             when g.end_of_game (or nothing was legal, hazard H3): a fresh Azul(players=P, rules=rules) + new_round()
 * GameRunner's shaped reward is two-player (game_runner.py:50): this stream carries none.  rec_after: the 256-byte wide record
 * after the move, before a restart. */
int oz_stream_np_start(oz_game *g, oz_rng *r, uint64_t seed, int players, int first_player, int tile_pool)
{
    oz_rng_seed(r, seed);
    int st = oz_init(g, players, first_player, tile_pool, r);
    if (st) return st;
    return oz_new_round(g, r);
}

int oz_stream_np_advance(oz_game *g, oz_rng *r, int first_player, int n_steps, uint8_t *mask, int32_t *action, uint8_t *done,
                         uint8_t *rec_after, uint64_t *stuck_count, uint64_t *episodes, double *stats_sum)
{
    const int players = g->players, tile_pool = g->tile_pool;
    for (int t = 0; t < n_steps; t++) {
        uint8_t m[180];
        oz_check_all_valid(g, m);
        if (mask) memcpy(mask + (size_t)t * 180, m, 180);
        int a = oz_random_agent(m, r);
        int dn;
        if (a < 0) {
            if (stuck_count) (*stuck_count)++;
            dn = 2;
        } else {
            int d, c, p;
            oz_deserialize(a, &d, &c, &p);
            int st = oz_step(g, d, c, p, r);
            if (st) return st;
            dn = g->end_of_game ? 1 : 0;
        }
        if (action) action[t] = a;
        if (done) done[t] = (uint8_t)dn;
        if (rec_after) oz_pack_np(g, rec_after + (size_t)t * 256);
        if (dn) {
            if (dn == 1) {
                if (stats_sum) {
                    double s[10];
                    oz_get_statistics(g, s);
                    for (int i = 0; i < 10; i++) stats_sum[i] += s[i];
                }
                if (episodes) (*episodes)++;
            }
            int st = oz_init(g, players, first_player, tile_pool, r);
            if (st) return st;
            st = oz_new_round(g, r);
            if (st) return st;
        }
    }
    return OZ_OK;
}

/* The same flat loop under extended rules (flags in g->ext; ext == 0 reproduces oz_stream_np_*): mask rows and the sampler follow the
 * game's action space (oz_num_actions). */
int oz_stream_x_start(oz_game *g, oz_rng *r, uint64_t seed, int players, int first_player, int tile_pool, int ext)
{
    oz_rng_seed(r, seed);
    int st = oz_init_ext(g, players, first_player, tile_pool, ext, r);
    if (st) return st;
    return oz_new_round(g, r);
}

int oz_stream_x_advance(oz_game *g, oz_rng *r, int first_player, int n_steps, uint8_t *mask, int32_t *action, uint8_t *done,
                        uint8_t *rec_after, uint64_t *stuck_count, uint64_t *episodes, double *stats_sum)
{
    const int players = g->players, tile_pool = g->tile_pool, ext = g->ext, na = oz_num_actions(g);
    for (int t = 0; t < n_steps; t++) {
        uint8_t m[OZ_MAX_ACTIONS];
        oz_check_all_valid_x(g, m);
        if (mask) memcpy(mask + (size_t)t * na, m, (size_t)na);
        int a = g->end_of_game ? -1 : oz_random_agent_x(m, na, r);
        int dn;
        if (a < 0) {
            if (stuck_count) (*stuck_count)++;
            dn = 2;
        } else {
            int d, c, p;
            oz_deserialize_x(g, a, &d, &c, &p);
            int st = oz_step(g, d, c, p, r);
            if (st) return st;
            dn = g->end_of_game ? 1 : 0;
        }
        if (action) action[t] = a;
        if (done) done[t] = (uint8_t)dn;
        if (rec_after) oz_pack_np(g, rec_after + (size_t)t * 256);
        if (dn) {
            if (dn == 1) {
                if (stats_sum) {
                    double s[10];
                    oz_get_statistics(g, s);
                    for (int i = 0; i < 10; i++) stats_sum[i] += s[i];
                }
                if (episodes) (*episodes)++;
            }
            int st = oz_init_ext(g, players, first_player, tile_pool, ext, r);
            if (st) return st;
            st = oz_new_round(g, r);
            if (st) return st;
        }
    }
    return OZ_OK;
}

typedef struct {
    uint64_t seed_base;
    int n_streams, n_steps, n_threads, tid, first_player, tile_pool;
    uint64_t moves, checksum;
} bench_arg;

static void *bench_worker(void *vp)
{
    bench_arg *a = (bench_arg *)vp;
    oz_runner *q = (oz_runner *)malloc(sizeof(oz_runner));
    oz_rng *r = (oz_rng *)malloc(sizeof(oz_rng));
    a->moves = 0;
    a->checksum = 0;
    for (int s = a->tid; s < a->n_streams; s += a->n_threads) {
        uint64_t stuck = 0;
        if (oz_stream_start(q, r, a->seed_base + (uint64_t)s, a->first_player, a->tile_pool)) continue;
        if (oz_stream_advance(q, r, a->n_steps, 0, 0, 0, 0, 0, &stuck, 0, 0)) continue;
        a->moves += (uint64_t)a->n_steps - stuck;
        uint8_t rec[128];
        oz_pack(q, rec);
        uint64_t h = 1469598103934665603ull;               /* FNV-1a over the final record */
        for (int i = 0; i < 128; i++) { h ^= rec[i]; h *= 1099511628211ull; }
        a->checksum += h * (uint64_t)(s + 1);
    }
    free(q);
    free(r);
    return 0;
}

uint64_t oz_bench_selfplay(uint64_t seed_base, int n_streams, int n_steps, int n_threads,
                           int first_player, int tile_pool, uint64_t *checksum)
{
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    bench_arg *args = (bench_arg *)malloc(sizeof(bench_arg) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        args[t].seed_base = seed_base; args[t].n_streams = n_streams; args[t].n_steps = n_steps;
        args[t].n_threads = n_threads; args[t].tid = t; args[t].first_player = first_player; args[t].tile_pool = tile_pool;
        pthread_create(&th[t], 0, bench_worker, &args[t]);
    }
    uint64_t moves = 0, cs = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], 0);
        moves += args[t].moves;
        cs += args[t].checksum;
    }
    if (checksum) *checksum = cs;
    free(th);
    free(args);
    return moves;
}
