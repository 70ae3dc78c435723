/*
 * azul_hip.h -- C ABI of libazulhip.so: the MI355X (gfx950) batched Azul environment.
 *
 * The reference (patello/azul_deep_reinforcement_learning) has no FFI; its boundary for this path is
 * the Python API of azulnet/azul.py and azulnet/game_runner.py.  Each entry point below is the batched
 * (N games, two games per 64-lane wavefront) replacement of one reference function, cited per entry as
 * file:line relative to the reference tree.  The Python mirror in azul_deep_reinforcement_learning_amd/
 * binds these with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every launch is asynchronous on `stream`; pointers named *_dev are DEVICE pointers owned by the
 *     caller (e.g. torch tensors via data_ptr()); pointers named *_host are host pointers;
 *   - optional outputs may be NULL; `active_dev` (uint8[N], NULL = all games) selects games;
 *   - return value: AZUL_SUCCESS or a negative AZUL_ERR_* (API misuse / HIP failure).  Per-game rule
 *     outcomes are reported in `status_dev` (uint8[N]) -- the single-game Python facade maps them to
 *     the reference's exceptions (IllegalMove / GameEnded, azul.py:8-15);
 *   - devices: a batch lives on the HIP device that was current in azul_batch_create; every entry that takes the batch
 *     runs there whatever device is current in the calling thread (the caller's current device is restored on return)
 *     and refuses a `stream` that belongs to another device (AZUL_ERR_INVALID).  The batch-less entries (policy /
 *     learner kernels) run on the device `stream` belongs to (NULL stream: the current device);
 *   - there is no CPU implementation behind this ABI.
 *
 * Game record: 128 bytes per game, little endian, array-of-records [N][128] (one cache line per game;
 * a wavefront reads its game with three coalesced loads):
 *
 *   off size field                         reference (azul.py / game_runner.py)
 *     0  25  u8  displays[5][5]            game_board_displays            azul.py:19
 *    25   6  u8  center[6]                 game_board_center ([5]=token)  azul.py:20,71
 *    31   1  u8  flags                     bits0-2 current_player, bits3-5 next_first_player, bit6 end_of_game
 *    32  50  u8  pattern_lines[2][5][5]                                   azul.py:21-22
 *    82   2  u8  floors[2]                                                azul.py:25
 *    84   8  u32 walls[2]                  bit 5*row+colour               azul.py:23-24
 *    92   4  i16 score[2]                                                 azul.py:26
 *    96   5  u8  box[5]                    box_tiles ("Lid" rule)         azul.py:51
 *   101   5  u8  lid[5]                    lid_tiles                      azul.py:52
 *   106   2  u16 turn_counter                                             azul.py:30
 *   108   4  u16 first_player_stats[2]                                    azul.py:31
 *   112   4  i16 floor_penalty[2]          (sum of the negative penalties) azul.py:32
 *   116   2  u8  max_combo[2]                                             azul.py:33
 *   118   6  u8  completed_lines[2][3]     0 row, 1 colour, 2 column      azul.py:58
 *   124   2  i16 player_score              GameRunner.player_score        game_runner.py:35
 *   126   2  u16 move_counter              GameRunner.move_counter        game_runner.py:36
 *
 * Wide record (batches of THREE or FOUR players, azul_batch_create_players; row N4 of SURVEY.md 8f): 256 bytes per game.
 * The reference deals five displays whatever the number of players (azul.py:19), so only the per-player fields grow:
 *
 *   off size field                         reference (azul.py)
 *     0  25  u8  displays[5][5]            azul.py:19
 *    25   6  u8  center[6]                 azul.py:20,71
 *    31   1  u8  flags                     bits0-2 current_player (0..4), bits3-5 next_first_player (0..4), bit6 end_of_game
 *    32 100  u8  pattern_lines[4][5][5]    azul.py:21-22   (players beyond `players` stay zero)
 *   132   4  u8  floors[4]                 azul.py:25
 *   136  16  u32 walls[4]                  azul.py:23-24
 *   152   8  i16 score[4]                  azul.py:26
 *   160   5  u8  box[5]                    azul.py:51
 *   165   5  u8  lid[5]                    azul.py:52
 *   170   2  u16 turn_counter              azul.py:30
 *   172   8  u16 first_player_stats[4]     azul.py:31
 *   180   8  i16 floor_penalty[4]          azul.py:32
 *   188   4  u8  max_combo[4]              azul.py:33
 *   192  12  u8  completed_lines[4][3]     azul.py:58
 *   204   1  u8  players                   azul.py:28
 *   205   1  u8  n_displays                0 = the reference's five (a five-display record is byte for byte what it was); else 7 / 9
 *   206   2  u8  reserved0                 zero
 *   208  20  u8  xdisplays[4][5]           factory displays 5 .. 8, beyond the reference's five (AZUL_RULE_DISPLAYS_2P1); zero otherwise
 *   228  28      reserved (zero)
 *
 * Random numbers: every game owns a CPython-exact MT19937 stream (624 words + index), i.e. what the
 * reference consumes through the process-global `random` module (azul.py:37,78,87; game_runner.py:97).
 */
#ifndef AZUL_HIP_H
#define AZUL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZUL_RECORD_BYTES 128
#define AZUL_RECORD_BYTES_WIDE 256   /* batches of 3 or 4 players */
#define AZUL_NUM_ACTIONS  180          /* the reference's action space: 6 sources x 5 colours x 6 rows (game_runner.py:102-117) */
#define AZUL_OBS_SIZE     136          /* the reference's two-player observation (game_runner.py:65-72) */
#define AZUL_MAX_ACTIONS  300          /* (9 + 1) sources x 5 x 6: four players with AZUL_RULE_DISPLAYS_2P1 (azul_batch_num_actions) */
#define AZUL_MAX_OBS      260          /* 5 * 9 + 6 + 52 * 4 + 1 (azul_batch_obs_size) */
#define AZUL_MT_WORDS     624
#define AZUL_NUM_STATS    10

/* return codes */
#define AZUL_SUCCESS        0
#define AZUL_ERR_INVALID   (-1)   /* bad argument */
#define AZUL_ERR_HIP       (-2)   /* HIP runtime failure (azul_last_error_string) */
#define AZUL_ERR_RANGE     (-3)   /* record field outside the representable domain */
#define AZUL_ERR_RULE      (-4)   /* IllegalRule, azul.py:41,54 */

/* per-game status (uint8) */
#define AZUL_OK            0
#define AZUL_ILLEGAL_MOVE  1      /* IllegalMove (azul.py:301-302); state untouched */
#define AZUL_GAME_ENDED    2      /* GameEnded   (azul.py:298-299) */
#define AZUL_STUCK         3      /* no legal move although the round is not over (SURVEY hazard H3) */
#define AZUL_BAD_ACTION    4      /* action outside [0,180) */
#define AZUL_BOX_EMPTY     5      /* "Lid" pool and lid both empty at a draw (reference raises) */
#define AZUL_TRUNCATED     6      /* azul_batch_set_move_limit: the episode was cut at the end of a round (beyond the reference) */

/* rules (azul.py:35-56) */
#define AZUL_POOL_RANDOM   0
#define AZUL_POOL_LID      1
#define AZUL_FIRST_RANDOM  0      /* "Random"; 1 or 2 = fixed first player */
/* Extended rules (azul_batch_create_rules), all OFF by default -- BEYOND THE REFERENCE, "parity unpinned": the reference implements none of
 * them (five displays for any number of players, azul.py:19 and the TODO at tests/test_azul.py:14; TODOs at azul.py:72,86; line bonuses
 * per round, azul.py:266-288).  Source: the published Azul rulebook, restated in oracle/azul_oracle.c (OZ_EXT_*) and, independently, in
 * tests/ext_rules_model.py. */
#define AZUL_RULE_DISPLAYS_2P1  1u  /* 5 / 7 / 9 factory displays for 2 / 3 / 4 players: (D + 1) * 30 actions, a = source + (D + 1) colour + 5 (D + 1) row */
#define AZUL_RULE_END_BONUS     2u  /* +2 per complete row, +7 per complete column, +10 per complete colour ONCE at the end of the game (not per round) */
#define AZUL_RULE_SHORT_DEAL    4u  /* bag and lid both empty at a draw: the round starts with what could be dealt (instead of AZUL_BOX_EMPTY) */
#define AZUL_RULE_FINITE_BAG    8u  /* tile_pool Random draws WITHOUT replacement from a 100-tile bag, discards go to the lid: _randbelow(tiles left) */

/* observation perspective (game_runner.py:56 `perspective`) */
#define AZUL_PERSP_PLAYER0  0
#define AZUL_PERSP_PLAYER1  1
#define AZUL_PERSP_CURRENT  2     /* current_player - 1, as opponent_move uses (game_runner.py:38); two-player reference batches */
#define AZUL_PERSP_MOVER    7     /* the same for ANY batch (with three / four players 2 is a player) */

/* flags written by azul_batch_flags */
#define AZUL_FLAG_END_OF_ROUND 1  /* azul.py:182-183 */
#define AZUL_FLAG_END_OF_GAME  2  /* azul.py:184-191 (walls) */
#define AZUL_FLAG_ENDED_FLAG   4  /* Azul.end_of_game attribute, azul.py:29 */

typedef struct azul_batch azul_batch_t;

const char *azul_last_error_string(void);
const char *azul_version(void);

/* ---- lifetime ------------------------------------------------------------------------------- */
/* N two-player games on the current HIP device, rules as in Azul(rules=...) (azul.py:35-56). */
int azul_batch_create(azul_batch_t **out, int n_games, int first_player, int tile_pool);
/* N games of `players` = 2, 3 or 4 players: Azul(players=..., rules=...) (azul.py:18-33; first_player 0 = "Random" or 1..players).
 * players = 2 is azul_batch_create.  For 3 and 4 players the records are the 256-byte wide records and the entries that
 * mirror Azul's own methods work -- azul_batch_init / _new_round / _move / _legal_mask / _next_player / _flags / _count_score /
 * _step / _statistics, the state and RNG I/O, azul_batch_random_action / _sample_mask -- while the entries that mirror
 * GameRunner (two-player in the reference: game_runner.py:50,57) return AZUL_ERR_INVALID. */
int azul_batch_create_players(azul_batch_t **out, int n_games, int players, int first_player, int tile_pool);
/* The same with extended-rule flags (AZUL_RULE_*; 0 = azul_batch_create_players).  Batches of three / four players and batches with any
 * flag set hold 256-byte wide records and support the entries that mirror Azul's own methods (azul_batch_init / _new_round / _move /
 * _legal_mask / _next_player / _flags / _count_score / _step / _statistics), azul_batch_random_action / _sample_mask, azul_batch_observe
 * (get_state for P players, game_runner.py:56-72), the state and RNG I/O and azul_batch_selfplay; mask rows are azul_batch_num_actions
 * bytes, observations azul_batch_obs_size floats.  The entries that mirror GameRunner.step / reset and the policy entries (compiled for
 * 180 actions / 136 observations) return AZUL_ERR_INVALID for them.  AZUL_RULE_FINITE_BAG with AZUL_POOL_LID: AZUL_ERR_RULE. */
int azul_batch_create_rules(azul_batch_t **out, int n_games, int players, int first_player, int tile_pool, unsigned rule_flags);
int azul_batch_players(const azul_batch_t *b);
int azul_batch_displays(const azul_batch_t *b);            /* 5, or 2 * players + 1 */
unsigned azul_batch_rule_flags(const azul_batch_t *b);
int azul_batch_num_actions(const azul_batch_t *b);         /* (displays + 1) * 30: 180 / 240 / 300   (game_runner.py:115) */
int azul_batch_obs_size(const azul_batch_t *b);            /* 5 displays + 6 + 52 players + 1: 136 for the reference's game (game_runner.py:65-72) */
int azul_batch_record_bytes(const azul_batch_t *b);       /* 128, or 256 for 3 / 4 players and extended-rule batches */
int azul_batch_destroy(azul_batch_t *b);
int azul_batch_size(const azul_batch_t *b);
/* device pointers of the resident arrays (for zero-copy views): records [N][128] u8, MT words [N][624] u32, MT index [N] u32 */
void *azul_batch_state_dev(azul_batch_t *b);
void *azul_batch_mt_dev(azul_batch_t *b);
void *azul_batch_mtpos_dev(azul_batch_t *b);

/* ---- state / RNG I/O (superset of export_JSON / import_JSON, azul.py:90-117) ------------------ */
int azul_batch_get_state(azul_batch_t *b, int first, int count, void *records_host, void *stream);
int azul_batch_set_state(azul_batch_t *b, int first, int count, const void *records_host, void *stream);
/* random.getstate() / random.setstate() of one game's stream: 624 words + index */
int azul_batch_get_rng(azul_batch_t *b, int game, uint32_t *mt_host, uint32_t *pos_host, void *stream);
int azul_batch_set_rng(azul_batch_t *b, int game, const uint32_t *mt_host, uint32_t pos, void *stream);
/* the same for games first .. first+count-1 (checkpoints): mt_host [count][624], pos_host [count] */
int azul_batch_get_rng_range(azul_batch_t *b, int first, int count, uint32_t *mt_host, uint32_t *pos_host, void *stream);
int azul_batch_set_rng_range(azul_batch_t *b, int first, int count, const uint32_t *mt_host, const uint32_t *pos_host, void *stream);
/* random.seed(int) per game (CPython init_by_array): seed[g] = seeds_host[g], or seed_base + g when seeds_host is NULL */
int azul_batch_seed(azul_batch_t *b, uint64_t seed_base, const uint64_t *seeds_host, void *stream);

/* ---- rules: one entry per reference method ---------------------------------------------------- */
int azul_batch_init(azul_batch_t *b, const uint8_t *active_dev, void *stream);                 /* Azul.__init__      azul.py:18-61  */
int azul_batch_new_round(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream);   /* new_round  azul.py:64-89  */
int azul_batch_move(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, void *stream); /* move (unchecked) azul.py:118-161 */
int azul_batch_legal_mask(azul_batch_t *b, uint8_t *mask_dev /*[N][180]*/, void *stream);     /* check_all_valid    game_runner.py:113-117, azul.py:162-176 */
int azul_batch_next_player(azul_batch_t *b, const uint8_t *active_dev, void *stream);          /* next_player        azul.py:177-181 */
int azul_batch_flags(azul_batch_t *b, uint8_t *flags_dev /*[N]*/, void *stream);               /* is_end_of_round / is_end_of_game azul.py:182-191 */
int azul_batch_count_score(azul_batch_t *b, const uint8_t *active_dev, void *stream);          /* count_score        azul.py:192-295 */
int azul_batch_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev,
                    uint8_t *status_dev, void *stream);                                         /* step               azul.py:296-313 */
int azul_batch_statistics(azul_batch_t *b, double *stats_dev /*[N][10]*/, void *stream);       /* get_statistics     azul.py:314-315 */

/* ---- env wrapper: GameRunner with the default RandomAgent opponent ----------------------------- */
int azul_batch_runner_init(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream);   /* GameRunner.__init__ game_runner.py:23-36 */
int azul_batch_runner_reset(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream);  /* reset               game_runner.py:76-85 */
int azul_batch_runner_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev,
                           int32_t *reward_dev, uint8_t *done_dev, uint8_t *status_dev, void *stream);       /* step                game_runner.py:43-55 */
int azul_batch_observe(azul_batch_t *b, int perspective, float *obs_dev /*[N][136]*/, void *stream);         /* get_state           game_runner.py:56-72 */
int azul_batch_random_action(azul_batch_t *b, const uint8_t *active_dev, int32_t *actions_dev,
                             void *stream);                                                     /* RandomAgent.get_a_output game_runner.py:94-97 (-1: no legal move) */
/* the same sampler on a caller-supplied legal mask (what RandomAgent.get_a_output receives, game_runner.py:94-97) */
int azul_batch_sample_mask(azul_batch_t *b, const uint8_t *mask_dev /*[N][180]*/, const uint8_t *active_dev,
                           int32_t *actions_dev, void *stream);
int azul_batch_score_preview(azul_batch_t *b, int32_t *potential_dev /*[N]*/, void *stream);   /* deepcopy+count_score, score[0]-score[1]  game_runner.py:48-50 */

/* ---- one method call of the reference's SINGLE-GAME API on one game of a batch -------------------
 * What the Python facade (Azul / GameRunner / RandomAgent of this package, i.e. BASELINE configs[0]: tests/test_azul.py,
 * tests/test_game_runner.py, nn_runner.py:22-30 on the unchanged agent.py) needs per method call, in ONE submission and ONE
 * host synchronisation: [record_in -> device] [mt_in -> device] [the rule kernel on game `game`] [results -> host].
 * `record_in` / `mt_in` NULL: the device copies are current (the caller compared them with what it last received);
 * results are only transferred when asked for in `want`; the 624 MT19937 words only come back when the call regenerated them
 * (`rng_regenerated`; otherwise only the index moved: `pos_out`).  All pointers are HOST pointers.
 *   op                       reference method                                   arg
 *   AZUL_CALL_QUERY          is_legal_move / check_all_valid (azul.py:162-176, game_runner.py:113-117), is_end_of_round /
 *                            is_end_of_game (azul.py:182-191), get_state (game_runner.py:56-72), get_statistics (azul.py:314-315),
 *                            the what-if potential (game_runner.py:48-50): selected by `want`   perspective (WANT_OBS)
 *   AZUL_CALL_INIT           Azul.__init__ first-player draw (azul.py:37)                       -
 *   AZUL_CALL_NEW_ROUND      new_round (azul.py:64-89)                                          -
 *   AZUL_CALL_MOVE           move (azul.py:118-161)                                             action d + 6 c + 30 r
 *   AZUL_CALL_NEXT_PLAYER    next_player (azul.py:177-181)                                      -
 *   AZUL_CALL_COUNT_SCORE    count_score (azul.py:192-295)                                      -
 *   AZUL_CALL_STEP           step (azul.py:296-313)                                             action
 *   AZUL_CALL_RUNNER_INIT    GameRunner.__init__ (game_runner.py:23-36)                         -
 *   AZUL_CALL_RUNNER_RESET   GameRunner.reset (game_runner.py:76-85)                            -
 *   AZUL_CALL_RUNNER_STEP    GameRunner.step with the RandomAgent opponent (game_runner.py:43-55)   action
 *   AZUL_CALL_SAMPLE_MASK    RandomAgent.get_a_output on `mask_in` (game_runner.py:87-97)       -                            */
enum { AZUL_CALL_QUERY = 0, AZUL_CALL_INIT, AZUL_CALL_NEW_ROUND, AZUL_CALL_MOVE, AZUL_CALL_NEXT_PLAYER, AZUL_CALL_COUNT_SCORE,
       AZUL_CALL_STEP, AZUL_CALL_RUNNER_INIT, AZUL_CALL_RUNNER_RESET, AZUL_CALL_RUNNER_STEP, AZUL_CALL_SAMPLE_MASK };
#define AZUL_WANT_RECORD     1u    /* the game's record after the call -> record_out */
#define AZUL_WANT_MASK       2u    /* legal mask of the state after the call -> mask[180] */
#define AZUL_WANT_OBS        4u    /* get_state(perspective: a player, or AZUL_PERSP_MOVER; AZUL_CALL_QUERY: in arg, other ops: in obs_persp) -> obs[azul_batch_obs_size] */
#define AZUL_WANT_FLAGS      8u    /* AZUL_FLAG_* -> flags */
#define AZUL_WANT_POTENTIAL 16u    /* game_runner.py:48-50 -> potential (two players) */
#define AZUL_WANT_STATS     32u    /* get_statistics -> stats[10] */
#define AZUL_WANT_NEXT_ACTION 64u  /* ops that draw: what RandomAgent.get_a_output (game_runner.py:87-97) answers on the
                                      state after the call -- the question a GameRunner loop asks next (nn_runner.py:22-30) -- computed from the two
                                      words the stream hands out next, WITHOUT moving its index -> next_action.  A caller that plays the answer (same
                                      mask, stream untouched in between) advances the index by one random(), two words, itself: AZUL_WANT_POS_IN on
                                      its next call */
#define AZUL_WANT_POS_IN    128u   /* ops that draw, mt_in NULL: install pos_in as the stream's index before the op (the words stay) */
typedef struct azul_call {
    /* in */
    int32_t op, game, arg;
    uint32_t want;
    const void *record_in;          /* NULL or the game's record (azul_batch_record_bytes bytes; validated like azul_batch_set_state) */
    const uint32_t *mt_in;          /* NULL or 624 words: the stream to draw from (random.getstate()[1][:624]) */
    uint32_t pos_in;                /* with mt_in or AZUL_WANT_POS_IN: the stream's index to install, random.getstate()[1][624]; ignored otherwise */
    const uint8_t *mask_in;         /* AZUL_CALL_SAMPLE_MASK: uint8[azul_batch_num_actions] */
    void *record_out;               /* AZUL_WANT_RECORD */
    uint32_t *mt_out;               /* NULL or room for 624 words, written only when rng_regenerated */
    /* out */
    uint32_t pos_out;               /* index of the game's stream after the call */
    int32_t rng_regenerated;        /* the call regenerated the 624 words (reported by the kernel itself; they are in mt_out if given) */
    int32_t status;                 /* AZUL_OK / AZUL_ILLEGAL_MOVE / ... */
    int32_t reward, done;           /* AZUL_CALL_RUNNER_STEP */
    int32_t action;                 /* AZUL_CALL_SAMPLE_MASK (-1: nothing legal) */
    int32_t flags, potential;
    int32_t next_action;            /* AZUL_WANT_NEXT_ACTION: the action; -1 nothing legal; -2 not available (the op failed or did not draw, or the draw
                                       would cross the regeneration of the 624 words: ask AZUL_CALL_SAMPLE_MASK) */
    int32_t obs_persp;              /* in: perspective of AZUL_WANT_OBS for ops whose `arg` is an action (AZUL_CALL_QUERY reads it from arg) */
    uint8_t mask[AZUL_MAX_ACTIONS + 4];   /* azul_batch_num_actions bytes are written */
    float obs[AZUL_MAX_OBS];              /* azul_batch_obs_size floats are written */
    double stats[AZUL_NUM_STATS];
} azul_call_t;
int azul_game_call(azul_batch_t *b, azul_call_t *call, void *stream);

/* ---- policy-driven self-play (BASELINE configs[2]; callers: agent.py:64-81, nn_runner.py:17-47) ---------------- */
/* One env move for every game with caller-chosen actions, fused: Azul.step for the current player (azul.py:296-313) ->
 * per-move shaped reward (delta of the what-if potential, game_runner.py:48-52) -> done -> statistics + auto-reset of
 * finished games (game_runner.py:76-82) -> the NEXT decision's inputs: observation from `perspective`
 * (game_runner.py:56-72), legal mask (game_runner.py:113-117) and the player to move.  An illegal action leaves the
 * game untouched (status AZUL_ILLEGAL_MOVE, reward 0) and still returns that game's current observation and mask. */
int azul_batch_policy_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, int32_t *reward_dev,
                           uint8_t *done_dev, uint8_t *status_dev, int perspective, float *obs_next_dev /*[N][136]*/,
                           uint8_t *mask_next_dev /*[N][180]*/, uint8_t *player_next_dev /*[N]*/, void *stream);
/* one AGENT step of NNRunner.run_episode for every game (nn_runner.py:24-29): GameRunner.step (game_runner.py:43-55: the
 * agent's move, the opponent's RandomAgent replies, shaped reward, done) and, when the episode ends, the GameRunner.reset()
 * that opens the next run_episode (nn_runner.py:20, game_runner.py:76-82, incl. the opponent's opening moves); then the NEXT
 * decision's observation / mask / player.  done = 2 and status STUCK when nobody could move (slot restarted, reward 0). */
int azul_batch_agent_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, int32_t *reward_dev,
                          uint8_t *done_dev, uint8_t *status_dev, int perspective, float *obs_next_dev /*[N][136]*/,
                          uint8_t *mask_next_dev /*[N][180]*/, uint8_t *player_next_dev /*[N]*/, void *stream);
/* observation + mask + player to move in one launch (the first decision of a rollout) */
int azul_batch_observe_all(azul_batch_t *b, int perspective, float *obs_dev, uint8_t *mask_dev, uint8_t *player_dev, void *stream);
/* `seed` value that switches the policy entries below from sampling ("Distribution") to np.argmax over the masked softmax
 * ("Max": the first maximum in action order), the two action_selection modes of Agent.get_ac_output (agent.py:64-72) */
#define AZUL_POLICY_ARGMAX 0xFFFFFFFFFFFFFFFFull
/* policy head for a batch of action logits [N][180] + legal masks [N][180]: masked softmax, ONE categorical sample per game
 * (agent.py:64-72), the log-probability of that action and the entropy term -mean(log p over legal actions)
 * (nn_runner.py:32-40).  fp32; randomness = Philox4x32-10(seed, counter [+ *counter_dev], game): keep the step counter in
 * device memory (counter_dev) when the call is replayed from a HIP graph.  `game` is the GLOBAL id game_id_base + row, so a game
 * draws the same numbers however the batch is sharded over GPUs or split into parts.  Rows without a legal action give -1. */
int azul_policy_head(const float *logits_dev, const uint8_t *mask_dev, uint64_t seed, uint64_t counter, const uint64_t *counter_dev,
                     int n_games, uint32_t game_id_base, int32_t *action_dev, float *logp_dev, float *entropy_dev, void *stream);
/* The whole ActorCritic forward (model.py:12-41) + the head above in ONE launch, 16 games per workgroup on the f32 matrix cores
 * (exact f32):  hidden = relu(obs @ w1t + b1), value = hidden[:, :H] . w2c + b2c, logits = hidden[:, H:] @ w2a_t + b2a, then
 * azul_policy_head's sampling on the logits.  Weight layouts (k-major, i.e. nn.Linear.weight transposed):
 *   w1t [num_inputs][2H]: columns 0..H-1 critic_linear1, H..2H-1 actor_linear1;  b1 [2H];  w2c [H] critic_linear2.weight;  b2c [1];
 *   w2a_t [H][num_actions] actor_linear2.weight^T;  b2a [num_actions].   Only (136, H = 180, 180) is compiled in (the reference's
 * net); other shapes return AZUL_ERR_INVALID.  counter_dev: optional uint64_t[2] in device memory -- [0] is added to `counter`
 * and, when advance_counter > 0, advanced by that much by the launch itself (graph replays then draw fresh numbers without
 * a host round trip); [1] is the launch's completion ticket and must start as 0.  logits_dev [N][180] is optional. */
int azul_policy_forward(const float *obs_dev /*[N][136]*/, const uint8_t *mask_dev /*[N][180]*/, const float *w1t_dev, const float *b1_dev,
                        const float *w2c_dev, const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs,
                        int hidden_size, int num_actions, uint64_t seed, uint64_t counter, uint64_t *counter_dev, int advance_counter,
                        int n_games, uint32_t game_id_base, float *value_dev /*[N]*/, int32_t *action_dev, float *logp_dev,
                        float *entropy_dev, float *logits_dev, void *stream);
/* A whole WINDOW of policy-driven moves in one launch (the batched NNRunner.run_episode loop, nn_runner.py:17-47, without a
 * kernel boundary per move): for t = 0 .. n_steps-1 every game's observation / mask / player are written to slot t, the network
 * of azul_policy_forward is evaluated, an action is sampled (Philox counter `counter` + *counter_dev + t), and the env advances
 * -- opponent_random = 0: Azul.step for the current player with auto-reset (like azul_batch_policy_step, perspective = current
 * player); opponent_random = 1: GameRunner.step incl. the RandomAgent opponent's replies and GameRunner.reset() at episode end
 * (like azul_batch_agent_step, perspective 0) -- then slot n_steps receives the state after the last move.  Games and RNG
 * streams stay in registers / LDS for the whole launch.  Layouts are time-major: obs [T+1][N][136], mask [T+1][N][180],
 * player [T+1][N], action / reward / done / value / logp / entropy [T][N]; status [N] (optional) is the last move's status.
 * The results are bit-identical to n_steps x (azul_policy_forward + azul_batch_policy_step / _agent_step).  obs_dev must be 16-byte
 * aligned, mask_dev 4-byte aligned (the slots are written in 16-byte / 4-byte pieces). */
int azul_batch_policy_rollout(azul_batch_t *b, int n_steps, int opponent_random, const float *w1t_dev, const float *b1_dev,
                              const float *w2c_dev, const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs,
                              int hidden_size, int num_actions, uint64_t seed, uint64_t counter, uint64_t *counter_dev, float *obs_dev,
                              uint8_t *mask_dev, uint8_t *player_dev, int32_t *action_dev, int32_t *reward_dev, uint8_t *done_dev,
                              float *value_dev, float *logp_dev, float *entropy_dev, uint8_t *status_dev, void *stream);
/* The same launch also writing the window's discounted returns (nn_runner.py:70-76: q = reward + gamma * q backwards within an episode,
 * no carry across the window's end: what azul_discounted_returns(reward, done, returns, NULL, gamma, n_steps, N) computes, bit for bit) into
 * returns_dev [T][N] -- for windows of up to 32 moves from the rewards the kernel still holds in registers, without a second launch
 * (longer windows: the separate scan is launched behind the kernel).  returns_dev NULL: azul_batch_policy_rollout. */
int azul_batch_policy_rollout_returns(azul_batch_t *b, int n_steps, int opponent_random, const float *w1t_dev, const float *b1_dev,
                                      const float *w2c_dev, const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs,
                                      int hidden_size, int num_actions, uint64_t seed, uint64_t counter, uint64_t *counter_dev, float *obs_dev,
                                      uint8_t *mask_dev, uint8_t *player_dev, int32_t *action_dev, int32_t *reward_dev, uint8_t *done_dev,
                                      float *value_dev, float *logp_dev, float *entropy_dev, uint8_t *status_dev, float *returns_dev, float gamma,
                                      void *stream);
/* ---- GameRunner with a NETWORK opponent (game_runner.py:27-30: GameRunner(opponent=Agent(...)); scripts/run_batch.py:6-10,
 * tests/test_nn_runner.py:63-67, 84-90) ------------------------------------------------------------------------------------------
 * With an Agent as opponent every opponent_move() (game_runner.py:37-42) -- the opponent's replies AND player 1's forced moves (:46: the
 * loop runs while current_player != 1 or player 1 has fewer than two legal moves) and, after reset(), the opening moves (:84-85) -- asks
 * a second ActorCritic for an action on the observation from the MOVER's perspective (:38) through forward_actor alone (agent.py:73-81).
 *
 * One launch per window: azul_batch_policy_rollout with opponent_random = 1 whose RandomAgent is replaced by the `opponent` weight set.
 * Per agent step the kernel evaluates `agent` (value, action, log-prob, entropy: the C1 record, as before), plays the action, and then
 * runs matrix phases on `opponent` while any of a workgroup's 16 games owes an opponent_move(); reply j of a step samples with Philox key
 * opponent_seed + j (AZUL_POLICY_ARGMAX: action_selection "Max", agent.py:79-80) at the step's counter and the game's global id, so the
 * trajectories do not depend on sharding.  Both weight sets use azul_policy_forward's layouts (the opponent's critic arrays may be NULL).
 * One record per AGENT step, observations from perspective 0, like opponent_random = 1.  Optional trace of the opponent's play:
 * opp_action / opp_logp [T][opp_slots][N] receive its answers and their log-probabilities in the order they were played (a step's
 * replies beyond opp_slots are played but not recorded; slots beyond a step's replies are not written), opp_replies [T][N] their number
 * (opening moves of the next episode included).  Bit-identical to the per-move entries below driven with azul_policy_forward. */
typedef struct azul_net_weights {
    const float *w1t, *b1, *w2c, *b2c, *w2a_t, *b2a;     /* device pointers, layouts of azul_policy_forward */
} azul_net_weights_t;
typedef struct azul_rollout_buffers {
    float *obs;            /* [T+1][N][136] */
    uint8_t *mask;         /* [T+1][N][180] */
    uint8_t *player;       /* [T+1][N] */
    int32_t *action;       /* [T][N] */
    int32_t *reward;       /* [T][N] */
    uint8_t *done;         /* [T][N] */
    float *value, *logp, *entropy;   /* [T][N] */
    uint8_t *status;       /* [N] optional: first status of the last step that was not AZUL_OK */
    float *returns;        /* [T][N] optional (nn_runner.py:70-76) */
    int32_t *opp_action;   /* [T][opp_slots][N] optional */
    float *opp_logp;       /* [T][opp_slots][N] optional */
    uint8_t *opp_replies;  /* [T][N] optional */
    int opp_slots;
} azul_rollout_buffers_t;
int azul_batch_policy_rollout_vs(azul_batch_t *b, int n_steps, const azul_net_weights_t *agent, const azul_net_weights_t *opponent, int num_inputs,
                                 int hidden_size, int num_actions, uint64_t seed, uint64_t opponent_seed, uint64_t counter, uint64_t *counter_dev,
                                 const azul_rollout_buffers_t *out, float gamma, void *stream);
/* The same protocol one launch per cut, for opponents evaluated OUTSIDE the library (any network as PyTorch modules, or azul_policy_forward
 * on a second weight set): GameRunner.step / reset are cut at their opponent_move() calls.  pending_dev (uint8 [N], in / out) holds per
 * game 0 = nothing owed (the agent's next decision), 1 = an opponent_move() is owed inside GameRunner.step's loop (game_runner.py:46-47),
 * 2 = inside reset()'s loop (:84-85); owing_dev (uint32 [1], optional) receives the number of games that still owe one.
 *   azul_batch_net_step_begin   the agent's move (:44-45), then the loop condition; a step that needs no reply is closed at once
 *   azul_batch_net_step_reply   one opponent_move() with opp_actions_dev for every game that owes one, then the loop condition again
 *   azul_batch_net_reset_begin  GameRunner.reset(): the fresh game (:79-82), then the opening loop's condition
 * After each launch obs_opp_dev / mask_opp_dev hold, for the games that owe a move, what opponent_move hands the opponent: the
 * observation from the mover's perspective (:38) and the legal mask (:39).  reward_dev / done_dev are written by the launch that
 * CLOSES the agent step (:48-55; done = 2: nobody could move, slot restarted), which also opens the next episode (nn_runner.py:20);
 * status_dev keeps the step's first status that was not AZUL_OK; replies_dev (optional) counts the opponent moves of the step.  An
 * opponent action that is not legal leaves game and debt untouched (status AZUL_ILLEGAL_MOVE / AZUL_BAD_ACTION): answer again. */
int azul_batch_net_step_begin(azul_batch_t *b, const int32_t *actions_dev, uint8_t *pending_dev, uint8_t *replies_dev, int32_t *reward_dev,
                              uint8_t *done_dev, uint8_t *status_dev, float *obs_opp_dev /*[N][136]*/, uint8_t *mask_opp_dev /*[N][180]*/,
                              uint32_t *owing_dev, void *stream);
int azul_batch_net_step_reply(azul_batch_t *b, const int32_t *opp_actions_dev, uint8_t *pending_dev, uint8_t *replies_dev, int32_t *reward_dev,
                              uint8_t *done_dev, uint8_t *status_dev, float *obs_opp_dev, uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream);
int azul_batch_net_reset_begin(azul_batch_t *b, const uint8_t *active_dev, uint8_t *pending_dev, uint8_t *status_dev, float *obs_opp_dev,
                               uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream);
/* The A2C update's gradients (Agent.update, agent.py:39-58) for n_samples recorded (observation, mask, action, q-value) samples:
 * forward and backward of  L = mean_i( -logp_i[a_i] * adv_i + 0.5 * adv_i^2 + 0.1 * (-mean_{j legal} logp_i[j]) ),  adv = q - V
 * (advantage not detached, like the reference), on the f32 matrix cores.  `inv_n_total` = 1 / (samples of the whole batch over
 * all ranks): with data parallelism every rank calls this on its share and the flat gradients are summed.  Weight layouts as in
 * azul_policy_forward plus w2a_dev = actor_linear2.weight as PyTorch stores it ([action][hidden]).  grad_dev receives
 * AZUL_A2C_FLAT_SIZE + 4 floats: dw1t [136][360] | db1 [360] | dw2c [180] | db2c [1] | 1 pad | dw2a_t [180][180] | db2a [180] | sums
 * over the samples used of the actor / critic / entropy terms and their count.  workspace_dev: workspace_parts x that many floats (one partial per workgroup; 256 parts use
 * every CU).  Rows without a legal action carry no sample.  Optional device-side inputs: index_dev [n] (sample s is row index_dev[s]
 * of the arrays), n_samples_dev (the count; n_samples is then only an upper bound), inv_n_total_dev.  Only (136, 180, 180) is compiled in. */
int azul_a2c_gradients(const float *obs_dev, const uint8_t *mask_dev, const int32_t *action_dev, const float *qvals_dev, int n_samples,
                       float inv_n_total, const float *w1t_dev, const float *b1_dev, const float *w2c_dev, const float *b2c_dev,
                       const float *w2a_t_dev, const float *b2a_dev, const float *w2a_dev, int num_inputs, int hidden_size, int num_actions,
                       float *workspace_dev, int workspace_parts, float *grad_dev, const int32_t *index_dev, const int32_t *n_samples_dev,
                       const float *inv_n_total_dev, void *stream);
#define AZUL_A2C_FLAT_SIZE 82082      /* 82081 parameters of ActorCritic(136, 180, 180) in the k-major layout above + 1 pad float */
/* torch.optim.Adam's step (defaults: betas 0.9 / 0.999, eps 1e-8, no weight decay; same arithmetic) on the flat k-major master copy of
 * the parameters (layout of azul_a2c_gradients' gradient; w1t / b1 / w2c / b2c / w2a_t / b2a of the policy entries are views of it),
 * with the two moment vectors in the same layout; `step` counts from 1.  The updated values are also written into the eight PyTorch
 * parameter tensors (nn.Linear layouts), so module, kernels and optimiser state stay in sync without re-layout launches.
 * step_dev (optional, int32[1] in device memory): the step counter lives on the device -- it is advanced, and the step applied, only when
 * *n_total_dev > 0 (n_total_dev optional: the update's global sample count as a float in device memory); an update without samples
 * then leaves parameters, moments and step untouched.  With step_dev == NULL the host passes `step` (>= 1).
 * stats_out_dev (optional, float[5]): the update's loss terms as the reference logs them (agent.py:51-58) -- actor, critic, entropy
 * loss (the gradient buffer's loss sums / n), ac_loss = 1 a + 0.5 c + 0.1 e, and the sample count n (*n_total_dev, else n_total_host). */
int azul_a2c_apply_adam(const float *grad_dev, float *flat_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float lr, float beta1, float beta2,
                        float eps, int step, float *critic1_w, float *critic1_b, float *critic2_w, float *critic2_b, float *actor1_w,
                        float *actor1_b, float *actor2_w, float *actor2_b, int32_t *step_dev, const float *n_total_dev, float n_total_host,
                        float *stats_out_dev, void *stream);
/* Which steps of a window feed the update (NNRunner.train uses whole episodes, nn_runner.py:59-76): the steps whose episode ends
 * inside the window and that carry an action (>= 0).  done / action are time-major [n_steps][n_games]; index_dev receives the flat
 * indices t * n_games + g (game by game, steps ascending), count_dev[0] their number.  Feeds azul_a2c_gradients' index_dev /
 * n_samples_dev without a host round trip. */
int azul_select_complete_samples(const uint8_t *done_dev, const int32_t *action_dev, int n_steps, int n_games, int32_t *index_dev,
                                 int32_t *count_dev, void *stream);
/* The same selection over a RING of `ring_windows` windows of `window_steps` agent steps each, so that EVERY step of EVERY episode is
 * trained exactly once, like NNRunner.train (nn_runner.py:59-76), although episodes straddle windows: the time-major arrays hold
 * ring_windows * window_steps slots, absolute step s lives in slot s % (ring_windows * window_steps); `steps_played` = absolute steps
 * recorded so far (a multiple of window_steps: the newest window is steps_played - window_steps .. steps_played - 1); pending_dev
 * [n_games] (int32, start at 0) holds per game the first step not trained yet and is advanced by the call.  A game contributes its
 * steps from pending up to its last episode end inside the newest window (the caller chains azul_discounted_returns backwards
 * through the ring with the carry, so those steps' returns are exact).  index_dev receives flat indices slot * n_games + game (game by
 * game, steps ascending), count_dev[0] their number, count_dev[1] ACCUMULATES the steps that had left the ring before their episode
 * ended (zero it once); countf_dev (optional, float[2]) receives the count and 1 / max(count, 1) as floats (what azul_a2c_gradients'
 * inv_n_total_dev and azul_a2c_apply_adam's n_total_dev read); scratch_dev: int32 [3 n_games + ceil(n_games / 4)]. */
int azul_select_episode_samples(const uint8_t *done_ring_dev, const int32_t *action_ring_dev, int window_steps, int ring_windows, int n_games,
                                int64_t steps_played, int32_t *pending_dev, int32_t *index_dev, int32_t *count_dev, float *countf_dev,
                                int32_t *scratch_dev, void *stream);
/* discounted returns q[t] = r[t] + gamma * q[t+1] within episodes over a time-major window [n_steps][n_games]
 * (nn_runner.py:70-76); done[t][g] != 0 closes an episode at move t; carry_dev[n_games] (optional) chains windows. */
int azul_discounted_returns(const int32_t *reward_dev, const uint8_t *done_dev, float *returns_dev, float *carry_dev,
                            float gamma, int n_steps, int n_games, void *stream);
/* the same over a ring of `ring_steps` time slots (absolute step s in slot s % ring_steps) in ONE launch: from the newest recorded
 * step (steps_played - 1) back over span_steps (<= ring_steps) steps, e.g. all the windows a trajectory ring holds; equals the
 * window-by-window calls chained through the carry. */
int azul_discounted_returns_ring(const int32_t *reward_ring_dev, const uint8_t *done_ring_dev, float *returns_ring_dev, float gamma,
                                 int ring_steps, int64_t steps_played, int span_steps, int n_games, void *stream);

/* The C1 WIRE RECORD of the trajectory all-gather BASELINE configs[4] names: one window of a policy rollout (the time-major arrays of
 * azul_batch_policy_rollout*, [n_steps][n_games]...) packed to 184 bytes per agent step -- what NNRunner.run_episode keeps per step
 * (nn_runner.py:17-47) and NNRunner.train concatenates over episodes before one update (nn_runner.py:59-78):
 *     0 obs[136] u8 (the observation's integers, game_runner.py:65-72: 0..255 for every state the rules reach) | 136 maskbits[24] (the 180
 *     legal-move bits, bit a & 7 of byte a >> 3) | 160 action u8 (0xff = none) | 161 done | 162 player to move | 163 zero |
 *     164 reward i32 | 168 value f32 | 172 log-prob f32 | 176 entropy f32 | 180 discounted return f32
 * records_dev: uint8 [n_steps][n_games][184].  One HBM-bound launch (750 bytes in, 184 out per step).  obs_dev 16-byte aligned,
 * mask_dev / records_dev 4-byte aligned. */
#define AZUL_C1_BYTES 184
int azul_pack_c1(const float *obs_dev, const uint8_t *mask_dev, const uint8_t *player_dev, const int32_t *action_dev, const int32_t *reward_dev,
                 const uint8_t *done_dev, const float *value_dev, const float *logp_dev, const float *entropy_dev, const float *returns_dev,
                 int n_steps, int n_games, uint8_t *records_dev, void *stream);

/* ---- flat random-agent self-play (the benchmarked hot path) ----------------------------------- */
/*
 * Advance every game by `n_steps` env moves in ONE launch: per move  mask -> RandomAgent -> Azul.step ->
 * reward (delta of the what-if potential) -> done, with auto-reset (GameRunner.reset semantics, same
 * stream) when a game ends.  Trajectory outputs are [n_steps][N]...; any may be NULL.
 *   mask_dev     uint8  [n_steps][N][180] legal moves before the move
 *   maskbits_dev uint64 [n_steps][N][3]   the same mask bit-packed (bit a&63 of word a>>6)
 *   action_dev int32 [n_steps][N]        chosen action (-1 when stuck)
 *   reward_dev int32 [n_steps][N]
 *   done_dev   uint8 [n_steps][N]        1 = game ended with this move, 2 = stuck (no move, reset), 3 = cut by the move limit (azul_batch_set_move_limit)
 *   packed_dev uint32 [n_steps][N]       compact record: action (0xff = none) | done << 8 | (reward & 0xffff) << 16
 *   rec_dev    uint8 [n_steps][N][128]   record after the move, before the auto-reset (tests only)
 * Batches of THREE or FOUR players (azul_batch_create_players; row N4) play the same flat loop -- mask -> RandomAgent
 * (game_runner.py:87-97, any mask) -> Azul.step (azul.py:296-313, P-generic), a fresh Azul(players = P, rules) + new_round() when
 * a game ends or nobody can move; start them with azul_batch_init + azul_batch_new_round -- in a persistent kernel with two games
 * per wavefront.  Their reward stream is all zero (the shaped reward is GameRunner's: two players, game_runner.py:50), rec_dev rows
 * are 256-byte wide records (bytes of absent players and the reserved tail are not written), mask rows are dense.  A game of such a
 * batch that a rule error stops (bag and lid empty without AZUL_RULE_SHORT_DEAL, where the reference raises: azul.py:86-87) plays no
 * further move in this launch: its remaining slots carry action -1 / done 2 and are counted in `stuck` like the slots of a stuck game.
 * The same holds for a TWO-player batch ("Lid" pool, box and lid both empty when a round has to be dealt) once the host has written
 * records into it (azul_batch_set_state, azul_game_call's record_in) or it has a move limit: play from azul_batch_init / _reset cannot
 * reach such a state, so a batch that was never handed a record runs the instantiation without that bookkeeping (the benchmarked one).
 */
int azul_batch_selfplay(azul_batch_t *b, int n_steps, uint8_t *mask_dev, uint64_t *maskbits_dev, int32_t *action_dev,
                        int32_t *reward_dev, uint8_t *done_dev, uint32_t *packed_dev, uint8_t *rec_dev, void *stream);
/* the same with a row pitch for the byte mask: mask_dev is [n_steps][N][mask_row_bytes] (>= 180; bytes 180.. of a row are not
 * written).  192 keeps every game's row 64-byte aligned, so a wave's mask stores cover whole 32-byte sectors (no partial
 * writes: DESIGN.md 3, write amplification). */
int azul_batch_selfplay_strided(azul_batch_t *b, int n_steps, uint8_t *mask_dev, int mask_row_bytes, uint64_t *maskbits_dev,
                                int32_t *action_dev, int32_t *reward_dev, uint8_t *done_dev, uint32_t *packed_dev, uint8_t *rec_dev,
                                void *stream);
/* per-game counters accumulated by selfplay / runner_step / the policy entries: episodes[N] u64, stuck[N] u32, sums of
 * get_statistics() over finished games [N][10] f64 (azul.py:314-315, the data behind GameStatistics, game_runner.py:10-22).
 * azul_batch_counters_dev hands out the DEVICE arrays themselves (zero-copy, no synchronisation: read them on a stream
 * ordered after the launches, like every other *_dev pointer of this ABI); azul_batch_counters is the convenience form
 * for hosts -- it copies into HOST buffers (NULL to skip) and synchronises `stream`. */
int azul_batch_counters_dev(azul_batch_t *b, uint64_t **episodes_dev, uint32_t **stuck_dev, double **stat_sums_dev);
int azul_batch_counters(azul_batch_t *b, uint64_t *episodes_host, uint32_t *stuck_host, double *stat_sums_host, void *stream);
int azul_batch_reset_counters(azul_batch_t *b, void *stream);

/* global id of the batch's game 0 (default 0).  Multi-GPU runs shard games by global id (rank r owns [G r, G (r+1)), seeds
 * seed_base + global id); the policy sampler of azul_batch_policy_rollout keys its Philox stream with id_base + local index, so
 * a game's sampled actions -- like its CPython stream -- do not depend on the GPU count. */
int azul_batch_set_id_base(azul_batch_t *b, uint32_t first_global_id);

/* MOVE LIMIT -- beyond the reference, OFF by default (0), "parity unpinned": under the reference's rules a game can reach a state from which
 * it NEVER ends -- e.g. all 20 tiles of one colour locked in pattern lines that can no longer be completed, so that no wall row can ever be
 * filled and is_end_of_game (azul.py:184-191) stays false for ever; GameRunner's callers loop `while not done` (nn_runner.py:24,
 * tests/test_game_runner.py:48) and would never return.  With random play one game in ~10^9 game-moves gets there (bench.py's `sustained`
 * names the game: seed 801 after ~310 k moves); it then keeps its slot for ever and -- short rounds -- slows its wavefront by ~10 %.
 * With max_moves > 0 (<= 65535) an episode whose move_counter (game_runner.py:36) has reached max_moves when a round ends WITHOUT ending the
 * game is cut there: the round is scored, no new round is dealt, the slot restarts like at a game end (GameRunner.reset semantics, same
 * RNG stream), the move reports done = 3 (policy entries: status AZUL_TRUNCATED, reward as for a stuck slot) and the slot is counted in
 * `stuck`, not in `episodes` / the statistics sums.  Applies to azul_batch_selfplay*, azul_batch_step, the GameRunner / policy entries and
 * the rollout kernels of two-player batches; trajectories do not depend on how moves are split over launches. */
int azul_batch_set_move_limit(azul_batch_t *b, uint32_t max_moves);

/* TEST KNOB: the factory draw decides a colour in integer arithmetic unless K*T lies within `margin` of a multiple of
 * 2^32 (then by the literal fp64 computation).  Default 8192 (proved sufficient); a wider margin only sends
 * more draws through the fp64 path -- results are identical, tests use it to exercise that path.  Range [8192, 2^31). */
int azul_batch_set_draw_margin(azul_batch_t *b, uint64_t margin);

/* DIAGNOSTIC: per-segment cycle sums of the self-play kernel (mask, sample, move, after-move, tail, new round, scoring,
 * reset, loop).  Only a library built with -DAZ_PROFILE_SEGMENTS stamps them (tools/segment_profile.sh); the shipped
 * build returns zeros.  Synchronises the device. */
int azul_batch_segment_profile(azul_batch_t *b, uint64_t *cycles_host, int n, int reset);

/* DIAGNOSTIC: what the flat self-play kernel that azul_batch_selfplay_strided launches for this batch (tile pool) and output shape
 * (padded_rows: mask rows of >= 192 bytes; mask_bits: bit-packed masks as well) occupies, read from the loaded code object by the HIP
 * runtime: allocated vector registers per lane, static LDS and scratch bytes per workgroup (= per wave: one-wave workgroups), and how many
 * of its waves a CU can hold at once (hipOccupancyMaxActiveBlocksPerMultiprocessor) -- the RESIDENT occupancy, whatever the grid size.
 * The reference has no counterpart. */
int azul_selfplay_kernel_resources(azul_batch_t *b, int padded_rows, int mask_bits, int *vgprs, int *lds_bytes, int *scratch_bytes,
                                   int *resident_waves_per_cu);

/* DIAGNOSTIC: the shader clock as a wave sees it.  One wavefront runs `spin_iterations` dependent vector operations between two readings of
 * s_memtime (shader-clock cycles) and s_memrealtime (constant 100 MHz); out_dev (uint64[3], device memory) receives {shader cycles,
 * 100 MHz ticks, the chain's result}: out[0] / out[1] x 100 = MHz.  Asynchronous on `stream`.  bench.py's sustained phase launches it
 * between blocks of the headline kernel, beside the driver-reported clock (amdsmi).  The reference has no counterpart. */
int azul_device_clock_probe(uint64_t *out_dev, int spin_iterations, void *stream);

/* device time of azul_batch_selfplay launches, measured with hipEvents on the launch stream: call azul_timing_begin, launch
 * any number of selfplay calls, then azul_timing_end (synchronises the stream).  total_ms / launches: the event bracket from
 * begin to end and the launches inside it; kernel_ms / kernel_launches: the sum of the event pairs recorded immediately
 * around each of the first 1024 launches (the kernel's own duration, what `rocprofv3 --kernel-trace` reports) and their number.
 * azul_timing_launch_ms: after azul_timing_end, the individual durations of those launches (at most `cap` of them are written;
 * *n = how many the last timed region bracketed).  The reference has no counterpart (it is timed from outside: scripts/ run
 * game_runner.py under a wall clock). */
int azul_timing_begin(azul_batch_t *b, void *stream);
int azul_timing_end(azul_batch_t *b, void *stream, float *total_ms, int *launches, float *kernel_ms, int *kernel_launches);
int azul_timing_launch_ms(azul_batch_t *b, float *launch_ms, int cap, int *n);

#ifdef __cplusplus
}
#endif
#endif
