// azul_kernels.hip -- gfx950 kernels and the C ABI of libazulhip.so (declared in include/azul_hip.h).
//
// Mapping: TWO GAMES PER 64-LANE WAVEFRONT (lanes 0..31 / 32..63), one wavefront per workgroup for the rule kernels and the self-play
// loops (grid = ceil(N / 2) workgroups: 2048 waves = two per SIMD on the 256 CUs of an MI355X at the benchmarked N = 4096), eight waves
// per workgroup for the policy rollout.  A game's board cells sit in the lanes of its half, everything else is replicated across the
// half ("half-uniform") and the rules run on the vector pipe; the rare paths (factory draw, scoring, episode reset) are ordinary
// divergent branches between the halves.  The rules exist ONCE per game shape (azul_common.hpp): azul_selfplay2.hpp + azul_env2.hpp
// for two players, azul_rules_x.hpp for 3 / 4 players and the extended rules.
//
// Kernels
//   azul_seed_kernel            one THREAD per game: CPython init_by_array is a strictly sequential 1247-step recurrence
//   azul_op_kernel              every single-call rule / runner entry point of the ABI for two-player batches (azul_ops2.hpp)
//   azul_x_op_kernel            the rule entries for batches of 3- / 4-player games and for extended-rule batches (row N4;
//                               azul_rules_x.hpp: 256-byte wide records)
//   azul_selfplay2_kernel       state register-resident across n_steps env moves (the hot path); azul_x_selfplay_kernel: row N4's
//   azul_returns_kernel         discounted returns over a trajectory window
//   azul_pack_c1_kernel         the C1 wire record of the opt-in trajectory all-gather (184 bytes per agent step); azul_clock_probe_kernel (diagnostic)
//   azul_policy.hpp             policy head, fused ActorCritic forward; azul_rollout2.hpp: persistent policy rollout (rows N1 / N2)
//   azul_learner.hpp            A2C gradients (forward + backward on the matrix cores), partial reduction, sample selection (row N2)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"

using namespace az;

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------
#include "azul_selfplay_kernels.hpp"

#include "azul_rules_x.hpp"

// Rule entries for batches of three / four players and for extended-rule batches (row N4): one launch = one rule call per game, two games
// per wavefront (azx::op_body_x).  grid = ceil(count / 2) one-wave workgroups.
template <u32 P, u32 D>
__global__ void __launch_bounds__(64) azul_x_op_kernel(azx::XBatchDev b, azx::XOp a)
{
    __shared__ u32 mt_lds[2][624];
    __shared__ double2 tab_lds[azx::Dim<D>::TROWS * T_STRIDE];
    azx::op_body_x<P, D>(b, a, blockIdx.x, mt_lds, tab_lds);
}

// Their flat random-agent self-play, persistent like azul_selfplay2_kernel (two games per wavefront, state in VGPRs, MT19937 streams and
// their tempered copies in LDS, XCD-aware game placement).
template <u32 P, u32 D, int OUT, bool PAD, bool BITS>
__global__ void __launch_bounds__(64) azul_x_selfplay_kernel(azx::XBatchDev b, azx::XTraj t)
{
    __shared__ u32 mt_lds[2][624];
    __shared__ u32 mtt_lds[2][624];
    __shared__ double2 tab_lds[azx::Dim<D>::TROWS * T_STRIDE];
    const u32 nb = gridDim.x, xcd = blockIdx.x & 7u, q8 = nb >> 3, rem = nb & 7u;
    const u32 wave_id = xcd * q8 + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);      // every XCD plays a contiguous range of games
    azx::selfplay_body_x<P, D, OUT, PAD, BITS>(b, t, wave_id, mt_lds, mtt_lds, tab_lds);
}

#include "azul_policy.hpp"
#include "azul_rollout2.hpp"
#include "azul_learner.hpp"

// ------------------------------------------------------------------------------------------------
// host side: C ABI
// ------------------------------------------------------------------------------------------------
struct azul_batch {
    BatchDev d;          // `tab` is written once by azul_batch_create
    int device;          // the device the batch's arrays live on: every entry runs there (DeviceGuard)
    int players;         // 2, 3 or 4
    int rec_bytes;       // 128 (two players under the reference's rules: every entry) or 256 (the wide record: the x path below)
    bool x;              // three / four players, or any extended rule: the azul_rules_x.hpp kernels (rule entries + flat self-play)
    unsigned ext;        // AZUL_RULE_* flags (0: the reference's rules)
    int displays;        // 5, or 2 * players + 1 with AZUL_RULE_DISPLAYS_2P1
    double *Tx;          // the sampling table, T_STRIDE pairs per row, for 5 (displays + 1) + 1 rows (31 for the reference's 180 actions)
    hipEvent_t ev0, ev1; // bracket of a timed region (azul_timing_begin / _end)
    std::vector<hipEvent_t> lev;   // event pairs around the individual self-play launches of a timed region
    int timed_launches;  // launches since azul_timing_begin
    int timed_pairs;     // of which bracketed by their own event pair (the first AZ_TIMED_PAIRS)
    bool timing;
    uint8_t *call_pin;   // azul_game_call: one call's arguments / results (CallScratch) in pinned, device-visible host memory
    bool handed_in;      // the host has written records into the batch (azul_batch_set_state, azul_game_call's record_in): flat self-play then
                         // runs the instantiation that marks and counts the slots of a rule-error-stopped game (a state play cannot reach)
};
enum { AZ_TIMED_PAIRS = 1024 };

static thread_local std::string g_err;

static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    g_err = what;
    if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
    return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(AZUL_ERR_HIP, #expr, e_); } while (0)

// Every entry runs on ONE device -- the batch's (recorded by azul_batch_create), or for the batch-less entries the
// device `stream` belongs to (NULL stream: the caller's current device) -- whatever device is current in the calling
// thread: the guard makes that device current for the duration of the call and restores the caller's on return.  A
// stream that belongs to another device than the batch is a caller bug and is refused (AZUL_ERR_INVALID).
struct DeviceGuard {
    int prev, rc;
    bool switched;
    DeviceGuard() : prev(-1), rc(AZUL_SUCCESS), switched(false) {}
    int enter(int want, void *stream, const char *who)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return rc = fail(AZUL_ERR_HIP, "hipGetDevice", e);
        int sdev = -1;
        if (stream) {
            hipDevice_t d;
            e = hipStreamGetDevice((hipStream_t)stream, &d);
            if (e != hipSuccess) return rc = fail(AZUL_ERR_HIP, "hipStreamGetDevice", e);
            sdev = (int)d;
        }
        if (want < 0) want = sdev >= 0 ? sdev : prev;
        else if (sdev >= 0 && sdev != want) {
            g_err = std::string(who) + ": the stream belongs to device " + std::to_string(sdev) + ", the batch lives on device " + std::to_string(want);
            return rc = AZUL_ERR_INVALID;
        }
        if (want != prev) {
            e = hipSetDevice(want);
            if (e != hipSuccess) return rc = fail(AZUL_ERR_HIP, "hipSetDevice", e);
            switched = true;
        }
        return AZUL_SUCCESS;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
// first statement of an entry that takes a batch (`b` may still be NULL: the entry's own checks report that)
#define BATCH_GUARD(b, stream) DeviceGuard guard_; if ((b) && guard_.enter((b)->device, (void *)(stream), __func__)) return guard_.rc
// first statement of an entry without a batch
#define STREAM_GUARD(stream) DeviceGuard guard_; if (guard_.enter(-1, (void *)(stream), __func__)) return guard_.rc

extern "C" {

const char *azul_last_error_string(void) { return g_err.c_str(); }
const char *azul_version(void)
{
    return "azul-mi355x 0.6 (gfx950; two games per wavefront: two-player rule entries, self-play, policy rollout with random / network opponent, 3 / 4 players and extended rules)";
}

static void batch_free(azul_batch *b)
{
    void *bufs[] = {b->d.state, b->d.mt, b->d.mtpos, b->d.episodes, b->d.stuck, b->d.stat_sum, b->d.prof, b->Tx};
    for (void *p : bufs) if (p) (void)hipFree(p);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    for (hipEvent_t e : b->lev) (void)hipEventDestroy(e);
    if (b->call_pin) (void)hipHostFree(b->call_pin);
    delete b;
}

static int batch_alloc(azul_batch *b, int n_games, int first_player, int tile_pool)
{
    HIP_TRY(hipGetDevice(&b->device));
    const size_t N = (size_t)n_games;
    const size_t RB = (size_t)b->rec_bytes;
    b->d.n = (u32)n_games;
    b->d.rules.first_player = (u32)first_player;
    b->d.rules.tile_pool = (u32)tile_pool;
    b->d.draw_margin = AZ_DRAW_MARGIN;
    HIP_TRY(hipMalloc((void **)&b->d.state, N * RB));
    HIP_TRY(hipMalloc((void **)&b->d.mt, N * 624 * sizeof(u32)));
    HIP_TRY(hipMalloc((void **)&b->d.mtpos, N * sizeof(u32)));
    HIP_TRY(hipMalloc((void **)&b->d.episodes, N * sizeof(u64)));
    HIP_TRY(hipMalloc((void **)&b->d.stuck, N * sizeof(u32)));
    HIP_TRY(hipMalloc((void **)&b->d.stat_sum, N * 10 * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&b->d.prof, AZ_PROF_SLOTS * sizeof(u64)));
    HIP_TRY(hipMemset(b->d.prof, 0, AZ_PROF_SLOTS * sizeof(u64)));
    {   // the sampler's table: built with CPython's very additions and checked entry by entry on this host (azul_tables.hpp)
        const int rows = 5 * (b->displays + 1) + 1;
        std::vector<double> hX((size_t)rows * T_STRIDE * 2);
        if (!build_sample_pairs(rows, hX.data())) return fail(AZUL_ERR_INVALID, "weight-table decomposition check failed on this host");
        HIP_TRY(hipMalloc((void **)&b->Tx, hX.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(b->Tx, hX.data(), hX.size() * sizeof(double), hipMemcpyHostToDevice));
        b->d.tab = (const double2 *)b->Tx;
    }
    HIP_TRY(hipMemset(b->d.state, 0, N * RB));
    HIP_TRY(hipMemset(b->d.mt, 0, N * 624 * sizeof(u32)));
    {   // a defined stream even before azul_batch_seed: index 624 over an all-zero state is never used un-seeded
        std::vector<u32> pos(N, 624u);
        HIP_TRY(hipMemcpy(b->d.mtpos, pos.data(), N * sizeof(u32), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(b->d.episodes, 0, N * sizeof(u64)));
    HIP_TRY(hipMemset(b->d.stuck, 0, N * sizeof(u32)));
    HIP_TRY(hipMemset(b->d.stat_sum, 0, N * 10 * sizeof(double)));
    HIP_TRY(hipEventCreate(&b->ev0));
    HIP_TRY(hipEventCreate(&b->ev1));
    return AZUL_SUCCESS;
}

int azul_batch_create(azul_batch_t **out, int n_games, int first_player, int tile_pool)
{
    return azul_batch_create_players(out, n_games, 2, first_player, tile_pool);
}

int azul_batch_create_players(azul_batch_t **out, int n_games, int players, int first_player, int tile_pool)
{
    return azul_batch_create_rules(out, n_games, players, first_player, tile_pool, 0u);
}

int azul_batch_create_rules(azul_batch_t **out, int n_games, int players, int first_player, int tile_pool, unsigned rule_flags)
{
    if (!out || n_games <= 0) return fail(AZUL_ERR_INVALID, "azul_batch_create: bad arguments");
    if (players < 2 || players > 4) return fail(AZUL_ERR_INVALID, "players must be 2, 3 or 4");
    if (first_player < 0 || first_player > players) return fail(AZUL_ERR_RULE, "first_player must be 0 (Random) or 1 .. players");
    if (tile_pool != AZUL_POOL_RANDOM && tile_pool != AZUL_POOL_LID) return fail(AZUL_ERR_RULE, "tile_pool must be AZUL_POOL_RANDOM or AZUL_POOL_LID");
    if (rule_flags & ~(AZUL_RULE_DISPLAYS_2P1 | AZUL_RULE_END_BONUS | AZUL_RULE_SHORT_DEAL | AZUL_RULE_FINITE_BAG))
        return fail(AZUL_ERR_RULE, "unknown AZUL_RULE_* flag");
    if ((rule_flags & AZUL_RULE_FINITE_BAG) && tile_pool != AZUL_POOL_RANDOM)
        return fail(AZUL_ERR_RULE, "AZUL_RULE_FINITE_BAG applies to tile_pool Random (the Lid pool already is a finite bag)");
    *out = nullptr;
    azul_batch *b = new azul_batch();
    memset(&b->d, 0, sizeof(b->d));
    b->device = -1;
    b->players = players;
    b->ext = rule_flags;
    b->displays = (rule_flags & AZUL_RULE_DISPLAYS_2P1) ? 2 * players + 1 : 5;
    b->x = players != 2 || rule_flags != 0u;
    b->rec_bytes = b->x ? AZUL_RECORD_BYTES_WIDE : AZUL_RECORD_BYTES;
    b->Tx = nullptr;
    b->ev0 = b->ev1 = nullptr;
    b->timing = false;
    b->timed_launches = 0;
    b->timed_pairs = 0;
    b->call_pin = nullptr;
    int rc = batch_alloc(b, n_games, first_player, tile_pool);
    if (rc != AZUL_SUCCESS) { batch_free(b); return rc; }      // nothing leaks when an allocation fails half way
    *out = b;
    return AZUL_SUCCESS;
}

int azul_batch_destroy(azul_batch_t *b)
{
    if (!b) return AZUL_SUCCESS;
    BATCH_GUARD(b, nullptr);
    batch_free(b);
    return AZUL_SUCCESS;
}

int azul_batch_size(const azul_batch_t *b) { return b ? (int)b->d.n : 0; }
int azul_batch_players(const azul_batch_t *b) { return b ? b->players : 0; }
int azul_batch_displays(const azul_batch_t *b) { return b ? b->displays : 0; }
unsigned azul_batch_rule_flags(const azul_batch_t *b) { return b ? b->ext : 0u; }
int azul_batch_num_actions(const azul_batch_t *b) { return b ? (b->displays + 1) * 30 : 0; }                              /* game_runner.py:115 */
int azul_batch_obs_size(const azul_batch_t *b) { return b ? 5 * b->displays + 6 + 52 * b->players + 1 : 0; }             /* game_runner.py:65-72 */
int azul_batch_record_bytes(const azul_batch_t *b) { return b ? b->rec_bytes : 0; }
void *azul_batch_state_dev(azul_batch_t *b) { return b ? b->d.state : nullptr; }
void *azul_batch_mt_dev(azul_batch_t *b) { return b ? b->d.mt : nullptr; }
void *azul_batch_mtpos_dev(azul_batch_t *b) { return b ? b->d.mtpos : nullptr; }

static int check_range(const azul_batch_t *b, int first, int count)
{
    if (!b || first < 0 || count < 0 || (u64)first + (u64)count > b->d.n) return fail(AZUL_ERR_INVALID, "game range outside the batch");
    return AZUL_SUCCESS;
}

// domain the kernels are exact on (documented in DESIGN.md)
static int record_in_domain(const azul_batch_t *b, const uint8_t *p)
{
    const u32 P = (u32)b->players;
    const bool wide = b->rec_bytes == AZUL_RECORD_BYTES_WIDE;
    u32 flags = p[31];
    if ((flags & 7u) > P || ((flags >> 3) & 7u) > P || (flags & 0x80u)) return fail(AZUL_ERR_RANGE, "flags: players are 0..P");
    const uint8_t *floors = p + (wide ? 132 : 82), *walls = p + (wide ? 136 : 84), *box = p + (wide ? 160 : 96);
    for (u32 q = 0; q < P; q++) {
        if (floors[q] > 7) return fail(AZUL_ERR_RANGE, "floors are 0..7 (azul.py:120-123)");
        u32 w;
        memcpy(&w, walls + 4 * q, 4);
        if (w >> 25) return fail(AZUL_ERR_RANGE, "walls are 25-bit boards");
    }
    u32 sb = 0, sl = 0;
    for (int c = 0; c < 5; c++) { sb += box[c]; sl += box[5 + c]; }
    if (sb > 255 || sl > 255) return fail(AZUL_ERR_RANGE, "box / lid hold at most 255 tiles in total");
    if (wide && p[204] != P) return fail(AZUL_ERR_RANGE, "wide record: byte 204 must hold the batch's number of players");
    if (wide && p[205] != (b->displays == 5 ? 0 : b->displays))
        return fail(AZUL_ERR_RANGE, "wide record: byte 205 must hold the batch's number of displays (0 for the reference's five)");
    return AZUL_SUCCESS;
}

int azul_batch_get_state(azul_batch_t *b, int first, int count, void *records_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, first, count)) return rc;
    if (!records_host) return fail(AZUL_ERR_INVALID, "records_host is NULL");
    const size_t RB = (size_t)b->rec_bytes;
    HIP_TRY(hipMemcpyAsync(records_host, b->d.state + (size_t)first * RB, (size_t)count * RB, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_set_state(azul_batch_t *b, int first, int count, const void *records_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, first, count)) return rc;
    if (!records_host) return fail(AZUL_ERR_INVALID, "records_host is NULL");
    const uint8_t *p = (const uint8_t *)records_host;
    const size_t RB = (size_t)b->rec_bytes;
    for (int i = 0; i < count; i++, p += RB)
        if (int rc = record_in_domain(b, p)) return rc;
    HIP_TRY(hipMemcpyAsync(b->d.state + (size_t)first * RB, records_host, (size_t)count * RB, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    b->handed_in = true;
    return AZUL_SUCCESS;
}

int azul_batch_get_rng(azul_batch_t *b, int game, uint32_t *mt_host, uint32_t *pos_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, game, 1)) return rc;
    if (mt_host) HIP_TRY(hipMemcpyAsync(mt_host, b->d.mt + (size_t)game * 624, 624 * sizeof(u32), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (pos_host) HIP_TRY(hipMemcpyAsync(pos_host, b->d.mtpos + game, sizeof(u32), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_set_rng(azul_batch_t *b, int game, const uint32_t *mt_host, uint32_t pos, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, game, 1)) return rc;
    if (!mt_host || pos > 624u) return fail(AZUL_ERR_INVALID, "azul_batch_set_rng: need 624 words and an index in 0..624");
    HIP_TRY(hipMemcpyAsync(b->d.mt + (size_t)game * 624, mt_host, 624 * sizeof(u32), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(b->d.mtpos + game, &pos, sizeof(u32), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_get_rng_range(azul_batch_t *b, int first, int count, uint32_t *mt_host, uint32_t *pos_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return AZUL_SUCCESS;
    if (mt_host) HIP_TRY(hipMemcpyAsync(mt_host, b->d.mt + (size_t)first * 624, (size_t)count * 624 * sizeof(u32), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (pos_host) HIP_TRY(hipMemcpyAsync(pos_host, b->d.mtpos + first, (size_t)count * sizeof(u32), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_set_rng_range(azul_batch_t *b, int first, int count, const uint32_t *mt_host, const uint32_t *pos_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (int rc = check_range(b, first, count)) return rc;
    if (count == 0) return AZUL_SUCCESS;
    if (!mt_host || !pos_host) return fail(AZUL_ERR_INVALID, "azul_batch_set_rng_range: need 624 words and an index per game");
    for (int g = 0; g < count; g++)
        if (pos_host[g] > 624u) return fail(AZUL_ERR_INVALID, "azul_batch_set_rng_range: index outside 0..624");
    HIP_TRY(hipMemcpyAsync(b->d.mt + (size_t)first * 624, mt_host, (size_t)count * 624 * sizeof(u32), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(b->d.mtpos + first, pos_host, (size_t)count * sizeof(u32), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_seed(azul_batch_t *b, uint64_t seed_base, const uint64_t *seeds_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    u64 *dseeds = nullptr;
    if (seeds_host) {
        HIP_TRY(hipMalloc((void **)&dseeds, (size_t)b->d.n * sizeof(u64)));
        hipError_t e = hipMemcpyAsync(dseeds, seeds_host, (size_t)b->d.n * sizeof(u64), hipMemcpyHostToDevice, (hipStream_t)stream);
        if (e != hipSuccess) { (void)hipFree(dseeds); return fail(AZUL_ERR_HIP, "hipMemcpyAsync(seeds)", e); }
    }
    hipLaunchKernelGGL(azul_seed_kernel, dim3((b->d.n + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, b->d, (u64)seed_base, (const u64 *)dseeds);
    hipError_t e = hipGetLastError();
    if (dseeds) {                                        // the staging copy is released on every path (hipFree waits for the device)
        const hipError_t e2 = e == hipSuccess ? hipStreamSynchronize((hipStream_t)stream) : hipSuccess;
        (void)hipFree(dseeds);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return fail(AZUL_ERR_HIP, "azul_batch_seed", e);
    return AZUL_SUCCESS;
}

static azx::XBatchDev xdev(const azul_batch_t *b)
{
    azx::XBatchDev x;
    x.state = b->d.state; x.mt = b->d.mt; x.mtpos = b->d.mtpos; x.episodes = b->d.episodes; x.stuck = b->d.stuck; x.stat_sum = b->d.stat_sum;
    x.n = b->d.n; x.draw_margin = b->d.draw_margin;
    x.rules.first_player = b->d.rules.first_player;
    x.rules.pool = (b->ext & AZUL_RULE_FINITE_BAG) ? (u32)azx::XPOOL_BAG : (b->d.rules.tile_pool == POOL_LID ? (u32)azx::XPOOL_LID : (u32)azx::XPOOL_RANDOM);
    x.rules.end_bonus = (b->ext & AZUL_RULE_END_BONUS) ? 1u : 0u;
    x.rules.short_deal = (b->ext & AZUL_RULE_SHORT_DEAL) ? 1u : 0u;
    x.tab = (const double2 *)b->Tx;
    x.prof = b->d.prof;
    return x;
}

// (players, displays) -> the instantiation; `call` is a macro body that uses PP / DD
#define AZ_X_DISPATCH(b, call) do { \
        const int pd_ = (b)->players * 16 + (b)->displays; \
        if (pd_ == 2 * 16 + 5) { constexpr u32 PP = 2, DD = 5; call; } \
        else if (pd_ == 3 * 16 + 5) { constexpr u32 PP = 3, DD = 5; call; } \
        else if (pd_ == 3 * 16 + 7) { constexpr u32 PP = 3, DD = 7; call; } \
        else if (pd_ == 4 * 16 + 5) { constexpr u32 PP = 4, DD = 5; call; } \
        else { constexpr u32 PP = 4, DD = 9; call; } } while (0)

static int launch_op_x(azul_batch_t *b, const OpArgs &a, void *stream, int count)
{
    // the entries that mirror Azul's own methods (and the RandomAgent sampler, check_all_valid, get_state): P-generic in the reference;
    // GameRunner.step / reset / the what-if potential are two-player there (game_runner.py:50) and stay with two-player reference batches
    azx::XOp x;
    memset(&x, 0, sizeof(x));
    switch (a.op) {
    case OP_QUERY: x.op = azx::XOP_QUERY; break;
    case OP_INIT: x.op = azx::XOP_INIT; break;
    case OP_NEW_ROUND: x.op = azx::XOP_NEW_ROUND; break;
    case OP_MOVE: x.op = azx::XOP_MOVE; break;
    case OP_NEXT_PLAYER: x.op = azx::XOP_NEXT_PLAYER; break;
    case OP_COUNT_SCORE: x.op = azx::XOP_COUNT_SCORE; break;
    case OP_STEP: x.op = azx::XOP_STEP; break;
    case OP_RANDOM_ACTION: x.op = azx::XOP_RANDOM_ACTION; break;
    case OP_SAMPLE_MASK: x.op = azx::XOP_SAMPLE_MASK; break;
    default: x.op = -1; break;
    }
    if (x.op < 0 || a.potential || a.reward || a.done)
        return fail(AZUL_ERR_INVALID, "this entry mirrors GameRunner's two-player step / reset / shaped reward (game_runner.py:43-55, 76-85): batches of "
                                      "three / four players and extended-rule batches support the Azul rule entries, the sampler, the mask and the observation");
    x.actions = a.actions; x.active = a.active; x.mask_in = a.mask_in; x.actions_out = a.actions_out; x.status = a.status; x.mask = a.mask;
    x.obs = a.obs; x.persp = a.persp; x.flags = a.flags; x.stats = a.stats; x.player = a.player; x.rng_dirty = a.rng_dirty;
    x.rec_out = a.rec_out; x.pos_out = a.pos_out; x.next_action = a.next_action; x.pos_set = a.pos_set;
    x.first = a.first; x.count = count < 0 ? b->d.n : (u32)count;
    const dim3 grid((x.count + 1u) / 2u), block(64);
    const azx::XBatchDev xb = xdev(b);
    AZ_X_DISPATCH(b, hipLaunchKernelGGL((azul_x_op_kernel<PP, DD>), grid, block, 0, (hipStream_t)stream, xb, x));
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

static int launch_op(azul_batch_t *b, const OpArgs &a, void *stream, int count = -1)
{
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    if (b->x) return launch_op_x(b, a, stream, count);
    OpArgs a2 = a;
    a2.count = count < 0 ? b->d.n : (u32)count;                       // games a.first .. a.first + count - 1, two per wavefront
    const dim3 grid((a2.count + 1u) / 2u), block(64);
    const hipStream_t st = (hipStream_t)stream;
    if (b->d.rules.tile_pool == POOL_LID) hipLaunchKernelGGL(azul_op_kernel<true>, grid, block, 0, st, b->d, a2);
    else hipLaunchKernelGGL(azul_op_kernel<false>, grid, block, 0, st, b->d, a2);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

static OpArgs op_args(int op)
{
    OpArgs a;
    memset(&a, 0, sizeof(a));
    a.op = op;
    return a;
}

// perspective of an observation: a player 0 .. P-1, AZUL_PERSP_MOVER (any batch), or -- two-player reference batches -- AZUL_PERSP_CURRENT
static bool persp_ok(const azul_batch_t *b, int p) { return p == AZUL_PERSP_MOVER || (p >= 0 && (b->x ? p < b->players : p <= 2)); }
static int persp_of(const azul_batch_t *b, int p) { return p == AZUL_PERSP_MOVER ? (b->x ? AZUL_PERSP_MOVER : AZUL_PERSP_CURRENT) : p; }

int azul_batch_init(azul_batch_t *b, const uint8_t *active_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_INIT); a.active = active_dev;
    return launch_op(b, a, stream);
}

int azul_batch_new_round(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_NEW_ROUND); a.active = active_dev; a.status = status_dev;
    return launch_op(b, a, stream);
}

int azul_batch_move(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev) return fail(AZUL_ERR_INVALID, "actions_dev is NULL");
    OpArgs a = op_args(OP_MOVE); a.actions = actions_dev; a.active = active_dev;
    return launch_op(b, a, stream);
}

int azul_batch_legal_mask(azul_batch_t *b, uint8_t *mask_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!mask_dev) return fail(AZUL_ERR_INVALID, "mask_dev is NULL");
    OpArgs a = op_args(OP_QUERY); a.mask = mask_dev;
    return launch_op(b, a, stream);
}

int azul_batch_next_player(azul_batch_t *b, const uint8_t *active_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_NEXT_PLAYER); a.active = active_dev;
    return launch_op(b, a, stream);
}

int azul_batch_flags(azul_batch_t *b, uint8_t *flags_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!flags_dev) return fail(AZUL_ERR_INVALID, "flags_dev is NULL");
    OpArgs a = op_args(OP_QUERY); a.flags = flags_dev;
    return launch_op(b, a, stream);
}

int azul_batch_count_score(azul_batch_t *b, const uint8_t *active_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_COUNT_SCORE); a.active = active_dev;
    return launch_op(b, a, stream);
}

int azul_batch_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, uint8_t *status_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev) return fail(AZUL_ERR_INVALID, "actions_dev is NULL");
    OpArgs a = op_args(OP_STEP); a.actions = actions_dev; a.active = active_dev; a.status = status_dev;
    return launch_op(b, a, stream);
}

int azul_batch_statistics(azul_batch_t *b, double *stats_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!stats_dev) return fail(AZUL_ERR_INVALID, "stats_dev is NULL");
    OpArgs a = op_args(OP_QUERY); a.stats = stats_dev;
    return launch_op(b, a, stream);
}

int azul_batch_runner_init(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_RUNNER_INIT); a.active = active_dev; a.status = status_dev;
    return launch_op(b, a, stream);
}

int azul_batch_runner_reset(azul_batch_t *b, const uint8_t *active_dev, uint8_t *status_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    OpArgs a = op_args(OP_RUNNER_RESET); a.active = active_dev; a.status = status_dev;
    return launch_op(b, a, stream);
}

int azul_batch_runner_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, int32_t *reward_dev,
                           uint8_t *done_dev, uint8_t *status_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev) return fail(AZUL_ERR_INVALID, "actions_dev is NULL");
    OpArgs a = op_args(OP_RUNNER_STEP);
    a.actions = actions_dev; a.active = active_dev; a.reward = reward_dev; a.done = done_dev; a.status = status_dev;
    return launch_op(b, a, stream);
}

int azul_batch_observe(azul_batch_t *b, int perspective, float *obs_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || !obs_dev || !persp_ok(b, perspective)) return fail(AZUL_ERR_INVALID, "azul_batch_observe: bad arguments");
    OpArgs a = op_args(OP_QUERY); a.obs = obs_dev; a.persp = persp_of(b, perspective);
    return launch_op(b, a, stream);
}

int azul_batch_random_action(azul_batch_t *b, const uint8_t *active_dev, int32_t *actions_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev) return fail(AZUL_ERR_INVALID, "actions_dev is NULL");
    OpArgs a = op_args(OP_RANDOM_ACTION); a.active = active_dev; a.actions_out = actions_dev;
    return launch_op(b, a, stream);
}

int azul_batch_policy_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, int32_t *reward_dev,
                           uint8_t *done_dev, uint8_t *status_dev, int perspective, float *obs_next_dev,
                           uint8_t *mask_next_dev, uint8_t *player_next_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev || perspective < 0 || perspective > 2) return fail(AZUL_ERR_INVALID, "azul_batch_policy_step: bad arguments");
    OpArgs a = op_args(OP_POLICY_STEP);
    a.actions = actions_dev; a.active = active_dev; a.reward = reward_dev; a.done = done_dev; a.status = status_dev;
    a.persp = perspective; a.obs = obs_next_dev; a.mask = mask_next_dev; a.player = player_next_dev;
    return launch_op(b, a, stream);
}

int azul_batch_agent_step(azul_batch_t *b, const int32_t *actions_dev, const uint8_t *active_dev, int32_t *reward_dev,
                          uint8_t *done_dev, uint8_t *status_dev, int perspective, float *obs_next_dev,
                          uint8_t *mask_next_dev, uint8_t *player_next_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!actions_dev || perspective < 0 || perspective > 2) return fail(AZUL_ERR_INVALID, "azul_batch_agent_step: bad arguments");
    OpArgs a = op_args(OP_AGENT_STEP);
    a.actions = actions_dev; a.active = active_dev; a.reward = reward_dev; a.done = done_dev; a.status = status_dev;
    a.persp = perspective; a.obs = obs_next_dev; a.mask = mask_next_dev; a.player = player_next_dev;
    return launch_op(b, a, stream);
}

// GameRunner.step / reset with an EXTERNAL opponent, cut at their opponent_move() calls (azul_env2.hpp: NET_*)
static int net_op(azul_batch_t *b, int op, const int32_t *actions_dev, const uint8_t *active_dev, uint8_t *pending_dev, uint8_t *replies_dev,
                  int32_t *reward_dev, uint8_t *done_dev, uint8_t *status_dev, float *obs_opp_dev, uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream)
{
    if (!b || !pending_dev || (op != OP_NET_RESET && !actions_dev)) return fail(AZUL_ERR_INVALID, "azul_batch_net_*: bad arguments");
    OpArgs a = op_args(op);
    a.actions = actions_dev; a.active = active_dev; a.pending = pending_dev; a.replies = replies_dev; a.reward = reward_dev; a.done = done_dev;
    a.status = status_dev; a.obs = obs_opp_dev; a.mask = mask_opp_dev; a.persp = AZUL_PERSP_CURRENT; a.owing = owing_dev;
    if (b->x) return fail(AZUL_ERR_INVALID, "azul_batch_net_*: GameRunner is two-player (game_runner.py:50)");
    if (owing_dev) HIP_TRY(hipMemsetAsync(owing_dev, 0, sizeof(uint32_t), (hipStream_t)stream));
    return launch_op(b, a, stream);
}

int azul_batch_net_step_begin(azul_batch_t *b, const int32_t *actions_dev, uint8_t *pending_dev, uint8_t *replies_dev, int32_t *reward_dev,
                              uint8_t *done_dev, uint8_t *status_dev, float *obs_opp_dev, uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    return net_op(b, OP_NET_BEGIN, actions_dev, nullptr, pending_dev, replies_dev, reward_dev, done_dev, status_dev, obs_opp_dev, mask_opp_dev, owing_dev, stream);
}

int azul_batch_net_step_reply(azul_batch_t *b, const int32_t *opp_actions_dev, uint8_t *pending_dev, uint8_t *replies_dev, int32_t *reward_dev,
                              uint8_t *done_dev, uint8_t *status_dev, float *obs_opp_dev, uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    return net_op(b, OP_NET_REPLY, opp_actions_dev, nullptr, pending_dev, replies_dev, reward_dev, done_dev, status_dev, obs_opp_dev, mask_opp_dev, owing_dev, stream);
}

int azul_batch_net_reset_begin(azul_batch_t *b, const uint8_t *active_dev, uint8_t *pending_dev, uint8_t *status_dev, float *obs_opp_dev,
                               uint8_t *mask_opp_dev, uint32_t *owing_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    return net_op(b, OP_NET_RESET, nullptr, active_dev, pending_dev, nullptr, nullptr, nullptr, status_dev, obs_opp_dev, mask_opp_dev, owing_dev, stream);
}

int azul_batch_observe_all(azul_batch_t *b, int perspective, float *obs_dev, uint8_t *mask_dev, uint8_t *player_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || !persp_ok(b, perspective)) return fail(AZUL_ERR_INVALID, "azul_batch_observe_all: bad perspective");
    OpArgs a = op_args(OP_QUERY); a.persp = persp_of(b, perspective); a.obs = obs_dev; a.mask = mask_dev; a.player = player_dev;
    return launch_op(b, a, stream);
}

int azul_discounted_returns(const int32_t *reward_dev, const uint8_t *done_dev, float *returns_dev, float *carry_dev,
                            float gamma, int n_steps, int n_games, void *stream)
{
    if (!reward_dev || !done_dev || !returns_dev || n_steps < 0 || n_games <= 0) return fail(AZUL_ERR_INVALID, "azul_discounted_returns: bad arguments");
    if (n_steps == 0) return AZUL_SUCCESS;
    STREAM_GUARD(stream);
    hipLaunchKernelGGL(azul_returns_kernel, dim3(((u32)n_games + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       reward_dev, done_dev, returns_dev, carry_dev, gamma, n_steps, (u32)n_games);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_discounted_returns_ring(const int32_t *reward_ring_dev, const uint8_t *done_ring_dev, float *returns_ring_dev, float gamma,
                                 int ring_steps, int64_t steps_played, int span_steps, int n_games, void *stream)
{
    if (!reward_ring_dev || !done_ring_dev || !returns_ring_dev || ring_steps <= 0 || steps_played <= 0 || span_steps < 0 ||
        span_steps > ring_steps || span_steps > steps_played || n_games <= 0 || (uint64_t)ring_steps * (uint64_t)n_games >= (1ull << 32))
        return fail(AZUL_ERR_INVALID, "azul_discounted_returns_ring: bad arguments");
    if (span_steps == 0) return AZUL_SUCCESS;
    STREAM_GUARD(stream);
    hipLaunchKernelGGL(azul_returns_ring_kernel, dim3(((u32)n_games + 63u) / 64u), dim3(64), 0, (hipStream_t)stream, reward_ring_dev, done_ring_dev,
                       returns_ring_dev, gamma, ring_steps, (int)(steps_played % ring_steps == 0 ? ring_steps : steps_played % ring_steps), span_steps,
                       (u32)n_games);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_policy_head(const float *logits_dev, const uint8_t *mask_dev, uint64_t seed, uint64_t counter, const uint64_t *counter_dev,
                     int n_games, uint32_t game_id_base, int32_t *action_dev, float *logp_dev, float *entropy_dev, void *stream)
{
    if (!logits_dev || !mask_dev || !action_dev || !logp_dev || !entropy_dev || n_games <= 0) return fail(AZUL_ERR_INVALID, "azul_policy_head: bad arguments");
    if (((uintptr_t)mask_dev & 3u) != 0) return fail(AZUL_ERR_INVALID, "azul_policy_head: mask_dev must be 4-byte aligned");
    STREAM_GUARD(stream);
    hipLaunchKernelGGL(azul_policy_head_kernel, dim3(((u32)n_games + 3u) / 4u), dim3(64), 0, (hipStream_t)stream, logits_dev, mask_dev,
                       (u64)seed, (u64)counter, (const u64 *)counter_dev, (u32)n_games, action_dev, logp_dev, entropy_dev, (u32)game_id_base);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_policy_forward(const float *obs_dev, const uint8_t *mask_dev, const float *w1t_dev, const float *b1_dev, const float *w2c_dev,
                        const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs, int hidden_size, int num_actions,
                        uint64_t seed, uint64_t counter, uint64_t *counter_dev, int advance_counter, int n_games, uint32_t game_id_base,
                        float *value_dev, int32_t *action_dev, float *logp_dev, float *entropy_dev, float *logits_dev, void *stream)
{
    if (num_inputs != PF_IN || hidden_size != PF_HID || num_actions != PF_ACT)
        return fail(AZUL_ERR_INVALID, "azul_policy_forward: only ActorCritic(136, 180, hidden 180) is compiled in");
    if (!obs_dev || !mask_dev || !w1t_dev || !b1_dev || !w2c_dev || !b2c_dev || !w2a_t_dev || !b2a_dev || !value_dev || !action_dev ||
        !logp_dev || !entropy_dev || n_games <= 0 || advance_counter < 0)
        return fail(AZUL_ERR_INVALID, "azul_policy_forward: bad arguments");
    if (((uintptr_t)w1t_dev & 7u) != 0 || ((uintptr_t)mask_dev & 3u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_policy_forward: w1t_dev must be 8-byte aligned, mask_dev 4-byte aligned");
    STREAM_GUARD(stream);
    PolicyWeights W = {w1t_dev, b1_dev, w2c_dev, b2c_dev, w2a_t_dev, b2a_dev};
    hipLaunchKernelGGL(azul_policy_forward_kernel, dim3(((u32)n_games + PF_GAMES - 1) / PF_GAMES), dim3(256), 0, (hipStream_t)stream,
                       obs_dev, mask_dev, W, (u64)seed, (u64)counter, (u64 *)counter_dev, advance_counter, (u32)n_games, value_dev,
                       action_dev, logp_dev, entropy_dev, logits_dev, (u32)game_id_base);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

static int launch_rollout(azul_batch_t *b, const PolicyWeights &W, const RolloutArgs &a, int opp, void *stream)
{
    const hipStream_t st = (hipStream_t)stream;
    const bool lid = b->d.rules.tile_pool == POOL_LID;
    const dim3 grid2((b->d.n + PF_GAMES - 1) / PF_GAMES), block2(64 * PR2_WAVES);
    if (lid) {
        if (opp == 2) hipLaunchKernelGGL((azul_policy_rollout2_kernel<true, 2>), grid2, block2, 0, st, b->d, W, a);
        else if (opp == 1) hipLaunchKernelGGL((azul_policy_rollout2_kernel<true, 1>), grid2, block2, 0, st, b->d, W, a);
        else hipLaunchKernelGGL((azul_policy_rollout2_kernel<true, 0>), grid2, block2, 0, st, b->d, W, a);
    } else {
        if (opp == 2) hipLaunchKernelGGL((azul_policy_rollout2_kernel<false, 2>), grid2, block2, 0, st, b->d, W, a);
        else if (opp == 1) hipLaunchKernelGGL((azul_policy_rollout2_kernel<false, 1>), grid2, block2, 0, st, b->d, W, a);
        else hipLaunchKernelGGL((azul_policy_rollout2_kernel<false, 0>), grid2, block2, 0, st, b->d, W, a);
    }
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_batch_policy_rollout(azul_batch_t *b, int n_steps, int opponent_random, const float *w1t_dev, const float *b1_dev,
                              const float *w2c_dev, const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs,
                              int hidden_size, int num_actions, uint64_t seed, uint64_t counter, uint64_t *counter_dev, float *obs_dev,
                              uint8_t *mask_dev, uint8_t *player_dev, int32_t *action_dev, int32_t *reward_dev, uint8_t *done_dev,
                              float *value_dev, float *logp_dev, float *entropy_dev, uint8_t *status_dev, void *stream)
{
    return azul_batch_policy_rollout_returns(b, n_steps, opponent_random, w1t_dev, b1_dev, w2c_dev, b2c_dev, w2a_t_dev, b2a_dev, num_inputs, hidden_size,
                                             num_actions, seed, counter, counter_dev, obs_dev, mask_dev, player_dev, action_dev, reward_dev, done_dev,
                                             value_dev, logp_dev, entropy_dev, status_dev, nullptr, 0.f, stream);
}

int azul_batch_policy_rollout_returns(azul_batch_t *b, int n_steps, int opponent_random, const float *w1t_dev, const float *b1_dev,
                                      const float *w2c_dev, const float *b2c_dev, const float *w2a_t_dev, const float *b2a_dev, int num_inputs,
                                      int hidden_size, int num_actions, uint64_t seed, uint64_t counter, uint64_t *counter_dev, float *obs_dev,
                                      uint8_t *mask_dev, uint8_t *player_dev, int32_t *action_dev, int32_t *reward_dev, uint8_t *done_dev,
                                      float *value_dev, float *logp_dev, float *entropy_dev, uint8_t *status_dev, float *returns_dev, float gamma,
                                      void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || n_steps < 0) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: bad arguments");
    if (b->x) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: the policy entries are compiled for the reference's two-player game "
                                            "(ActorCritic(136, 180): 180 actions, 136 observations; game_runner.py:50) -- not for 3 / 4 players or extended rules");
    if (num_inputs != PF_IN || hidden_size != PF_HID || num_actions != PF_ACT)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: only ActorCritic(136, 180, hidden 180) is compiled in");
    if (!w1t_dev || !b1_dev || !w2c_dev || !b2c_dev || !w2a_t_dev || !b2a_dev || !obs_dev || !mask_dev || !player_dev || !action_dev ||
        !reward_dev || !done_dev || !value_dev || !logp_dev || !entropy_dev)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: NULL pointer");
    if (((uintptr_t)w1t_dev & 7u) != 0) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: w1t_dev must be 8-byte aligned");
    if (((uintptr_t)obs_dev & 15u) != 0 || ((uintptr_t)mask_dev & 3u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout: obs_dev must be 16-byte aligned, mask_dev 4-byte aligned");
    PolicyWeights W = {w1t_dev, b1_dev, w2c_dev, b2c_dev, w2a_t_dev, b2a_dev};
    RolloutArgs a = {n_steps, obs_dev, mask_dev, player_dev, action_dev, reward_dev, done_dev, value_dev, logp_dev, entropy_dev, status_dev,
                     returns_dev, gamma, (u64)seed, (u64)counter, (u64 *)counter_dev};
    int rc = launch_rollout(b, W, a, opponent_random ? 1 : 0, stream);
    if (rc == AZUL_SUCCESS && returns_dev && n_steps > 32)       // the kernel keeps a window's rewards in 32 lanes: longer windows get the separate scan
        return azul_discounted_returns(reward_dev, done_dev, returns_dev, nullptr, gamma, n_steps, (int)b->d.n, stream);
    return rc;
}

/* GameRunner(opponent=Agent(...)) inside the persistent rollout (game_runner.py:27-30, 37-47, 84-85; scripts/run_batch.py:6-10) */
int azul_batch_policy_rollout_vs(azul_batch_t *b, int n_steps, const azul_net_weights_t *agent, const azul_net_weights_t *opponent, int num_inputs,
                                 int hidden_size, int num_actions, uint64_t seed, uint64_t opponent_seed, uint64_t counter, uint64_t *counter_dev,
                                 const azul_rollout_buffers_t *out, float gamma, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || n_steps < 0 || !agent || !opponent || !out) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: bad arguments");
    if (b->x) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: the policy entries are compiled for the reference's two-player game");
    if (num_inputs != PF_IN || hidden_size != PF_HID || num_actions != PF_ACT)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: only ActorCritic(136, 180, hidden 180) is compiled in");
    if (!agent->w1t || !agent->b1 || !agent->w2c || !agent->b2c || !agent->w2a_t || !agent->b2a || !opponent->w1t || !opponent->b1 || !opponent->w2a_t ||
        !opponent->b2a || !out->obs || !out->mask || !out->player || !out->action || !out->reward || !out->done || !out->value || !out->logp || !out->entropy)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: NULL pointer");
    if (((uintptr_t)agent->w1t & 15u) != 0 || ((uintptr_t)opponent->w1t & 15u) != 0 || ((uintptr_t)agent->w2a_t & 7u) != 0 || ((uintptr_t)opponent->w2a_t & 7u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: w1t must be 16-byte aligned, w2a_t 8-byte aligned");
    if (((uintptr_t)out->obs & 15u) != 0 || ((uintptr_t)out->mask & 3u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: obs must be 16-byte aligned, mask 4-byte aligned");
    if ((out->opp_action || out->opp_logp) && out->opp_slots <= 0) return fail(AZUL_ERR_INVALID, "azul_batch_policy_rollout_vs: opp_slots must be positive with a trace");
    PolicyWeights W = {agent->w1t, agent->b1, agent->w2c, agent->b2c, agent->w2a_t, agent->b2a};
    RolloutArgs a = {n_steps, out->obs, out->mask, out->player, out->action, out->reward, out->done, out->value, out->logp, out->entropy, out->status,
                     out->returns, gamma, (u64)seed, (u64)counter, (u64 *)counter_dev};
    a.Wopp = {opponent->w1t, opponent->b1, opponent->w2c, opponent->b2c, opponent->w2a_t, opponent->b2a};
    a.opp_seed = (u64)opponent_seed;
    a.opp_action = out->opp_action; a.opp_logp = out->opp_logp; a.opp_replies = out->opp_replies; a.opp_slots = out->opp_slots;
    int rc = launch_rollout(b, W, a, 2, stream);
    if (rc == AZUL_SUCCESS && out->returns && n_steps > 32)
        return azul_discounted_returns(out->reward, out->done, out->returns, nullptr, gamma, n_steps, (int)b->d.n, stream);
    return rc;
}

int azul_a2c_gradients(const float *obs_dev, const uint8_t *mask_dev, const int32_t *action_dev, const float *qvals_dev, int n_samples,
                       float inv_n_total, const float *w1t_dev, const float *b1_dev, const float *w2c_dev, const float *b2c_dev,
                       const float *w2a_t_dev, const float *b2a_dev, const float *w2a_dev, int num_inputs, int hidden_size, int num_actions,
                       float *workspace_dev, int workspace_parts, float *grad_dev, const int32_t *index_dev, const int32_t *n_samples_dev,
                       const float *inv_n_total_dev, void *stream)
{
    if (num_inputs != PF_IN || hidden_size != PF_HID || num_actions != PF_ACT)
        return fail(AZUL_ERR_INVALID, "azul_a2c_gradients: only ActorCritic(136, 180, hidden 180) is compiled in");
    if (!w1t_dev || !b1_dev || !w2c_dev || !b2c_dev || !w2a_t_dev || !b2a_dev || !w2a_dev || !workspace_dev || !grad_dev || n_samples < 0 ||
        workspace_parts <= 0 || (n_samples > 0 && (!obs_dev || !mask_dev || !action_dev || !qvals_dev)))
        return fail(AZUL_ERR_INVALID, "azul_a2c_gradients: bad arguments");
    if (((uintptr_t)w2a_t_dev & 7u) != 0 || ((uintptr_t)w2a_dev & 7u) != 0 || ((uintptr_t)mask_dev & 3u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_a2c_gradients: weights must be 8-byte aligned, mask_dev 4-byte aligned");
    STREAM_GUARD(stream);
    const hipStream_t st = (hipStream_t)stream;
    const u32 tiles = ((u32)n_samples + LG_M - 1) / LG_M;
    const u32 parts = tiles < (u32)workspace_parts ? tiles : (u32)workspace_parts;
    if (parts == 0) { HIP_TRY(hipMemsetAsync(grad_dev, 0, sizeof(float) * LG_P_TOTAL, st)); return AZUL_SUCCESS; }
    PolicyWeights W = {w1t_dev, b1_dev, w2c_dev, b2c_dev, w2a_t_dev, b2a_dev};
    LearnerArgs a = {obs_dev, mask_dev, action_dev, qvals_dev, (u32)n_samples, inv_n_total, w2a_dev, workspace_dev, index_dev, n_samples_dev,
                     inv_n_total_dev};
    hipLaunchKernelGGL(azul_a2c_grad_kernel, dim3(parts), dim3(64 * LG_WAVES), 0, st, W, a);
    hipLaunchKernelGGL(azul_a2c_reduce_kernel, dim3((LG_P_TOTAL + 255) / 256), dim3(256), 0, st, workspace_dev, parts, grad_dev);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

#if defined(AZ_LG_PROFILE)
extern "C" int azul_debug_lg_profile(uint64_t *host, int n, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(host, HIP_SYMBOL(lg_prof_dev), sizeof(uint64_t) * (size_t)n) != hipSuccess) return AZUL_ERR_HIP;
    if (reset) { static const unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lg_prof_dev), z, sizeof(z)) != hipSuccess) return AZUL_ERR_HIP; }
    return AZUL_SUCCESS;
}
#endif

int azul_a2c_apply_adam(const float *grad_dev, float *flat_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float lr, float beta1, float beta2,
                        float eps, int step, float *critic1_w, float *critic1_b, float *critic2_w, float *critic2_b, float *actor1_w,
                        float *actor1_b, float *actor2_w, float *actor2_b, int32_t *step_dev, const float *n_total_dev, float n_total_host,
                        float *stats_out_dev, void *stream)
{
    if (!grad_dev || !flat_dev || !exp_avg_dev || !exp_avg_sq_dev || (!step_dev && step < 1) || !critic1_w || !critic1_b || !critic2_w ||
        !critic2_b || !actor1_w || !actor1_b || !actor2_w || !actor2_b)
        return fail(AZUL_ERR_INVALID, "azul_a2c_apply_adam: bad arguments");
    STREAM_GUARD(stream);
    const double st = step_dev ? 1.0 : (double)step;
    const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
    ModuleParams P = {critic1_w, critic1_b, critic2_w, critic2_b, actor1_w, actor1_b, actor2_w, actor2_b};
    if (step_dev) hipLaunchKernelGGL(azul_a2c_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, n_total_dev);
    hipLaunchKernelGGL(azul_a2c_apply_kernel, dim3((LG_P_PARAMS + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_dev, flat_dev, exp_avg_dev,
                       exp_avg_sq_dev, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), P, (const i32 *)step_dev, n_total_dev, n_total_host, stats_out_dev);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_select_episode_samples(const uint8_t *done_ring_dev, const int32_t *action_ring_dev, int window_steps, int ring_windows, int n_games,
                                int64_t steps_played, int32_t *pending_dev, int32_t *index_dev, int32_t *count_dev, float *countf_dev,
                                int32_t *scratch_dev, void *stream)
{
    if (!done_ring_dev || !action_ring_dev || !pending_dev || !index_dev || !count_dev || !scratch_dev || window_steps <= 0 || ring_windows <= 0 ||
        n_games <= 0 || steps_played < window_steps || steps_played % window_steps != 0 || steps_played > 0x7fff0000ll)
        return fail(AZUL_ERR_INVALID, "azul_select_episode_samples: bad arguments");
    STREAM_GUARD(stream);
    const u32 N = (u32)n_games, blocks = (N + 3u) / 4u;      // one wave per game, four games per workgroup
    const int R = window_steps * ring_windows;
    hipLaunchKernelGGL(azul_select_ring_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, done_ring_dev, action_ring_dev, window_steps, R, N,
                       (i32)steps_played, (const i32 *)pending_dev, scratch_dev, count_dev);
    hipLaunchKernelGGL(azul_select_ring_write_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, action_ring_dev, R, N, pending_dev,
                       (const i32 *)scratch_dev, index_dev, count_dev, countf_dev);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_select_complete_samples(const uint8_t *done_dev, const int32_t *action_dev, int n_steps, int n_games, int32_t *index_dev,
                                 int32_t *count_dev, void *stream)
{
    if (!done_dev || !action_dev || !index_dev || !count_dev || n_steps < 0 || n_games <= 0)
        return fail(AZUL_ERR_INVALID, "azul_select_complete_samples: bad arguments");
    STREAM_GUARD(stream);
    hipLaunchKernelGGL(azul_select_complete_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, done_dev, action_dev, n_steps, (u32)n_games,
                       index_dev, count_dev);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_batch_sample_mask(azul_batch_t *b, const uint8_t *mask_dev, const uint8_t *active_dev, int32_t *actions_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!mask_dev || !actions_dev) return fail(AZUL_ERR_INVALID, "azul_batch_sample_mask: NULL pointer");
    OpArgs a = op_args(OP_SAMPLE_MASK); a.mask_in = mask_dev; a.active = active_dev; a.actions_out = actions_dev;
    return launch_op(b, a, stream);
}

int azul_batch_score_preview(azul_batch_t *b, int32_t *potential_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!potential_dev) return fail(AZUL_ERR_INVALID, "potential_dev is NULL");
    OpArgs a = op_args(OP_QUERY); a.potential = potential_dev;
    return launch_op(b, a, stream);
}

// ---- azul_game_call: one method call of the single-game API in one submission + one synchronisation ---------------------------
struct CallScratch {             // one call's arguments and results in PINNED, device-visible host memory: the rule kernel reads its action /
    i32 action_in;               // mask straight from it and writes every result straight into it (zero-copy over PCIe: a few hundred bytes),
    u32 pos;                     // so a call is [record / stream uploads when stale] -> ONE launch -> ONE synchronisation
    i32 reward, action_out, potential, next_action;
    uint8_t status, done, flags, player;
    uint8_t rng_dirty, pad_[3];  // the kernel regenerated the 624 words (written by every call: 0 for ops that do not draw)
    uint8_t mask_in[AZUL_MAX_ACTIONS + 4];
    uint8_t mask[AZUL_MAX_ACTIONS + 4];
    float obs[AZUL_MAX_OBS];
    double stats[AZUL_NUM_STATS];
    uint8_t record[AZUL_RECORD_BYTES_WIDE];      // staging of record_in, then the record after the call
    u32 mt[AZUL_MT_WORDS];                       // staging of mt_in / mt_out
};

int azul_game_call(azul_batch_t *b, azul_call_t *c, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || !c) return fail(AZUL_ERR_INVALID, "azul_game_call: batch / call is NULL");
    if (int rc = check_range(b, c->game, 1)) return rc;
    static const int op_of[] = {OP_QUERY, OP_INIT, OP_NEW_ROUND, OP_MOVE, OP_NEXT_PLAYER, OP_COUNT_SCORE, OP_STEP, OP_RUNNER_INIT, OP_RUNNER_RESET,
                                OP_RUNNER_STEP, OP_SAMPLE_MASK};
    if (c->op < 0 || c->op > AZUL_CALL_SAMPLE_MASK) return fail(AZUL_ERR_INVALID, "azul_game_call: unknown op");
    if (c->op == AZUL_CALL_SAMPLE_MASK && !c->mask_in) return fail(AZUL_ERR_INVALID, "azul_game_call: AZUL_CALL_SAMPLE_MASK needs mask_in");
    if ((c->want & AZUL_WANT_RECORD) && !c->record_out) return fail(AZUL_ERR_INVALID, "azul_game_call: AZUL_WANT_RECORD needs record_out");
    if (c->mt_in && c->pos_in > 624u) return fail(AZUL_ERR_INVALID, "azul_game_call: index outside 0..624");
    const int obs_persp = c->op == AZUL_CALL_QUERY ? c->arg : c->obs_persp;
    if ((c->want & AZUL_WANT_OBS) && !persp_ok(b, obs_persp))
        return fail(AZUL_ERR_INVALID, "azul_game_call: AZUL_WANT_OBS needs a perspective (AZUL_CALL_QUERY: arg; other ops: obs_persp)");
    if ((c->want & AZUL_WANT_POS_IN) && c->pos_in > 624u) return fail(AZUL_ERR_INVALID, "azul_game_call: index outside 0..624");
    const size_t NA = (size_t)azul_batch_num_actions(b), NOBS = (size_t)azul_batch_obs_size(b);
    if (c->record_in) if (int rc = record_in_domain(b, (const uint8_t *)c->record_in)) return rc;
    const hipStream_t st = (hipStream_t)stream;
    if (!b->call_pin) {
        HIP_TRY(hipHostMalloc((void **)&b->call_pin, sizeof(CallScratch), hipHostMallocDefault));      // pinned + mapped: kernels address it directly
        memset(b->call_pin, 0, sizeof(CallScratch));
    }
    CallScratch *H = (CallScratch *)b->call_pin;
    const size_t RB = (size_t)b->rec_bytes;
    const size_t g = (size_t)c->game;
    // ---- inputs that live in device arrays: only when the caller says the device copies are stale
    if (c->record_in) {
        memcpy(H->record, c->record_in, RB);
        HIP_TRY(hipMemcpyAsync(b->d.state + g * RB, H->record, RB, hipMemcpyHostToDevice, st));
        b->handed_in = true;
    }
    if (c->mt_in) {
        memcpy(H->mt, c->mt_in, sizeof(H->mt));
        H->pos = c->pos_in;
        HIP_TRY(hipMemcpyAsync(b->d.mt + g * 624, H->mt, sizeof(H->mt), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b->d.mtpos + g, &H->pos, sizeof(u32), hipMemcpyHostToDevice, st));
        if (c->record_in || (c->want & AZUL_WANT_RECORD)) HIP_TRY(hipStreamSynchronize(st));      // H->record / H->pos are written by the kernel below
    } else if (c->record_in && (c->want & AZUL_WANT_RECORD)) {
        HIP_TRY(hipStreamSynchronize(st));               // the staged record must have left H->record before the kernel overwrites it
    }
    // ---- the rule kernel on this one game: arguments read from, results written to the pinned block
    H->action_in = c->arg;
    if (c->op == AZUL_CALL_SAMPLE_MASK) memcpy(H->mask_in, c->mask_in, NA);
    OpArgs a = op_args(op_of[c->op]);
    a.first = (u32)c->game;
    a.actions = &H->action_in;
    a.status = &H->status;
    if (c->op == AZUL_CALL_RUNNER_STEP) { a.reward = &H->reward; a.done = &H->done; }
    if (c->op == AZUL_CALL_SAMPLE_MASK) { a.mask_in = H->mask_in; a.actions_out = &H->action_out; }
    if (c->want & AZUL_WANT_MASK) a.mask = H->mask;
    if (c->want & AZUL_WANT_OBS) { a.obs = H->obs; a.persp = persp_of(b, obs_persp); }
    if (c->want & AZUL_WANT_FLAGS) a.flags = &H->flags;
    if (c->want & AZUL_WANT_POTENTIAL) a.potential = &H->potential;
    if (c->want & AZUL_WANT_STATS) a.stats = H->stats;
    if (c->want & AZUL_WANT_RECORD) a.rec_out = H->record;
    if (c->want & AZUL_WANT_NEXT_ACTION) a.next_action = &H->next_action;
    if ((c->want & AZUL_WANT_POS_IN) && !c->mt_in) a.pos_set = 1u + c->pos_in;
    a.rng_dirty = &H->rng_dirty;
    a.pos_out = &H->pos;
    H->status = AZUL_OK;
    if (int rc = launch_op(b, a, stream, 1)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    // the result head is shared by all ops: only what THIS op / `want` produced is handed out, the rest reads zero
    c->status = H->status;
    c->reward = c->op == AZUL_CALL_RUNNER_STEP ? H->reward : 0;
    c->done = c->op == AZUL_CALL_RUNNER_STEP ? H->done : 0;
    c->action = c->op == AZUL_CALL_SAMPLE_MASK ? H->action_out : 0;
    c->flags = (c->want & AZUL_WANT_FLAGS) ? H->flags : 0;
    c->potential = (c->want & AZUL_WANT_POTENTIAL) ? H->potential : 0;
    c->next_action = (c->want & AZUL_WANT_NEXT_ACTION) ? H->next_action : -2;
    c->pos_out = H->pos;
    if (c->want & AZUL_WANT_MASK) memcpy(c->mask, H->mask, NA);
    if (c->want & AZUL_WANT_OBS) memcpy(c->obs, H->obs, NOBS * sizeof(float));
    if (c->want & AZUL_WANT_STATS) memcpy(c->stats, H->stats, sizeof(H->stats));
    if (c->want & AZUL_WANT_RECORD) memcpy(c->record_out, H->record, RB);
    // whether the 624 words were regenerated is the kernel's own statement (Rng::dirty, written next to the status): it does not
    // depend on the caller's `pos_in`, which is only the index uploaded together with mt_in
    c->rng_regenerated = H->rng_dirty ? 1 : 0;
    if (c->rng_regenerated && c->mt_out) {
        HIP_TRY(hipMemcpyAsync(H->mt, b->d.mt + g * 624, sizeof(H->mt), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        memcpy(c->mt_out, H->mt, sizeof(H->mt));
    }
    return AZUL_SUCCESS;
}

int azul_batch_selfplay_strided(azul_batch_t *b, int n_steps, uint8_t *mask_dev, int mask_row_bytes, uint64_t *maskbits_dev, int32_t *action_dev,
                                int32_t *reward_dev, uint8_t *done_dev, uint32_t *packed_dev, uint8_t *rec_dev, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b || n_steps < 0) return fail(AZUL_ERR_INVALID, "azul_batch_selfplay: bad arguments");
    if (mask_row_bytes < AZUL_NUM_ACTIONS) return fail(AZUL_ERR_INVALID, "azul_batch_selfplay: mask rows hold 180 bytes, mask_row_bytes >= 180");
    if (n_steps == 0) return AZUL_SUCCESS;
    if (b->x) {
        // three / four players and extended-rule batches (row N4): the flat loop mask -> RandomAgent -> Azul.step with a fresh Azul + new_round()
        // at each game end; two games per wavefront on the wide record (azul_rules_x.hpp); `reward` all zero (the shaped reward is
        // GameRunner's: two players)
        const int NA = azul_batch_num_actions(b), NAP = (NA + 7) / 8 * 8;
        if (mask_row_bytes < NA) return fail(AZUL_ERR_INVALID, "azul_batch_selfplay: mask rows hold azul_batch_num_actions bytes, mask_row_bytes must not be smaller");
        if ((u64)n_steps * b->d.n * (u64)(rec_dev && mask_row_bytes < AZUL_RECORD_BYTES_WIDE ? AZUL_RECORD_BYTES_WIDE : mask_row_bytes) >= (1ull << 32))
            return fail(AZUL_ERR_INVALID, "azul_batch_selfplay: a trajectory stream of one launch must stay below 4 GiB (use fewer moves per launch)");
        const azx::XBatchDev xb = xdev(b);
        const azx::XTraj t = {n_steps, mask_dev, (u64 *)maskbits_dev, action_dev, reward_dev, done_dev, rec_dev, packed_dev, (u32)mask_row_bytes};
        const bool none = !mask_dev && !maskbits_dev && !action_dev && !reward_dev && !done_dev && !rec_dev && !packed_dev;
        const bool core = mask_dev && action_dev && reward_dev && done_dev && packed_dev && !rec_dev;     // + maskbits_dev or not
        // padded rows (8-byte aligned, room for ceil(NA / 8) 8-byte chunks; five displays: >= 192 like the two-player kernel) take the
        // one-store-per-row path
        const bool pad = mask_row_bytes % 8 == 0 && mask_row_bytes >= (b->displays == 5 ? 192 : NAP) && ((uintptr_t)mask_dev & 7u) == 0u;
        const dim3 grid((b->d.n + 1u) / 2u), block(64);
        const hipStream_t st = (hipStream_t)stream;
        const bool pair = b->timing && b->timed_pairs < AZ_TIMED_PAIRS;       // per-launch event pair inside a timed region (as below)
        if (pair) {
            while ((int)b->lev.size() < 2 * (b->timed_pairs + 1)) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); b->lev.push_back(e); }
            HIP_TRY(hipEventRecord(b->lev[2 * b->timed_pairs], st));
        }
        if (none) AZ_X_DISPATCH(b, hipLaunchKernelGGL((azul_x_selfplay_kernel<PP, DD, 0, false, false>), grid, block, 0, st, xb, t));
        else if (core && pad && maskbits_dev) AZ_X_DISPATCH(b, hipLaunchKernelGGL((azul_x_selfplay_kernel<PP, DD, 1, true, true>), grid, block, 0, st, xb, t));
        else if (core && pad) AZ_X_DISPATCH(b, hipLaunchKernelGGL((azul_x_selfplay_kernel<PP, DD, 1, true, false>), grid, block, 0, st, xb, t));
        else AZ_X_DISPATCH(b, hipLaunchKernelGGL((azul_x_selfplay_kernel<PP, DD, 2, false, false>), grid, block, 0, st, xb, t));
        HIP_TRY(hipGetLastError());
        if (pair) { HIP_TRY(hipEventRecord(b->lev[2 * b->timed_pairs + 1], st)); b->timed_pairs++; }
        if (b->timing) b->timed_launches++;
        return AZUL_SUCCESS;
    }
    if ((u64)n_steps * b->d.n * (u64)(rec_dev && mask_row_bytes < AZUL_RECORD_BYTES ? AZUL_RECORD_BYTES : mask_row_bytes) >= (1ull << 32))
        return fail(AZUL_ERR_INVALID, "azul_batch_selfplay: a trajectory stream of one launch must stay below 4 GiB (use fewer moves per launch)");
    TrajArgs t = {n_steps, mask_dev, (u64 *)maskbits_dev, action_dev, reward_dev, done_dev, rec_dev, packed_dev};
    const bool none = !mask_dev && !maskbits_dev && !action_dev && !reward_dev && !done_dev && !rec_dev && !packed_dev;
    const bool core = mask_dev && action_dev && reward_dev && done_dev && packed_dev && !rec_dev;     // + maskbits_dev or not
    const bool full = core && maskbits_dev;
    const dim3 grid((b->d.n + 1u) / 2u), block(64);
    const hipStream_t st = (hipStream_t)stream;
    const u32 ms = (u32)mask_row_bytes;
    // padded rows (>= 192 bytes, 8-byte aligned: alloc_trajectory(mask_pitch = 192)) take the one-store-per-row path
    const bool pad = ms >= 192u && ms % 8u == 0u && ((uintptr_t)mask_dev & 7u) == 0u;
#define AZ_LAUNCH(LID, LIM) do { \
        if (none) hipLaunchKernelGGL((azul_selfplay2_kernel<LID, 0, false, false, LIM>), grid, block, 0, st, b->d, t, ms); \
        else if (full && pad) hipLaunchKernelGGL((azul_selfplay2_kernel<LID, 1, true, true, LIM>), grid, block, 0, st, b->d, t, ms); \
        else if (core && pad) hipLaunchKernelGGL((azul_selfplay2_kernel<LID, 1, true, false, LIM>), grid, block, 0, st, b->d, t, ms); \
        else if (full) hipLaunchKernelGGL((azul_selfplay2_kernel<LID, 1, false, true, LIM>), grid, block, 0, st, b->d, t, ms); \
        else hipLaunchKernelGGL((azul_selfplay2_kernel<LID, 2, false, false, LIM>), grid, block, 0, st, b->d, t, ms); \
    } while (0)
    // inside a timed region the first AZ_TIMED_PAIRS launches are bracketed by their own event pair (the kernel's duration,
    // not the distance between launches)
    const bool pair = b->timing && b->timed_pairs < AZ_TIMED_PAIRS;
    if (pair) {
        while ((int)b->lev.size() < 2 * (b->timed_pairs + 1)) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); b->lev.push_back(e); }
        HIP_TRY(hipEventRecord(b->lev[2 * b->timed_pairs], st));
    }
    // (a batch with a move limit runs the instantiation that carries the limit's code: azul_batch_set_move_limit; so does a batch whose
    // records the host has written -- only a handed-in state can stop on a rule error, and that instantiation marks and counts the slots a
    // stopped game no longer plays; move_limit == 0 is "no limit" there too)
    if (b->d.move_limit || b->handed_in) { if (b->d.rules.tile_pool == POOL_LID) AZ_LAUNCH(true, true); else AZ_LAUNCH(false, true); }
    else { if (b->d.rules.tile_pool == POOL_LID) AZ_LAUNCH(true, false); else AZ_LAUNCH(false, false); }
#undef AZ_LAUNCH
    HIP_TRY(hipGetLastError());
    if (pair) { HIP_TRY(hipEventRecord(b->lev[2 * b->timed_pairs + 1], st)); b->timed_pairs++; }
    if (b->timing) b->timed_launches++;
    return AZUL_SUCCESS;
}

int azul_batch_selfplay(azul_batch_t *b, int n_steps, uint8_t *mask_dev, uint64_t *maskbits_dev, int32_t *action_dev,
                        int32_t *reward_dev, uint8_t *done_dev, uint32_t *packed_dev, uint8_t *rec_dev, void *stream)
{
    return azul_batch_selfplay_strided(b, n_steps, mask_dev, b ? azul_batch_num_actions(b) : AZUL_NUM_ACTIONS, maskbits_dev, action_dev, reward_dev,
                                       done_dev, packed_dev, rec_dev, stream);
}

int azul_batch_counters(azul_batch_t *b, uint64_t *episodes_host, uint32_t *stuck_host, double *stat_sums_host, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    const size_t N = b->d.n;
    if (episodes_host) HIP_TRY(hipMemcpyAsync(episodes_host, b->d.episodes, N * sizeof(u64), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (stuck_host) HIP_TRY(hipMemcpyAsync(stuck_host, b->d.stuck, N * sizeof(u32), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (stat_sums_host) HIP_TRY(hipMemcpyAsync(stat_sums_host, b->d.stat_sum, N * 10 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_reset_counters(azul_batch_t *b, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    const size_t N = b->d.n;
    HIP_TRY(hipMemsetAsync(b->d.episodes, 0, N * sizeof(u64), (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(b->d.stuck, 0, N * sizeof(u32), (hipStream_t)stream));
    HIP_TRY(hipMemsetAsync(b->d.stat_sum, 0, N * 10 * sizeof(double), (hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_batch_counters_dev(azul_batch_t *b, uint64_t **episodes_dev, uint32_t **stuck_dev, double **stat_sums_dev)
{
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    if (episodes_dev) *episodes_dev = (uint64_t *)b->d.episodes;
    if (stuck_dev) *stuck_dev = (uint32_t *)b->d.stuck;
    if (stat_sums_dev) *stat_sums_dev = b->d.stat_sum;
    return AZUL_SUCCESS;
}

int azul_batch_set_id_base(azul_batch_t *b, uint32_t first_global_id)
{
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    b->d.id_base = first_global_id;
    return AZUL_SUCCESS;
}

int azul_batch_set_move_limit(azul_batch_t *b, uint32_t max_moves)
{
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    if (b->x && max_moves) return fail(AZUL_ERR_INVALID, "azul_batch_set_move_limit: two-player reference batches (GameRunner.move_counter is two-player, game_runner.py:36)");
    if (max_moves > 65535u) return fail(AZUL_ERR_INVALID, "azul_batch_set_move_limit: the record's move counter is 16 bits wide");
    b->d.move_limit = max_moves;
    return AZUL_SUCCESS;
}

int azul_batch_set_draw_margin(azul_batch_t *b, uint64_t margin)
{
    BATCH_GUARD(b, nullptr);
    if (!b || margin < AZ_DRAW_MARGIN || margin > 0x7fffffffull) return fail(AZUL_ERR_INVALID, "margin must be in [8192, 2^31)");
    b->d.draw_margin = margin;
    return AZUL_SUCCESS;
}

int azul_batch_segment_profile(azul_batch_t *b, uint64_t *cycles_host, int n, int reset)
{
    BATCH_GUARD(b, nullptr);
    if (!b || !cycles_host || n < 1) return fail(AZUL_ERR_INVALID, "azul_batch_segment_profile: bad arguments");
    if (n > AZ_PROF_SLOTS) n = AZ_PROF_SLOTS;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles_host, b->d.prof, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(b->d.prof, 0, AZ_PROF_SLOTS * sizeof(u64)));
    return AZUL_SUCCESS;
}

int azul_selfplay_kernel_resources(azul_batch_t *b, int padded_rows, int mask_bits, int *vgprs, int *lds_bytes, int *scratch_bytes,
                                   int *resident_waves_per_cu)
{
    BATCH_GUARD(b, nullptr);
    if (!b || b->x) return fail(AZUL_ERR_INVALID, "azul_selfplay_kernel_resources: two-player reference batches");
    const bool lid = b->d.rules.tile_pool == POOL_LID, lim = b->d.move_limit != 0 || b->handed_in;
    const void *fn;
#define AZ_PICK(LID, LIM) (padded_rows ? (mask_bits ? (const void *)azul_selfplay2_kernel<LID, 1, true, true, LIM> : (const void *)azul_selfplay2_kernel<LID, 1, true, false, LIM>) \
                                       : (mask_bits ? (const void *)azul_selfplay2_kernel<LID, 1, false, true, LIM> : (const void *)azul_selfplay2_kernel<LID, 2, false, false, LIM>))
    if (lim) fn = lid ? AZ_PICK(true, true) : AZ_PICK(false, true);
    else fn = lid ? AZ_PICK(true, false) : AZ_PICK(false, false);
#undef AZ_PICK
    hipFuncAttributes at;
    HIP_TRY(hipFuncGetAttributes(&at, fn));
    int blocks = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 64, 0));
    if (vgprs) *vgprs = at.numRegs;
    if (lds_bytes) *lds_bytes = (int)at.sharedSizeBytes;
    if (scratch_bytes) *scratch_bytes = (int)at.localSizeBytes;
    if (resident_waves_per_cu) *resident_waves_per_cu = blocks;      // one-wave workgroups
    return AZUL_SUCCESS;
}

int azul_pack_c1(const float *obs_dev, const uint8_t *mask_dev, const uint8_t *player_dev, const int32_t *action_dev, const int32_t *reward_dev,
                 const uint8_t *done_dev, const float *value_dev, const float *logp_dev, const float *entropy_dev, const float *returns_dev,
                 int n_steps, int n_games, uint8_t *records_dev, void *stream)
{
    if (!obs_dev || !mask_dev || !player_dev || !action_dev || !reward_dev || !done_dev || !value_dev || !logp_dev || !entropy_dev || !returns_dev ||
        !records_dev || n_steps < 0 || n_games <= 0 || (uint64_t)n_steps * (uint64_t)n_games * 46u >= (1ull << 32))
        return fail(AZUL_ERR_INVALID, "azul_pack_c1: bad arguments");
    if (((uintptr_t)obs_dev & 15u) != 0 || ((uintptr_t)mask_dev & 3u) != 0 || ((uintptr_t)records_dev & 3u) != 0)
        return fail(AZUL_ERR_INVALID, "azul_pack_c1: obs_dev must be 16-byte aligned, mask_dev and records_dev 4-byte aligned");
    if (n_steps == 0) return AZUL_SUCCESS;
    STREAM_GUARD(stream);
    PackC1Args a = {obs_dev, mask_dev, player_dev, action_dev, reward_dev, done_dev, value_dev, logp_dev, entropy_dev, returns_dev, (u32 *)records_dev,
                    (u32)n_steps * (u32)n_games};
    hipLaunchKernelGGL(azul_pack_c1_kernel, dim3((a.cells * 46u + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_device_clock_probe(uint64_t *out_dev, int spin_iterations, void *stream)
{
    if (!out_dev || spin_iterations <= 0) return fail(AZUL_ERR_INVALID, "azul_device_clock_probe: bad arguments");
    STREAM_GUARD(stream);
    hipLaunchKernelGGL(azul_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (u64 *)out_dev, (u32)spin_iterations);
    HIP_TRY(hipGetLastError());
    return AZUL_SUCCESS;
}

int azul_timing_begin(azul_batch_t *b, void *stream)
{
    BATCH_GUARD(b, stream);
    if (!b) return fail(AZUL_ERR_INVALID, "batch is NULL");
    b->timing = true;
    b->timed_launches = 0;
    b->timed_pairs = 0;
    HIP_TRY(hipEventRecord(b->ev0, (hipStream_t)stream));
    return AZUL_SUCCESS;
}

int azul_timing_end(azul_batch_t *b, void *stream, float *total_ms, int *launches, float *kernel_ms, int *kernel_launches)
{
    BATCH_GUARD(b, stream);
    if (!b || !b->timing) return fail(AZUL_ERR_INVALID, "azul_timing_end without azul_timing_begin");
    HIP_TRY(hipEventRecord(b->ev1, (hipStream_t)stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    if (total_ms) *total_ms = ms;
    if (launches) *launches = b->timed_launches;
    float ksum = 0.f;
    for (int i = 0; i < b->timed_pairs; i++) {
        float k = 0.f;
        HIP_TRY(hipEventElapsedTime(&k, b->lev[2 * i], b->lev[2 * i + 1]));
        ksum += k;
    }
    if (kernel_ms) *kernel_ms = ksum;
    if (kernel_launches) *kernel_launches = b->timed_pairs;
    b->timing = false;
    return AZUL_SUCCESS;
}

int azul_timing_launch_ms(azul_batch_t *b, float *launch_ms, int cap, int *n)
{
    BATCH_GUARD(b, nullptr);
    if (!b || b->timing || cap < 0 || (cap > 0 && !launch_ms)) return fail(AZUL_ERR_INVALID, "azul_timing_launch_ms: after azul_timing_end, with room for `cap` values");
    for (int i = 0; i < b->timed_pairs && i < cap; i++) HIP_TRY(hipEventElapsedTime(launch_ms + i, b->lev[2 * i], b->lev[2 * i + 1]));
    if (n) *n = b->timed_pairs;
    return AZUL_SUCCESS;
}

} // extern "C"
