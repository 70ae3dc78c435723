// simt_selfplay2.cpp -- TEST-ONLY: THE BENCHMARKED KERNEL, azul_selfplay2_kernel (csrc/azul_selfplay_kernels.hpp on csrc/azul_selfplay2.hpp,
// azul_wave.hpp and azul_core.hpp, all UNMODIFIED), compiled by g++ and run lane by lane in lockstep (simt/simt.hpp), one emulated
// workgroup per pair of games with the kernel's own blockIdx -> game placement, so that it can be diffed against the oracle -- and run under
// UBSan / ASan -- in the build container, before a GPU sees it.  (wave_body below restates the kernel's body for ONE purpose: the opt-in
// rotated loop, a compile-time variant of the kernel -- -DAZ2_ROTATED_LOOP, DESIGN.md 3 -- that the default build does not contain.)
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_wave.hpp"
#include "azul_core.hpp"
#include "azul_tables.hpp"
using namespace az;
#include "azul_ops.hpp"
#include "azul_selfplay_kernels.hpp"

struct WaveJob {
    // the batch (BatchDev of the kernel)
    uint8_t *state;          // [N][128]
    u32 *mt;                 // [N][624]
    u32 *mtpos;              // [N]
    const double *T;
    u64 *episodes; u32 *stuck; double *stat_sum;
    u32 n;
    u32 first_player;
    u64 margin;
    // the launch
    int n_steps;
    uint8_t *mask; u32 pitch; u64 *maskbits; i32 *action, *reward; uint8_t *done; u32 *packed; uint8_t *rec;
    int variant;             // 0: OUT 1 / PAD / BITS   1: OUT 1 / PAD   2: OUT 1 / dense / BITS   3: OUT 2 (run-time subset)   4: OUT 0
    int rotated;
    u32 wave_id;
    // "LDS" of the wave
    u32 mt_lds[2][624];
    u32 mtt_lds[2][624];
    double tab_lds[T_WORDS];
    double2 tabfs_lds[T_ROWS * T_BINADES];
};

template <bool LID, int OUT, bool PAD, bool BITS>
static void wave_body(WaveJob *j)
{
    const u32 lane = wv::lane(), l = lane & 31u, half = lane >> 5;
    for (u32 i = lane; i < (u32)T_WORDS; i += 64u) j->tab_lds[i] = j->T[i];
    for (u32 i = lane; i < (u32)(T_ROWS * T_BINADES); i += 64u) j->tabfs_lds[i] = make_double2(j->T[i], j->T[T_ROWS * T_BINADES + i / T_BINADES]);
    az2::lds_sync();
    const u32 gi = j->wave_id * 2u + half;
    if (gi >= j->n) return;                                  // odd batch: the last wave plays one game
    uint8_t *rec = j->state + (size_t)gi * AZUL_RECORD_BYTES;
    az2::K2 k;
    az2::k2_init(k);
    az2::Tab2 tab = {j->tab_lds, j->tab_lds + T_ROWS * T_BINADES, j->tabfs_lds};
    az2::G2 g;
    az2::g2_load(g, rec, l);
    az2::prime2(g, k);
    az2::Rng2 r;
    u32 *gmt = j->mt + (size_t)gi * 624u;
    az2::rng2_open(r, gmt, j->mt_lds[half], j->mtpos[gi], l);
    az2::rng2_attach_tempered(r, j->mtt_lds[half], l);
    az2::Counters2 cnt;
    az2::counters2_open(cnt, j->episodes + gi, j->stuck + gi, j->stat_sum + (size_t)gi * 10, l);
    az2::Out2 o = {j->mask, j->maskbits, j->action, j->reward, j->done, j->packed, j->rec, j->pitch, gi,
                   l == 0u ? (u32 *)j->action : (l == 1u ? (u32 *)j->reward : j->packed)};
    if (!j->rotated) {
        bool dead = false;
        for (int s = 0; s < j->n_steps; s++) {
            if (!dead) az2::selfplay_step2<LID, OUT, PAD, BITS>(g, j->first_player, k, r, tab, j->margin, cnt, o, nullptr, dead);
            o.e += j->n;
        }
    } else {
        az2::Prep2 P;
        az2::prepare2(g, k, r, tab, P);
        for (int s = 0; s < j->n_steps; s++) {
            u32 f = az2::selfplay_rotated2<LID, OUT, PAD, BITS>(g, P, j->first_player, k, r, tab, j->margin, cnt, o, nullptr);
            if (f & 0x100u) break;
            o.e += j->n;
        }
    }
    az2::g2_store(g, rec, l);
    az2::rng2_close(r, gmt, j->mtpos + gi, l);
    az2::counters2_close(cnt, l);
}

struct KernelJob { BatchDev b; TrajArgs t; u32 pitch; int variant; };
template <bool LID>
static void kernel_main_t(void *arg)
{
    KernelJob *j = (KernelJob *)arg;
    switch (j->variant) {
    case 0: azul_selfplay2_kernel<LID, 1, true, true>(j->b, j->t, j->pitch); break;
    case 1: azul_selfplay2_kernel<LID, 1, true, false>(j->b, j->t, j->pitch); break;
    case 2: azul_selfplay2_kernel<LID, 1, false, true>(j->b, j->t, j->pitch); break;
    case 3: azul_selfplay2_kernel<LID, 2, false, false>(j->b, j->t, j->pitch); break;
    default: azul_selfplay2_kernel<LID, 0, false, false>(j->b, j->t, j->pitch); break;
    }
}

// round 1's kernel, one game per wavefront (AZUL_SELFPLAY_KERNEL=1: the A/B partner): OUT 1 writes every stream (dense mask rows), 2 a subset
template <bool LID>
static void kernel1_main_t(void *arg)
{
    KernelJob *j = (KernelJob *)arg;
    switch (j->variant) {
    case 2: azul_selfplay_kernel<LID, 1>(j->b, j->t); break;
    case 3: azul_selfplay_kernel<LID, 2>(j->b, j->t); break;
    default: azul_selfplay_kernel<LID, 0>(j->b, j->t); break;
    }
}

template <bool LID>
static void lane_main_t(void *arg)
{
    WaveJob *j = (WaveJob *)arg;
    switch (j->variant) {
    case 0: wave_body<LID, 1, true, true>(j); break;
    case 1: wave_body<LID, 1, true, false>(j); break;
    case 2: wave_body<LID, 1, false, true>(j); break;
    case 3: wave_body<LID, 2, false, false>(j); break;
    default: wave_body<LID, 0, false, false>(j); break;
    }
}

extern "C" {

// n_games games (records [N][128], MT19937 states [N][624] + positions [N], counters) advance by n_steps moves, two per wave;
// trajectory streams are [n_steps][N]... like the kernel's.  Returns the number of cross-lane operations executed (a size check
// for the test), or a negative number on bad arguments.
long long sh2_selfplay(int n_games, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum, int first_player,
                       int tile_pool, unsigned long long margin, int n_steps, int variant, int rotated, uint8_t *mask, int pitch,
                       u64 *maskbits, i32 *action, i32 *reward, uint8_t *done, u32 *packed, uint8_t *rec)
{
    if (n_games <= 0 || n_steps < 0) return -1;
    static double T[T_WORDS];
    if (!build_sample_tab(T)) return -2;
    long long ops = 0;
    if (rotated != 1) {
        // the kernel itself: one one-wave workgroup per pair of games, blockIdx.x as the launch gives it (the kernel maps it to its games)
        KernelJob kj;
        memset(&kj, 0, sizeof(kj));
        kj.b.state = state; kj.b.mt = mt; kj.b.mtpos = mtpos; kj.b.T = T; kj.b.episodes = episodes; kj.b.stuck = stuck; kj.b.stat_sum = stat_sum;
        kj.b.n = (u32)n_games; kj.b.rules.first_player = (u32)first_player; kj.b.rules.tile_pool = (u32)tile_pool;
        kj.b.draw_margin = margin ? margin : AZ_DRAW_MARGIN;
        kj.t.n_steps = n_steps; kj.t.mask = mask; kj.t.maskbits = maskbits; kj.t.action = action; kj.t.reward = reward; kj.t.done = done;
        kj.t.rec = rec; kj.t.packed = packed;
        kj.pitch = (u32)pitch; kj.variant = variant;
        const bool v1 = rotated == 2;                      // (2: the one-game-per-wave kernel, one workgroup per game)
        const unsigned blocks = v1 ? (unsigned)n_games : ((unsigned)n_games + 1u) / 2u;
        simt::g_grid_dim = {blocks, 1, 1};
        for (unsigned blk = 0; blk < blocks; blk++) {
            simt::g_block_idx = {blk, 0, 0};
            if (v1) ops += (long long)simt::run_workgroup(tile_pool == POOL_LID ? kernel1_main_t<true> : kernel1_main_t<false>, &kj, 1, simt::STACK_BYTES);
            else ops += (long long)simt::run_workgroup(tile_pool == POOL_LID ? kernel_main_t<true> : kernel_main_t<false>, &kj, 1, simt::STACK_BYTES);
        }
        return ops;
    }
    for (u32 w = 0; w < ((u32)n_games + 1u) / 2u; w++) {
        WaveJob *j = (WaveJob *)calloc(1, sizeof(WaveJob));
        j->state = state; j->mt = mt; j->mtpos = mtpos; j->T = T; j->episodes = episodes; j->stuck = stuck; j->stat_sum = stat_sum;
        j->n = (u32)n_games; j->first_player = (u32)first_player; j->margin = margin ? margin : AZ_DRAW_MARGIN;
        j->n_steps = n_steps; j->mask = mask; j->pitch = (u32)pitch; j->maskbits = maskbits; j->action = action; j->reward = reward;
        j->done = done; j->packed = packed; j->rec = rec; j->variant = variant; j->rotated = rotated; j->wave_id = w;
        ops += (long long)simt::run_wave(tile_pool == POOL_LID ? lane_main_t<true> : lane_main_t<false>, j);
        free(j);
    }
    return ops;
}

// self-test of the emulated cross-lane operations against their definitions (lane l holds 100 + l)
int sh2_selftest_result[8];
static void selftest_lane(void *)
{
    const u32 l = wv::lane();
    const u32 v = 100u + l;
    u64 b = __builtin_amdgcn_ballot_w64((l % 3u) == 0u);
    u32 rl = (u32)__builtin_amdgcn_readlane((int)v, 37);
    u32 bp = (u32)__builtin_amdgcn_ds_bpermute((int)(((l + 5u) & 63u) << 2), (int)v);
    u32 shr = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    u32 s = az2::hsum(l < 32u ? l : 2u * l);
    u32 mx = az2::hmax(l ^ 21u);
    bool ok = true;
    u64 want = 0;
    for (u32 i = 0; i < 64u; i += 3u) want |= 1ull << i;
    ok = ok && b == want && rl == 137u && bp == 100u + ((l + 5u) & 63u);
    ok = ok && shr == ((l & 15u) ? 99u + l : 0u);
    ok = ok && s == (l < 32u ? 496u : 3040u);                   // 0 + .. + 31, 2 (32 + .. + 63)
    u32 m = 0;
    for (u32 i = (l & 32u); i < (l & 32u) + 32u; i++) m = (i ^ 21u) > m ? (i ^ 21u) : m;
    ok = ok && mx == m;
    // divergence: only the upper half runs a ballot; then everybody meets again
    u64 inner = 0;
    if (l >= 32u) inner = __builtin_amdgcn_ballot_w64(true);
    u64 all = __builtin_amdgcn_ballot_w64(true);
    ok = ok && (l >= 32u ? inner == 0xffffffff00000000ull : inner == 0) && all == ~0ull;
    // a loop with per-half trip counts
    u32 it = 0, acc = 0;
    while (it < (l >= 32u ? 5u : 2u)) { acc += (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)); it++; }
    ok = ok && acc == (l >= 32u ? 64u + 64u + 32u * 3u : 128u);
    if (!ok) sh2_selftest_result[0] += 1;
}
int sh2_selftest() { memset(sh2_selftest_result, 0, sizeof(sh2_selftest_result)); simt::run_wave(selftest_lane, nullptr); return sh2_selftest_result[0]; }

}
