// simt_selfplay2.cpp -- TEST-ONLY: THE BENCHMARKED KERNEL, azul_selfplay2_kernel (csrc/azul_selfplay_kernels.hpp on csrc/azul_selfplay2.hpp
// and azul_common.hpp, all UNMODIFIED), compiled by g++ and run lane by lane in lockstep (simt/simt.hpp), one emulated workgroup per pair of
// games with the kernel's own blockIdx -> game placement, so that it can be diffed against the oracle -- and run under UBSan / ASan -- in
// the build container, before a GPU sees it.
#define __HIPCC__ 1
#include "azul_hip.h"
#include "azul_common.hpp"
#include "azul_tables.hpp"
using namespace az;
#include "azul_selfplay_kernels.hpp"

struct KernelJob { BatchDev b; TrajArgs t; u32 pitch; int variant; };
template <bool LID>
static void kernel_main_t(void *arg)
{
    KernelJob *j = (KernelJob *)arg;
    if (j->b.move_limit) {                               // (the host's dispatch: a batch with a move limit runs the LIM instantiation)
        switch (j->variant) {
        case 0: azul_selfplay2_kernel<LID, 1, true, true, true>(j->b, j->t, j->pitch); break;
        case 1: azul_selfplay2_kernel<LID, 1, true, false, true>(j->b, j->t, j->pitch); break;
        case 2: azul_selfplay2_kernel<LID, 1, false, true, true>(j->b, j->t, j->pitch); break;
        case 3: azul_selfplay2_kernel<LID, 2, false, false, true>(j->b, j->t, j->pitch); break;
        default: azul_selfplay2_kernel<LID, 0, false, false, true>(j->b, j->t, j->pitch); break;
        }
        return;
    }
    switch (j->variant) {
    case 0: azul_selfplay2_kernel<LID, 1, true, true>(j->b, j->t, j->pitch); break;
    case 1: azul_selfplay2_kernel<LID, 1, true, false>(j->b, j->t, j->pitch); break;
    case 2: azul_selfplay2_kernel<LID, 1, false, true>(j->b, j->t, j->pitch); break;
    case 3: azul_selfplay2_kernel<LID, 2, false, false>(j->b, j->t, j->pitch); break;
    default: azul_selfplay2_kernel<LID, 0, false, false>(j->b, j->t, j->pitch); break;
    }
}

static u32 g_move_limit = 0;       // BatchDev::move_limit of the next launches (azul_batch_set_move_limit)

extern "C" {
void sh2_set_move_limit(unsigned m) { g_move_limit = m; }


// n_games games (records [N][128], MT19937 states [N][624] + positions [N], counters) advance by n_steps moves, two per wave;
// trajectory streams are [n_steps][N]... like the kernel's.  Returns the number of cross-lane operations executed (a size check
// for the test), or a negative number on bad arguments.
long long sh2_selfplay(int n_games, uint8_t *state, u32 *mt, u32 *mtpos, u64 *episodes, u32 *stuck, double *stat_sum, int first_player,
                       int tile_pool, unsigned long long margin, int n_steps, int variant, uint8_t *mask, int pitch,
                       u64 *maskbits, i32 *action, i32 *reward, uint8_t *done, u32 *packed, uint8_t *rec)
{
    if (n_games <= 0 || n_steps < 0) return -1;
    static double T[T_PAIRS * 2];
    if (!build_sample_pairs(T_ROWS, T)) return -2;
    long long ops = 0;
    // the kernel itself: one one-wave workgroup per pair of games, blockIdx.x as the launch gives it (the kernel maps it to its games)
    KernelJob kj;
    memset(&kj, 0, sizeof(kj));
    kj.b.state = state; kj.b.mt = mt; kj.b.mtpos = mtpos; kj.b.tab = (const double2 *)T; kj.b.episodes = episodes; kj.b.stuck = stuck; kj.b.stat_sum = stat_sum;
    kj.b.n = (u32)n_games; kj.b.rules.first_player = (u32)first_player; kj.b.rules.tile_pool = (u32)tile_pool;
    kj.b.draw_margin = margin ? margin : AZ_DRAW_MARGIN;
    kj.b.move_limit = g_move_limit;
    kj.t.n_steps = n_steps; kj.t.mask = mask; kj.t.maskbits = maskbits; kj.t.action = action; kj.t.reward = reward; kj.t.done = done;
    kj.t.rec = rec; kj.t.packed = packed;
    kj.pitch = (u32)pitch; kj.variant = variant;
    const unsigned blocks = ((unsigned)n_games + 1u) / 2u;
    simt::g_grid_dim = {blocks, 1, 1};
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(tile_pool == POOL_LID ? kernel_main_t<true> : kernel_main_t<false>, &kj, 1, simt::STACK_BYTES);
    }
    return ops;
}

// azul_pack_c1_kernel (the C1 wire record of the opt-in trajectory all-gather), four emulated waves per workgroup
static void pack_main(void *arg) { azul_pack_c1_kernel(*(PackC1Args *)arg); }
long long sh2_pack_c1(const float *obs, const uint8_t *mask, const uint8_t *player, const i32 *action, const i32 *reward, const uint8_t *done,
                      const float *value, const float *logp, const float *entropy, const float *returns, int cells, uint8_t *out)
{
    PackC1Args a = {obs, mask, player, action, reward, done, value, logp, entropy, returns, (u32 *)out, (u32)cells};
    const unsigned blocks = ((unsigned)cells * 46u + 255u) / 256u;
    simt::g_grid_dim = {blocks, 1, 1};
    long long ops = 0;
    for (unsigned blk = 0; blk < blocks; blk++) {
        simt::g_block_idx = {blk, 0, 0};
        ops += (long long)simt::run_workgroup(pack_main, &a, 4, simt::STACK_BYTES) + 1;
    }
    return ops;
}

// azul_clock_probe_kernel: runs to completion and writes its three words (the emulation's timers read 0)
static void probe_main(void *arg) { azul_clock_probe_kernel((u64 *)arg, 50u); }
int sh2_clock_probe(u64 *out3)
{
    simt::g_grid_dim = {1, 1, 1};
    simt::g_block_idx = {0, 0, 0};
    simt::run_workgroup(probe_main, out3, 1, simt::STACK_BYTES);
    return 0;
}

// az2::sample_slow2 (the boundary search of RandomAgent's draw: bisect_right over the cumulative weights) on n points, 64 per emulated wave
struct SampleJob { const double *T; const double *x; const int *J; const int *M; int *out; int base, n; };
static void sample_lane(void *arg)
{
    SampleJob *j = (SampleJob *)arg;
    const int i = j->base + (int)wv::lane();
    if (i >= j->n) return;
    az2::Tab2 T = {(const double2 *)j->T};
    const u32 J = (u32)j->J[i], M = (u32)j->M[i];
    j->out[i] = (int)az2::sample_slow2(T, j->x[i], T.fs[8u * J].y, J, M, J + M);
}
int sh2_sample_slow(int n, const double *x, const int *J, const int *M, int *out)
{
    static double T[T_PAIRS * 2];
    if (!build_sample_pairs(T_ROWS, T)) return -2;
    SampleJob j = {T, x, J, M, out, 0, n};
    for (j.base = 0; j.base < n; j.base += 64) simt::run_wave(sample_lane, &j);
    return 0;
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
