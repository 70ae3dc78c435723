// azul_ops.hpp -- the body of the two-player rule kernel (azul_op_kernel: one rule call per game and launch, one game per wavefront) and
// the env moves it shares with the one-game-per-wave policy rollout, as a header: azul_kernels.hip wraps op_body in the __global__
// function, and tests/hostcheck/hostcheck.cpp runs THIS FILE, unmodified, on the CPU behind the 64-lane vector emulation of
// azul_wave.hpp (hc_op) -- the facade's emulated device in the CPU suite and under ASan / UBSan is the product's own dispatch.
// Reference lines: azulnet/azul.py (every rule method), azulnet/game_runner.py:23-97.
#pragma once

struct BatchDev {
    uint8_t *state;      // [N][128]
    u32 *mt;             // [N][624]
    u32 *mtpos;          // [N]
    const double *T;     // SampleTab source: Fr[31][8] then S[31]
    u64 *episodes;       // [N]
    u32 *stuck;          // [N]
    double *stat_sum;    // [N][10]
    u32 n;
    Rules rules;
    u64 draw_margin;     // AZ_DRAW_MARGIN; tests widen it to force the literal fp64 factory draw
    u64 *prof;           // [SEG_COUNT] segment cycle sums (only written by the -DAZ_PROFILE_SEGMENTS diagnostic build)
    u32 id_base;         // global id of game 0 (azul_batch_set_id_base): keys the policy sampler's Philox stream
};

enum {
    OP_QUERY = 0, OP_INIT, OP_NEW_ROUND, OP_MOVE, OP_NEXT_PLAYER, OP_COUNT_SCORE, OP_STEP,
    OP_RUNNER_INIT, OP_RUNNER_RESET, OP_RUNNER_STEP, OP_RANDOM_ACTION, OP_SAMPLE_MASK, OP_POLICY_STEP, OP_AGENT_STEP
};

struct OpArgs {
    int op;
    const i32 *actions;      // [N]   in  (MOVE / STEP / RUNNER_STEP)
    const uint8_t *active;   // [N]   in, optional
    const uint8_t *mask_in;  // [N][180] in (SAMPLE_MASK)
    i32 *actions_out;        // [N]   out (RANDOM_ACTION)
    uint8_t *status;         // [N]   out
    i32 *reward;             // [N]   out
    uint8_t *done;           // [N]   out
    uint8_t *mask;           // [N][180] out (after the op)
    float *obs;              // [N][136] out (after the op)
    int persp;
    uint8_t *flags;          // [N]   out
    i32 *potential;          // [N]   out
    double *stats;           // [N][10] out
    uint8_t *player;         // [N]   out: current_player after the op
    uint8_t *rng_dirty;      // [N]   out: the op regenerated the game's 624 MT19937 words (Rng::dirty; 0 for ops that do not draw)
    uint8_t *rec_out;        // [N][record bytes] out: the game's record after the op
    u32 *pos_out;            // [N]   out: index of the game's MT19937 stream after the op
    i32 *next_action;        // [N]   out: RandomAgent's choice on the state after the op, drawn at the stream's index after the op WITHOUT moving it
                             //       (-1: nothing legal, -2: not available -- the op failed / did not draw, or the draw would cross a regeneration)
    u32 pos_set;             // 0, or 1 + the stream index to install before the op (single-game calls: the host's index is the authority)
    u32 first;               // the launch covers games first .. first + grid - 1; row i of the arrays above belongs to game first + i
};

AZ_FN bool op_needs_rng(int op)
{
    return op == OP_INIT || op == OP_NEW_ROUND || op == OP_STEP || op == OP_RUNNER_INIT || op == OP_RUNNER_RESET ||
           op == OP_RUNNER_STEP || op == OP_RANDOM_ACTION || op == OP_SAMPLE_MASK || op == OP_POLICY_STEP || op == OP_AGENT_STEP;
}

// One env move of policy-driven self-play on a primed, register-resident game: Azul.step (azul.py:296-313) for the current
// player, the shaped reward of game_runner.py:48-52 per move, done, statistics and the auto-reset of game_runner.py:76-82.
// Shared by the per-call kernel (OP_POLICY_STEP) and the persistent policy-rollout kernel.
template <bool LID>
AZ_FN u32 env_policy_step(Game &g, const LaneConst &k, Rng &r, const BatchDev &b, u32 gi, i32 av, i32 &rew, u32 &dn,
                                               bool &dirty_state)
{
    rew = 0;
    dn = 0;
    u32 st = ST_OK;
    bool stuck = false;
    if (av < 0 && !g.eog) {                      // "no action": legitimate only when nothing is legal (hazard H3)
        Mask m;
        legal_mask(g, k, m);
        stuck = mask_count(m) == 0u;
    }
    if (stuck) {
        AZ_LANE0(b.stuck[gi] += 1u);
        dn = 2u;
        st = episode_reset<LID>(g, b.rules.first_player, r);
        game_prime<LID>(g, k);
        if (!st) st = ST_STUCK;
        dirty_state = true;
        return st;
    }
    st = checked_step<LID>(g, k, r, av);
    dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
    if (dirty_state) {
        g.moves += 1u;
        i32 phi = potential<LID>(g, k);
        rew = phi - g.pscore;
        g.pscore = phi;
        dn = is_end_of_game(g) ? 1u : 0u;
        if (dn && st == ST_OK) {
            for (u32 q = 0; q < 10u; q++) { double sv = game_stat(g, q); AZ_LANE0(b.stat_sum[(size_t)gi * 10 + q] += sv); }
            AZ_LANE0(b.episodes[gi] += 1ull);
            st = episode_reset<LID>(g, b.rules.first_player, r);
            game_prime<LID>(g, k);
        }
    } else if (st == ST_GAME_ENDED) {
        // a finished game handed in (e.g. after set_state): restart the slot, report done
        dn = 1u;
        st = episode_reset<LID>(g, b.rules.first_player, r);
        game_prime<LID>(g, k);
        dirty_state = true;
    }
    return st;
}

// One AGENT step of NNRunner.run_episode (nn_runner.py:24-29): GameRunner.step -- the agent's move, the opponent's RandomAgent
// replies, the shaped reward, done (game_runner.py:43-55) -- and, when the episode ends, the GameRunner.reset() that opens the
// next run_episode (nn_runner.py:20 -> game_runner.py:76-82, incl. the opponent's opening moves), so the observation / mask
// taken afterwards are the next decision's.
template <bool LID>
AZ_FN u32 env_agent_step(Game &g, const LaneConst &k, Rng &r, const SampleTab &tab, const BatchDev &b, u32 gi, i32 av,
                                              i32 &rew, u32 &dn, bool &dirty_state)
{
    rew = 0;
    dn = 0;
    u32 st = runner_step<LID>(g, k, r, tab, av, rew, dn);
    dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_BAD_ACTION);
    if (st == ST_STUCK) { AZ_LANE0(b.stuck[gi] += 1u); dn = 2u; rew = 0; }       // hazard H3: nobody can move
    else if (st == ST_GAME_ENDED) dn = 1u;       // a finished game handed in: restart the slot, report done
    else if (st == ST_OK && dn) {
        for (u32 q = 0; q < 10u; q++) { double sv = game_stat(g, q); AZ_LANE0(b.stat_sum[(size_t)gi * 10 + q] += sv); }
        AZ_LANE0(b.episodes[gi] += 1ull);
    }
    if (dirty_state && dn) {
        u32 st2 = episode_reset<LID>(g, b.rules.first_player, r);
        game_prime<LID>(g, k);
        if (!st2) st2 = runner_opponent_loop<LID>(g, k, r, tab, true);
        if (st == ST_OK) st = st2;
    }
    return st;
}

// One rule call on game `oi` of the launch (row oi of the caller's arrays, game a.first + oi of the batch): the op, then the queries
// on the state it leaves.  mt_lds: 624 words of LDS for the game's MT19937 state; fr_lds: the sampler's table rows.
template <bool LID>
AZ_FN void op_body(const BatchDev &b, const OpArgs &a, u32 oi, u32 *mt_lds, double *fr_lds)
{
    const u32 gi = oi + a.first;                     // game of the batch / row of the caller's arrays
    const bool act = a.active ? (a.active[oi] != 0) : true;
    uint8_t *rec = b.state + (size_t)gi * AZUL_RECORD_BYTES;
    LaneConst k;
    lane_consts(k);
    SampleTab tab;
    sample_tab_load(tab, b.T, fr_lds);
    Game g;
    game_load(g, rec);
    game_prime<LID>(g, k);
    u32 st = ST_OK, rdirty = 0;
    i32 spec = -2;
    if (act && a.op != OP_QUERY) {
        Rng r;
        const bool use_rng = op_needs_rng(a.op);
        // the 2.5 KB MT19937 state is staged into LDS lazily, by the first draw (most single steps never draw)
        rng_attach(r, b.mt + (size_t)gi * 624u, mt_lds, use_rng ? (a.pos_set ? a.pos_set - 1u : b.mtpos[gi]) : 0u);
        r.margin = b.draw_margin;
        bool dirty_state = true;
        switch (a.op) {
        case OP_INIT:
            game_ctor<LID>(g, b.rules.first_player, r);
            break;
        case OP_NEW_ROUND:
            st = new_round<LID>(g, r);
            break;
        case OP_MOVE: {
            i32 av = a.actions[oi];
            if (av < 0 || av >= 180) { st = ST_BAD_ACTION; dirty_state = false; break; }
            do_move<LID>(g, action_code((u32)av));
        } break;
        case OP_NEXT_PLAYER:
            g.cur = (g.cur < 2u) ? g.cur + 1u : 1u;
            break;
        case OP_COUNT_SCORE:
            count_score<LID>(g, k);
            break;
        case OP_STEP:
            st = checked_step<LID>(g, k, r, a.actions[oi]);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
            break;
        case OP_RUNNER_INIT:
            st = episode_reset<LID>(g, b.rules.first_player, r);
            break;
        case OP_RUNNER_RESET:
            st = episode_reset<LID>(g, b.rules.first_player, r);
            if (!st) st = runner_opponent_loop<LID>(g, k, r, tab, true);
            break;
        case OP_RUNNER_STEP: {
            i32 rew = 0;
            u32 dn = 0;
            st = runner_step<LID>(g, k, r, tab, a.actions[oi], rew, dn);
            dirty_state = !(st == ST_ILLEGAL_MOVE || st == ST_GAME_ENDED || st == ST_BAD_ACTION);
            if (a.reward) AZ_LANE0(a.reward[oi] = rew);
            if (a.done) AZ_LANE0(a.done[oi] = (uint8_t)dn);
            if (!st && dn) {
                for (u32 q = 0; q < 10u; q++) { double sv = game_stat(g, q); AZ_LANE0(b.stat_sum[(size_t)gi * 10 + q] += sv); }
                AZ_LANE0(b.episodes[gi] += 1ull);
            }
            if (st == ST_STUCK) AZ_LANE0(b.stuck[gi] += 1u);
        } break;
        case OP_RANDOM_ACTION: {
            Mask m;
            legal_mask(g, k, m);
            u32 code;
            i32 av = random_agent(m, r, tab, k, code);
            AZ_LANE0(a.actions_out[oi] = av);
            dirty_state = false;
        } break;
        case OP_POLICY_STEP: {
            i32 rew = 0;
            u32 dn = 0;
            st = env_policy_step<LID>(g, k, r, b, gi, a.actions[oi], rew, dn, dirty_state);
            if (a.reward) AZ_LANE0(a.reward[oi] = rew);
            if (a.done) AZ_LANE0(a.done[oi] = (uint8_t)dn);
        } break;
        case OP_AGENT_STEP: {
            i32 rew = 0;
            u32 dn = 0;
            st = env_agent_step<LID>(g, k, r, tab, b, gi, a.actions[oi], rew, dn, dirty_state);
            if (a.reward) AZ_LANE0(a.reward[oi] = rew);
            if (a.done) AZ_LANE0(a.done[oi] = (uint8_t)dn);
        } break;
        case OP_SAMPLE_MASK: {
            const uint8_t *mi = a.mask_in + (size_t)oi * AZUL_NUM_ACTIONS;
            vu32 l = lane();
            Mask m;
            m.b0 = sel(ld_u8(mi, l, l < 64u) != 0u, splat(1u), splat(0u));
            m.b1 = sel(ld_u8(mi, l + 64u, l < 64u) != 0u, splat(1u), splat(0u));
            m.b2 = sel(ld_u8(mi, l + 128u, l < 52u) != 0u, splat(1u), splat(0u));
            m.m0 = ballot(m.b0 != 0u); m.m1 = ballot(m.b1 != 0u); m.m2 = ballot(m.b2 != 0u);
            u32 code;
            i32 av = random_agent(m, r, tab, k, code);
            AZ_LANE0(a.actions_out[oi] = av);
            dirty_state = false;
        } break;
        default:
            dirty_state = false;
            break;
        }
        if (dirty_state) game_store(g, rec);
        if (a.next_action && use_rng && st == ST_OK && r.pos + 2u <= 624u) {
            // the question a GameRunner loop asks next (nn_runner.py:22-30 with RandomAgent: get_valid_moves -> get_a_output): answered
            // here from the two words the stream would hand out next, index restored -- the caller advances it when it plays the answer
            const u32 keep = r.pos;
            Mask m;
            legal_mask(g, k, m);
            u32 code;
            spec = random_agent(m, r, tab, k, code);
            r.pos = keep;
        }
        if (use_rng) rng_close(r, b.mtpos + gi);
        rdirty = r.dirty & 1u;
    }
    if (a.next_action) AZ_LANE0(a.next_action[oi] = spec);
    if (a.rng_dirty) AZ_LANE0(a.rng_dirty[oi] = (uint8_t)rdirty);
    if (a.status && act) AZ_LANE0(a.status[oi] = (uint8_t)st);
    if (a.rec_out) game_store(g, a.rec_out + (size_t)oi * AZUL_RECORD_BYTES);
    if (a.pos_out) AZ_LANE0(a.pos_out[oi] = b.mtpos[gi]);      // (written by rng_close above when the op drew)
    // queries on the post-op state
    if (a.mask) {
        Mask m;
        legal_mask(g, k, m);
        mask_write(m, a.mask + (size_t)oi * AZUL_NUM_ACTIONS);
    }
    if (a.obs) {
        u32 p = (a.persp == AZUL_PERSP_CURRENT) ? me_index(g) : (u32)a.persp;
        observe(g, p, a.obs + (size_t)oi * AZUL_OBS_SIZE);
    }
    if (a.flags) {
        u32 f = (sources_board(g) == 0u ? AZUL_FLAG_END_OF_ROUND : 0) | (is_end_of_game(g) ? AZUL_FLAG_END_OF_GAME : 0) |
                (g.eog ? AZUL_FLAG_ENDED_FLAG : 0);
        AZ_LANE0(a.flags[oi] = (uint8_t)f);
    }
    if (a.potential) {
        i32 phi = potential<LID>(g, k);
        AZ_LANE0(a.potential[oi] = phi);
    }
    if (a.stats) {
        for (u32 q = 0; q < 10u; q++) { double sv = game_stat(g, q); AZ_LANE0(a.stats[(size_t)oi * 10 + q] = sv); }
    }
    if (a.player) AZ_LANE0(a.player[oi] = (uint8_t)g.cur);
}
