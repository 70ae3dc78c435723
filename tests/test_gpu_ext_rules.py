"""Row N4's extended rules on the GPU -- BEYOND THE REFERENCE, PARITY UNPINNED: 2P+1 factory displays, end-of-game bonuses, the short deal
and the finite bag as default-off flags of azul_batch_create_rules (include/azul_hip.h AZUL_RULE_*; kernels csrc/azul_rules_x.hpp, two
games per wavefront).  The reference implements none of them (azulnet/azul.py:19, 72, 86, 266-288; tests/test_azul.py:14), so the
comparison is the C oracle's OZ_EXT_* restatement of the rulebook, itself cross-checked by the independent Python model of
tests/ext_rules_model.py (tests/test_ext_rules_model.py).

  * every flag combination x P = 2, 3, 4 x three rule sets, 32+ streams each: masks, actions, done flags, compact records, bit-packed
    masks, 256-byte record snapshots, final records, all 624 MT19937 words + positions, episode / stuck counters, statistics sums;
  * tile conservation (100) as a property of every snapshot with a tracked pool;
  * the rule entries one call at a time (init / new_round / move / next_player / count_score / step, sampler, mask, observation, flags,
    statistics) against the oracle;
  * flags off is byte-identical to the batches of azul_batch_create_players / azul_batch_create (their own tests run unchanged;
    here: the two constructors give the same bytes);
  * API: sizes, refusals, rule errors."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as oz

pytestmark = pytest.mark.gpu

RULESETS = [(oz.FIRST_RANDOM, oz.POOL_LID), (1, oz.POOL_RANDOM), (2, oz.POOL_LID)]
FLAGSETS = [oz.EXT_DISPLAYS_2P1, oz.EXT_END_BONUS, oz.EXT_SHORT_DEAL, oz.EXT_FINITE_BAG,
            oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL,
            oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL | oz.EXT_FINITE_BAG]


def _rules(first, pool):
    return {"first_player": "Random" if first == oz.FIRST_RANDOM else first, "tile_pool": "Lid" if pool == oz.POOL_LID else "Random"}


def _conserved(recs, P, D):
    """bag + lid + displays + centre + pattern lines + walls == 100 for every record (uint8 [..., 256] viewed as the wide dtype)."""
    r = recs
    walls = r["walls"][..., :P].astype(np.uint32)
    on_walls = sum(((walls >> b) & 1) for b in range(25)).sum(axis=-1)
    tiles = (r["box"].sum(axis=-1).astype(int) + r["lid"].sum(axis=-1) + r["displays"].sum(axis=(-1, -2)) + r["xdisplays"].sum(axis=(-1, -2))
             + r["center"][..., :5].sum(axis=-1) + r["pattern_lines"][..., :P, :, :].sum(axis=(-1, -2, -3)) + on_walls)
    return tiles


@pytest.mark.parametrize("players", [2, 3, 4])
@pytest.mark.parametrize("ext", FLAGSETS)
def test_beyond_the_reference_parity_unpinned_selfplay_equals_the_oracle(players, ext):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    P = players
    D = 2 * P + 1 if ext & oz.EXT_DISPLAYS_2P1 else 5
    NA = (D + 1) * 30
    n, T = 32, 300
    if P == 4 and (ext & oz.EXT_DISPLAYS_2P1):
        T = 700                                              # nine displays drain bag and lid: short deals / BOX_EMPTY happen
    short_deals = box_empty = 0
    for (first, pool) in RULESETS:
        if pool == oz.POOL_LID and ext & oz.EXT_FINITE_BAG:
            continue
        if first > P:
            continue
        for variant in ("records", "full", "none"):
            env = BatchedAzul(n, rules=_rules(first, pool), players=P, ext_rules=ext, device="cuda:0")
            assert env.num_actions == NA and env.displays == D and env.record_dtype.itemsize == 256
            seed0 = 5000 + 100 * ext
            env.seed(seed0)
            env.init()
            assert (env.new_round().cpu().numpy() == 0).all()
            if variant == "records":
                tr = env.alloc_trajectory(T, with_records=True)
                env.selfplay(T, tr["mask"], tr["action"], tr["reward"], tr["done"], records=tr["records"])
            elif variant == "full":
                tr = env.alloc_trajectory(T, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[D])
                env.selfplay(T, tr["mask"], tr["action"], tr["reward"], tr["done"], maskbits=tr["maskbits"], packed=tr["packed"])
            else:
                env.selfplay(T // 2)
                env.selfplay(T - T // 2)
            torch.cuda.synchronize()
            recs = env.get_records()
            cnt = env.counters()
            mts, poss = env.get_rng_range()
            if variant != "none":
                mask_all, act_all, done_all = tr["mask"].cpu().numpy(), tr["action"].cpu().numpy(), tr["done"].cpu().numpy()
                rew_all = tr["reward"].cpu().numpy()
            for g in range(n):
                tag = (P, ext, first, pool, variant, g)
                s = oz.StreamX(seed0 + g, P, first_player=first, tile_pool=pool, ext=ext)
                try:
                    o = s.advance(T)
                except RuntimeError:
                    # bag and lid ran dry without the short-deal rule: the oracle stops (OZ_BOX_EMPTY, where the reference raises); the device
                    # game stays as it was.  Compare the moves before it.
                    assert not ext & oz.EXT_SHORT_DEAL, tag
                    box_empty += 1
                    s2 = oz.StreamX(seed0 + g, P, first_player=first, tile_pool=pool, ext=ext)
                    ok = 0
                    while True:
                        try:
                            o1 = s2.advance(1)
                        except RuntimeError:
                            break
                        if variant != "none":
                            assert np.array_equal(mask_all[ok, g], o1["mask"][0]) and act_all[ok, g] == o1["action"][0], tag + (ok,)
                        ok += 1
                    assert ok < T
                    continue
                if variant != "none":
                    assert np.array_equal(mask_all[:, g], o["mask"]), tag
                    assert np.array_equal(act_all[:, g], o["action"]) and np.array_equal(done_all[:, g], o["done"]), tag
                    assert not rew_all[:, g].any(), tag          # GameRunner's shaped reward is two-player (game_runner.py:50)
                if variant == "records":
                    got = tr["records"][:, g].cpu().numpy()
                    assert got.tobytes() == o["rec_after"].tobytes(), tag
                    if pool == oz.POOL_LID or ext & oz.EXT_FINITE_BAG:
                        assert (_conserved(o["rec_after"], P, D) == 100).all(), tag
                        r = o["rec_after"]
                        fresh = (r["center"][:, 5] == 1) & (r["center"][:, :5].sum(axis=1) == 0)
                        dealt = r["displays"].sum(axis=(1, 2)) + r["xdisplays"].sum(axis=(1, 2))
                        short_deals += int((fresh & (dealt < 4 * D) & (o["done"] == 0)).sum())
                if variant == "full":
                    p = tr["packed"][:, g].cpu().numpy().view(np.uint32)
                    a = np.where((p & 0xFF) == 0xFF, (p >> 16).astype(np.int32), (p & 0xFF).astype(np.int32))
                    a[a == 0xFFFF] = -1
                    assert np.array_equal(a, o["action"]) and np.array_equal((p >> 8) & 0xFF, o["done"]), tag
                    bits = tr["maskbits"][:, g].cpu().numpy().view(np.uint8).reshape(T, -1)[:, :(NA + 7) // 8]
                    assert np.array_equal(bits, np.packbits(o["mask"].astype(bool), axis=1, bitorder="little")), tag
                assert recs[g].tobytes() == s.record().tobytes(), tag
                assert int(poss[g]) == s.rng_state()[1] and np.array_equal(mts[g], s.rng_state()[0]), tag
                assert int(cnt["episodes"][g]) == int(s.episodes.value) and int(cnt["stuck"][g]) == int(s.stuck.value), tag
                assert np.allclose(cnt["stat_sums"][g], s.stats_sum, rtol=0, atol=1e-9), tag
    if ext & oz.EXT_SHORT_DEAL:
        assert box_empty == 0
        if P == 4 and ext & oz.EXT_DISPLAYS_2P1:
            assert short_deals > 0


@pytest.mark.parametrize("players", [2, 3, 4])
@pytest.mark.parametrize("ext", [oz.EXT_END_BONUS, oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL | oz.EXT_FINITE_BAG])
def test_beyond_the_reference_parity_unpinned_rule_entries_against_the_oracle(players, ext):
    """init / new_round / move / next_player / count_score / step one call at a time, the RandomAgent draw on the game's own mask and on
    a caller's mask, masks, observations from every perspective, flags and statistics, for 48 games driven by the sampler's picks."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    Lz = oz.lib()
    P, n = players, 48
    pool = oz.POOL_RANDOM if ext & oz.EXT_FINITE_BAG else oz.POOL_LID
    env = BatchedAzul(n, rules=_rules(oz.FIRST_RANDOM, pool), players=P, ext_rules=ext)
    NA, NOBS = env.num_actions, env.obs_size
    env.seed(7100)
    env.init()
    env.new_round()
    rngs = [oz.seeded_rng(7100 + g) for g in range(n)]
    games = [oz.Game() for _ in range(n)]
    for g in range(n):
        assert Lz.oz_init_ext(C.byref(games[g]), P, 0, pool, ext, C.byref(rngs[g])) == 0
        assert Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g])) == 0
        assert Lz.oz_num_actions(C.byref(games[g])) == NA and Lz.oz_obs_size(C.byref(games[g])) == NOBS
    for it in range(150):
        recs = env.get_records()
        for g in range(n):
            assert recs[g].tobytes() == oz.pack_np(games[g]).tobytes(), (it, g)
        mask = env.get_valid_moves().cpu().numpy()
        for g in range(n):
            assert np.array_equal(mask[g], oz.check_all_valid_x(games[g])), (it, g)
        if it % 5 == 0:
            for persp in list(range(P)) + [L.PERSP_MOVER]:
                obs = env.get_state(persp).cpu().numpy()
                for g in range(0, n, 7):
                    pp = persp if persp < P else (games[g].current_player - 1) % P
                    assert np.array_equal(obs[g].astype(np.int64), oz.get_state_x(games[g], pp)), (it, g, persp)
        # the sampler on the game's own mask, or on the same mask handed in
        a = (env.random_action() if it % 2 == 0 else env.sample_mask(mask.astype(np.uint8))).cpu().numpy()
        for g in range(n):
            m = np.ascontiguousarray(mask[g].astype(np.uint8))
            want = Lz.oz_random_agent_x(m.ctypes.data_as(C.POINTER(C.c_uint8)), NA, C.byref(rngs[g]))
            assert a[g] == want, (it, g)
        S = NA // 30
        playable = a >= 0
        act = np.where(playable, a, 0).astype(np.int32)
        ended = np.array([bool(games[g].end_of_game) for g in range(n)])
        sel = playable & ~ended
        if it % 3 == 0:
            # the unchecked single methods instead of step
            env.move(act, active=sel.astype(np.uint8))
            for g in np.flatnonzero(sel):
                Lz.oz_move(C.byref(games[g]), int(act[g]) % S, (int(act[g]) // S) % 5, int(act[g]) // (5 * S))
            flags = env.flags().cpu().numpy()
            eor = np.array([bool(Lz.oz_is_end_of_round(C.byref(games[g]))) for g in range(n)])
            assert np.array_equal((flags & L.FLAG_END_OF_ROUND) != 0, eor)
            env.count_score(active=(sel & eor).astype(np.uint8))
            env.next_player(active=(sel & ~eor).astype(np.uint8))
            deal = np.zeros(n, dtype=bool)
            for g in np.flatnonzero(sel):
                if eor[g]:
                    Lz.oz_count_score(C.byref(games[g]))
                    if not Lz.oz_is_end_of_game(C.byref(games[g])):
                        deal[g] = True
                else:
                    Lz.oz_next_player(C.byref(games[g]))
            st = env.new_round(active=deal.astype(np.uint8)).cpu().numpy()
            for g in np.flatnonzero(deal):
                assert st[g] == Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g])), (it, g)
            # games whose walls say "over" after count_score: restart them on both sides (the flag-setting path is step's)
            over = np.array([bool(Lz.oz_is_end_of_game(C.byref(games[g]))) for g in range(n)]) & sel & eor
            assert np.array_equal(((env.flags().cpu().numpy() & L.FLAG_END_OF_GAME) != 0)[sel & eor], over[sel & eor])
            env.init(active=over.astype(np.uint8))
            st = env.new_round(active=over.astype(np.uint8)).cpu().numpy()
            for g in np.flatnonzero(over):
                assert Lz.oz_init_ext(C.byref(games[g]), P, 0, pool, ext, C.byref(rngs[g])) == 0
                assert st[g] == Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g]))
        else:
            st = env.azul_step(act, active=sel.astype(np.uint8)).cpu().numpy()
            for g in np.flatnonzero(sel):
                want = Lz.oz_step(C.byref(games[g]), int(act[g]) % S, (int(act[g]) // S) % 5, int(act[g]) // (5 * S), C.byref(rngs[g]))
                assert st[g] == want, (it, g)
            fin = np.array([bool(games[g].end_of_game) for g in range(n)])
            if fin.any():
                stats = env.statistics().cpu().numpy()
                for g in np.flatnonzero(fin):
                    assert np.allclose(stats[g], list(oz.get_statistics(games[g]).values()), rtol=0, atol=1e-12), (it, g)
                assert (env.azul_step(np.zeros(n, np.int32), active=fin.astype(np.uint8)).cpu().numpy()[fin] == L.GAME_ENDED).all()
                env.init(active=fin.astype(np.uint8))
                st = env.new_round(active=fin.astype(np.uint8)).cpu().numpy()
                for g in np.flatnonzero(fin):
                    assert Lz.oz_init_ext(C.byref(games[g]), P, 0, pool, ext, C.byref(rngs[g])) == 0
                    assert st[g] == Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g]))
        # stuck games (nothing legal): restart on both sides
        stuck = ~playable & ~ended
        if stuck.any():
            env.init(active=stuck.astype(np.uint8))
            env.new_round(active=stuck.astype(np.uint8))
            for g in np.flatnonzero(stuck):
                assert Lz.oz_init_ext(C.byref(games[g]), P, 0, pool, ext, C.byref(rngs[g])) == 0
                Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g]))
        mts, poss = env.get_rng_range()
        for g in range(0, n, 5):
            assert int(poss[g]) == rngs[g].idx and np.array_equal(mts[g], np.ctypeslib.as_array(rngs[g].mt)), (it, g)


def test_flags_off_is_the_players_batch_byte_for_byte():
    """azul_batch_create_rules(..., 0) IS azul_batch_create_players: same kernels, same bytes (whose reference-pinned tests are
    tests/test_gpu_players.py); two players without flags stay on the 128-byte record and the two-player kernels."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    for P in (3, 4):
        a = BatchedAzul(16, players=P)
        b = BatchedAzul(16, players=P, ext_rules=0, rules={"first_player": "Random", "tile_pool": "Lid", "displays": 5, "bonuses": "round"})
        for env in (a, b):
            env.seed(3)
            env.init()
            env.new_round()
            env.selfplay(200)
        assert a.get_records().tobytes() == b.get_records().tobytes() and L.lib.azul_batch_rule_flags(b._h) == 0
    two = BatchedAzul(4)
    assert two.record_dtype.itemsize == 128 and two.num_actions == 180 and two.obs_size == 136 and not two.wide
    flagged = BatchedAzul(4, ext_rules=L.RULE_END_BONUS)
    assert flagged.record_dtype.itemsize == 256 and flagged.wide and L.lib.azul_batch_rule_flags(flagged._h) == L.RULE_END_BONUS


def test_sizes_refusals_and_rule_errors():
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    from azul_deep_reinforcement_learning_amd.batch import IllegalRule
    for P in (2, 3, 4):
        env = BatchedAzul(4, players=P, rules={"displays": "2P+1"})
        D = 2 * P + 1
        assert (env.displays, env.num_actions, env.obs_size) == (D, (D + 1) * 30, 5 * D + 6 + 52 * P + 1)
        assert L.lib.azul_batch_num_actions(env._h) == env.num_actions and L.lib.azul_batch_obs_size(env._h) == env.obs_size
    env = BatchedAzul(4, players=3, rules={"displays": "2P+1", "bonuses": "end"})
    env.seed(1)
    env.init()
    env.new_round()
    # GameRunner's step / reset / what-if potential and the policy entries are two-player, 180 actions (game_runner.py:50)
    for call in (lambda: env.step(np.zeros(4, np.int32)), lambda: env.reset(), lambda: env.runner_init(), lambda: env.score_preview()):
        with pytest.raises(L.AzulHipError):
            call()
    obs = torch.zeros((4, env.obs_size), device=env.device)
    act = torch.zeros(4, dtype=torch.int32, device=env.device)
    u8 = torch.zeros(4, dtype=torch.uint8, device=env.device)
    msk = torch.zeros((4, env.num_actions), dtype=torch.uint8, device=env.device)
    with pytest.raises(L.AzulHipError):
        env.policy_step(act, act.clone(), u8, u8.clone(), obs, msk, u8.clone())
    # a mask row pitch smaller than the action space is refused; an action outside it is BAD_ACTION
    tr = env.alloc_trajectory(4)
    with pytest.raises(L.AzulHipError):
        L.check(L.lib.azul_batch_selfplay_strided(env._h, 4, C.c_void_p(tr["mask"].data_ptr()), 180, None, None, None, None, None, None, env._stream()))
    st = env.azul_step(np.full(4, env.num_actions, np.int32)).cpu().numpy()
    assert (st == L.BAD_ACTION).all()
    # rule errors
    h = C.c_void_p()
    assert L.lib.azul_batch_create_rules(C.byref(h), 4, 2, 1, L.POOL_LID, L.RULE_FINITE_BAG) == L.ERR_RULE
    assert L.lib.azul_batch_create_rules(C.byref(h), 4, 2, 1, L.POOL_LID, 16) == L.ERR_RULE
    with pytest.raises(IllegalRule):
        BatchedAzul(4, rules={"tile_pool": "Lid", "finite_bag": True})
    with pytest.raises(IllegalRule):
        BatchedAzul(4, rules={"displays": 6})
    with pytest.raises(IllegalRule):
        BatchedAzul(4, rules={"bonuses": "never"})
    # a record of another display count is refused
    rec = env.get_records()
    rec["n_displays"][0] = 0
    with pytest.raises(L.AzulHipError):
        env.set_records(rec)


@pytest.mark.parametrize("players,ext", [(3, 0), (4, oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL)])
def test_full_size_batches_are_shard_invariant_and_conserve_tiles(players, ext):
    """BASELINE-size properties for row N4's kernels (the counterpart of tests/test_full_size_configs.py): 32,768 games in one batch vs
    eight batches of 4096 seeded by global id -- compact records, final records, RNG positions and counters byte-identical per global
    id (a game does not depend on the GPU count or on its wave's sibling) --, sampled ids replayed through the oracle, and 100 tiles
    conserved in every one of the 32,768 final records."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    P, G, SHARDS, T, base = players, 4096, 8, 256, 77000
    big = BatchedAzul(SHARDS * G, players=P, ext_rules=ext)
    big.seed(base)
    big.init()
    big.new_round()
    tb = big.alloc_trajectory(T, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[big.displays], mask_bits=False)
    packs = []
    for _ in range(2):
        big.selfplay(T, tb["mask"], tb["action"], tb["reward"], tb["done"], packed=tb["packed"])
        packs.append(tb["packed"].clone())
    torch.cuda.synchronize()
    big_rec, big_pos, big_cnt = big.get_records(), big.get_rng_range()[1], big.counters()
    assert (_conserved(big_rec, P, big.displays) == 100).all()
    assert int(big_cnt["episodes"].sum()) > SHARDS * G * 4
    for k in range(SHARDS):
        env = BatchedAzul(G, players=P, ext_rules=ext)
        env.seed(base + k * G)
        env.init()
        env.new_round()
        ts = env.alloc_trajectory(T, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[env.displays], mask_bits=False)
        for i in range(2):
            env.selfplay(T, ts["mask"], ts["action"], ts["reward"], ts["done"], packed=ts["packed"])
            assert torch.equal(ts["packed"], packs[i][:, k * G:(k + 1) * G]), (k, i)
        torch.cuda.synchronize()
        assert env.get_records().tobytes() == big_rec[k * G:(k + 1) * G].tobytes(), k
        assert np.array_equal(env.get_rng_range()[1], big_pos[k * G:(k + 1) * G]), k
        c = env.counters()
        assert np.array_equal(c["episodes"], big_cnt["episodes"][k * G:(k + 1) * G]) and np.array_equal(c["stuck"], big_cnt["stuck"][k * G:(k + 1) * G])
        del env, ts
    pk = torch.cat(packs).cpu().numpy().view(np.uint32)
    for gid in list(range(0, SHARDS * G, 2731)) + [G - 1, G, SHARDS * G - 1]:
        s = oz.StreamX(base + gid, P, ext=ext)
        o = s.advance(2 * T, want_records=False)
        p = pk[:, gid]
        a = np.where((p & 0xFF) == 0xFF, (p >> 16).astype(np.int32), (p & 0xFF).astype(np.int32))
        a[a == 0xFFFF] = -1
        assert np.array_equal(a, o["action"]) and np.array_equal((p >> 8) & 0xFF, o["done"]), gid
        assert s.record().tobytes() == big_rec[gid].tobytes() and s.rng_state()[1] == int(big_pos[gid]), gid
