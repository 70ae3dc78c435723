"""Row N4 on the GPU: batches of three- and four-player games (azul_batch_create_players, 256-byte wide records, kernels of
csrc/azul_rules_x.hpp: two games per wavefront) replay the reference's Azul(players=3|4) streams (tests/golden/traj_players.npz, generated from the real
reference) through the C ABI: azul_batch_init / _new_round / _legal_mask / _step / _flags / _statistics, IllegalMove and
GameEnded statuses, RNG positions; the single rule methods (move, count_score, next_player) against the oracle; and the
entries that mirror the two-player GameRunner are refused.  Reference: azulnet/azul.py:18-33, 64-89, 118-191, 192-315."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as oz

pytestmark = pytest.mark.gpu

RULES = {"default": {}, "lid_randomfirst": {"first_player": "Random", "tile_pool": "Lid"}}


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, "traj_players.npz"))


def _expected(gold, key, t, P):
    rec = np.zeros((), dtype=oz.RECORD_NP_DTYPE)
    rec["displays"], rec["center"] = gold[key + "_displays"][t], gold[key + "_center"][t]
    rec["flags"] = int(gold[key + "_cur"][t]) | (int(gold[key + "_nfp"][t]) << 3) | (int(gold[key + "_eog_flag"][t]) << 6)
    rec["pattern_lines"][:P] = gold[key + "_pattern_lines"][t]
    rec["floors"][:P] = gold[key + "_floors"][t]
    w = gold[key + "_walls"][t].reshape(P, 25).astype(np.uint32)
    rec["walls"][:P] = (w << np.arange(25, dtype=np.uint32)).sum(axis=1)
    rec["score"][:P] = gold[key + "_score"][t]
    rec["box"], rec["lid"], rec["turn_counter"] = gold[key + "_box"][t], gold[key + "_lid"][t], gold[key + "_turn_counter"][t]
    rec["first_player_stats"][:P] = gold[key + "_first_player_stats"][t]
    rec["floor_penalty"][:P] = gold[key + "_floor_penalty"][t]
    rec["max_combo"][:P] = gold[key + "_max_combo"][t]
    rec["completed_lines"][:P] = gold[key + "_completed_lines"][t]
    rec["players"] = P
    return rec


@pytest.mark.parametrize("players", [3, 4])
@pytest.mark.parametrize("ruleset", ["default", "lid_randomfirst", "lid_firstP"])
def test_batched_games_replay_the_reference(golden_dir, players, ruleset):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    gold = _gold(golden_dir)
    P = players
    rules = dict(RULES[ruleset]) if ruleset in RULES else {"first_player": P, "tile_pool": "Lid"}
    keys = ["p%d_%s_s%d" % (P, ruleset, s) for s in range(8)]
    n = len(keys)
    env = BatchedAzul(n, rules=rules, players=P)
    assert env.record_dtype.itemsize == 256 and L.lib.azul_batch_record_bytes(env._h) == 256 and L.lib.azul_batch_players(env._h) == P
    env.seed(seeds=np.arange(n, dtype=np.uint64))
    env.init()
    recs = env.get_records()
    for g, key in enumerate(keys):
        assert int(recs[g]["flags"]) >> 3 == int(gold[key + "_init_nfp"]) and int(recs[g]["players"]) == P
    assert (env.new_round().cpu().numpy() == 0).all()
    recs = env.get_records()
    for g, key in enumerate(keys):
        assert np.array_equal(recs[g]["displays"], gold[key + "_first_displays"]) and np.array_equal(recs[g]["center"], gold[key + "_first_center"])
    lengths = [len(gold[k + "_action"]) for k in keys]
    for t in range(max(lengths)):
        live = np.array([t < ln for ln in lengths])
        mask = env.get_valid_moves().cpu().numpy()
        flags = env.flags().cpu().numpy()
        acts = np.zeros(n, dtype=np.int32)
        for g, key in enumerate(keys):
            if live[g]:
                assert np.array_equal(np.packbits(mask[g], bitorder="little"), gold[key + "_mask"][t]), (key, t)
                assert bool(flags[g] & L.FLAG_END_OF_ROUND) == bool(gold[key + "_eor_before_step"][t])
                acts[g] = int(gold[key + "_action"][t])
        if t % 6 == 1:                                       # IllegalMove: status 1, record and RNG position untouched
            bad = np.array([int(np.flatnonzero(~mask[g])[0]) if not mask[g].all() else 0 for g in range(n)], dtype=np.int32)
            sel = live & ~mask.all(axis=1)
            before, pos0 = env.get_records(), env.get_rng_range()[1]
            st = env.azul_step(bad, active=sel.astype(np.uint8)).cpu().numpy()
            assert (st[sel] == L.ILLEGAL_MOVE).all()
            assert env.get_records().tobytes() == before.tobytes() and np.array_equal(env.get_rng_range()[1], pos0)
        st = env.azul_step(acts, active=live.astype(np.uint8)).cpu().numpy()
        assert (st[live] == L.OK).all(), t
        recs = env.get_records()
        flags = env.flags().cpu().numpy()
        for g, key in enumerate(keys):
            if live[g]:
                assert recs[g].tobytes() == _expected(gold, key, t, P).tobytes(), (key, t)
                assert bool(flags[g] & L.FLAG_END_OF_GAME) == bool(gold[key + "_eog_walls"][t])
                assert bool(flags[g] & L.FLAG_ENDED_FLAG) == bool(gold[key + "_eog_flag"][t])
    # every game has ended: GameEnded, statistics, RNG words consumed
    st = env.azul_step(np.zeros(n, dtype=np.int32)).cpu().numpy()
    assert (st == L.GAME_ENDED).all()
    stats = env.statistics().cpu().numpy()
    for g, key in enumerate(keys):
        assert np.allclose(stats[g], gold[key + "_stats"], rtol=0, atol=1e-12), key
        r = oz.seeded_rng(g)
        for _ in range(int(gold[key + "_rng_words"][-1])):
            oz.lib().oz_rng_u32(C.byref(r))
        mt, pos = env.get_rng(g)
        assert pos == r.idx and np.array_equal(mt, np.ctypeslib.as_array(r.mt)), key


@pytest.mark.parametrize("players", [3, 4])
def test_single_rule_methods_and_random_agent_against_the_oracle(players):
    """move / count_score / next_player / new_round one call at a time (not through step), the RandomAgent draw on the
    game's own mask, and JSON round trips, against the oracle for 64 games driven by RandomAgent picks."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    Lz = oz.lib()
    P, n = players, 64
    env = BatchedAzul(n, rules={"first_player": "Random", "tile_pool": "Lid"}, players=P)
    env.seed(900)
    env.init()
    env.new_round()
    rngs = [oz.seeded_rng(900 + g) for g in range(n)]
    games = [oz.Game() for _ in range(n)]
    for g in range(n):
        assert Lz.oz_init(C.byref(games[g]), P, 0, oz.POOL_LID, C.byref(rngs[g])) == 0
        assert Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g])) == 0
    alive = np.ones(n, dtype=bool)
    for it in range(140):
        a = env.random_action(active=alive.astype(np.uint8)).cpu().numpy()
        for g in range(n):
            if alive[g]:
                m = oz.check_all_valid(games[g]).astype(np.uint8)
                assert a[g] == Lz.oz_random_agent(m.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(rngs[g])), (it, g)
        stuck = alive & (a < 0)
        alive &= ~stuck
        env.move(np.where(alive, a, 0), active=alive.astype(np.uint8))
        for g in range(n):
            if alive[g]:
                Lz.oz_move(C.byref(games[g]), int(a[g]) % 6, (int(a[g]) // 6) % 5, int(a[g]) // 30)
        eor = env.is_end_of_round().cpu().numpy() & alive
        for g in range(n):
            if alive[g]:
                assert bool(eor[g]) == bool(Lz.oz_is_end_of_round(C.byref(games[g])))
        env.count_score(active=eor.astype(np.uint8))
        env.next_player(active=(alive & ~eor).astype(np.uint8))
        for g in range(n):
            if eor[g]:
                Lz.oz_count_score(C.byref(games[g]))
            elif alive[g]:
                Lz.oz_next_player(C.byref(games[g]))
        over = env.is_end_of_game().cpu().numpy() & eor
        deal = eor & ~over
        st = env.new_round(active=deal.astype(np.uint8)).cpu().numpy()
        for g in range(n):
            if over[g]:
                assert Lz.oz_is_end_of_game(C.byref(games[g]))
            if deal[g]:
                assert Lz.oz_new_round(C.byref(games[g]), C.byref(rngs[g])) == st[g] == 0
        alive &= ~over
        recs = env.get_records()
        for g in range(n):
            assert recs[g].tobytes() == oz.pack_np(games[g]).tobytes(), (it, g)
        if not alive.any():
            break
    assert not alive.any()
    pos = env.get_rng_range()[1]
    assert all(int(pos[g]) == rngs[g].idx for g in range(n))
    # JSON round trip of the wide record (the reference's schema + x_* keys)
    data = env.export_json()
    assert data[0]["players"] == P and len(data[0]["pattern_lines"]) == P
    other = BatchedAzul(n, rules={"first_player": "Random", "tile_pool": "Lid"}, players=P)
    other.import_json(data)
    assert other.get_records().tobytes() == env.get_records().tobytes()


def test_game_runner_entries_are_refused_for_more_than_two_players():
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    env = BatchedAzul(4, players=3)
    env.seed(1)
    env.init()
    with pytest.raises(L.AzulHipError):
        env.reset()                                                     # GameRunner.reset / step / the shaped reward: two players (game_runner.py:50)
    with pytest.raises(L.AzulHipError):
        env.step(np.zeros(4, np.int32))
    with pytest.raises(L.AzulHipError):
        env.score_preview()
    env.new_round()
    assert env.get_state().shape == (4, 5 * 5 + 6 + 52 * 3 + 1)         # get_state itself is P-generic in the reference (game_runner.py:56-72)
    with pytest.raises(L.AzulHipError):
        env.get_state(perspective=3)                                    # players are 0 .. P-1
    tr = env.alloc_trajectory(4, mask_pitch=192)                        # padded mask rows are fine (round 3 refused them for 3 / 4 players)
    env.selfplay(4, tr["mask"], tr["action"], tr["reward"], tr["done"])
    with pytest.raises(L.AzulHipError):
        BatchedAzul(4, players=5)
    with pytest.raises(Exception):
        BatchedAzul(4, players=3, rules={"first_player": 4})            # IllegalRule: 1..players


def test_persistent_selfplay_for_three_and_four_players_replays_the_reference(golden_dir):
    """azul_batch_selfplay on 3- and 4-player batches (azul_x_selfplay_kernel: one launch, two games per wavefront resident in registers) against
    tests/golden/traj_players_selfplay.npz -- streams played by the REAL reference (Azul(players=P).step with the reference's
    RandomAgent on the global stream, fresh game at each game end) -- and against the oracle's records, RNG positions and counters."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    gold = np.load(os.path.join(golden_dir, "traj_players_selfplay.npz"))
    groups = {}
    for i, key in enumerate(gold["index_key"]):
        groups.setdefault((int(gold["index_players"][i]), int(gold["index_first"][i]), int(gold["index_pool"][i])), []).append(
            (int(gold["index_seed"][i]), str(key)))
    assert len(groups) == 6
    for (P, first, pool), members in groups.items():
        members.sort()
        seeds = [m[0] for m in members]
        assert seeds == list(range(seeds[0], seeds[0] + len(seeds)))
        rules = {"tile_pool": "Lid" if pool == 1 else "Random"}
        if first >= 0:
            rules["first_player"] = "Random" if first == 0 else first
        T = len(gold[members[0][1] + "_action"])
        for variant in ("records", "full", "none"):
            env = BatchedAzul(len(seeds), rules=rules, players=P, device="cuda:0")
            env.seed(seeds[0])
            env.init()                                            # Azul(players=P, rules)
            env.new_round()
            if variant == "records":                              # OUT == 2: any subset of the streams + record snapshots
                tr = env.alloc_trajectory(T, with_records=True)
                env.selfplay(T, tr["mask"], tr["action"], tr["reward"], tr["done"], records=tr["records"])
            elif variant == "full":                               # OUT == 1: all streams, lane-distributed stores
                tr = env.alloc_trajectory(T, packed_mask=True)
                env.selfplay(T, tr["mask"], tr["action"], tr["reward"], tr["done"], maskbits=tr["maskbits"], packed=tr["packed"])
            else:                                                 # OUT == 0, in two launches
                env.selfplay(T // 2)
                env.selfplay(T - T // 2)
            torch.cuda.synchronize()
            recs = env.get_records()
            cnt = env.counters()
            for g, (seed, key) in enumerate(members):
                s = oz.StreamNP(seed, P, first if first >= 0 else oz.FIRST_ABSENT, pool)
                o = s.advance(T)
                if variant != "none":
                    mask = tr["mask"][:, g].cpu().numpy()
                    assert np.array_equal(np.packbits(mask.astype(bool), axis=1, bitorder="little"), gold[key + "_mask"]), (key, variant)
                    assert np.array_equal(tr["action"][:, g].cpu().numpy(), gold[key + "_action"]), (key, variant)
                    assert np.array_equal(tr["done"][:, g].cpu().numpy(), gold[key + "_done"]), (key, variant)
                    assert not tr["reward"][:, g].any()
                if variant == "records":
                    got = tr["records"][:, g].cpu().numpy()
                    assert got.tobytes() == o["rec_after"].tobytes(), key
                if variant == "full":
                    from azul_deep_reinforcement_learning_amd.parallel import unpack_moves
                    a, d, r = unpack_moves(tr["packed"][:, g])
                    assert np.array_equal(a.cpu().numpy(), gold[key + "_action"]) and np.array_equal(d.cpu().numpy(), gold[key + "_done"])
                    bits = tr["maskbits"][:, g].cpu().numpy().view(np.uint8).reshape(T, 24)[:, :23]
                    assert np.array_equal(bits, gold[key + "_mask"]), key
                assert recs[g].tobytes() == s.record().tobytes(), (key, variant)
                assert env.get_rng(g)[1] == s.rng_state()[1] and np.array_equal(env.get_rng(g)[0], s.rng_state()[0]), (key, variant)
                assert int(cnt["episodes"][g]) == int(gold[key + "_episodes"]) and int(cnt["stuck"][g]) == int(gold[key + "_stuck"])
                assert np.allclose(cnt["stat_sums"][g], gold[key + "_stats_sum"], rtol=0, atol=1e-9), key
