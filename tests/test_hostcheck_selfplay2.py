"""The benchmarked code on CPU: csrc/azul_selfplay2.hpp (two games per wavefront: azul_selfplay2_kernel itself) compiled
UNMODIFIED by g++ and run under the lockstep 64-lane emulation of tests/hostcheck/simt, against the oracle -- masks, actions,
rewards, done flags, record snapshots, final records, all 624 MT19937 words + positions, episode counters and statistics sums,
for every output variant of the kernel (padded one-store mask rows, bit-packed masks, dense rows, run-time subsets, no outputs),
an odd batch (the last wave plays one game) and three rule sets.  Also: the emulator's own cross-lane operations against their
definitions, and the same runs under UBSan / ASan (tests/hostcheck/Makefile; the logs are kept under profiles/).
Reference: azulnet/game_runner.py:43-55, 76-97; azulnet/azul.py:64-313."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as oz

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")
RULES = {"lid_randomfirst": (0, 1), "random_first1": (1, 0), "lid_first2": (2, 1)}      # (first_player code, tile_pool code)


def load(name=None):
    name = name or os.environ.get("AZUL_SIMT_LIB", "libsimt_selfplay2.so")        # run_sanitizers.sh: the _ubsan / _asan builds
    subprocess.check_call(["make", "-s", "-C", HERE, name], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, name))
    L.sh2_selfplay.restype = C.c_longlong
    L.sh2_selfplay.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_case(L, first, pool, n, T, variant, seed0, margin=0, prepare=None):
    """n games seeded seed0 + g (the oracle provides the state after random.seed; GameRunner(); reset()), T moves through the emulated
    wave code; returns everything the kernel would have written."""
    streams = [oz.Stream(seed0 + g, first_player=first if first else oz.FIRST_RANDOM, tile_pool=pool) for g in range(n)]
    if prepare:
        prepare(streams)
    state = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], dtype=np.uint32)
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    pitch = 192 if variant in (0, 1) else 180
    out = {}
    if variant != 4:
        out = {"mask": np.full((T, n, pitch), 0xEE, np.uint8), "action": np.full((T, n), -7, np.int32), "reward": np.full((T, n), -7, np.int32),
               "done": np.full((T, n), 9, np.uint8)}
        if variant in (0, 2):
            out["maskbits"] = np.zeros((T, n, 3), np.uint64)
        if variant != 3:
            out["packed"] = np.zeros((T, n), np.uint32)
        else:
            out["rec"] = np.zeros((T, n, 128), np.uint8)
    ops = L.sh2_selfplay(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), first, pool, margin, T, variant,
                         ptr(out.get("mask")), pitch, ptr(out.get("maskbits")), ptr(out.get("action")), ptr(out.get("reward")),
                         ptr(out.get("done")), ptr(out.get("packed")), ptr(out.get("rec")))
    assert ops > 0
    return streams, state, mt, pos, ep, stuck, ss, out, ops


def check_case(L, first, pool, n, T, variant, seed0, margin=0, prepare=None):
    streams, state, mt, pos, ep, stuck, ss, out, ops = run_case(L, first, pool, n, T, variant, seed0, margin, prepare)
    for g, s in enumerate(streams):
        o = s.advance(T)
        tag = (first, pool, variant, g)
        if variant != 4:
            assert np.array_equal(out["mask"][:, g, :180], o["mask"]), tag
            assert np.array_equal(out["action"][:, g], o["action"]) and np.array_equal(out["reward"][:, g], o["reward"]), tag
            assert np.array_equal(out["done"][:, g], o["done"]), tag
        if "maskbits" in out:
            bits = out["maskbits"][:, g].view(np.uint8).reshape(T, 24)[:, :23]
            assert np.array_equal(bits, np.packbits(o["mask"].astype(bool), axis=1, bitorder="little")), tag
        if "packed" in out:
            p = out["packed"][:, g]
            a = (p & 0xFF).astype(np.int32)
            a[a == 0xFF] = -1
            assert np.array_equal(a, o["action"]) and np.array_equal((p >> 8) & 0xFF, o["done"]), tag
            assert np.array_equal((p >> 16).astype(np.uint16).view(np.int16).astype(np.int32), o["reward"]), tag
        if "rec" in out:
            assert out["rec"][:, g].tobytes() == o["rec_after"].tobytes(), tag
        assert state[g].tobytes() == s.record().tobytes(), tag
        assert np.array_equal(mt[g], s.rng_state()[0]) and int(pos[g]) == s.rng_state()[1], tag
        assert int(ep[g]) == int(s.episodes.value) and int(stuck[g]) == int(s.stuck.value), tag
        assert np.allclose(ss[g], s.stats_sum, rtol=0, atol=1e-9), tag
    if variant in (0, 1):
        assert not out["mask"][:, :, 184:].any() or (out["mask"][:, :, 184:] == 0xEE).all()      # bytes 184.. of a padded row are never written
    return ops


def test_emulated_cross_lane_operations_match_their_definitions():
    assert load().sh2_selftest() == 0


@pytest.mark.parametrize("ruleset", sorted(RULES))
def test_selfplay_step2_under_lockstep_emulation_equals_the_oracle(ruleset):
    L = load()
    first, pool = RULES[ruleset]
    total = 0
    for variant in (0, 1, 2, 3, 4):
        total += check_case(L, first, pool, n=5, T=130, variant=variant, seed0=300 + 10 * variant)
    assert total > 30000                                   # cross-lane operations emulated


def test_factory_draw_fp64_path_under_emulation():
    """A draw margin that covers every draw sends the whole factory draw through the literal fp64 code and the sequential loop."""
    L = load()
    check_case(L, 0, 1, n=2, T=90, variant=3, seed0=55, margin=0x7fffffff)


@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_factory_draw_across_an_mt19937_regeneration(ruleset):
    """The forty words of a round's factory draw straddle the regeneration of the 624-word state (CPython's index anywhere in
    586..624 when the round starts): the words before it are read, the state is regenerated (three groups of chunks), the words
    after it are read.  34 games whose streams start at every second index from 558 on reach their first round end -- and
    several more -- inside the run; every word of the final states is compared."""
    L = load()
    first, pool = RULES[ruleset]

    def prepare(streams):
        for g, s in enumerate(streams):
            s.r.idx = 558 + 2 * g + (g & 1)

    for variant in (3, 0):
        check_case(L, first, pool, n=34, T=40, variant=variant, seed0=1200, prepare=prepare)


def test_stuck_slot_and_finished_game_under_emulation():
    """Hazard H3 (nothing legal although the round is not over: only the first-player token is left): the wave restarts the slot
    (done == 2, action -1, no random() consumed by the failed decision) in the half concerned while the sibling half plays on --
    the rare block that sits behind a scalar flag in selfplay_step2.  (A game handed in with its end_of_game flag set takes the
    same block on the device; the reference raises GameEnded there, so the oracle's stream has no counterpart: GPU tests cover it.)"""
    L = load()
    n, T = 4, 40
    streams = [oz.Stream(900 + g) for g in range(n)]
    for s in streams:
        s.advance(7)
    # games 1 and 2 (different waves, different halves): nothing legal
    recs = []
    for g in (1, 2):
        r_ = streams[g].record().copy()
        r_["displays"][:] = 0
        r_["center"][:] = [0, 0, 0, 0, 0, 1]
        recs.append((streams[g], r_))
    for s, rec in recs:
        q = oz.unpack(rec, tile_pool=oz.POOL_LID, first_player=oz.FIRST_RANDOM)
        C.memmove(C.byref(s.q), C.byref(q), C.sizeof(q))
    state = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], dtype=np.uint32)
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    mask = np.zeros((T, n, 180), np.uint8)
    action, reward, done = np.zeros((T, n), np.int32), np.zeros((T, n), np.int32), np.zeros((T, n), np.uint8)
    rec = np.zeros((T, n, 128), np.uint8)
    assert L.sh2_selfplay(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), 0, 1, 0, T, 3, ptr(mask), 180, None, ptr(action),
                          ptr(reward), ptr(done), None, ptr(rec)) > 0
    assert action[0, 1] == -1 and done[0, 1] == 2 and action[0, 2] == -1 and done[0, 2] == 2
    for g, s in enumerate(streams):
        stuck0 = int(s.stuck.value)
        o = s.advance(T)
        assert np.array_equal(mask[:, g], o["mask"]) and np.array_equal(action[:, g], o["action"]), g
        assert np.array_equal(reward[:, g], o["reward"]) and np.array_equal(done[:, g], o["done"]), g
        assert rec[:, g].tobytes() == o["rec_after"].tobytes(), g
        assert state[g].tobytes() == s.record().tobytes() and int(pos[g]) == s.rng_state()[1] and np.array_equal(mt[g], s.rng_state()[0]), g
        assert int(stuck[g]) == int(s.stuck.value) - stuck0


@pytest.mark.parametrize("variant,limit", [(3, 0), (3, 5000), (1, 5000), (0, 5000)])
def test_rule_error_stops_one_game_and_leaves_its_sibling_alone(variant, limit):
    """"Lid" pool with box and lid both empty when a round has to be dealt (the reference raises inside random.choices, azul.py:85-87; play
    cannot get there: 100 tiles, at most 80 of them on walls and lines when a round is dealt): the game concerned stops where it is -- its
    last move's outputs are written, nothing after -- and the other game of the same wave plays on, move for move like the oracle.  (The
    loop's per-game exit sits behind a wave-uniform flag that only the rare blocks set.)  In the instantiation a batch with a move limit runs,
    every LATER slot of the launch is also marked like a stuck slot (empty mask row, action -1, reward 0, done 2) and counted in `stuck`; the
    default instantiation leaves them untouched (that bookkeeping cost the benchmarked kernel ~1 %: profiles/round6_headline_ab.txt).
    Variants: dense rows with run-time subsets (3), the padded one-store rows (1), + bit-packed masks (0)."""
    L = load()
    L.sh2_set_move_limit.argtypes = [C.c_uint]
    n, T = 2, (60 if variant == 3 else 8)
    streams = [oz.Stream(4000 + g) for g in range(n)]
    for s in streams:
        s.advance(5)
    rec = streams[0].record().copy()
    rec["displays"][:] = 0
    rec["center"][:] = [1, 0, 0, 0, 0, 0]                 # one tile left, no token: the next move ends the round
    rec["box"][:] = 0
    rec["lid"][:] = 0
    rec["pattern_lines"][:] = 0                           # no full line returns tiles to the lid
    q = oz.unpack(rec, tile_pool=oz.POOL_LID, first_player=oz.FIRST_RANDOM)
    C.memmove(C.byref(streams[0].q), C.byref(q), C.sizeof(q))
    state = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], dtype=np.uint32)
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    pitch = 180 if variant == 3 else 192
    mask = np.full((T, n, pitch), 0xEE, np.uint8)
    action, reward, done = np.full((T, n), -7, np.int32), np.full((T, n), -7, np.int32), np.full((T, n), 9, np.uint8)
    bits = np.full((T, n, 3), 0xEEEEEEEEEEEEEEEE, np.uint64) if variant == 0 else None
    packed = np.full((T, n), 0xEEEEEEEE, np.uint32) if variant != 3 else None
    try:
        L.sh2_set_move_limit(limit)
        assert L.sh2_selfplay(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), 0, 1, 0, T, variant, ptr(mask), pitch, ptr(bits), ptr(action),
                              ptr(reward), ptr(done), ptr(packed), None) > 0
    finally:
        L.sh2_set_move_limit(0)
    assert 0 <= action[0, 0] < 180 and int(done[0, 0]) == 0 and int(ep[0]) == 0                      # its last move
    if limit:
        assert (action[1:, 0] == -1).all() and (done[1:, 0] == 2).all() and (reward[1:, 0] == 0).all()  # ... then marked slots
        assert not mask[1:, 0, :180].any() and int(stuck[0]) == T - 1
        if bits is not None:
            assert not bits[1:, 0].any()
        if packed is not None:
            assert (packed[1:, 0] == (0xff | (2 << 8))).all()                                          # action none | done 2 | reward 0
    else:
        assert (action[1:, 0] == -7).all() and (done[1:, 0] == 9).all() and int(stuck[0]) == 0       # ... nothing after it
    after = state[0].view(oz.RECORD_DTYPE)[0]
    assert not after["box"].any() and int(after["center"][0]) == 0
    o = streams[1].advance(T)                                                                          # its sibling: the oracle's game
    assert np.array_equal(action[:, 1], o["action"]) and np.array_equal(reward[:, 1], o["reward"]) and np.array_equal(done[:, 1], o["done"])
    assert np.array_equal(mask[:, 1, :180], o["mask"]) and state[1].tobytes() == streams[1].record().tobytes()
    assert int(pos[1]) == streams[1].rng_state()[1] and int(stuck[1]) == 0


def test_pack_c1_kernel_under_emulation_equals_the_torch_restatement():
    """azul_pack_c1_kernel (the producer in front of the C1 trajectory all-gather, nn_runner.py:17-47 / :59-78) compiled unmodified and run
    as emulated workgroups, against parallel.pack_c1's host restatement: 184 bytes per (step, game), incl. a ragged last workgroup, negative
    actions (0xff), every mask bit position and the four floats bit for bit."""
    import torch
    from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, pack_c1, unpack_c1
    L = load()
    L.sh2_pack_c1.restype = C.c_longlong
    L.sh2_pack_c1.argtypes = [C.c_void_p] * 10 + [C.c_int, C.c_void_p]
    rs = np.random.RandomState(4)
    T, G = 3, 37
    tr = {"obs": torch.from_numpy(rs.randint(0, 256, size=(T + 1, G, 136)).astype(np.float32)),
          "mask": torch.from_numpy((rs.rand(T + 1, G, 180) < 0.3).astype(np.uint8)), "player": torch.from_numpy(rs.randint(1, 3, size=(T + 1, G)).astype(np.uint8)),
          "action": torch.from_numpy(rs.randint(-1, 180, size=(T, G)).astype(np.int32)), "reward": torch.from_numpy(rs.randint(-40, 40, size=(T, G)).astype(np.int32)),
          "done": torch.from_numpy(rs.randint(0, 3, size=(T, G)).astype(np.uint8)), "value": torch.from_numpy(rs.randn(T, G, 1).astype(np.float32)),
          "log_prob": torch.from_numpy(rs.randn(T, G).astype(np.float32)), "entropy": torch.from_numpy(rs.rand(T, G).astype(np.float32)),
          "returns": torch.from_numpy(rs.randn(T, G).astype(np.float32) * 9)}
    tr["mask"][0, 0] = 1
    tr["mask"][1, 5, 179] = 1
    want = pack_c1(tr, T)
    out = np.full((T, G, C1_BYTES), 0xEE, np.uint8)
    a = {k: np.ascontiguousarray(v[:T].numpy()) for k, v in tr.items()}
    assert L.sh2_pack_c1(ptr(a["obs"]), ptr(a["mask"]), ptr(a["player"]), ptr(a["action"]), ptr(a["reward"]), ptr(a["done"]), ptr(a["value"]),
                         ptr(a["log_prob"]), ptr(a["entropy"]), ptr(a["returns"]), T * G, ptr(out)) > 0
    assert np.array_equal(out, want.numpy())
    u = unpack_c1(torch.from_numpy(out))
    assert torch.equal(u["mask"], tr["mask"][:T]) and torch.equal(u["action"], tr["action"]) and torch.equal(u["returns"], tr["returns"])
    probe = np.full(3, 7, np.uint64)
    L.sh2_clock_probe.argtypes = [C.c_void_p]
    assert L.sh2_clock_probe(ptr(probe)) == 0 and int(probe[0]) == 0 and int(probe[1]) == 0 and int(probe[2]) != 7      # (the emulation's timers read 0)


def test_move_limit_cuts_episodes_at_round_ends_like_the_oracle():
    """azul_batch_set_move_limit (beyond the reference, off by default; some games never end under the reference's rules): with a limit of
    30 moves every episode is cut at its first end of a round at or after move 30 -- the round is scored, no new round is dealt, done = 3,
    the slot restarts on the same stream and counts as `stuck`, not as an episode -- exactly like the oracle's restatement
    (oz_stream_advance_limited), masks / actions / rewards / records / MT19937 words included; limit 0 is the reference's behaviour."""
    L = load()
    L.sh2_set_move_limit.argtypes = [C.c_uint]
    n, T, limit = 3, 150, 30
    try:
        L.sh2_set_move_limit(limit)
        streams, state, mt, pos, ep, stuck, ss, out, ops = run_case(L, 0, 1, n, T, 3, 5100)
    finally:
        L.sh2_set_move_limit(0)
    cuts = 0
    for g, s in enumerate(streams):
        o = s.advance(T, move_limit=limit)
        assert np.array_equal(out["action"][:, g], o["action"]) and np.array_equal(out["reward"][:, g], o["reward"]), g
        assert np.array_equal(out["done"][:, g], o["done"]) and np.array_equal(out["mask"][:, g, :180], o["mask"]), g
        assert np.array_equal(out["rec"][:, g].view(oz.RECORD_DTYPE).reshape(T), o["rec_after"]), g
        assert state[g].tobytes() == s.record().tobytes() and int(pos[g]) == s.rng_state()[1] and np.array_equal(mt[g], s.rng_state()[0]), g
        assert int(ep[g]) == int(s.episodes.value) and int(stuck[g]) == int(s.stuck.value), g
        cuts += int((o["done"] == 3).sum())
        where = np.flatnonzero(o["done"] == 3)
        assert all(int(o["rec_after"][t]["move_counter"]) >= limit for t in where)          # cut at the end of the round that reached the limit
    assert cuts >= 6


def test_boundary_search_of_the_random_agents_draw_equals_cpythons_bisect():
    """az2::sample_slow2 -- what decides a RandomAgent draw that the one-compare fast path does not (x inside the 0.01-weight floor moves, x within
    1e-9 of a boundary, the clamp at the last weight) -- against the literal CPython computation (random.py:506-541: accumulate + bisect_right(cum,
    x, 0, n - 1)) for every (J, M) and x at, one and two ulps around, and between EVERY cumulative weight, plus random x; a game whose moves are all
    floor moves (hazard H9's never-ending games) takes this path on every decision (game_runner.py:87-97)."""
    from bisect import bisect_right
    from itertools import accumulate
    L = load()
    L.sh2_sample_slow.argtypes = [C.c_int] + [C.c_void_p] * 4
    rs = np.random.RandomState(11)
    xs, Js, Ms, want = [], [], [], []
    for J in range(0, 31):
        for M in [0, 1, 2, 8, 33, 150 - J]:
            if J + M == 0 or J + M > 180:
                continue
            cum = list(accumulate([0.01] * J + [1.0] * M))
            total = cum[-1] + 0.0
            pts = [0.0, total, np.nextafter(total, 0.0)]
            for c in (cum if len(cum) <= 64 else cum[:J + 3] + cum[J + 3::7] + cum[-2:]):
                pts += [c, np.nextafter(c, 0.0), np.nextafter(c, 9.0), np.nextafter(np.nextafter(c, 0.0), 0.0), c - 1e-10, c + 1e-10, c - 2e-9, c + 2e-9, c - 0.005]
            for k in range(1, J + 1):
                pts += [k * 0.01, np.nextafter(k * 0.01, 0.0), np.nextafter(k * 0.01, 9.0), k / 100.0 - 5e-10, k / 100.0 + 5e-10]
            pts += list(rs.rand(40) * total)
            for x in pts:
                x = float(x)
                if not 0.0 <= x <= total:
                    continue
                xs.append(x); Js.append(J); Ms.append(M)
                want.append(bisect_right(cum, x, 0, len(cum) - 1) + 1)
    xs, Js, Ms = np.array(xs), np.array(Js, np.int32), np.array(Ms, np.int32)
    out = np.full(len(xs), -1, np.int32)
    assert L.sh2_sample_slow(len(xs), ptr(xs), ptr(Js), ptr(Ms), ptr(out)) == 0
    bad = np.flatnonzero(out != np.array(want))
    assert len(bad) == 0, (len(bad), [(xs[i], Js[i], Ms[i], out[i], want[i]) for i in bad[:5]])
    assert len(xs) > 50000


def test_floor_only_decisions_of_a_never_ending_game_under_emulation():
    """Hazard H9's state (game 801 of the benchmark batch after ~310 k moves, profiles/round6_never_ending_game_801.txt: both players hold r
    tiles of colour 2 in pattern row r, so every legal action is a floor move for ever): every decision is a draw among 0.01 weights only
    (M = 0), which selfplay_step2 decides in its one-compare path from the table's floor-only row {fl(100 S[J]), 0}
    (azul_tables.hpp: build_floor_pairs; game_runner.py:87-97 + random.choices).  Six streams continue from that state for 150 moves each:
    masks, actions, rewards, records and RNG positions equal the oracle's, and every action IS a floor move."""
    L = load()
    rec = np.zeros(1, dtype=oz.RECORD_DTYPE)[0]
    rec["displays"] = [[0, 2, 0, 1, 1], [2, 0, 0, 1, 1], [0, 2, 0, 1, 1], [3, 0, 0, 0, 1], [1, 1, 0, 1, 1]]
    rec["center"] = [0, 0, 0, 0, 0, 1]
    rec["flags"] = 2
    lines = np.zeros((2, 5, 5), np.uint8)
    for r in range(1, 5):
        lines[:, r, 2] = r
    rec["pattern_lines"] = lines
    rec["walls"] = [16785787, 18779]
    rec["box"] = [8, 4, 0, 6, 5]
    rec["lid"] = [3, 6, 0, 5, 6]
    rec["turn_counter"] = 3949
    rec["first_player_stats"] = [1926, 2023]
    rec["floor_penalty"] = [10751, 10794]
    rec["max_combo"] = [4, 2]
    rec["move_counter"] = 38949
    seen = []

    def prepare(streams):
        for s in streams:
            s.q = oz.unpack(rec, tile_pool=oz.POOL_LID, first_player=oz.FIRST_RANDOM)
            seen.append(s)

    for variant in (0, 3):
        streams, state, mt, pos, ep, stuck, ss, out, ops = run_case(L, 0, 1, n=6, T=150, variant=variant, seed0=4242, prepare=prepare)
        for g, s in enumerate(streams):
            o = s.advance(150)
            assert np.array_equal(out["mask"][:, g, :180], o["mask"]) and np.array_equal(out["action"][:, g], o["action"]), (variant, g)
            assert np.array_equal(out["reward"][:, g], o["reward"]) and np.array_equal(out["done"][:, g], o["done"]), (variant, g)
            assert (o["action"] < 30).all() and not o["done"].any() and not o["mask"][:, 30:].any(), (variant, g)
            assert state[g].tobytes() == s.record().tobytes() and int(pos[g]) == s.rng_state()[1], (variant, g)
