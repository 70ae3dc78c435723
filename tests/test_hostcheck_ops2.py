"""The two-player rule kernel on CPU: azul_op_kernel (csrc/azul_selfplay_kernels.hpp on azul_ops2.hpp -> azul_env2.hpp -> azul_selfplay2.hpp,
all UNMODIFIED) compiled by g++ and run under the lockstep 64-lane emulation of tests/hostcheck/simt, against the oracle: every op of the
C ABI's two-player rule entries on states of real trajectories -- mask, observation (both perspectives), what-if potential, flags, count_score,
Azul.step / GameRunner.step / reset with an explicit MT19937 stream, the RandomAgent sampler, policy / agent steps with their counters -- plus
the refusal paths, a batch launch with an odd number of games, and the literal-fp64 factory draw.  The `-m gpu` tests repeat it on the real
kernel through the C ABI (tests/test_gpu_batch_ops.py).  Reference: azulnet/azul.py:18-315, azulnet/game_runner.py:23-117."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as oz
from tests.hostcheck import hostcheck as hc

OP = hc.EmuBackend.OP
OP_POLICY_STEP, OP_AGENT_STEP = 12, 13


def dev(fp=oz.FIRST_RANDOM, pool=oz.POOL_LID):
    return hc.EmuBackend(fp, pool)


def test_seed_stream_matches_cpython():
    for seed in [0, 1, 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 63 + 11]:
        mt = np.zeros(624, np.uint32)
        hc.lib().sh2_seed(seed, hc.ptr(mt))
        r = oz.seeded_rng(seed)
        assert np.array_equal(mt, np.ctypeslib.as_array(r.mt))


def test_weight_table_is_cpython_accumulate():
    from itertools import accumulate
    T = np.zeros((31, 151))
    hc.lib().sh2_weight_table(hc.ptr(T))
    for J in (0, 1, 7, 30):
        cum = list(accumulate([0.01] * J + [1.0] * 150))
        assert T[J, 0] == (cum[J - 1] if J else 0.0)
        for m in (1, 2, 77, 150):
            assert T[J, m] == cum[J + m - 1]
    # the kernels' compact form m + Fr[J][floor(log2 m)] reproduces all 31 x 150 sums exactly
    assert hc.lib().sh2_sample_tab_ok() == 1


def test_single_record_ops_on_trajectory_states():
    o = oz.Stream(11, oz.FIRST_RANDOM, oz.POOL_LID)
    recs = o.advance(400)["rec_after"]
    e = dev()
    for rec in recs[::5]:
        q = oz.unpack(rec, oz.POOL_LID, oz.FIRST_RANDOM)
        e.put(rec)
        out = e._op("query", want_mask=True, want_obs=0, want_flags=True, want_potential=True, want_stats=True)
        assert np.array_equal(out["mask"].astype(bool), oz.check_all_valid(q.game))
        assert np.array_equal(out["obs"].astype(np.int64), oz.get_state(q.game, 0))
        assert np.array_equal(e.op_observe(1), oz.get_state(q.game, 1))
        assert out["potential"] == oz.lib().oz_potential(C.byref(q.game))
        assert bool(out["flags"] & 1) == bool(oz.lib().oz_is_end_of_round(C.byref(q.game)))
        assert bool(out["flags"] & 2) == bool(oz.lib().oz_is_end_of_game(C.byref(q.game)))
        assert out["player"] == int(rec["flags"]) & 7
        assert out["stats"].tolist() == list(oz.get_statistics(q.game).values())
        # real scoring: the record afterwards, and the derived state the next query sees
        e.op_count_score()
        oz.lib().oz_count_score(C.byref(q.game))
        assert e.get().tobytes() == oz.pack(q).tobytes()
        assert np.array_equal(e.op_mask(), oz.check_all_valid(q.game))


def test_step_and_runner_step_with_explicit_rng():
    OL = oz.lib()
    for seed in range(3):
        r = oz.seeded_rng(seed)
        q = oz.Runner()
        assert OL.oz_runner_init(C.byref(q), oz.FIRST_RANDOM, oz.POOL_LID, C.byref(r)) == 0
        assert OL.oz_runner_reset(C.byref(q), C.byref(r)) == 0
        e = dev()
        hc.lib().sh2_seed(seed, hc.ptr(e.mt))
        e.pos[0] = 624
        assert e.op_runner_init() == 0          # GameRunner()
        assert e.op_runner_reset() == 0         # reset()
        assert e.get().tobytes() == oz.pack(q).tobytes()
        done = False
        rs = np.random.RandomState(seed)
        while not done:
            mask = oz.check_all_valid(q.game)
            legal = np.flatnonzero(mask)
            a = int(rs.choice(legal))
            illegal = np.flatnonzero(~mask)
            if len(illegal):              # an illegal action first: state and stream must stay untouched
                before, pos_before, mt_before = e.rec.copy(), int(e.pos[0]), e.mt.copy()
                assert e.op_runner_step(int(illegal[0]))[2] == 1
                assert e.op_step(int(illegal[-1])) == 1
                assert np.array_equal(before, e.rec) and pos_before == int(e.pos[0]) and np.array_equal(mt_before, e.mt)
            rew, dn, st = e.op_runner_step(a)
            orew, odn = C.c_int64(0), C.c_int(0)
            assert OL.oz_runner_step(C.byref(q), a, C.byref(r), C.byref(orew), C.byref(odn)) == 0
            assert st == 0 and rew == orew.value and dn == bool(odn.value)
            assert e.get().tobytes() == oz.pack(q).tobytes()
            done = bool(odn.value)
        assert int(e.pos[0]) == r.idx
        assert np.array_equal(e.mt, np.ctypeslib.as_array(r.mt))
        assert e.op_step(0) == 2               # GameEnded (azul.py:298-299)
        assert e.op_step(-1) == 2 and e._op("move", 180)["status"] == 4      # out of range: BAD_ACTION where the game is still running


def test_rule_methods_one_by_one_equal_the_oracle():
    """Azul.__init__ / new_round / move / next_player / is_end_of_round / count_score called one after the other, the way tests/test_azul.py
    drives the reference (azul.py:296-313 spelled out by hand)."""
    OL = oz.lib()
    for seed, fp, pool in [(5, oz.FIRST_RANDOM, oz.POOL_LID), (6, 1, oz.POOL_RANDOM)]:
        r = oz.seeded_rng(seed)
        q = oz.Runner()
        assert OL.oz_init(C.byref(q.game), 2, fp, pool, C.byref(r)) == 0
        q.first_player, q.tile_pool = fp, pool
        e = dev(fp, pool)
        hc.lib().sh2_seed(seed, hc.ptr(e.mt))
        e.pos[0] = 624
        e.op_init()
        assert e.get().tobytes() == oz.pack(q).tobytes()
        assert e.op_new_round() == 0 and OL.oz_new_round(C.byref(q.game), C.byref(r)) == 0
        rs = np.random.RandomState(seed)
        for _ in range(70):
            assert e.get().tobytes() == oz.pack(q).tobytes()
            if OL.oz_is_end_of_game(C.byref(q.game)):
                break
            legal = np.flatnonzero(oz.check_all_valid(q.game))
            a = int(rs.choice(legal))
            e.op_move(a)
            OL.oz_move(C.byref(q.game), a % 6, (a // 6) % 5, a // 30)
            assert e.get().tobytes() == oz.pack(q).tobytes()
            if e.op_flags() & 1:
                assert OL.oz_is_end_of_round(C.byref(q.game))
                e.op_count_score()
                OL.oz_count_score(C.byref(q.game))
                if not (e.op_flags() & 2):
                    assert e.op_new_round() == 0 and OL.oz_new_round(C.byref(q.game), C.byref(r)) == 0
            else:
                assert not OL.oz_is_end_of_round(C.byref(q.game))
                e.op_next_player()
                OL.oz_next_player(C.byref(q.game))
        assert int(e.pos[0]) == r.idx


@pytest.mark.parametrize("pool", [oz.POOL_LID, oz.POOL_RANDOM])
def test_random_agent_and_sample_mask_draw_like_random_choices(pool):
    """RandomAgent.get_a_output on the game's own mask and on a caller's mask (game_runner.py:87-97): same action, same stream position."""
    OL = oz.lib()
    o = oz.Stream(21, oz.FIRST_RANDOM, pool)
    recs = o.advance(120)["rec_after"]
    e = dev(oz.FIRST_RANDOM, pool)
    for i, rec in enumerate(recs[::7]):
        q = oz.unpack(rec, pool, oz.FIRST_RANDOM)
        mask = oz.check_all_valid(q.game)
        for seed_pos in (0, 300, 621, 622, 623, 624):
            r = oz.seeded_rng(1000 + i)
            r.idx = seed_pos
            want = OL.oz_random_agent(mask.astype(np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(r))
            e.put(rec)
            hc.lib().sh2_seed(1000 + i, hc.ptr(e.mt))
            e.pos[0] = seed_pos
            got = e._op("random_action")["action"]
            assert got == want and int(e.pos[0]) == r.idx, (i, seed_pos)
            assert np.array_equal(e.mt, np.ctypeslib.as_array(r.mt))
            hc.lib().sh2_seed(1000 + i, hc.ptr(e.mt))
            e.pos[0] = seed_pos
            assert e.op_sample_mask(mask.astype(np.uint8)) == want
    # nothing legal: -1, no word consumed (the reference raises before random())
    e.pos[0] = 17
    assert e.op_sample_mask(np.zeros(180, np.uint8)) == -1 and int(e.pos[0]) == 17


def test_batch_launch_with_an_odd_number_of_games_and_inactive_rows():
    """Five games = three waves, the last one half empty; rows with active == 0 stay as they are."""
    L = hc.lib()
    n = 5
    streams = [oz.Stream(40 + g, oz.FIRST_RANDOM, oz.POOL_LID) for g in range(n)]
    for s in streams:
        s.advance(9 + 3 * streams.index(s), want_records=False)
    recs = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], np.uint32)
    masks = [oz.check_all_valid(oz.unpack(s.record(), oz.POOL_LID, oz.FIRST_RANDOM).game) for s in streams]
    actions = np.array([int(np.flatnonzero(m)[0]) for m in masks], np.int32)
    actions[1] = int(np.flatnonzero(~masks[1])[0])               # illegal: refused, state untouched
    active = np.array([1, 1, 1, 0, 1], np.uint8)
    status = np.full(n, 99, np.uint8)
    mask_out = np.zeros((n, 180), np.uint8)
    before = recs.copy()
    ep, sk, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    assert L.sh2_op_batch(n, hc.ptr(recs), oz.FIRST_RANDOM, oz.POOL_LID, OP["step"], hc.ptr(actions), hc.ptr(active), hc.ptr(mt), hc.ptr(pos),
                          hc.ptr(status), hc.ptr(mask_out), None, None, hc.ptr(ep), hc.ptr(sk), hc.ptr(ss)) == 0
    assert list(status) == [0, 1, 0, 99, 0]
    for g in range(n):
        q = oz.unpack(np.frombuffer(before[g].tobytes(), oz.RECORD_DTYPE)[0], oz.POOL_LID, oz.FIRST_RANDOM)
        if g in (1, 3):
            assert np.array_equal(recs[g], before[g])
        else:
            r = oz.seeded_rng(0)
            np.ctypeslib.as_array(r.mt)[:] = streams[g].rng_state()[0]
            r.idx = streams[g].rng_state()[1]
            a = int(actions[g])
            assert oz.lib().oz_step(C.byref(q.game), a % 6, (a // 6) % 5, a // 30, C.byref(r)) == 0
            assert recs[g].tobytes() == oz.pack(q).tobytes() and int(pos[g]) == r.idx
        assert np.array_equal(mask_out[g].astype(bool), oz.check_all_valid(q.game)), g


def test_factory_draw_literal_fp64_path_in_the_rule_kernel():
    """A draw margin that covers every draw sends new_round's factory draw through the literal fp64 code (azul.py:85-87)."""
    OL = oz.lib()
    for seed in range(3):
        r = oz.seeded_rng(seed)
        q = oz.Runner()
        assert OL.oz_runner_init(C.byref(q), oz.FIRST_RANDOM, oz.POOL_LID, C.byref(r)) == 0
        e = dev()
        hc.lib().sh2_seed(seed, hc.ptr(e.mt))
        e.pos[0] = 624
        assert e._op("runner_init", margin=0x7fffffff)["status"] == 0
        assert e.get().tobytes() == oz.pack(q).tobytes() and int(e.pos[0]) == r.idx
