"""The device core (csrc/azul_core.hpp), compiled for the HOST with the 64-lane emulation of
csrc/azul_wave.hpp, must agree bit for bit with the oracle.  This is a logic check of the wave-level
code in the build container; the `-m gpu` tests repeat it on the real kernels through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as oz
from tests.hostcheck import hostcheck as hc

RULESETS = [(oz.FIRST_RANDOM, oz.POOL_LID), (1, oz.POOL_RANDOM), (2, oz.POOL_LID), (oz.FIRST_RANDOM, oz.POOL_RANDOM)]


def test_seed_stream_matches_cpython():
    for seed in [0, 1, 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 63 + 11]:
        mt = np.zeros(624, np.uint32)
        hc.lib().hc_seed(seed, hc.ptr(mt))
        r = oz.seeded_rng(seed)
        assert np.array_equal(mt, np.ctypeslib.as_array(r.mt))


@pytest.mark.parametrize("fp,pool", RULESETS)
def test_selfplay_stream_bit_exact(fp, pool):
    for seed in list(range(6)) + [2 ** 32 + 7]:
        n = 700                                    # ~12 episodes, several MT regenerations
        o = oz.Stream(seed, fp if fp else oz.FIRST_RANDOM, pool)
        h = hc.HostStream(seed, fp, pool)
        eo = o.advance(n)
        eh = h.advance(n)
        assert np.array_equal(eo["action"], eh["action"]), (seed, fp, pool)
        assert np.array_equal(eo["mask"], eh["mask"])
        assert np.array_equal(eo["reward"], eh["reward"])
        assert np.array_equal(eo["done"], eh["done"])
        assert np.array_equal(eo["rec_after"].view(np.uint8).reshape(n, 128), eh["rec_after"])
        st = h.get()
        assert st["rec"].tobytes() == o.record().tobytes()
        mt, pos = o.rng_state()
        assert np.array_equal(st["mt"], mt) and st["pos"] == pos
        assert st["episodes"] == int(o.episodes.value) and st["stuck"] == 0
        assert np.array_equal(st["stat_sum"], o.stats_sum)


def test_chunked_advance_equals_one_shot():
    a = hc.HostStream(3, 0, 1)
    b = hc.HostStream(3, 0, 1)
    whole = a.advance(300)
    parts = [b.advance(n) for n in (1, 7, 64, 100, 128)]
    for key in whole:
        assert np.array_equal(whole[key], np.concatenate([p[key] for p in parts]))
    assert a.get()["rec"].tobytes() == b.get()["rec"].tobytes()


def test_single_record_ops_on_trajectory_states():
    L = hc.lib()
    o = oz.Stream(11, oz.FIRST_RANDOM, oz.POOL_LID)
    recs = o.advance(400)["rec_after"]
    for rec in recs[::3]:
        raw = np.frombuffer(rec.tobytes(), np.uint8).copy()
        q = oz.unpack(rec, oz.POOL_LID, oz.FIRST_RANDOM)
        m = np.zeros(180, np.uint8)
        L.hc_mask(hc.ptr(raw), hc.ptr(m))
        assert np.array_equal(m.astype(bool), oz.check_all_valid(q.game))
        for persp in (0, 1):
            obs = np.zeros(136, np.float32)
            L.hc_observe(hc.ptr(raw), persp, hc.ptr(obs))
            assert np.array_equal(obs.astype(np.int64), oz.get_state(q.game, persp))
        assert L.hc_potential(hc.ptr(raw), 1) == oz.lib().oz_potential(C.byref(q.game))
        fl = L.hc_flags(hc.ptr(raw))
        assert bool(fl & 1) == bool(oz.lib().oz_is_end_of_round(C.byref(q.game)))
        assert bool(fl & 2) == bool(oz.lib().oz_is_end_of_game(C.byref(q.game)))
        # real scoring
        raw2 = raw.copy()
        L.hc_count_score(hc.ptr(raw2), 1)
        oz.lib().oz_count_score(C.byref(q.game))
        assert raw2.tobytes() == oz.pack(q).tobytes()


def test_step_and_runner_step_with_explicit_rng():
    L = hc.lib()
    OL = oz.lib()
    for seed in range(4):
        r = oz.seeded_rng(seed)
        q = oz.Runner()
        assert OL.oz_runner_init(C.byref(q), oz.FIRST_RANDOM, oz.POOL_LID, C.byref(r)) == 0
        assert OL.oz_runner_reset(C.byref(q), C.byref(r)) == 0
        # same through the core
        mt = np.zeros(624, np.uint32)
        L.hc_seed(seed, hc.ptr(mt))
        pos = np.array([624], np.uint32)
        rec = np.zeros(128, np.uint8)
        assert L.hc_runner_reset(hc.ptr(rec), 0, 1, hc.ptr(mt), hc.ptr(pos), 1) == 0     # __init__
        assert L.hc_runner_reset(hc.ptr(rec), 0, 1, hc.ptr(mt), hc.ptr(pos), 0) == 0     # reset()
        assert rec.tobytes() == oz.pack(q).tobytes()
        done = False
        rs = np.random.RandomState(seed)
        while not done:
            mask = oz.check_all_valid(q.game)
            legal = np.flatnonzero(mask)
            a = int(rs.choice(legal))
            # an illegal action first: state and stream must stay untouched
            illegal = np.flatnonzero(~mask)
            if len(illegal):
                bad = int(illegal[0])
                before, pos_before = rec.copy(), pos.copy()
                rew, dn = C.c_int(0), C.c_int(0)
                assert L.hc_runner_step(hc.ptr(rec), bad, 0, 1, hc.ptr(mt), hc.ptr(pos), C.byref(rew), C.byref(dn)) == 1
                assert np.array_equal(before, rec) and pos_before[0] == pos[0]
            rew, dn = C.c_int(0), C.c_int(0)
            assert L.hc_runner_step(hc.ptr(rec), a, 0, 1, hc.ptr(mt), hc.ptr(pos), C.byref(rew), C.byref(dn)) == 0
            orew, odn = C.c_int64(0), C.c_int(0)
            assert OL.oz_runner_step(C.byref(q), a, C.byref(r), C.byref(orew), C.byref(odn)) == 0
            assert rew.value == orew.value and dn.value == odn.value
            assert rec.tobytes() == oz.pack(q).tobytes()
            done = bool(odn.value)
        assert pos[0] == r.idx


def test_weight_table_is_cpython_accumulate():
    from itertools import accumulate
    T = np.zeros((31, 151))
    hc.lib().hc_weight_table(hc.ptr(T))
    for J in (0, 1, 7, 30):
        cum = list(accumulate([0.01] * J + [1.0] * 150))
        assert T[J, 0] == (cum[J - 1] if J else 0.0)
        for m in (1, 2, 77, 150):
            assert T[J, m] == cum[J + m - 1]
    # the kernels' compact form m + Fr[J][floor(log2 m)] reproduces all 31 x 150 sums exactly
    assert hc.lib().hc_sample_tab_ok() == 1


def test_factory_draw_literal_fp64_build_matches_oracle():
    """libhostcheck_fp.so = the same core with the integer fast path of the factory draw disabled."""
    import os
    import subprocess
    here = os.path.dirname(hc.__file__)
    subprocess.check_call(["make", "-s", "-C", here, "libhostcheck_fp.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(here, "libhostcheck_fp.so"))
    L.hc_stream_new.restype = C.c_void_p
    L.hc_stream_new.argtypes = [C.c_ulonglong, C.c_int, C.c_int]
    L.hc_stream_advance.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    for seed in range(4):
        h = L.hc_stream_new(seed, 0, 1)
        n = 600
        act = np.zeros(n, np.int32)
        rec = np.zeros((n, 128), np.uint8)
        assert L.hc_stream_advance(h, n, None, hc.ptr(act), None, None, hc.ptr(rec)) == 0
        eo = oz.Stream(seed, 0, 1).advance(n)
        assert np.array_equal(eo["action"], act) and eo["rec_after"].tobytes() == rec.tobytes()


def test_vector_pointer_output_path():
    """OUT == 1 (per-lane trajectory pointers, all-lane stores) writes the same streams as the scalar form."""
    for seed, fp, pool in [(9, 0, 1), (10, 1, 0)]:
        o = oz.Stream(seed, fp, pool).advance(900)
        h = hc.HostStream(seed, fp, pool).advance(900, want_records=False)
        for key in ("mask", "action", "reward", "done"):
            assert np.array_equal(o[key], h[key]), key
