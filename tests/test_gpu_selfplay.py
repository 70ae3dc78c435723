"""GPU parity (through the C ABI): flat random-agent self-play kernel vs the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as oz

pytestmark = pytest.mark.gpu

RULESETS = [
    ({"first_player": "Random", "tile_pool": "Lid"}, oz.FIRST_RANDOM, oz.POOL_LID),
    ({}, 1, oz.POOL_RANDOM),
    ({"first_player": 2, "tile_pool": "Lid"}, 2, oz.POOL_LID),
    ({"first_player": "Random"}, oz.FIRST_RANDOM, oz.POOL_RANDOM),
]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _start(env, seed_base):
    env.seed(seed_base)
    env.runner_init()   # GameRunner()
    env.runner_init()   # reset() without the opponent pre-moves (they are ordinary env moves here)


@pytest.mark.parametrize("rules,fp,pool", RULESETS)
def test_selfplay_trajectories_bit_exact(torch_cuda, rules, fp, pool):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n, steps, base = 96, 300, 7000
    env = BatchedAzul(n, rules=rules)
    _start(env, base)
    t = env.alloc_trajectory(steps, with_records=True)
    env.selfplay(steps, t["mask"], t["action"], t["reward"], t["done"], t["records"])
    torch_cuda.cuda.synchronize()
    act, rew, dn = t["action"].cpu().numpy(), t["reward"].cpu().numpy(), t["done"].cpu().numpy()
    msk, rec = t["mask"].cpu().numpy(), t["records"].cpu().numpy()
    final = env.get_records()
    cnt = env.counters()
    for g in range(n):
        s = oz.Stream(base + g, fp, pool)
        o = s.advance(steps)
        assert np.array_equal(o["action"], act[:, g]), g
        assert np.array_equal(o["mask"], msk[:, g]), g
        assert np.array_equal(o["reward"], rew[:, g]), g
        assert np.array_equal(o["done"], dn[:, g]), g
        assert o["rec_after"].tobytes() == rec[:, g].tobytes(), g
        assert s.record().tobytes() == final[g].tobytes(), g
        mt, pos = s.rng_state()
        gmt, gpos = env.get_rng(g) if g % 16 == 0 else (mt, pos)
        assert np.array_equal(mt, gmt) and pos == gpos
        assert int(s.episodes.value) == int(cnt["episodes"][g]) and cnt["stuck"][g] == 0
        assert np.array_equal(s.stats_sum, cnt["stat_sums"][g])


def test_selfplay_chunking_is_invisible(torch_cuda):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n = 64
    a, b = BatchedAzul(n), BatchedAzul(n)
    _start(a, 5)
    _start(b, 5)
    ta = a.alloc_trajectory(200)
    a.selfplay(200, ta["mask"], ta["action"], ta["reward"], ta["done"])
    parts = []
    for k in (1, 3, 60, 136):
        tb = b.alloc_trajectory(k)
        b.selfplay(k, tb["mask"], tb["action"], tb["reward"], tb["done"])
        parts.append(tb)
    torch_cuda.cuda.synchronize()
    for key in ("mask", "action", "reward", "done"):
        assert torch_cuda.equal(ta[key], torch_cuda.cat([p[key] for p in parts], dim=0))
    assert a.get_records().tobytes() == b.get_records().tobytes()


def test_selfplay_without_outputs_matches(torch_cuda):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n = 64
    a, b = BatchedAzul(n), BatchedAzul(n)
    _start(a, 77)
    _start(b, 77)
    ta = a.alloc_trajectory(150)
    a.selfplay(150, ta["mask"], ta["action"], ta["reward"], ta["done"])
    b.selfplay(150)
    torch_cuda.cuda.synchronize()
    assert a.get_records().tobytes() == b.get_records().tobytes()


def test_factory_draw_fp64_path_equals_integer_path(torch_cuda):
    """Both decision paths of the Lid factory draw (integer fast path / literal fp64) give the oracle's games."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n, steps = 64, 400
    a, b = BatchedAzul(n), BatchedAzul(n)
    b.set_draw_margin(0x7fffffff)            # every draw near a multiple of 2^32 within 2^31: always the fp64 path
    _start(a, 321)
    _start(b, 321)
    a.selfplay(steps)
    b.selfplay(steps)
    torch_cuda.cuda.synchronize()
    ra, rb = a.get_records(), b.get_records()
    assert ra.tobytes() == rb.tobytes()
    for g in range(0, n, 7):
        s = oz.Stream(321 + g)
        s.advance(steps, want_records=False)
        assert s.record().tobytes() == ra[g].tobytes()


def test_packed_record_and_maskbits_match_plain_outputs(torch_cuda):
    """The compact 4-byte record and the bit-packed mask (what the all-gather ships) carry the plain outputs."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd.parallel import unpack_moves
    env = BatchedAzul(128)
    _start(env, 900)
    t = env.alloc_trajectory(300, packed_mask=True)
    env.selfplay(300, t["mask"], t["action"], t["reward"], t["done"], maskbits=t["maskbits"], packed=t["packed"])
    torch_cuda.cuda.synchronize()
    action, done, reward = unpack_moves(t["packed"])
    assert torch_cuda.equal(action, t["action"]) and torch_cuda.equal(done, t["done"]) and torch_cuda.equal(reward, t["reward"])
    bits = t["maskbits"].cpu().numpy().view(np.uint8).reshape(300, 128, 24)
    assert np.array_equal(np.unpackbits(bits, axis=2, bitorder="little")[:, :, :180], t["mask"].cpu().numpy())
