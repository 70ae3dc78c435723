"""The arithmetic facts the kernels' RandomAgent sampler rests on (csrc/azul_tables.hpp, azul_selfplay2.hpp: selfplay_step2 / sample_slow2), checked
against CPython's own random.choices arithmetic (random.py:506-541: accumulate + bisect_right(cum, x, 0, n - 1)) -- no device code involved:
  * S[k], k additions of 0.01, stays within 1e-13 of k / 100 (times 100) for every row a table can have (k <= 60);
  * FLOOR-ONLY masks (every legal action a 0.01-weight floor move; game_runner.py:87-97): with x100 = random() * fl(100 S[J]) the ordinal is
    floor(x100) + 1 wherever x100 is further than 1e-9 from an integer and floor(x100) + 1 <= J -- for random draws and for draws placed at,
    and 1e-16 .. 1e-10 around, every boundary k / J."""
from fractions import Fraction
from itertools import accumulate

import numpy as np


def test_cumulative_floor_weights_stay_within_1e13_of_the_hundredths():
    s, worst = 0.0, Fraction(0)
    for k in range(1, 61):
        s = s + 0.01
        worst = max(worst, abs(Fraction(s) * 100 - k))
    assert float(worst) < 1e-13


def test_floor_only_one_compare_form_equals_bisect_right_outside_its_margin():
    rs = np.random.RandomState(5)
    checked = unsafe = 0
    for J in range(1, 51):                                # 30 floor actions on the reference's five displays, 50 on nine
        S = np.array([0.0] + list(accumulate([0.01] * J)))
        t100 = 100.0 * S[J]
        u = rs.randint(0, 2 ** 53, size=100000, dtype=np.int64).astype(np.float64) * (1.0 / 9007199254740992.0)
        near = np.concatenate([np.arange(1, J + 1) / J + d for d in (-3e-16, -1e-16, 0.0, 1e-16, 3e-16, 1e-12, -1e-12, 1e-10, -1e-10)])
        u = np.concatenate([u, near[(near >= 0) & (near < 1)]])
        x = u * (S[J] + 0.0)                              # CPython: random() * total, total = cum[-1] + 0.0
        want = np.minimum(np.searchsorted(S[1:], x, side="right"), J - 1) + 1      # bisect_right(cum, x, 0, n - 1) + 1
        y = u * t100
        fy = np.floor(y)
        safe = (np.abs((y - fy) - 0.5) < 0.5 - 1e-9) & (fy + 1 <= J)
        assert np.array_equal((fy + 1)[safe], want[safe]), J
        checked += int(safe.sum())
        unsafe += int((~safe).sum())
    assert checked > 4_000_000 and 0 < unsafe < 20000     # the draws inside the margin are the boundary search's (tests/test_hostcheck_selfplay2.py)
