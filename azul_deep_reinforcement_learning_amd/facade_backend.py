"""Single-game bridge between the Python facade (azul.py / game_runner.py of this package) and the C ABI.

A facade call = pack the object's numpy attributes into one record (128 bytes for two players, the 256-byte wide
record for three and four), run ONE kernel on a 1-game batch, unpack.  Randomness stays the reference's: the process-global CPython ``random`` stream.  Before a
call that draws, the generator's 624 words + index are pushed into the game's device stream
(``azul_batch_set_rng``); afterwards the advanced state is pulled back and installed with
``random.setstate`` -- so ``random.seed(1); Azul().new_round()`` gives the reference's board exactly
(reference tests/test_azul.py:36-39) while every draw is computed on the GPU.
"""
import random

import numpy as np

from . import _lib as L

# bytes that crossed PCIe / launches / host-device synchronisations caused by facade calls (bench.py reports them per episode)
_TRAFFIC = {"h2d": 0, "d2h": 0, "launches": 0, "syncs": 0}


def traffic():
    return dict(_TRAFFIC)


def reset_traffic():
    for k in _TRAFFIC:
        _TRAFFIC[k] = 0


def _count(h2d=0, d2h=0, launches=0, syncs=0):
    _TRAFFIC["h2d"] += h2d
    _TRAFFIC["d2h"] += d2h
    _TRAFFIC["launches"] += launches
    _TRAFFIC["syncs"] += syncs


class HipBackend:
    """1-game BatchedAzul per rule set, created lazily on the current CUDA device."""

    def __init__(self, first_player, tile_pool, players=2):
        import torch
        from .batch import BatchedAzul
        if not torch.cuda.is_available():
            raise RuntimeError("the Azul facade runs its rules on an MI355X through libazulhip.so; no GPU is visible "
                               "(there is no CPU path)")
        self.torch = torch
        rules = {"first_player": "Random" if first_player == L.FIRST_RANDOM else int(first_player),
                 "tile_pool": "Lid" if tile_pool == L.POOL_LID else "Random"}
        self.env = BatchedAzul(1, rules=rules, players=players)

    # --- RNG bridging -------------------------------------------------------------------------
    def push_rng(self):
        st = random.getstate()
        words = np.array(st[1][:624], dtype=np.uint32)
        self.env.set_rng(0, words, st[1][624])
        _count(h2d=2500, syncs=1)
        self._gauss = st[2]

    def pull_rng(self):
        mt, pos = self.env.get_rng(0)
        _count(d2h=2500, syncs=1)
        random.setstate((3, tuple(int(x) for x in mt) + (int(pos),), self._gauss))

    # --- record in / out ------------------------------------------------------------------------
    def put(self, rec):
        self.env.set_records(rec)
        _count(h2d=rec.nbytes, syncs=1)

    def get(self):
        _count(d2h=self.env.record_dtype.itemsize, syncs=1)
        return self.env.get_records()[0]

    # --- operations (each one launch) -------------------------------------------------------------
    def _one(self, t, h2d=0):
        _count(h2d=h2d, d2h=t.numel() * t.element_size(), launches=1, syncs=1 + (1 if h2d else 0))
        return t.cpu().numpy()[0]

    def op_init(self):
        self.env.init()
        _count(launches=1)

    def op_new_round(self):
        return int(self._one(self.env.new_round()))

    def op_move(self, action):
        self.env.move([action])
        _count(h2d=4, launches=1, syncs=1)

    def op_next_player(self):
        self.env.next_player()
        _count(launches=1)

    def op_count_score(self):
        self.env.count_score()
        _count(launches=1)

    def op_step(self, action):
        return int(self._one(self.env.azul_step([action]), h2d=4))

    def op_flags(self):
        return int(self._one(self.env.flags()))

    def op_mask(self):
        return self._one(self.env.get_valid_moves()).astype(bool)

    def op_observe(self, perspective):
        return self._one(self.env.get_state(perspective)).astype(np.int64)

    def op_statistics(self):
        return self._one(self.env.statistics())

    def op_potential(self):
        return int(self._one(self.env.score_preview()))

    def op_runner_init(self):
        return int(self._one(self.env.runner_init()))

    def op_runner_reset(self):
        return int(self._one(self.env.reset()))

    def op_runner_step(self, action):
        reward, done, st = self.env.step([action])
        _count(launches=-2)                             # one launch, three results
        return int(self._one(reward, h2d=4)), bool(self._one(done)), int(self._one(st))

    def op_sample_mask(self, mask):
        return int(self._one(self.env.sample_mask(np.asarray(mask, dtype=np.uint8).reshape(1, 180)), h2d=180))


_FACTORY = HipBackend      # tests/hostcheck swaps in its 64-lane host emulation of the SAME core for CPU-only logic checks
_CACHE = {}


def backend(first_player, tile_pool, players=2):
    key = (_FACTORY, int(first_player), int(tile_pool), int(players))
    if key not in _CACHE:
        _CACHE[key] = _FACTORY(int(first_player), int(tile_pool), int(players))
    return _CACHE[key]
