"""Single-game bridge between the Python facade (azul.py / game_runner.py of this package) and the C ABI.

A facade call = ONE ``azul_game_call``: the object's numpy attributes packed into one record (128 bytes for two players, the
256-byte wide record for three and four), ONE kernel on a 1-game batch, the results back -- one submission, one host
synchronisation.  The record is only sent when it differs from what the device already holds (the facade's attributes are
caller-writable numpy arrays, so the packed bytes are compared, not a dirty flag).

A call that changes a game also answers the two questions the reference's loops ask next (nn_runner.py:22-30,
game_runner.py:37-42): the legal mask of the state it leaves (AZUL_WANT_MASK) and what ``RandomAgent.get_a_output`` draws on that
mask from the stream as the call leaves it (AZUL_WANT_NEXT_ACTION: computed, the stream's index NOT moved).  ``get_valid_moves()``
and ``RandomAgent.get_a_output(mask)`` are then served from these answers -- no submission -- when, and only when, their inputs are
the ones the answers were computed from: the same record bytes; the same mask bytes and a global ``random`` state equal to the one
this backend installed.  Playing the remembered draw advances the global stream by the one ``random()`` it stands for, and the next
drawing call tells the device (AZUL_WANT_POS_IN).  Anything else -- a host draw in between, a re-seed, an edited mask or board --
takes the ordinary path.  ``GameRunner.step`` brings the agent's observation as well (AZUL_WANT_OBS, ``obs_persp`` 0) for the
``get_state()`` that opens the next iteration of nn_runner.py:22-30, served while no other submission has happened and the record is
unchanged.  A GameRunner loop -- RandomAgent's or a network's -- is one submission per agent step.

Randomness stays the reference's: the process-global CPython ``random`` stream.  A call that draws runs on the game's device
stream; afterwards the advanced state is installed with ``random.setstate`` -- so ``random.seed(1); Azul().new_round()`` gives
the reference's board exactly (reference tests/test_azul.py:36-39) while every draw is computed on the GPU.  The 624 words
cross PCIe only when they have to: host -> device when ``random.getstate()`` is not the state this backend installed last
(somebody seeded or drew on the host in between), device -> host when the call regenerated them (every 624 draws); otherwise
only the index moves.  `RandomAgent` samples on whichever backend holds the stream, so a GameRunner loop keeps it resident.
"""
import ctypes as C
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


# who holds the global `random` stream on a device: the backend that ran the last drawing call and the state it installed
_RNG = {"be": None, "state": None, "pos": 0}


# facade op -> (AZUL_CALL_*, results wanted, takes an argument)
_OPS = {
    "op_init": (L.CALL_INIT, 0), "op_new_round": (L.CALL_NEW_ROUND, 0), "op_move": (L.CALL_MOVE, 0),
    "op_next_player": (L.CALL_NEXT_PLAYER, 0), "op_count_score": (L.CALL_COUNT_SCORE, 0), "op_step": (L.CALL_STEP, 0),
    "op_flags": (L.CALL_QUERY, L.WANT_FLAGS), "op_mask": (L.CALL_QUERY, L.WANT_MASK), "op_observe": (L.CALL_QUERY, L.WANT_OBS),
    "op_statistics": (L.CALL_QUERY, L.WANT_STATS), "op_potential": (L.CALL_QUERY, L.WANT_POTENTIAL),
    "op_runner_init": (L.CALL_RUNNER_INIT, 0), "op_runner_reset": (L.CALL_RUNNER_RESET, 0),
    # GameRunner.step also brings back the legal mask of the state it leaves: the caller's next question is get_valid_moves()
    # (nn_runner.py:22-30, game_runner.py:73-75), which is then answered without a submission while the record is unchanged
    "op_runner_step": (L.CALL_RUNNER_STEP, L.WANT_MASK),
}
# two-player reference games: the calls after which a loop asks "which moves are legal" and "what does RandomAgent play" next
ASK_AHEAD = True          # (False: every question is its own submission -- the A/B of tests/test_facade_ask_ahead.py and bench.py)
_ASK_OBS = {"op_runner_step": 0}     # ... and "what does the agent see" (GameRunner.get_state(): the agent is player 1, perspective 0; nn_runner.py:22)
_ASK_AHEAD = {"op_new_round", "op_step", "op_runner_reset", "op_runner_step"}
_WANT_D2H = {}
_WANT_BYTES = ((L.WANT_MASK, 180), (L.WANT_OBS, 544), (L.WANT_STATS, 80), (L.WANT_NEXT_ACTION, 4), (L.WANT_FLAGS, 1))          # (the reference's sizes; traffic accounting only)


class HipBackend:
    """1-game BatchedAzul per rule set, created lazily on the current CUDA device; every facade call is one azul_game_call."""

    def __init__(self, first_player, tile_pool, players=2, ext=0):
        import torch
        from .batch import BatchedAzul
        if not torch.cuda.is_available():
            raise RuntimeError("the Azul facade runs its rules on an MI355X through libazulhip.so; no GPU is visible "
                               "(there is no CPU path)")
        self.torch = torch
        rules = {"first_player": "Random" if first_player == L.FIRST_RANDOM else int(first_player),
                 "tile_pool": "Lid" if tile_pool == L.POOL_LID else "Random"}
        self.env = BatchedAzul(1, rules=rules, players=players, ext_rules=ext)
        self._setup(self.env.num_actions, self.env.obs_size, self.env.record_dtype)

    def _setup(self, num_actions, obs_size, record_dtype):
        self.num_actions, self.obs_size = num_actions, obs_size
        self.ask_ahead = True                          # (a backend's own switch next to the module's ASK_AHEAD)
        self.c = L.AzulCall()
        self.c.game = 0
        self._rec_in = np.zeros(1, dtype=record_dtype)
        self._rec_out = np.zeros(1, dtype=record_dtype)
        self._mt_in = np.zeros(624, dtype=np.uint32)
        self._mt_out = np.zeros(624, dtype=np.uint32)
        self._mask_in = np.zeros(self.num_actions, dtype=np.uint8)
        self.c.record_out = self._rec_out.ctypes.data
        self.c.mt_out = self._mt_out.ctypes.data
        self._resident = None                # bytes of the record the device holds
        self._mask_for, self._mask = None, None   # the legal mask a call brought back, and the record bytes it belongs to
        self._ahead = None                   # (mask bytes, stream state installed by that call, its index, RandomAgent's draw on them)
        self._obs_for = None                 # (record bytes, perspective, submission number): self.c.obs still holds that observation
        self._flags_for = None               # (record bytes, AZUL_FLAG_* of that state)
        self._seq = 0                        # submissions so far

    def _game_call(self):
        """The one place that touches the device: self.c through azul_game_call on this backend's 1-game batch."""
        L.check(L.lib.azul_game_call(self.env._h, C.byref(self.c), self.env._stream()))

    def _submit(self, draws):
        """Run self.c; for a drawing call: hand the global stream over (only if the device copy is stale) and install the advanced
        state afterwards."""
        c = self.c
        h2d = 0
        if draws:
            st = random.getstate()
            if _RNG["be"] is self and _RNG["state"] == st:
                c.mt_in, c.pos_in = None, _RNG["pos"]              # the device's words ARE the global stream's; the host's index is the authority
                c.want |= L.WANT_POS_IN                            # (it moves without the device when a remembered draw is played)
            else:
                self._mt_in[:] = st[1][:624]
                c.mt_in, c.pos_in = self._mt_in.ctypes.data, st[1][624]
                h2d += 2500
        else:
            c.mt_in, c.pos_in = None, 0
        self._seq += 1                       # (the remembered draw stays: it is checked against mask bytes and stream state when it is used)
        self._game_call()
        if draws:
            words = tuple(self._mt_out.tolist()) if c.rng_regenerated else st[1][:624]
            new = (3, words + (int(c.pos_out),), st[2])
            random.setstate(new)
            _RNG["be"], _RNG["state"], _RNG["pos"] = self, new, int(c.pos_out)
        return h2d

    def call(self, op, args=(), rec=None, draws=False, mutates=True):
        """One facade method: `rec` = the caller's packed attributes; returns (result, record after the call or None)."""
        c = self.c
        rb = rec.tobytes()
        # GameRunner's player_score / move_counter (the last four bytes of the 128-byte record) only matter to GameRunner's own calls: the
        # Azul-level calls pack them as zero, and must neither re-send the record nor miss the cached mask because of them
        game = len(rb) - 4 if (len(rb) == L.RECORD_BYTES and not op.startswith("op_runner")) else len(rb)
        if op == "op_mask" and self._mask_for is not None and rb[:game] == self._mask_for[:game]:
            return self._mask.copy(), None   # the previous call already computed this state's mask (the bytes are compared, not a flag)
        if op == "op_flags" and self._flags_for is not None and rb[:game] == self._flags_for[0][:game]:
            return self._flags_for[1], None  # is_end_of_round / is_end_of_game of the state the last call left
        if op == "op_observe" and self._obs_for is not None:
            of = self._obs_for
            if of[2] == self._seq and int(args[0]) == of[1] and rb[:game] == of[0][:game]:
                return np.array(c.obs[:self.obs_size], dtype=np.float32).astype(np.int64), None      # brought back by the last submission
        c.op, want = _OPS[op]
        c.arg = int(args[0]) if args else 0
        c.mask_in = None
        h2d = 4 if args else 0
        if self._resident is None or rb[:game] != self._resident[:game]:
            self._rec_in[0] = rec
            c.record_in = self._rec_in.ctypes.data
            h2d += len(rb)
        else:
            c.record_in = None
        ahead = ASK_AHEAD and self.ask_ahead and op in _ASK_AHEAD
        if ahead:
            # (the draw is only asked of a call that DRAWS: such a call tells the device the stream index the host holds -- a non-drawing
            # call would answer from the device's stale index after a played-ahead draw)
            want |= L.WANT_MASK | L.WANT_FLAGS | (L.WANT_NEXT_ACTION if draws else 0)
            if op in _ASK_OBS:
                want |= L.WANT_OBS
                c.obs_persp = _ASK_OBS[op]
        c.want = want | (L.WANT_RECORD if mutates else 0)
        try:
            h2d += self._submit(draws)
        except Exception:
            self._resident = None            # the library refused the call (e.g. a record outside the kernels' domain)
            raise
        wb = _WANT_D2H.get(want)
        if wb is None:
            wb = _WANT_D2H[want] = 24 + sum(n for bit, n in _WANT_BYTES if want & bit)
        regen = 1 if (draws and c.rng_regenerated) else 0
        t = _TRAFFIC
        t["h2d"] += h2d
        t["d2h"] += wb + (len(rb) if mutates else 0) + 2496 * regen
        t["launches"] += 1
        t["syncs"] += 1 + regen
        new = None
        if mutates:
            new = self._rec_out[0].copy()
            self._resident = new.tobytes()
        elif c.record_in is not None:
            self._resident = rb
        if want & L.WANT_MASK and new is not None:
            m8 = np.frombuffer(bytes(c.mask), dtype=np.uint8)[:self.num_actions]
            self._mask_for, self._mask = self._resident, m8.astype(bool)
            if ahead and draws and c.next_action >= 0 and _RNG["be"] is self:
                self._ahead = (m8.tobytes(), _RNG["state"], _RNG["pos"], int(c.next_action))
            if ahead and op in _ASK_OBS:
                self._obs_for = (self._resident, _ASK_OBS[op], self._seq)
            if ahead:
                self._flags_for = (self._resident, int(c.flags))
        if op == "op_runner_step":
            return (int(c.reward), bool(c.done), int(c.status)), new
        if op == "op_mask":
            return np.frombuffer(bytes(c.mask), dtype=np.uint8)[:self.num_actions].astype(bool), new
        if op == "op_observe":
            return np.array(c.obs[:self.obs_size], dtype=np.float32).astype(np.int64), new
        if op == "op_statistics":
            return np.array(c.stats[:], dtype=np.float64), new
        if op == "op_flags":
            return int(c.flags), new
        if op == "op_potential":
            return int(c.potential), new
        return int(c.status), new

    def sample(self, mask):
        """RandomAgent.get_a_output on a caller's mask (game_runner.py:87-97): one random.choices draw on this backend's stream."""
        c = self.c
        m = np.asarray(mask, dtype=np.uint8).reshape(-1)
        if m.size != self.num_actions:
            raise ValueError("this backend samples masks of %d actions" % self.num_actions)
        ah, self._ahead = self._ahead, None
        if ah is not None and _RNG["be"] is self and _RNG["state"] is ah[1]:
            st = random.getstate()
            if st == ah[1] and m.tobytes() == ah[0]:
                # the draw the last call already made on exactly this mask and this stream: play it -- the stream moves on by one random()
                new = (3, st[1][:624] + (ah[2] + 2,), st[2])
                random.setstate(new)
                _RNG["state"], _RNG["pos"] = new, ah[2] + 2
                return ah[3]
        c.op, c.arg, c.want, c.record_in = L.CALL_SAMPLE_MASK, 0, 0, None
        self._mask_in[:] = m
        c.mask_in = self._mask_in.ctypes.data
        h2d = 184 + self._submit(True)
        _count(h2d=h2d, d2h=24 + (2496 if c.rng_regenerated else 0), launches=1, syncs=1 + (1 if c.rng_regenerated else 0))
        return int(c.action)


_FACTORY = HipBackend      # the class behind backend(): the product has exactly this one (no CPU path)
_CACHE = {}


def backend(first_player, tile_pool, players=2, ext=0):
    key = (_FACTORY, int(first_player), int(tile_pool), int(players), int(ext))
    if key not in _CACHE:
        _CACHE[key] = _FACTORY(int(first_player), int(tile_pool), int(players), int(ext))
    return _CACHE[key]


def sampling_backend(num_actions=180):
    """The backend RandomAgent draws on: the one that holds the global stream if there is one (any rule set with the same action space
    can run the sampler), so that a GameRunner loop never moves the 624 words.  Masks of 240 / 300 actions (seven / nine displays,
    beyond the reference) are sampled on a backend of that action space."""
    be = _RNG["be"]
    if be is not None and type(be) is _FACTORY and getattr(be, "num_actions", 180) == num_actions:
        return be
    if num_actions == 180:
        return backend(1, L.POOL_RANDOM)
    players = {240: 3, 300: 4}.get(num_actions)
    if players is None:
        raise ValueError("a legal mask holds 180, 240 or 300 actions")
    return backend(1, L.POOL_RANDOM, players, L.RULE_DISPLAYS_2P1)
