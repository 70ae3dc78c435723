"""``Azul`` -- host-side mirror of the reference's rules class (azulnet/azul.py:17-315), backed by the GPU.

Same constructor, attributes (numpy arrays that callers read AND write), methods, exceptions and JSON schema
as the reference; every rule evaluation (new_round, move, is_legal_move, is_end_of_round, is_end_of_game,
count_score, step, get_statistics) is one kernel launch through libazulhip.so (azul_game_call: one submission, one host
synchronisation) on the current state of the attributes.  What stays on the host is bookkeeping only: rule parsing, attribute <-> record conversion, JSON
I/O and ``__eq__``.  ``Azul(players=3)`` / ``Azul(players=4)`` behave like the reference's (five displays, turn order
1..P; azul.py:18-33, 177-181; SURVEY.md hazard H5) on the 3 / 4 player kernels and the 256-byte wide record.
"""
import json
import random
import struct

import numpy as np

from . import _lib as L
from . import facade_backend as fb
from .batch import IllegalRule, parse_ext_rules, parse_rules
from .records import RECORD_DTYPE, RECORD_NP_DTYPE, all_displays, bits_to_walls, pack_flags, unpack_flags, walls_to_bits


def _ilist(x, name):
    """An attribute's values as Python ints.  The reference's attributes are int arrays, but callers assign freely: a float array
    holding integers (game.score = np.array([3., 0.])) is accepted like the reference accepts it; a fractional value is named."""
    a = x if type(x) is np.ndarray else np.asarray(x)
    k = a.dtype.kind
    if k == "i" or k == "u":
        return a.ravel().tolist()
    if k == "f":                                     # (a handful of values: plain Python is quicker than two numpy passes)
        vals = a.ravel().tolist()
        ints = [int(v) for v in vals]
        if ints != vals:
            raise ValueError("%s must hold integral values" % name)
        return ints
    return a.ravel().astype(np.int64).tolist()


class IllegalMove(Exception):
    pass


class GameEnded(Exception):
    pass


_STATUS_EXC = {L.ILLEGAL_MOVE: IllegalMove, L.GAME_ENDED: GameEnded, L.BAD_ACTION: IllegalMove}


class Azul:
    def __init__(self, players=2, state_file=None, rules={}):
        if players not in (2, 3, 4):
            raise ValueError("Azul is a game for 2, 3 or 4 players")
        self._first_code, self._pool_code = parse_rules(rules, players)      # raises IllegalRule (azul.py:41,54)
        # extended rules (beyond the reference, "parity unpinned"; batch.parse_ext_rules): all off unless the rules dict asks for them
        self._ext = parse_ext_rules(rules, players)
        self._displays = 2 * players + 1 if self._ext & L.RULE_DISPLAYS_2P1 else 5
        self._sources = self._displays + 1                                   # action = display + sources * colour + 5 * sources * pattern
        self.game_board_displays = np.zeros((self._displays, 5), dtype=int)
        self.game_board_center = np.zeros(6, dtype=int)
        self.pattern_lines = np.zeros((players, 5, 5), dtype=int)
        self.walls = np.zeros((players, 5, 5), dtype=bool)
        self.floors = np.zeros(players, dtype=int)
        self.score = np.zeros(players, dtype=int)
        self.current_player = 0
        self.players = players
        self.end_of_game = False
        self.turn_counter = 0
        self.first_player_stats = np.zeros(players)
        self.floor_penalty = np.zeros(players)
        self.max_combo = np.zeros(players)
        self.completed_lines = np.zeros((players, 3))
        self.rules = rules
        if self._first_code == L.FIRST_RANDOM:
            self.next_first_player = random.choice(list(range(1, players + 1)))   # azul.py:37, the global stream
        else:
            self.next_first_player = self._first_code
        self.tile_pool = "Lid" if self._pool_code == L.POOL_LID else "Random"
        self._tracked = self.tile_pool == "Lid" or bool(self._ext & L.RULE_FINITE_BAG)
        if self._tracked:
            self.box_tiles = np.array([20, 20, 20, 20, 20])
            self.lid_tiles = np.array([0, 0, 0, 0, 0])
        if state_file is not None:
            self.import_JSON(state_file)

    # ------------------------------------------------------------------------------------------
    # host bookkeeping
    # ------------------------------------------------------------------------------------------
    def __eq__(self, other):
        # azul.py:62-63: box/lid/statistics/rules are not part of equality
        return (np.array_equal(self.game_board_displays, other.game_board_displays)
                and np.array_equal(self.game_board_center, other.game_board_center)
                and np.array_equal(self.pattern_lines, other.pattern_lines)
                and np.array_equal(self.walls, other.walls)
                and np.array_equal(self.floors, other.floors)
                and np.array_equal(self.score, other.score)
                and self.current_player == other.current_player
                and self.next_first_player == other.next_first_player
                and self.players == other.players
                and self.end_of_game == other.end_of_game
                and self.turn_counter == other.turn_counter)

    __hash__ = None

    def import_JSON(self, path):
        with open(path) as fh:
            data = json.load(fh)
        self.game_board_displays = np.array(data["game_board_displays"], dtype=int)
        self.game_board_center = np.array(data["game_board_center"], dtype=int)
        self.pattern_lines = np.array(data["pattern_lines"], dtype=int)
        self.walls = np.array(data["walls"], dtype=bool)
        self.floors = np.array(data["floors"], dtype=int)
        self.score = np.array(data["score"], dtype=int)
        self.current_player = data["current_player"]
        self.next_first_player = data["next_first_player"]
        self.players = data["players"]
        self.turn_counter = data["turn_counter"]

    def export_JSON(self, path):
        data = {
            "game_board_displays": np.asarray(self.game_board_displays).tolist(),
            "game_board_center": np.asarray(self.game_board_center).tolist(),
            "pattern_lines": np.asarray(self.pattern_lines).tolist(),
            "walls": np.asarray(self.walls).tolist(),
            "floors": np.asarray(self.floors).tolist(),
            "score": np.asarray(self.score).tolist(),
            "current_player": int(self.current_player),
            "next_first_player": int(self.next_first_player),
            "players": int(self.players),
            "turn_counter": int(self.turn_counter),
        }
        with open(path, "w+") as fh:
            fh.write(json.dumps(data))

    # ------------------------------------------------------------------------------------------
    # attributes <-> 128-byte record
    # ------------------------------------------------------------------------------------------
    def _backend(self):
        return fb.backend(self._first_code, self._pool_code, self.players, self._ext)

    # the record as a struct format (little endian, no padding; include/azul_hip.h): struct.pack range-checks every value for free
    _FMT2 = struct.Struct("<25B6BB50B2B2I2h5B5BH2H2h2B6BhH")                      # 128 bytes, two players
    _FMTN = struct.Struct("<25B6BB100B4B4I4h5B5BH4H4h4B12BBB2x20B28x")             # 256 bytes, the wide record (absent players / displays: zero)
    _RANGES = (("game_board_displays", 0, 255), ("game_board_center", 0, 255), ("pattern_lines", 0, 255), ("floors", 0, 7),
               ("score", -32768, 32767), ("box_tiles", 0, 255), ("lid_tiles", 0, 255), ("first_player_stats", 0, 65535),
               ("floor_penalty", -32768, 32767), ("max_combo", 0, 255), ("completed_lines", 0, 255))

    _ARRAYS = ("game_board_displays", "game_board_center", "pattern_lines", "walls", "floors", "score", "first_player_stats",
               "floor_penalty", "max_combo", "completed_lines", "box_tiles", "lid_tiles")

    def _scalars(self, runner):
        return (self.current_player, self.next_first_player, self.end_of_game, self.turn_counter, self.players,
                (runner.player_score, runner.move_counter) if runner is not None else None)

    def _remember(self, rec, runner):
        """The record the attributes were just unpacked from, with what it takes to tell later that nobody touched them: the array
        objects themselves and their bytes (callers write them in place), the scalars."""
        d = self.__dict__
        self._memo = [rec, [(n, d[n], d[n].tobytes()) for n in self._ARRAYS if n in d], self._scalars(runner), runner is not None]

    def _remembered(self, runner):
        m = getattr(self, "_memo", None)
        if m is None:
            return None
        d = self.__dict__
        for (n, a, raw) in m[1]:
            if d.get(n) is not a or a.tobytes() != raw:
                return None
        sc = self._scalars(runner)
        if sc[:5] != m[2][:5]:
            return None
        if sc[5] == m[2][5]:
            return m[0]
        if not m[3] or runner is not None:
            return None                                   # (GameRunner's counters were edited)
        # an Azul-level call on a GameRunner's game: the record as the runner's last call left it -- its last four bytes hold the
        # runner's counters where a fresh pack writes zeros, and facade_backend.call ignores exactly those bytes for Azul-level calls
        return m[0]

    def _to_record(self, runner=None):
        """The attributes packed into the game record (128 bytes; 256 for three / four players).  Callers write the attributes
        freely (numpy arrays, Python ints), so every value is range-checked against what its record field holds: struct.pack does
        that while it packs (one C call instead of a dozen numpy reductions and field assignments).  Attributes that are still
        the arrays the last call unpacked, byte for byte, are not packed again."""
        rec = self._remembered(runner)
        if rec is not None:
            return rec
        P = self.players
        wide = P != 2 or self._ext != 0
        pad = 4 - P if wide else 0
        lid = self._tracked
        D = self._displays
        try:
            fl = _ilist(self.floors, "floors")
            if max(fl) > 7:
                raise struct.error("floors")
            z = [0] * pad
            disp = _ilist(self.game_board_displays, "game_board_displays")
            if len(disp) != 5 * D:
                raise struct.error("displays")
            vals = (disp[:25] + _ilist(self.game_board_center, "game_board_center")
                    + [pack_flags(self._player(self.current_player), self._player(self.next_first_player), self.end_of_game)]
                    + _ilist(self.pattern_lines, "pattern_lines") + [0] * (25 * pad) + fl + z
                    + walls_to_bits(self.walls).tolist() + z + _ilist(self.score, "score") + z
                    + (_ilist(self.box_tiles, "box_tiles") + _ilist(self.lid_tiles, "lid_tiles") if lid else [0] * 10)
                    + [int(self.turn_counter)]
                    + _ilist(self.first_player_stats, "first_player_stats") + z
                    + _ilist(self.floor_penalty, "floor_penalty") + z
                    + _ilist(self.max_combo, "max_combo") + z
                    + _ilist(self.completed_lines, "completed_lines") + [0] * (3 * pad))
            if wide:
                raw = self._FMTN.pack(*vals, P, 0 if D == 5 else D, *(disp[25:] + [0] * (45 - 5 * D)))
            else:
                raw = self._FMT2.pack(*vals, int(runner.player_score) if runner is not None else 0,
                                      int(runner.move_counter) if runner is not None else 0)
        except (struct.error, TypeError, ValueError) as err:
            if "must hold integral values" in str(err):
                raise
            for (name, l, h) in self._RANGES:             # name the offender the slow way
                if not hasattr(self, name):
                    continue
                v = np.asarray(getattr(self, name))
                if v.size and (v.min() < l or v.max() > h):
                    raise ValueError("%s outside the range the GPU record holds [%d, %d]" % (name, l, h))
            raise ValueError("an attribute does not fit the GPU record (shape of a %d-player game, current_player / next_first_player in "
                             "0..%d, turn_counter / player_score / move_counter in their 16-bit fields)" % (P, P))
        return np.frombuffer(raw, dtype=RECORD_NP_DTYPE if wide else RECORD_DTYPE)[0]

    def _action(self, display, color, pattern):
        """nn_serialize (game_runner.py:102-103) with 6 -> displays + 1"""
        return int(display) + self._sources * int(color) + 5 * self._sources * int(pattern)

    def _player(self, v):
        v = int(v)
        if not 0 <= v <= self.players:
            raise ValueError("player")
        return v

    _TAIL2 = struct.Struct("<2I2h10xH2H2h8xhH")      # the two-player record from byte 84: walls, score, turn counter, stats, GameRunner's counters

    def _from_record2(self, rec, runner):
        """_from_record for the 128-byte record: one widening of the bytes, the attributes are slices of it (same dtypes and shapes)."""
        raw = rec.tobytes()
        b = np.frombuffer(raw, dtype=np.uint8).astype(int)
        f = b[106:124].astype(float)
        w0, w1, s0, s1, tc, f0, f1, p0, p1, ps, mc = self._TAIL2.unpack_from(raw, 84)
        self.game_board_displays = b[0:25].reshape(5, 5)
        self.game_board_center = b[25:31]
        self.current_player, self.next_first_player, self.end_of_game = unpack_flags(raw[31])
        self.pattern_lines = b[32:82].reshape(2, 5, 5)
        self.floors = b[82:84]
        self.walls = bits_to_walls(rec["walls"])
        self.score = np.array((s0, s1))
        if self._tracked:
            self.box_tiles = b[96:101]
            self.lid_tiles = b[101:106]
        self.turn_counter = tc
        self.first_player_stats = np.array((f0, f1), dtype=float)
        self.floor_penalty = np.array((p0, p1), dtype=float)
        self.max_combo = f[10:12]
        self.completed_lines = f[12:18].reshape(2, 3)
        if runner is not None:
            runner.player_score = ps
            runner.move_counter = mc
        self._remember(rec, runner)

    def _from_record(self, rec, runner=None):
        P = self.players
        if rec.dtype.itemsize == 128:
            return self._from_record2(rec, runner)
        self.game_board_displays = all_displays(rec).astype(int)
        self.game_board_center = rec["center"].astype(int)
        self.current_player, self.next_first_player, self.end_of_game = unpack_flags(rec["flags"])
        self.pattern_lines = rec["pattern_lines"][:P].astype(int)
        self.floors = rec["floors"][:P].astype(int)
        self.walls = bits_to_walls(rec["walls"][:P])
        self.score = rec["score"][:P].astype(int)
        if self._tracked:
            self.box_tiles = rec["box"].astype(int)
            self.lid_tiles = rec["lid"].astype(int)
        self.turn_counter = int(rec["turn_counter"])
        self.first_player_stats = rec["first_player_stats"][:P].astype(float)
        self.floor_penalty = rec["floor_penalty"][:P].astype(float)
        self.max_combo = rec["max_combo"][:P].astype(float)
        self.completed_lines = rec["completed_lines"][:P].astype(float)
        if runner is not None:
            runner.player_score = int(rec["player_score"])
            runner.move_counter = int(rec["move_counter"])
        self._remember(rec, runner)

    def _run(self, op, *args, draws=False, mutates=True, runner=None):
        out, rec = self._backend().call(op, args, self._to_record(runner), draws, mutates)
        if mutates:
            self._from_record(rec, runner)
        return out

    # ------------------------------------------------------------------------------------------
    # rules (GPU)
    # ------------------------------------------------------------------------------------------
    def new_round(self):
        st = self._run("op_new_round", draws=True)
        if st == L.BOX_EMPTY:
            raise ValueError("Total of weights must be finite")      # what random.choices raises in the reference

    def move(self, display, color, pattern):
        self._run("op_move", self._action(display, color, pattern))

    def is_legal_move(self, display, color, pattern):
        mask = self._run("op_mask", mutates=False)
        return bool(mask[self._action(display, color, pattern)])

    def legal_mask(self):
        """All is_legal_move answers at once (what check_all_valid returns): 180, or (displays + 1) * 30 with the 2P+1 displays rule."""
        return self._run("op_mask", mutates=False)

    def next_player(self):
        self._run("op_next_player")

    def is_end_of_round(self):
        return bool(self._run("op_flags", mutates=False) & L.FLAG_END_OF_ROUND)

    def is_end_of_game(self):
        return bool(self._run("op_flags", mutates=False) & L.FLAG_END_OF_GAME)

    def count_score(self):
        self._run("op_count_score")

    def step(self, display, color, pattern):
        st = self._run("op_step", self._action(display, color, pattern), draws=True)
        if st in _STATUS_EXC:
            raise _STATUS_EXC[st]
        if st == L.BOX_EMPTY:
            raise ValueError("Total of weights must be finite")

    def get_statistics(self):
        s = self._run("op_statistics", mutates=False)
        keys = ["player_score", "opponent_score", "rounds", "percent_first_player", "floor_penalty", "max_combo",
                "completed_rows", "completed_columns", "completed_colors", "win_percent"]
        out = dict(zip(keys, (float(x) for x in s)))
        out["win_percent"] = bool(out["win_percent"])
        return out


__all__ = ["Azul", "IllegalMove", "GameEnded", "IllegalRule"]
