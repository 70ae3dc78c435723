"""The canonical 128-byte game record (include/azul_hip.h) as a numpy structured dtype, and converters
between it and the reference's attribute / JSON schema (azul.py:19-33, 90-117)."""
import numpy as np

STAT_KEYS = ("player_score", "opponent_score", "rounds", "percent_first_player", "floor_penalty",
             "max_combo", "completed_rows", "completed_columns", "completed_colors", "win_percent")

RECORD_DTYPE = np.dtype([
    ("displays", "u1", (5, 5)),
    ("center", "u1", (6,)),
    ("flags", "u1"),
    ("pattern_lines", "u1", (2, 5, 5)),
    ("floors", "u1", (2,)),
    ("walls", "<u4", (2,)),
    ("score", "<i2", (2,)),
    ("box", "u1", (5,)),
    ("lid", "u1", (5,)),
    ("turn_counter", "<u2"),
    ("first_player_stats", "<u2", (2,)),
    ("floor_penalty", "<i2", (2,)),
    ("max_combo", "u1", (2,)),
    ("completed_lines", "u1", (2, 3)),
    ("player_score", "<i2"),
    ("move_counter", "<u2"),
])
assert RECORD_DTYPE.itemsize == 128

_WALL_SHIFTS = np.arange(25, dtype=np.uint32)


def pack_flags(current_player, next_first_player, end_of_game):
    return (int(current_player) & 7) | ((int(next_first_player) & 7) << 3) | ((1 if end_of_game else 0) << 6)


def unpack_flags(flags):
    flags = int(flags)
    return flags & 7, (flags >> 3) & 7, bool((flags >> 6) & 1)


def walls_to_bits(walls):
    """bool[2][5][5] -> uint32[2] (bit 5*row+colour)."""
    w = np.asarray(walls).astype(bool).reshape(2, 25).astype(np.uint32)
    return (w << _WALL_SHIFTS).sum(axis=1).astype(np.uint32)


def bits_to_walls(bits):
    b = np.asarray(bits, dtype=np.uint32).reshape(2, 1)
    return (((b >> _WALL_SHIFTS) & 1) != 0).reshape(2, 5, 5)
