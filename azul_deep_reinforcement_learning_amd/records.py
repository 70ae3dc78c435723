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

# the 256-byte wide record of 3- and 4-player batches (include/azul_hip.h; row N4)
RECORD_NP_DTYPE = np.dtype([
    ("displays", "u1", (5, 5)),
    ("center", "u1", (6,)),
    ("flags", "u1"),
    ("pattern_lines", "u1", (4, 5, 5)),
    ("floors", "u1", (4,)),
    ("walls", "<u4", (4,)),
    ("score", "<i2", (4,)),
    ("box", "u1", (5,)),
    ("lid", "u1", (5,)),
    ("turn_counter", "<u2"),
    ("first_player_stats", "<u2", (4,)),
    ("floor_penalty", "<i2", (4,)),
    ("max_combo", "u1", (4,)),
    ("completed_lines", "u1", (4, 3)),
    ("players", "u1"),
    ("n_displays", "u1"),            # 0 = the reference's five; 7 / 9 with the 2P+1 displays rule (beyond the reference)
    ("reserved0", "u1", (2,)),
    ("xdisplays", "u1", (4, 5)),     # factory displays 5 .. 8
    ("reserved", "u1", (28,)),
])
assert RECORD_NP_DTYPE.itemsize == 256

_WALL_SHIFTS = np.arange(25, dtype=np.uint32)


def pack_flags(current_player, next_first_player, end_of_game):
    return (int(current_player) & 7) | ((int(next_first_player) & 7) << 3) | ((1 if end_of_game else 0) << 6)


def unpack_flags(flags):
    flags = int(flags)
    return flags & 7, (flags >> 3) & 7, bool((flags >> 6) & 1)


def walls_to_bits(walls):
    """bool[P][5][5] -> uint32[P] (bit 5*row+colour)."""
    w = np.asarray(walls).astype(bool)
    w = w.reshape(w.shape[0], 25).astype(np.uint32)
    return (w << _WALL_SHIFTS).sum(axis=1).astype(np.uint32)


def bits_to_walls(bits):
    b = np.asarray(bits, dtype=np.uint32).reshape(-1, 1)
    return (((b >> _WALL_SHIFTS) & 1) != 0).reshape(b.shape[0], 5, 5)


def record_displays(rec):
    """Number of factory displays a record describes: 5 unless the wide record says otherwise."""
    return (int(rec["n_displays"]) or 5) if "n_displays" in rec.dtype.names else 5


def all_displays(rec):
    """displays[D][5] of a record (the reference's five, then the extra ones of the 2P+1 rule)."""
    D = record_displays(rec)
    return rec["displays"] if D == 5 else np.concatenate([rec["displays"], rec["xdisplays"][:D - 5]])


def set_displays(rec, displays):
    d = np.asarray(displays)
    D = d.shape[0]
    rec["displays"] = d[:5]
    if D != 5:
        rec["n_displays"] = D
        rec["xdisplays"][:D - 5] = d[5:]


def record_players(rec):
    """Number of players a record describes: 2 for the 128-byte record, the `players` byte of the 256-byte wide record."""
    return int(rec["players"]) if "players" in rec.dtype.names else 2


def record_to_json(rec):
    """One record (128 bytes, or the 256-byte wide record of 3 / 4 players) -> the dict Azul.export_JSON writes
    (azul.py:106-117: ten keys) plus `x_*` keys for what the reference's schema leaves out (tile pools, end_of_game flag,
    per-game statistics, GameRunner's score/move counters)."""
    cur, nfp, eog = unpack_flags(rec["flags"])
    P = record_players(rec)
    d = {
        "game_board_displays": all_displays(rec).astype(int).tolist(),
        "game_board_center": rec["center"].astype(int).tolist(),
        "pattern_lines": rec["pattern_lines"][:P].astype(int).tolist(),
        "walls": bits_to_walls(rec["walls"][:P]).astype(int).tolist(),
        "floors": rec["floors"][:P].astype(int).tolist(),
        "score": rec["score"][:P].astype(int).tolist(),
        "current_player": int(cur),
        "next_first_player": int(nfp),
        "players": P,
        "turn_counter": int(rec["turn_counter"]),
        "x_end_of_game": bool(eog),
        "x_box_tiles": rec["box"].astype(int).tolist(),
        "x_lid_tiles": rec["lid"].astype(int).tolist(),
        "x_first_player_stats": rec["first_player_stats"][:P].astype(int).tolist(),
        "x_floor_penalty": rec["floor_penalty"][:P].astype(int).tolist(),
        "x_max_combo": rec["max_combo"][:P].astype(int).tolist(),
        "x_completed_lines": rec["completed_lines"][:P].astype(int).tolist(),
    }
    if "player_score" in rec.dtype.names:
        d["x_player_score"] = int(rec["player_score"])
        d["x_move_counter"] = int(rec["move_counter"])
    elif P == 2:
        d["x_wide"] = True           # a two-player game of an extended-rule batch lives in the wide record
    return d


def json_to_record(d):
    """Inverse of record_to_json; a plain reference file (no `x_*` keys) loads with empty pools / zero statistics, like
    Azul.import_JSON (azul.py:90-104), which restores exactly those ten keys.  Two players give a 128-byte record, three and
    four the 256-byte wide record."""
    P = int(d.get("players", 2))
    if P not in (2, 3, 4):
        raise ValueError("Azul is a game for 2, 3 or 4 players")
    D = len(d["game_board_displays"])
    wide = P != 2 or D != 5 or bool(d.get("x_wide", False))
    rec = np.zeros((), dtype=RECORD_NP_DTYPE if wide else RECORD_DTYPE)
    if wide:
        rec["players"] = P
    set_displays(rec, d["game_board_displays"]) if wide else rec.__setitem__("displays", np.asarray(d["game_board_displays"]))
    rec["center"] = np.asarray(d["game_board_center"])
    rec["pattern_lines"][:P] = np.asarray(d["pattern_lines"])
    rec["walls"][:P] = walls_to_bits(np.asarray(d["walls"]))
    rec["floors"][:P] = np.asarray(d["floors"])
    rec["score"][:P] = np.asarray(d["score"])
    rec["flags"] = pack_flags(d["current_player"], d["next_first_player"], d.get("x_end_of_game", False))
    rec["turn_counter"] = d["turn_counter"]
    for key, field in (("x_box_tiles", "box"), ("x_lid_tiles", "lid")):
        if key in d:
            rec[field] = np.asarray(d[key])
    for key, field in (("x_first_player_stats", "first_player_stats"), ("x_floor_penalty", "floor_penalty"), ("x_max_combo", "max_combo"),
                       ("x_completed_lines", "completed_lines")):
        if key in d:
            rec[field][:P] = np.asarray(d[key])
    if not wide:
        for key, field in (("x_player_score", "player_score"), ("x_move_counter", "move_counter")):
            if key in d:
                rec[field] = np.asarray(d[key])
    return rec
