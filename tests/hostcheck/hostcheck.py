"""ctypes loaders of the TEST-ONLY lockstep emulations of the product's kernels (tests/hostcheck/simt_*.cpp) and the facade's emulated device."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    """The two-player rule kernel (azul_op_kernel on csrc/azul_ops2.hpp) under the lockstep wave emulation: simt_ops2.cpp."""
    global _lib
    if _lib is None:
        name = os.environ.get("AZUL_SIMT_OPS_LIB", "libsimt_ops2.so")      # tests/hostcheck/run_sanitizers.sh: libsimt_ops2_asan.so
        subprocess.check_call(["make", "-s", "-C", _HERE, name], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, name))
        L.sh2_op.restype = C.c_int
        L.sh2_op.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 13
        L.sh2_op_batch.restype = C.c_int
        L.sh2_op_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11
        L.sh2_seed.argtypes = [C.c_ulonglong, C.c_void_p]
        L.sh2_weight_table.argtypes = [C.c_void_p]
        L.sh2_sample_tab_ok.restype = C.c_int
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- the P-player / D-display rules (csrc/azul_rules_x.hpp) under the lockstep wave emulation (simt_rules_x.cpp) ----
_xlib = None
XOP = {"query": 0, "init": 1, "new_round": 2, "move": 3, "next_player": 4, "count_score": 5, "step": 6, "random_action": 7, "sample_mask": 8}
EXT_DISPLAYS_2P1, EXT_END_BONUS, EXT_SHORT_DEAL, EXT_FINITE_BAG = 1, 2, 4, 8


def xlib():
    global _xlib
    if _xlib is None:
        name = os.environ.get("AZUL_SIMT_X_LIB", "libsimt_rules_x.so")
        subprocess.check_call(["make", "-s", "-C", _HERE, name], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, name))
        L.shx_op.restype = C.c_int
        L.shx_op.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_ulonglong, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_uint]
        _xlib = L
    return _xlib


def x_op(rec, players, first, pool, ext, op, action, mt, pos, mask_in=None, want_mask=False, want_flags=False, want_stats=False, want_obs=None,
         margin=0, want_next=False, pos_set=None):
    """One rule call of the emulated azul_x_op_kernel body on one 256-byte record.  `pool` / `ext`: the ABI's tile_pool and AZUL_RULE_*
    flags.  Returns a dict: status, mask, flags, stats, obs, action, player, rng_dirty."""
    D = 2 * players + 1 if ext & EXT_DISPLAYS_2P1 else 5
    NA, NOBS = (D + 1) * 30, 5 * D + 6 + 52 * players + 1
    mask = np.zeros(NA, np.uint8) if want_mask else None
    obs = np.zeros(NOBS, np.float32) if want_obs is not None else None
    stats = np.zeros(10) if want_stats else None
    flags, act, player, dirty, nxt = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-2)
    xpool = 2 if ext & EXT_FINITE_BAG else int(pool)
    mi = None if mask_in is None else np.ascontiguousarray(mask_in, dtype=np.uint8)
    st = xlib().shx_op(ptr(rec), players, D, int(first), xpool, int(bool(ext & EXT_END_BONUS)), int(bool(ext & EXT_SHORT_DEAL)), margin,
                       XOP[op] if isinstance(op, str) else op, int(action), ptr(mt), ptr(pos), ptr(mi), ptr(mask), ptr(obs),
                       int(want_obs) if want_obs is not None else 0, C.cast(C.byref(flags), C.c_void_p), ptr(stats),
                       C.cast(C.byref(act), C.c_void_p), C.cast(C.byref(player), C.c_void_p), C.cast(C.byref(dirty), C.c_void_p),
                       C.cast(C.byref(nxt), C.c_void_p) if want_next else None, 0 if pos_set is None else 1 + int(pos_set))
    return {"status": st, "mask": mask, "flags": flags.value, "stats": stats, "obs": obs, "action": act.value, "player": player.value,
            "rng_dirty": dirty.value, "next_action": nxt.value}


class EmuBackend:
    """TEST-ONLY stand-in for the DEVICE side of a facade call: the product's rule kernel, compiled for the host under the lockstep 64-lane
    emulation, one method per rule-kernel op on a record / MT19937 state it holds like the 1-game batch does.  EmuCallBackend below puts the
    product's own host logic (facade_backend.HipBackend) on top of it.  Never selected by the product."""

    def __new__(cls, first_player, tile_pool, players=2, ext=0):
        return super().__new__(EmuBackendX if (int(players) != 2 or int(ext)) and cls is EmuBackend else cls)

    def __init__(self, first_player, tile_pool, players=2, ext=0):
        self.fp, self.pool = int(first_player), int(tile_pool)
        self.num_actions, self.obs_size = 180, 136
        self.rec = np.zeros(128, np.uint8)
        self.mt = np.zeros(624, np.uint32)
        self.pos = np.array([624], np.uint32)
        self._gauss = None

    def push_rng(self):
        import random
        st = random.getstate()
        self.mt[:] = np.array(st[1][:624], dtype=np.uint32)
        self.pos[0] = st[1][624]
        self._gauss = st[2]

    def pull_rng(self):
        import random
        random.setstate((3, tuple(int(x) for x in self.mt) + (int(self.pos[0]),), self._gauss))

    def put(self, rec):
        self.rec[:] = np.frombuffer(np.asarray(rec).tobytes(), np.uint8)

    def get(self):
        from azul_deep_reinforcement_learning_amd.records import RECORD_DTYPE
        return self.rec.copy().view(RECORD_DTYPE)[0]

    # ---- one rule call = one launch of azul_op_kernel (csrc/azul_ops2.hpp: op_body2) on the emulated wave: sh2_op ----
    OP = {"query": 0, "init": 1, "new_round": 2, "move": 3, "next_player": 4, "count_score": 5, "step": 6, "runner_init": 7,
          "runner_reset": 8, "runner_step": 9, "random_action": 10, "sample_mask": 11}
    want_next, last_next = False, -2       # AZUL_WANT_NEXT_ACTION of the call being interpreted / the emulated kernel's answer
    pos_set = None                         # AZUL_WANT_POS_IN: the index the next drawing op installs first

    def _op(self, op, action=0, mask_in=None, want_mask=False, want_obs=None, want_flags=False, want_potential=False, want_stats=False, margin=0):
        mask = np.zeros(180, np.uint8) if want_mask else None
        obs = np.zeros(136, np.float32) if want_obs is not None else None
        stats = np.zeros(10) if want_stats else None
        ints = {k: C.c_int(0) for k in ("flags", "potential", "reward", "done", "action", "player", "dirty")}
        nxt = C.c_int(-2)
        mi = None if mask_in is None else np.ascontiguousarray(mask_in, dtype=np.uint8)
        ref = lambda k: C.cast(C.byref(ints[k]), C.c_void_p)
        st = lib().sh2_op(ptr(self.rec), self.fp, self.pool, int(margin), self.OP[op], int(action), ptr(self.mt), ptr(self.pos),
                         0 if self.pos_set is None else 1 + int(self.pos_set), ptr(mi), ptr(mask), ptr(obs),
                         int(want_obs) if want_obs is not None else 0, ref("flags") if want_flags else None,
                         ref("potential") if want_potential else None, ptr(stats), ref("reward"), ref("done"), ref("action"), ref("player"),
                         ref("dirty"), C.cast(C.byref(nxt), C.c_void_p) if self.want_next else None, None, None, None, None)
        self.pos_set = None
        self.last_next = nxt.value
        out = {k: v.value for k, v in ints.items()}
        out.update(status=st, mask=mask, obs=obs, stats=stats)
        return out

    def op_init(self):
        self._op("init")

    def op_new_round(self):
        return self._op("new_round")["status"]

    def op_move(self, action):
        self._op("move", action)

    def op_next_player(self):
        self._op("next_player")

    def op_count_score(self):
        self._op("count_score")

    def op_step(self, action):
        return self._op("step", action)["status"]

    def op_flags(self):
        return self._op("query", want_flags=True)["flags"]

    def op_mask(self):
        return self._op("query", want_mask=True)["mask"].astype(bool)

    def op_observe(self, perspective):
        return self._op("query", want_obs=int(perspective))["obs"].astype(np.int64)

    def op_statistics(self):
        return self._op("query", want_stats=True)["stats"]

    def op_potential(self):
        return self._op("query", want_potential=True)["potential"]

    def op_runner_init(self):
        return self._op("runner_init")["status"]

    def op_runner_reset(self):
        return self._op("runner_reset")["status"]

    def op_runner_step(self, action):
        o = self._op("runner_step", action)
        return o["reward"], bool(o["done"]), o["status"]

    def op_sample_mask(self, mask):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).reshape(-1)[:180])
        return self._op("sample_mask", mask_in=m)["action"]


class EmuBackendX(EmuBackend):
    """Three / four players and extended rules (csrc/azul_rules_x.hpp: the body of azul_x_op_kernel under the lockstep wave emulation)
    behind the same interface: 256-byte wide records, Azul's own methods, the sampler, get_state."""

    def __init__(self, first_player, tile_pool, players, ext=0):
        super().__init__(first_player, tile_pool)
        self.players, self.ext = int(players), int(ext)
        D = 2 * self.players + 1 if self.ext & EXT_DISPLAYS_2P1 else 5
        self.num_actions, self.obs_size = (D + 1) * 30, 5 * D + 6 + 52 * self.players + 1
        self.rec = np.zeros(256, np.uint8)

    def get(self):
        from azul_deep_reinforcement_learning_amd.records import RECORD_NP_DTYPE
        return self.rec.copy().view(RECORD_NP_DTYPE)[0]

    def _op(self, op, action=0, **kw):
        out = x_op(self.rec, self.players, self.fp, self.pool, self.ext, op, action, self.mt, self.pos, want_next=self.want_next,
                   pos_set=self.pos_set, **kw)
        self.pos_set = None
        self.last_next = out["next_action"]
        return out

    def op_init(self):
        self._op("init")

    def op_new_round(self):
        return self._op("new_round")["status"]

    def op_move(self, action):
        self._op("move", action)

    def op_next_player(self):
        self._op("next_player")

    def op_count_score(self):
        self._op("count_score")

    def op_step(self, action):
        return self._op("step", action)["status"]

    def op_flags(self):
        return self._op("query", want_flags=True)["flags"]

    def op_mask(self):
        return self._op("query", want_mask=True)["mask"].astype(bool)

    def op_statistics(self):
        return self._op("query", want_stats=True)["stats"]

    def op_observe(self, perspective):
        return self._op("query", want_obs=int(perspective))["obs"].astype(np.int64)

    def op_sample_mask(self, mask):
        m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).reshape(-1))
        if m.size != self.num_actions:
            raise ValueError("this backend samples masks of %d actions" % self.num_actions)
        return self._op("sample_mask", mask_in=m)["action"]

    def _two_players_only(self, *a, **k):
        raise RuntimeError("GameRunner.step / reset and the what-if potential are two-player (game_runner.py:43-55, 76-85)")

    op_potential = op_runner_init = op_runner_reset = op_runner_step = _two_players_only


def _call_backend_class():
    """facade_backend.HipBackend with its ONE device-touching method (_game_call: azul_game_call on a 1-game batch) replaced by an
    interpreter of the same call block on the emulated device core above -- every other line (what is sent, what is remembered, when
    the global random stream moves) is the product's, so the CPU suite and the sanitizer passes run the facade's real host logic."""
    import ctypes as C
    import azul_deep_reinforcement_learning_amd.facade_backend as fb
    from azul_deep_reinforcement_learning_amd import _lib as L
    from azul_deep_reinforcement_learning_amd.records import RECORD_DTYPE, RECORD_NP_DTYPE

    NAMES = {L.CALL_QUERY: "query", L.CALL_INIT: "init", L.CALL_NEW_ROUND: "new_round", L.CALL_MOVE: "move", L.CALL_NEXT_PLAYER: "next_player",
             L.CALL_COUNT_SCORE: "count_score", L.CALL_STEP: "step", L.CALL_RUNNER_INIT: "runner_init", L.CALL_RUNNER_RESET: "runner_reset",
             L.CALL_RUNNER_STEP: "runner_step", L.CALL_SAMPLE_MASK: "sample_mask"}
    DRAWING = (L.CALL_INIT, L.CALL_NEW_ROUND, L.CALL_STEP, L.CALL_RUNNER_INIT, L.CALL_RUNNER_RESET, L.CALL_RUNNER_STEP, L.CALL_SAMPLE_MASK)

    class EmuCallBackend(fb.HipBackend):
        def __init__(self, first_player, tile_pool, players=2, ext=0):
            self.dev = EmuBackend(first_player, tile_pool, players, ext)
            self._setup(self.dev.num_actions, self.dev.obs_size, RECORD_DTYPE if self.dev.rec.size == 128 else RECORD_NP_DTYPE)
            self.calls = 0

        def _game_call(self):          # (include/azul_hip.h: azul_game_call, statement by statement)
            c, e = self.c, self.dev
            self.calls += 1
            NA, RB = self.num_actions, e.rec.size
            draws = c.op in DRAWING
            if c.record_in:
                e.rec[:] = np.frombuffer(C.string_at(c.record_in, RB), np.uint8)
            if c.mt_in:
                e.mt[:] = np.frombuffer(C.string_at(c.mt_in, 2496), np.uint32)
                e.pos[0] = c.pos_in
            elif (c.want & L.WANT_POS_IN) and draws:
                e.pos_set = int(c.pos_in)                  # (the kernel body installs it: OpArgs::pos_set / XOp::pos_set)
            mt0 = e.mt.copy()
            # ONE call of the kernel body, like the product's one launch: the op, then the queries on the state it leaves
            name = NAMES[c.op]
            if isinstance(e, EmuBackendX) and (name.startswith("runner") or c.want & L.WANT_POTENTIAL):
                e._two_players_only()                       # (three / four players: refused like the library refuses it)
            kw = {"want_mask": bool(c.want & L.WANT_MASK), "want_flags": bool(c.want & L.WANT_FLAGS), "want_stats": bool(c.want & L.WANT_STATS),
                  "want_obs": (int(c.arg if c.op == L.CALL_QUERY else c.obs_persp) if c.want & L.WANT_OBS else None)}
            if c.want & L.WANT_POTENTIAL:
                kw["want_potential"] = True
            if c.op == L.CALL_SAMPLE_MASK:
                kw["mask_in"] = np.frombuffer(C.string_at(c.mask_in, NA), np.uint8)
            e.want_next = bool(c.want & L.WANT_NEXT_ACTION)
            out = e._op(name, c.arg, **kw)
            e.want_next = False
            c.status = int(out["status"])
            c.reward, c.done = (int(out.get("reward", 0)), int(out.get("done", 0))) if c.op == L.CALL_RUNNER_STEP else (0, 0)
            c.action = int(out["action"]) if c.op == L.CALL_SAMPLE_MASK else 0
            base = C.addressof(c)
            if c.want & L.WANT_MASK:
                C.memmove(base + L.AzulCall.mask.offset, np.ascontiguousarray(out["mask"], dtype=np.uint8).ctypes.data, NA)
            if c.want & L.WANT_OBS:
                C.memmove(base + L.AzulCall.obs.offset, np.ascontiguousarray(out["obs"], dtype=np.float32).ctypes.data, 4 * self.obs_size)
            c.flags = int(out["flags"]) if c.want & L.WANT_FLAGS else 0
            c.potential = int(out.get("potential", 0)) if c.want & L.WANT_POTENTIAL else 0
            if c.want & L.WANT_STATS:
                C.memmove(base + L.AzulCall.stats.offset, np.ascontiguousarray(out["stats"], dtype=np.float64).ctypes.data, 80)
            if c.want & L.WANT_RECORD:
                C.memmove(c.record_out, e.rec.ctypes.data, RB)
            c.next_action = int(e.last_next) if (c.want & L.WANT_NEXT_ACTION) else -2
            c.pos_out = int(e.pos[0])
            c.rng_regenerated = int(bool((e.mt != mt0).any()))
            if c.rng_regenerated and c.mt_out:
                C.memmove(c.mt_out, e.mt.ctypes.data, 2496)

    return EmuCallBackend


_ECB = None


def call_backend_class():
    global _ECB
    if _ECB is None:
        _ECB = _call_backend_class()
    return _ECB
