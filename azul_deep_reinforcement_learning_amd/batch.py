"""BatchedAzul: N concurrent Azul games (two players like the reference's GameRunner; three / four players and extended rules for the
rule methods and flat self-play) resident on one MI355X.

Host-side mirror of the reference's env wrapper for the batched case: the method names follow
``azulnet.GameRunner`` (``reset`` / ``step`` / ``get_state`` / ``get_valid_moves``; reference
azulnet/game_runner.py:43-85) and ``azulnet.Azul`` (``new_round`` / ``move`` / ``count_score`` / ...;
azulnet/azul.py), every call is one kernel launch over all games through the C ABI of libazulhip.so.
PyTorch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .records import RECORD_DTYPE, RECORD_NP_DTYPE, STAT_KEYS

_RULE_POOL = {"Random": L.POOL_RANDOM, "Lid": L.POOL_LID}
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # current stream of a device as an integer handle


class IllegalRule(Exception):
    pass


def parse_rules(rules, players=2):
    """rules dict of the reference (azul.py:35-56) -> (first_player code, tile_pool code)."""
    first = 1
    if "first_player" in rules:
        fp = rules["first_player"]
        if fp == "Random":
            first = L.FIRST_RANDOM
        elif type(fp) == int and 1 <= fp <= players:
            first = fp
        else:
            raise IllegalRule
    pool = L.POOL_RANDOM
    if "tile_pool" in rules:
        if rules["tile_pool"] not in _RULE_POOL:
            raise IllegalRule
        pool = _RULE_POOL[rules["tile_pool"]]
    return first, pool


def parse_ext_rules(rules, players=2):
    """The extended-rule keys of a rules dict -> AZUL_RULE_* flags.  BEYOND THE REFERENCE ("parity unpinned"): the reference knows none of
    these keys and ignores unknown ones (azul.py:35-56); all default to the reference's behaviour.
        "displays": 5 (default) | "2P+1" (or the number 2 * players + 1)    the rulebook's 5 / 7 / 9 factory displays
        "bonuses": "round" (default, azul.py:266-288) | "end"               +2 / +7 / +10 once, when the game has ended
        "short_deal": False | True                                          bag and lid empty: deal what is left instead of raising
        "finite_bag": False | True                                          tile_pool "Random" draws from a 100-tile bag"""
    flags = 0
    d = rules.get("displays", 5)
    if d == "2P+1" or (type(d) == int and d == 2 * players + 1 and d != 5):
        flags |= L.RULE_DISPLAYS_2P1
    elif d != 5:
        raise IllegalRule
    b = rules.get("bonuses", "round")
    if b == "end":
        flags |= L.RULE_END_BONUS
    elif b != "round":
        raise IllegalRule
    if rules.get("short_deal", False):
        flags |= L.RULE_SHORT_DEAL
    if rules.get("finite_bag", False):
        if rules.get("tile_pool", "Random") != "Random":
            raise IllegalRule
        flags |= L.RULE_FINITE_BAG
    return flags


def ext_rules_dict(flags):
    """AZUL_RULE_* flags -> the rules-dict keys parse_ext_rules reads."""
    d = {}
    if flags & L.RULE_DISPLAYS_2P1:
        d["displays"] = "2P+1"
    if flags & L.RULE_END_BONUS:
        d["bonuses"] = "end"
    if flags & L.RULE_SHORT_DEAL:
        d["short_deal"] = True
    if flags & L.RULE_FINITE_BAG:
        d["finite_bag"] = True
    return d


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _DevArray:
    """A device array owned by the library, exposed through __cuda_array_interface__ so torch can view it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _dev_view(ptr, shape, typestr, device):
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


class BatchedAzul:
    def __init__(self, n_games, rules={"first_player": "Random", "tile_pool": "Lid"}, device=None, seed=None, players=2, ext_rules=None):
        """players = 3 or 4: the reference's Azul(players=...) (five displays, azul.py:19).  Extended rules (beyond the reference,
        "parity unpinned") through the rules dict (parse_ext_rules: "displays": "2P+1", "bonuses": "end", "short_deal", "finite_bag")
        or as AZUL_RULE_* flags in `ext_rules`.  Batches of 3 / 4 players and extended-rule batches hold 256-byte wide records
        (records.RECORD_NP_DTYPE) and support the Azul rule methods, the sampler, masks, observations and selfplay -- not the
        two-player GameRunner / policy entries."""
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedAzul needs an MI355X: there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = int(n_games)
        self.rules = dict(rules)
        self.players = int(players)
        first, pool = parse_rules(rules, self.players)
        self.ext = parse_ext_rules(rules, self.players) | (int(ext_rules) if ext_rules else 0)
        if (self.ext & L.RULE_FINITE_BAG) and pool != L.POOL_RANDOM:
            raise IllegalRule
        self.rules.update(ext_rules_dict(self.ext))
        self.wide = self.players != 2 or self.ext != 0
        self.record_dtype = RECORD_NP_DTYPE if self.wide else RECORD_DTYPE
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib.azul_batch_create_rules(C.byref(self._h), self.n, self.players, first, pool, self.ext))
        self.displays = int(L.lib.azul_batch_displays(self._h))
        self.num_actions = int(L.lib.azul_batch_num_actions(self._h))
        self.obs_size = int(L.lib.azul_batch_obs_size(self._h))
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:   # module globals may be gone at interpreter exit
            L.lib.azul_batch_destroy(h)
            self._h = None

    # -- plumbing --------------------------------------------------------------------------------
    def _stream(self):
        if _RAW_STREAM is not None and self.device.index is not None:
            return C.c_void_p(_RAW_STREAM(self.device.index))      # the same handle, without building a torch.cuda.Stream per call
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _dev(self, x, dtype):
        if x is None:
            return None
        t = torch.as_tensor(x)
        return t.to(device=self.device, dtype=dtype).contiguous()

    # -- RNG / state I/O -------------------------------------------------------------------------
    def seed(self, seed_base=0, seeds=None):
        """random.seed(seeds[g]) (or seed_base + g) for every game's private CPython-exact stream."""
        arr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        if arr is not None and arr.shape != (self.n,):
            raise ValueError("seeds must have shape (n_games,)")
        L.check(L.lib.azul_batch_seed(self._h, int(seed_base), None if arr is None else arr.ctypes.data_as(C.c_void_p), self._stream()))

    def get_records(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=self.record_dtype)
        L.check(L.lib.azul_batch_get_state(self._h, first, count, out.ctypes.data_as(C.c_void_p), self._stream()))
        return out

    def set_records(self, records, first=0):
        rec = np.ascontiguousarray(records, dtype=self.record_dtype).reshape(-1)
        L.check(L.lib.azul_batch_set_state(self._h, first, len(rec), rec.ctypes.data_as(C.c_void_p), self._stream()))

    def get_rng(self, game):
        mt = np.zeros(624, dtype=np.uint32)
        pos = np.zeros(1, dtype=np.uint32)
        L.check(L.lib.azul_batch_get_rng(self._h, game, mt.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), self._stream()))
        return mt, int(pos[0])

    def set_rng(self, game, mt, pos):
        mt = np.ascontiguousarray(mt, dtype=np.uint32)
        L.check(L.lib.azul_batch_set_rng(self._h, game, mt.ctypes.data_as(C.c_void_p), int(pos), self._stream()))

    def get_rng_range(self, first=0, count=None):
        """(mt [count][624] uint32, pos [count] uint32): every game's random.getstate() in one copy (checkpoints)."""
        count = self.n - first if count is None else count
        mt = np.zeros((count, 624), dtype=np.uint32)
        pos = np.zeros(count, dtype=np.uint32)
        L.check(L.lib.azul_batch_get_rng_range(self._h, first, count, mt.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), self._stream()))
        return mt, pos

    def set_rng_range(self, mt, pos, first=0):
        mt = np.ascontiguousarray(mt, dtype=np.uint32).reshape(-1, 624)
        pos = np.ascontiguousarray(pos, dtype=np.uint32).reshape(-1)
        if len(mt) != len(pos):
            raise ValueError("mt and pos disagree on the number of games")
        L.check(L.lib.azul_batch_set_rng_range(self._h, first, len(pos), mt.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), self._stream()))

    def export_json(self, path=None, first=0, count=None):
        """Games as a list of dicts in the reference's JSON schema (azul.py:90-117) + the fields the schema lacks (`x_*`)."""
        from .records import record_to_json
        data = [record_to_json(r) for r in self.get_records(first, count)]
        if path is not None:
            import json
            with open(path, "w") as fh:
                json.dump(data, fh)
        return data

    def import_json(self, src, first=0):
        """`src`: a path or the list export_json returns; games first.. are overwritten (RNG streams are left alone)."""
        from .records import json_to_record
        if isinstance(src, (str, bytes)):
            import json
            with open(src) as fh:
                src = json.load(fh)
        self.set_records(np.array([json_to_record(d) for d in src], dtype=self.record_dtype), first)

    # -- Azul methods, batched --------------------------------------------------------------------
    def init(self, active=None):
        L.check(L.lib.azul_batch_init(self._h, _ptr(self._dev(active, torch.uint8)), self._stream()))

    def new_round(self, active=None):
        st = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        L.check(L.lib.azul_batch_new_round(self._h, _ptr(self._dev(active, torch.uint8)), _ptr(st), self._stream()))
        return st

    def move(self, actions, active=None):
        a = self._dev(actions, torch.int32)
        L.check(L.lib.azul_batch_move(self._h, _ptr(a), _ptr(self._dev(active, torch.uint8)), self._stream()))

    def next_player(self, active=None):
        L.check(L.lib.azul_batch_next_player(self._h, _ptr(self._dev(active, torch.uint8)), self._stream()))

    def count_score(self, active=None):
        L.check(L.lib.azul_batch_count_score(self._h, _ptr(self._dev(active, torch.uint8)), self._stream()))

    def flags(self):
        f = self._new((self.n,), torch.uint8)
        L.check(L.lib.azul_batch_flags(self._h, _ptr(f), self._stream()))
        return f

    def is_end_of_round(self):
        return (self.flags() & L.FLAG_END_OF_ROUND) != 0

    def is_end_of_game(self):
        return (self.flags() & L.FLAG_END_OF_GAME) != 0

    def azul_step(self, actions, active=None):
        """Azul.step for every game; returns the uint8 status vector (OK / ILLEGAL_MOVE / GAME_ENDED / ...)."""
        a = self._dev(actions, torch.int32)
        st = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        L.check(L.lib.azul_batch_step(self._h, _ptr(a), _ptr(self._dev(active, torch.uint8)), _ptr(st), self._stream()))
        return st

    def statistics(self):
        s = self._new((self.n, L.NUM_STATS), torch.float64)
        L.check(L.lib.azul_batch_statistics(self._h, _ptr(s), self._stream()))
        return s

    # -- GameRunner methods, batched --------------------------------------------------------------
    def runner_init(self, active=None):
        st = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        L.check(L.lib.azul_batch_runner_init(self._h, _ptr(self._dev(active, torch.uint8)), _ptr(st), self._stream()))
        return st

    def reset(self, active=None):
        st = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        L.check(L.lib.azul_batch_runner_reset(self._h, _ptr(self._dev(active, torch.uint8)), _ptr(st), self._stream()))
        return st

    def step(self, actions, active=None):
        """GameRunner.step for every game -> (reward int32[N], done bool[N], status uint8[N])."""
        a = self._dev(actions, torch.int32)
        reward = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        done = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        st = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
        L.check(L.lib.azul_batch_runner_step(self._h, _ptr(a), _ptr(self._dev(active, torch.uint8)), _ptr(reward), _ptr(done), _ptr(st), self._stream()))
        return reward, done.bool(), st

    def get_state(self, perspective=0, out=None):
        obs = self._new((self.n, self.obs_size), torch.float32) if out is None else out
        L.check(L.lib.azul_batch_observe(self._h, int(perspective), _ptr(obs), self._stream()))
        return obs

    def get_valid_moves(self, out=None):
        m = self._new((self.n, self.num_actions), torch.uint8) if out is None else out
        L.check(L.lib.azul_batch_legal_mask(self._h, _ptr(m), self._stream()))
        return m if out is not None else m.bool()

    def random_action(self, active=None):
        a = torch.full((self.n,), -1, dtype=torch.int32, device=self.device)
        L.check(L.lib.azul_batch_random_action(self._h, _ptr(self._dev(active, torch.uint8)), _ptr(a), self._stream()))
        return a

    def sample_mask(self, mask, active=None):
        """RandomAgent.get_a_output for caller-supplied masks [N][num_actions] (one random.choices draw per game)."""
        m = self._dev(mask, torch.uint8)
        if tuple(m.shape) != (self.n, self.num_actions):
            raise ValueError("mask must be [N][%d]" % self.num_actions)
        a = torch.full((self.n,), -1, dtype=torch.int32, device=self.device)
        L.check(L.lib.azul_batch_sample_mask(self._h, _ptr(m), _ptr(self._dev(active, torch.uint8)), _ptr(a), self._stream()))
        return a

    def score_preview(self):
        p = self._new((self.n,), torch.int32)
        L.check(L.lib.azul_batch_score_preview(self._h, _ptr(p), self._stream()))
        return p

    # -- policy-driven self-play (config 3) ----------------------------------------------------------
    def observe_all(self, perspective=L.PERSP_CURRENT, obs=None, mask=None, player=None):
        obs = self._new((self.n, self.obs_size), torch.float32) if obs is None else obs
        mask = self._new((self.n, self.num_actions), torch.uint8) if mask is None else mask
        if self.wide and perspective == L.PERSP_CURRENT:
            perspective = L.PERSP_MOVER
        player = self._new((self.n,), torch.uint8) if player is None else player
        L.check(L.lib.azul_batch_observe_all(self._h, int(perspective), _ptr(obs), _ptr(mask), _ptr(player), self._stream()))
        return obs, mask, player

    def policy_step(self, actions, reward, done, status, obs_next, mask_next, player_next, perspective=L.PERSP_CURRENT, active=None):
        """One fused env move with caller-chosen actions (all arguments are preallocated device tensors)."""
        L.check(L.lib.azul_batch_policy_step(self._h, _ptr(actions), _ptr(self._dev(active, torch.uint8)), _ptr(reward), _ptr(done),
                                             _ptr(status), int(perspective), _ptr(obs_next), _ptr(mask_next), _ptr(player_next),
                                             self._stream()))

    def agent_step(self, actions, reward, done, status, obs_next, mask_next, player_next=None, perspective=0, active=None):
        """One AGENT step of NNRunner.run_episode (GameRunner.step incl. the opponent's replies; GameRunner.reset() when
        the episode ends) + the next decision's observation / mask; all arguments are preallocated device tensors."""
        L.check(L.lib.azul_batch_agent_step(self._h, _ptr(actions), _ptr(self._dev(active, torch.uint8)), _ptr(reward), _ptr(done),
                                            _ptr(status), int(perspective), _ptr(obs_next), _ptr(mask_next), _ptr(player_next),
                                            self._stream()))

    # -- GameRunner with an external (network) opponent, cut at its opponent_move() calls (game_runner.py:27-30, 37-47, 84-85) ------
    def net_state(self):
        """Preallocated device tensors of the protocol: pending / replies [N] u8, owing [1] (int32 view of the ABI's uint32), the
        opponent's observation [N][136] / mask [N][180] and its answer [N]."""
        d = self.device
        return {"pending": torch.zeros(self.n, dtype=torch.uint8, device=d), "replies": torch.zeros(self.n, dtype=torch.uint8, device=d),
                "owing": torch.zeros(1, dtype=torch.int32, device=d), "obs": torch.zeros(self.n, self.obs_size, device=d),
                "mask": torch.zeros(self.n, self.num_actions, dtype=torch.uint8, device=d),
                "action": torch.zeros(self.n, dtype=torch.int32, device=d)}

    def net_step_begin(self, actions, net, reward, done, status):
        """The agent's move of GameRunner.step (game_runner.py:44-45) for every game, then the loop condition (:46): net["pending"] says
        who owes an opponent_move(), net["obs"] / net["mask"] what it is handed, net["owing"] how many games owe one."""
        L.check(L.lib.azul_batch_net_step_begin(self._h, _ptr(actions), _ptr(net["pending"]), _ptr(net["replies"]), _ptr(reward), _ptr(done),
                                                _ptr(status), _ptr(net["obs"]), _ptr(net["mask"]), _ptr(net["owing"]), self._stream()))

    def net_step_reply(self, opp_actions, net, reward, done, status):
        """One opponent_move() (game_runner.py:37-42) with `opp_actions` for every game that owes one, then the loop condition again."""
        L.check(L.lib.azul_batch_net_step_reply(self._h, _ptr(opp_actions), _ptr(net["pending"]), _ptr(net["replies"]), _ptr(reward), _ptr(done),
                                                _ptr(status), _ptr(net["obs"]), _ptr(net["mask"]), _ptr(net["owing"]), self._stream()))

    def net_reset_begin(self, net, status, active=None):
        """GameRunner.reset() (game_runner.py:76-85) up to its first opponent_move()."""
        L.check(L.lib.azul_batch_net_reset_begin(self._h, _ptr(self._dev(active, torch.uint8)), _ptr(net["pending"]), _ptr(status), _ptr(net["obs"]),
                                                 _ptr(net["mask"]), _ptr(net["owing"]), self._stream()))

    # -- flat self-play rollout -------------------------------------------------------------------
    def selfplay(self, n_steps, mask=None, action=None, reward=None, done=None, records=None, maskbits=None, packed=None):
        """`n_steps` env moves for every game in one launch; outputs are preallocated tensors or None.  `mask` may be a
        [T][N][180] view of a wider [T][N][pitch] buffer (alloc_trajectory(mask_pitch=192)): the row pitch is taken from its
        strides.  Batches of 3 or 4 players and extended-rule batches play the same flat loop (mask -> RandomAgent -> Azul.step, a
        fresh Azul + new_round() at each game end: start them with init() + new_round()); their `reward` stream is all zero (the shaped
        reward is GameRunner's, two players: game_runner.py:50), `records` rows are 256-byte wide records, mask rows num_actions bytes
        (a padded pitch -- 192 / 256 / 320 for 5 / 7 / 9 displays -- takes the one-store-per-row path)."""
        pitch = self.num_actions
        if mask is not None:
            pitch = mask.stride(-2)
            if mask.stride(-1) != 1 or mask.shape[-1] != self.num_actions or (mask.dim() == 3 and mask.shape[0] > 1 and mask.stride(0) != self.n * pitch):
                raise ValueError("mask must be [T][N][%d] uint8 with contiguous rows and a uniform row pitch" % self.num_actions)
        L.check(L.lib.azul_batch_selfplay_strided(self._h, int(n_steps), _ptr(mask), int(pitch), _ptr(maskbits), _ptr(action), _ptr(reward),
                                                  _ptr(done), _ptr(packed), _ptr(records), self._stream()))

    def alloc_trajectory(self, n_steps, with_records=False, packed_mask=False, mask_pitch=None, mask_bits=None):
        """Trajectory buffers [n_steps][N]....  With `packed_mask` also the bit-packed mask `maskbits` (int64 [T][N][3])
        and the compact per-move record `packed` (int32 [T][N]: action | done << 8 | reward << 16) -- the two
        contiguous arrays the multi-GPU all-gather ships.  `mask_pitch` (e.g. 192): the byte mask is a [T][N][180] view of a
        [T][N][mask_pitch] buffer, so that every game's row starts 64-byte aligned (whole-sector stores)."""
        n = self.n
        NA = self.num_actions
        pitch = NA if mask_pitch is None else int(mask_pitch)
        store = torch.zeros((n_steps, n, pitch), dtype=torch.uint8, device=self.device) if pitch != NA else \
            self._new((n_steps, n, NA), torch.uint8)
        t = {"mask": store[:, :, :NA],
             "action": self._new((n_steps, n), torch.int32),
             "reward": self._new((n_steps, n), torch.int32),
             "done": self._new((n_steps, n), torch.uint8)}
        if packed_mask:
            if mask_bits is None or mask_bits:          # mask_bits=False: only the compact record (what the all-gather ships by default)
                t["maskbits"] = torch.zeros((n_steps, n, (NA + 63) // 64), dtype=torch.int64, device=self.device)
            t["packed"] = torch.zeros((n_steps, n), dtype=torch.int32, device=self.device)
        if with_records:
            # zeroed: a 3-player batch leaves the fourth player's bytes and the reserved tail of a wide record untouched
            t["records"] = torch.zeros((n_steps, n, self.record_dtype.itemsize), dtype=torch.uint8, device=self.device)
        return t

    def counters(self):
        ep = np.zeros(self.n, dtype=np.uint64)
        stuck = np.zeros(self.n, dtype=np.uint32)
        ss = np.zeros((self.n, L.NUM_STATS), dtype=np.float64)
        L.check(L.lib.azul_batch_counters(self._h, ep.ctypes.data_as(C.c_void_p), stuck.ctypes.data_as(C.c_void_p),
                                          ss.ctypes.data_as(C.c_void_p), self._stream()))
        return {"episodes": ep, "stuck": stuck, "stat_sums": ss, "keys": list(STAT_KEYS)}

    def counters_dev(self):
        """Zero-copy torch views of the device-resident counters (no synchronisation): episodes int64 [N], stuck int32 [N],
        stat_sums float64 [N][10].  The views alias the batch's arrays: keep the batch alive while they are in use."""
        ep, stuck, ss = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib.azul_batch_counters_dev(self._h, C.byref(ep), C.byref(stuck), C.byref(ss)))
        return {"episodes": _dev_view(ep.value, (self.n,), "<i8", self.device),
                "stuck": _dev_view(stuck.value, (self.n,), "<i4", self.device),
                "stat_sums": _dev_view(ss.value, (self.n, L.NUM_STATS), "<f8", self.device), "keys": list(STAT_KEYS)}

    def records_dev(self):
        """Zero-copy torch view of the device-resident game records, uint8 [N][record bytes] (azul_batch_state_dev; no synchronisation).  A
        caller that writes through it keeps the records in the domain set_records validates AND within what play can produce: unlike
        set_records it does not switch the batch's flat self-play to the instantiation that marks the slots of a rule-error-stopped game."""
        return _dev_view(L.lib.azul_batch_state_dev(self._h), (self.n, self.record_dtype.itemsize), "|u1", self.device)

    def reset_counters(self):
        L.check(L.lib.azul_batch_reset_counters(self._h, self._stream()))

    def set_id_base(self, first_global_id):
        """Global id of this batch's game 0 (multi-GPU shards / stream parts): keys the policy sampler's Philox stream."""
        L.check(L.lib.azul_batch_set_id_base(self._h, int(first_global_id) & 0xFFFFFFFF))

    def set_move_limit(self, max_moves):
        """Beyond the reference, off by default (0): cut an episode at the first end of a round with move_counter >= max_moves -- some games
        never end under the reference's rules (include/azul_hip.h: azul_batch_set_move_limit).  done = 3 marks the cut."""
        L.check(L.lib.azul_batch_set_move_limit(self._h, int(max_moves)))

    def set_draw_margin(self, margin):
        """Test knob: widen the window in which the factory draw falls back to the literal fp64 computation."""
        L.check(L.lib.azul_batch_set_draw_margin(self._h, int(margin)))

    def kernel_resources(self, padded_rows=True, mask_bits=False):
        """Registers / LDS / scratch of the flat self-play kernel for this output shape and how many of its waves a CU holds at once
        (the HIP runtime's occupancy calculator on the loaded code object)."""
        v, lds, sc, wv = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(L.lib.azul_selfplay_kernel_resources(self._h, int(padded_rows), int(mask_bits), C.byref(v), C.byref(lds), C.byref(sc), C.byref(wv)))
        return {"vgprs": v.value, "lds_bytes": lds.value, "scratch_bytes": sc.value, "resident_waves_per_cu": wv.value,
                "resident_waves_per_simd": wv.value / 4.0}

    def clock_probe(self, out, spin_iterations=20000):
        """Enqueue the on-device shader-clock probe; `out`: int64[3] device tensor -> {shader cycles, 100 MHz ticks, chain result}."""
        L.check(L.lib.azul_device_clock_probe(_ptr(out), int(spin_iterations), self._stream()))

    def timing_begin(self):
        L.check(L.lib.azul_timing_begin(self._h, self._stream()))

    def timing_end(self):
        """-> (bracket ms, launches inside it, summed per-launch kernel ms, launches that carried their own event pair)."""
        ms, n, kms, kn = C.c_float(0), C.c_int(0), C.c_float(0), C.c_int(0)
        L.check(L.lib.azul_timing_end(self._h, self._stream(), C.byref(ms), C.byref(n), C.byref(kms), C.byref(kn)))
        return float(ms.value), int(n.value), float(kms.value), int(kn.value)

    def timing_launch_ms(self):
        """After timing_end: the durations (ms) of the launches of that timed region that carried their own event pair (the first 1024)."""
        n = C.c_int(0)
        L.check(L.lib.azul_timing_launch_ms(self._h, None, 0, C.byref(n)))
        out = (C.c_float * max(n.value, 1))()
        L.check(L.lib.azul_timing_launch_ms(self._h, out, n.value, C.byref(n)))
        return [float(out[i]) for i in range(n.value)]
