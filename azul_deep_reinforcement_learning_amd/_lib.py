"""ctypes binding of libazulhip.so (C ABI: include/azul_hip.h).

The library is the product: there is NO fallback.  If it is missing or fails to load, importing this
module raises -- build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc, gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libazulhip.so")

RECORD_BYTES, NUM_ACTIONS, OBS_SIZE, MT_WORDS, NUM_STATS = 128, 180, 136, 624, 10
RECORD_BYTES_WIDE = 256         # batches of 3 or 4 players and extended-rule batches
MAX_ACTIONS, MAX_OBS = 300, 260
# extended rules (beyond the reference, "parity unpinned": include/azul_hip.h AZUL_RULE_*)
RULE_DISPLAYS_2P1, RULE_END_BONUS, RULE_SHORT_DEAL, RULE_FINITE_BAG = 1, 2, 4, 8
SUCCESS, ERR_INVALID, ERR_HIP, ERR_RANGE, ERR_RULE = 0, -1, -2, -3, -4
OK, ILLEGAL_MOVE, GAME_ENDED, STUCK, BAD_ACTION, BOX_EMPTY, TRUNCATED = 0, 1, 2, 3, 4, 5, 6
POOL_RANDOM, POOL_LID = 0, 1
FIRST_RANDOM = 0
PERSP_PLAYER0, PERSP_PLAYER1, PERSP_CURRENT = 0, 1, 2
PERSP_MOVER = 7
FLAG_END_OF_ROUND, FLAG_END_OF_GAME, FLAG_ENDED_FLAG = 1, 2, 4
A2C_FLAT_SIZE = 82082          # k-major flat layout of the parameters / gradient / Adam moments (82081 + 1 pad)
POLICY_ARGMAX = 0xFFFFFFFFFFFFFFFF        # `seed` value: np.argmax instead of sampling (agent.py action_selection="Max")

_vp, _i, _u64, _u32 = C.c_void_p, C.c_int, C.c_uint64, C.c_uint32

# azul_game_call (include/azul_hip.h): ops, result bits and the call block
(CALL_QUERY, CALL_INIT, CALL_NEW_ROUND, CALL_MOVE, CALL_NEXT_PLAYER, CALL_COUNT_SCORE, CALL_STEP, CALL_RUNNER_INIT, CALL_RUNNER_RESET,
 CALL_RUNNER_STEP, CALL_SAMPLE_MASK) = range(11)
WANT_RECORD, WANT_MASK, WANT_OBS, WANT_FLAGS, WANT_POTENTIAL, WANT_STATS, WANT_NEXT_ACTION, WANT_POS_IN = 1, 2, 4, 8, 16, 32, 64, 128


class AzulCall(C.Structure):
    _fields_ = [("op", C.c_int32), ("game", C.c_int32), ("arg", C.c_int32), ("want", C.c_uint32),
                ("record_in", C.c_void_p), ("mt_in", C.c_void_p), ("pos_in", C.c_uint32), ("mask_in", C.c_void_p),
                ("record_out", C.c_void_p), ("mt_out", C.c_void_p),
                ("pos_out", C.c_uint32), ("rng_regenerated", C.c_int32), ("status", C.c_int32), ("reward", C.c_int32), ("done", C.c_int32),
                ("action", C.c_int32), ("flags", C.c_int32), ("potential", C.c_int32), ("next_action", C.c_int32), ("obs_persp", C.c_int32),
                ("mask", C.c_uint8 * (MAX_ACTIONS + 4)), ("obs", C.c_float * MAX_OBS), ("stats", C.c_double * 10)]


class NetWeights(C.Structure):           # azul_net_weights_t
    _fields_ = [(k, C.c_void_p) for k in ("w1t", "b1", "w2c", "b2c", "w2a_t", "b2a")]


class RolloutBuffers(C.Structure):       # azul_rollout_buffers_t
    _fields_ = [(k, C.c_void_p) for k in ("obs", "mask", "player", "action", "reward", "done", "value", "logp", "entropy", "status", "returns",
                                          "opp_action", "opp_logp", "opp_replies")] + [("opp_slots", C.c_int)]


NET_READY, NET_REPLY, NET_OPENING = 0, 1, 2      # pending_dev of the azul_batch_net_* entries

# name -> (restype, argtypes); every symbol declared in include/azul_hip.h
SIGNATURES = {
    "azul_last_error_string": (C.c_char_p, []),
    "azul_version": (C.c_char_p, []),
    "azul_batch_create": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "azul_batch_create_players": (_i, [C.POINTER(_vp), _i, _i, _i, _i]),
    "azul_batch_create_rules": (_i, [C.POINTER(_vp), _i, _i, _i, _i, C.c_uint]),
    "azul_batch_players": (_i, [_vp]),
    "azul_batch_displays": (_i, [_vp]),
    "azul_batch_rule_flags": (C.c_uint, [_vp]),
    "azul_batch_num_actions": (_i, [_vp]),
    "azul_batch_obs_size": (_i, [_vp]),
    "azul_batch_record_bytes": (_i, [_vp]),
    "azul_batch_destroy": (_i, [_vp]),
    "azul_batch_size": (_i, [_vp]),
    "azul_batch_state_dev": (_vp, [_vp]),
    "azul_batch_mt_dev": (_vp, [_vp]),
    "azul_batch_mtpos_dev": (_vp, [_vp]),
    "azul_batch_get_state": (_i, [_vp, _i, _i, _vp, _vp]),
    "azul_batch_set_state": (_i, [_vp, _i, _i, _vp, _vp]),
    "azul_batch_get_rng": (_i, [_vp, _i, _vp, _vp, _vp]),
    "azul_batch_set_rng": (_i, [_vp, _i, _vp, _u32, _vp]),
    "azul_batch_get_rng_range": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "azul_batch_set_rng_range": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "azul_batch_seed": (_i, [_vp, _u64, _vp, _vp]),
    "azul_batch_init": (_i, [_vp, _vp, _vp]),
    "azul_batch_new_round": (_i, [_vp, _vp, _vp, _vp]),
    "azul_batch_move": (_i, [_vp, _vp, _vp, _vp]),
    "azul_batch_legal_mask": (_i, [_vp, _vp, _vp]),
    "azul_batch_next_player": (_i, [_vp, _vp, _vp]),
    "azul_batch_flags": (_i, [_vp, _vp, _vp]),
    "azul_batch_count_score": (_i, [_vp, _vp, _vp]),
    "azul_batch_step": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "azul_batch_statistics": (_i, [_vp, _vp, _vp]),
    "azul_batch_runner_init": (_i, [_vp, _vp, _vp, _vp]),
    "azul_batch_runner_reset": (_i, [_vp, _vp, _vp, _vp]),
    "azul_batch_runner_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azul_batch_observe": (_i, [_vp, _i, _vp, _vp]),
    "azul_batch_random_action": (_i, [_vp, _vp, _vp, _vp]),
    "azul_batch_policy_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "azul_batch_agent_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "azul_batch_observe_all": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "azul_policy_head": (_i, [_vp, _vp, _u64, _u64, _vp, _i, _u32, _vp, _vp, _vp, _vp]),
    "azul_policy_forward": (_i, [_vp] * 8 + [_i, _i, _i, _u64, _u64, _vp, _i, _i, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azul_batch_policy_rollout": (_i, [_vp, _i, _i] + [_vp] * 6 + [_i, _i, _i, _u64, _u64, _vp] + [_vp] * 10 + [_vp]),
    "azul_batch_policy_rollout_returns": (_i, [_vp, _i, _i] + [_vp] * 6 + [_i, _i, _i, _u64, _u64, _vp] + [_vp] * 10 + [_vp, C.c_float, _vp]),
    "azul_batch_policy_rollout_vs": (_i, [_vp, _i, C.POINTER(NetWeights), C.POINTER(NetWeights), _i, _i, _i, _u64, _u64, _u64, _vp,
                                          C.POINTER(RolloutBuffers), C.c_float, _vp]),
    "azul_batch_net_step_begin": (_i, [_vp] * 11),
    "azul_batch_net_step_reply": (_i, [_vp] * 11),
    "azul_batch_net_reset_begin": (_i, [_vp] * 8),
    "azul_a2c_gradients": (_i, [_vp, _vp, _vp, _vp, _i, C.c_float] + [_vp] * 7 + [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "azul_a2c_apply_adam": (_i, [_vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _i] + [_vp] * 8 + [_vp, _vp, C.c_float, _vp, _vp]),
    "azul_select_episode_samples": (_i, [_vp, _vp, _i, _i, _i, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azul_select_complete_samples": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "azul_discounted_returns": (_i, [_vp, _vp, _vp, _vp, C.c_float, _i, _i, _vp]),
    "azul_discounted_returns_ring": (_i, [_vp, _vp, _vp, C.c_float, _i, C.c_int64, _i, _i, _vp]),
    "azul_batch_sample_mask": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "azul_batch_score_preview": (_i, [_vp, _vp, _vp]),
    "azul_game_call": (_i, [_vp, C.POINTER(AzulCall), _vp]),
    "azul_batch_selfplay": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azul_batch_selfplay_strided": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "azul_batch_counters": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "azul_batch_counters_dev": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "azul_batch_reset_counters": (_i, [_vp, _vp]),
    "azul_batch_set_id_base": (_i, [_vp, _u32]),
    "azul_batch_set_draw_margin": (_i, [_vp, _u64]),
    "azul_batch_set_move_limit": (_i, [_vp, _u32]),
    "azul_batch_segment_profile": (_i, [_vp, _vp, _i, _i]),
    "azul_selfplay_kernel_resources": (_i, [_vp, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "azul_device_clock_probe": (_i, [_vp, _i, _vp]),
    "azul_pack_c1": (_i, [_vp] * 10 + [_i, _i, _vp, _vp]),
    "azul_timing_begin": (_i, [_vp, _vp]),
    "azul_timing_end": (_i, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(_i), C.POINTER(C.c_float), C.POINTER(_i)]),
    "azul_timing_launch_ms": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
}


class AzulHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libazulhip.so not found at %s: the HIP extension is the product and has no fallback. "
            "Build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    if rc != SUCCESS:
        raise AzulHipError("libazulhip error %d: %s" % (rc, lib.azul_last_error_string().decode()))
    return rc
