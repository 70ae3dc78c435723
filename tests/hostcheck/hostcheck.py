"""ctypes loader of the TEST-ONLY host emulation of the device core (tests/hostcheck/hostcheck.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(_HERE, "libhostcheck.so"))
        L.hc_stream_new.restype = C.c_void_p
        L.hc_stream_new.argtypes = [C.c_ulonglong, C.c_int, C.c_int]
        L.hc_stream_free.argtypes = [C.c_void_p]
        L.hc_stream_advance.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.hc_stream_get.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.hc_seed.argtypes = [C.c_ulonglong, C.c_void_p]
        L.hc_mask.argtypes = [C.c_void_p, C.c_void_p]
        L.hc_observe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.hc_potential.argtypes = [C.c_void_p, C.c_int]
        L.hc_flags.argtypes = [C.c_void_p]
        L.hc_count_score.argtypes = [C.c_void_p, C.c_int]
        L.hc_move.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.hc_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.hc_runner_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hc_runner_reset.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.hc_random_action.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hc_weight_table.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostStream:
    def __init__(self, seed, first_player, tile_pool):
        self.h = lib().hc_stream_new(int(seed), first_player, tile_pool)

    def __del__(self):
        if getattr(self, "h", None):
            lib().hc_stream_free(self.h)
            self.h = None

    def advance(self, n):
        out = {"mask": np.zeros((n, 180), np.uint8), "action": np.zeros(n, np.int32), "reward": np.zeros(n, np.int32),
               "done": np.zeros(n, np.uint8), "rec_after": np.zeros((n, 128), np.uint8)}
        rc = lib().hc_stream_advance(self.h, n, ptr(out["mask"]), ptr(out["action"]), ptr(out["reward"]),
                                     ptr(out["done"]), ptr(out["rec_after"]))
        assert rc == 0, rc
        return out

    def get(self):
        rec = np.zeros(128, np.uint8)
        mt = np.zeros(624, np.uint32)
        pos = np.zeros(1, np.uint32)
        ep = np.zeros(1, np.uint64)
        stuck = np.zeros(1, np.uint32)
        ss = np.zeros(10, np.float64)
        lib().hc_stream_get(self.h, ptr(rec), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss))
        return {"rec": rec, "mt": mt, "pos": int(pos[0]), "episodes": int(ep[0]), "stuck": int(stuck[0]), "stat_sum": ss}
