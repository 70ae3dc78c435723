"""CPU suite: the C-ABI library loads and exports every symbol include/azul_hip.h declares (no compute calls
without a GPU), and the host-side pieces (record dtype, codec, rule parsing, sharding) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "azul_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(azul_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from azul_deep_reinforcement_learning_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libazulhip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in _lib.lib.azul_version()


def test_product_has_no_cpu_path():
    import torch
    from azul_deep_reinforcement_learning_amd import Azul, BatchedAzul
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        BatchedAzul(4)
    with pytest.raises(RuntimeError):
        Azul().new_round()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "azul_deep_reinforcement_learning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libazul_oracle", "azul_oracle.h", "oracle/", "oz_"):
                    assert needle not in src, "%s references the oracle (%s)" % (f, needle)


def test_record_dtype_matches_header_and_oracle():
    from azul_deep_reinforcement_learning_amd.records import (RECORD_DTYPE, RECORD_NP_DTYPE, bits_to_walls, pack_flags, unpack_flags,
                                                              walls_to_bits)
    from oracle import oracle as oz
    assert RECORD_DTYPE == oz.RECORD_DTYPE
    assert [(n, RECORD_NP_DTYPE.fields[n][1], RECORD_NP_DTYPE.fields[n][0].itemsize) for n in RECORD_NP_DTYPE.names[:-1]] == \
           [(n, oz.RECORD_NP_DTYPE.fields[n][1], oz.RECORD_NP_DTYPE.fields[n][0].itemsize) for n in oz.RECORD_NP_DTYPE.names[:-1]]
    header = open(os.path.join(ROOT, "include", "azul_hip.h")).read()
    narrow, wide = header.split(" * Wide record", 1)
    for text, dt in ((narrow, RECORD_DTYPE), (wide.split("*/", 1)[0], RECORD_NP_DTYPE)):
        offs = {m.group(3): (int(m.group(1)), int(m.group(2))) for m in
                re.finditer(r"^ \*\s+(\d+)\s+(\d+)\s+\w+\s+(\w+)", text, flags=re.M)}
        for name in dt.names:
            if name == "reserved":
                continue
            assert name in offs, name
            off, size = offs[name]
            assert dt.fields[name][1] == off and dt.fields[name][0].itemsize == size, (dt.itemsize, name)
    for P in (2, 3, 4):
        w = np.random.RandomState(P).rand(P, 5, 5) < 0.4
        assert np.array_equal(bits_to_walls(walls_to_bits(w)), w)
    assert unpack_flags(pack_flags(2, 1, True)) == (2, 1, True) and unpack_flags(pack_flags(4, 3, False)) == (4, 3, False)


def test_rule_parsing():
    from azul_deep_reinforcement_learning_amd import IllegalRule, parse_rules
    assert parse_rules({}) == (1, 0)
    assert parse_rules({"first_player": "Random", "tile_pool": "Lid"}) == (0, 1)
    assert parse_rules({"first_player": 2}) == (2, 0)
    for bad in ({"first_player": 3}, {"first_player": 0}, {"first_player": "random"}, {"tile_pool": "Bag"}, {"first_player": 1.0}):
        with pytest.raises(IllegalRule):
            parse_rules(bad)
    assert parse_rules({"first_player": 4}, players=4) == (4, 0)


def test_global_id_sharding():
    from azul_deep_reinforcement_learning_amd.parallel import shard_seed_base
    assert [shard_seed_base(10, 4096, r) for r in range(3)] == [10, 4106, 8202]


def test_argument_validation_needs_no_gpu():
    """API misuse is rejected on the host, before anything is launched: a negative AZUL_ERR_* and a message."""
    from azul_deep_reinforcement_learning_amd import _lib as L
    lib = L.lib
    one = ctypes.c_void_p(8)                                   # a non-NULL, 8-byte "aligned" placeholder; never dereferenced
    # wrong network shape / NULL pointers
    assert lib.azul_policy_forward(one, one, one, one, one, one, one, one, 136, 128, 180, 1, 0, None, 0, 16, 0, one, one, one, one, None, None) == L.ERR_INVALID
    assert b"136, 180" in lib.azul_last_error_string()
    assert lib.azul_policy_forward(None, one, one, one, one, one, one, one, 136, 180, 180, 1, 0, None, 0, 16, 0, one, one, one, one, None, None) == L.ERR_INVALID
    assert lib.azul_policy_head(None, one, 1, 0, None, 4, 0, one, one, one, None) == L.ERR_INVALID
    assert lib.azul_a2c_gradients(one, one, one, one, 16, ctypes.c_float(1.0), one, one, one, one, one, one, one, 136, 180, 181,
                                  one, 256, one, None, None, None, None) == L.ERR_INVALID
    assert lib.azul_a2c_gradients(one, one, one, one, 16, ctypes.c_float(1.0), one, one, one, one, one, one, one, 136, 180, 180,
                                  None, 256, one, None, None, None, None) == L.ERR_INVALID
    assert lib.azul_a2c_apply_adam(one, one, one, one, ctypes.c_float(3e-4), ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8), 0,
                                   one, one, one, one, one, one, one, one, None, None, ctypes.c_float(0), None, None) == L.ERR_INVALID   # steps count from 1
    assert lib.azul_select_episode_samples(one, one, 32, 3, 64, 48, one, one, one, None, one, None) == L.ERR_INVALID    # not a whole number of windows
    assert lib.azul_select_complete_samples(one, one, 8, 0, one, one, None) == L.ERR_INVALID
    assert lib.azul_discounted_returns(None, one, one, None, ctypes.c_float(0.99), 4, 4, None) == L.ERR_INVALID
    assert lib.azul_batch_policy_rollout(None, 4, 0, one, one, one, one, one, one, 136, 180, 180, 1, 0, None, one, one, one, one, one, one, one, one,
                                         one, None, None) == L.ERR_INVALID
    with pytest.raises(L.AzulHipError):
        L.check(lib.azul_batch_selfplay(None, 4, None, None, None, None, None, None, None, None))


def test_product_sources_carry_no_experiment_switches_and_no_kernel_selection_by_environment():
    """One kernel per entry point: no A/B variants selected by environment variables, no timing-experiment macros that change results, no
    second implementation of the rules behind a switch (rounds 1-4 kept such things in the shipped sources; their history is in git and
    LABNOTES.md).  The only compile-time diagnostics left are the stamp builds (-DAZ_PROFILE_SEGMENTS, -DAZ_LG_PROFILE, -DAZ_PF_PROFILE),
    which leave every result as it is."""
    import glob
    import re
    from azul_deep_reinforcement_learning_amd import _lib as L
    v = L.lib.azul_version().decode()
    assert v.startswith("azul-mi355x") and "EXPERIMENT" not in v.upper(), v
    csrc = os.path.join(ROOT, "azul_deep_reinforcement_learning_amd", "csrc")
    allowed = {"AZ_PROFILE_SEGMENTS", "AZ_LG_PROFILE", "AZ_PF_PROFILE", "__HIPCC__", "AZ_DRAW_MARGIN"}
    for f in sorted(glob.glob(os.path.join(csrc, "*"))):
        src = open(f).read()
        assert "getenv" not in src, f
        for m in re.finditer(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b(.*)$", src, re.M):
            names = set(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", m.group(1))) - {"defined"}
            assert names <= allowed, (os.path.basename(f), m.group(0).strip())
    assert sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*"))) == [
        "azul_common.hpp", "azul_env2.hpp", "azul_kernels.hip", "azul_learner.hpp", "azul_ops2.hpp", "azul_policy.hpp", "azul_rollout2.hpp",
        "azul_rules_x.hpp", "azul_selfplay2.hpp", "azul_selfplay_kernels.hpp", "azul_tables.hpp"]


def test_call_block_layout_is_the_headers(tmp_path):
    """azul_call_t as the Python mirror (_lib.AzulCall) declares it == as a C compiler lays out include/azul_hip.h: every field's
    offset and the size, so a facade call never reads a result from the wrong bytes."""
    import subprocess
    from azul_deep_reinforcement_learning_amd import _lib as L
    names = [f[0] for f in L.AzulCall._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "azul_hip.h"\nint main(void) {\n' +
                   "".join('  printf("%s %%zu\\n", offsetof(azul_call_t, %s));\n' % (n, n) for n in names) +
                   '  printf("sizeof %zu\\n", sizeof(azul_call_t));\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert int(got[n]) == getattr(L.AzulCall, n).offset, n
    assert int(got["sizeof"]) == ctypes.sizeof(L.AzulCall)
    # every want bit of the header is mirrored
    text = open(os.path.join(ROOT, "include", "azul_hip.h")).read()
    for name, val in re.findall(r"#define AZUL_(WANT_[A-Z_]+)\s+(\d+)u", text):
        assert getattr(L, name) == int(val), name


def test_the_benchmark_never_hands_records_into_the_batch_it_measures():
    """A batch the host has written records into (azul_batch_set_state / azul_game_call's record_in -> azul_batch::handed_in) plays its flat
    self-play on the instantiation that also marks the slots of a rule-error-stopped game; the benchmarked batch only ever holds play's own
    records, so bench.py must not call set_records on it (its never-ending-games A/B copies records on the device: BatchedAzul.records_dev)."""
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "set_records(" not in bench and "set_state" not in bench
    host = open(os.path.join(ROOT, "azul_deep_reinforcement_learning_amd", "csrc", "azul_kernels.hip")).read()
    assert host.count("b->handed_in = true;") == 2                      # the two entries that write a caller's record into the batch
    assert "if (b->d.move_limit || b->handed_in)" in host               # ... and the launch that follows them
