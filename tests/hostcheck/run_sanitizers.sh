#!/bin/bash
# TEST-ONLY: the CPU logic checks of the device code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers do not
# exist on this pool).  Builds the sanitizer variants of the lockstep emulations (simt/) of every kernel family -- the rule
# kernel, the self-play kernel, the rollout's env side, the P-player rules, the rollout kernel, the learner kernels -- and runs the
# emulation test files against them.  Usage: tests/hostcheck/run_sanitizers.sh [log file]
set -u
cd "$(dirname "$0")/../.."
LOG=${1:-profiles/round6_sanitizers.txt}
make -s -C tests/hostcheck libsimt_ops2_asan.so libsimt_ops2_ubsan.so libsimt_selfplay2_asan.so libsimt_selfplay2_ubsan.so libsimt_env2_asan.so libsimt_env2_ubsan.so libsimt_rules_x_asan.so libsimt_rules_x_ubsan.so libsimt_rollout2_asan.so libsimt_rollout2_ubsan.so libsimt_learner_asan.so libsimt_learner_ubsan.so || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
{
  echo "# $(date -u +%FT%TZ)  g++ $(g++ -dumpversion)  -fsanitize=address,undefined -fno-sanitize-recover=all (an error aborts the test process)"
  echo "## csrc sha256 $(python tools/provenance.py)"
  echo "## the two-player rule kernel (azul_op_kernel: azul_ops2.hpp on azul_env2.hpp / azul_selfplay2.hpp, unmodified) under the lockstep emulation + the facade's runner scenarios on it, UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_OPS_LIB=libsimt_ops2_ubsan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_ops2.py tests/test_random_states.py tests/test_facade_runner.py -m "not gpu" -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same, ASan + UBSan"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_OPS_LIB=libsimt_ops2_asan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_ops2.py tests/test_random_states.py tests/test_facade_runner.py -m "not gpu" -q -p no:cacheprovider 2>&1 | tail -4
  echo "## P-player / D-display rules + extended rules (azul_rules_x.hpp, unmodified: the bodies of azul_x_op_kernel / azul_x_selfplay_kernel) under the lockstep emulation, UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_X_LIB=libsimt_rules_x_ubsan.so timeout 2400 python -m pytest tests/test_hostcheck_rules_x.py tests/test_hostcheck_players.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same + the facade on it (the product's host logic, facade_backend.HipBackend, on the emulated device: scenarios, ask-ahead interference, random API sequences), ASan + UBSan"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_X_LIB=libsimt_rules_x_asan.so AZUL_SIMT_OPS_LIB=libsimt_ops2_asan.so \
    timeout 2400 python -m pytest tests/test_hostcheck_rules_x.py tests/test_hostcheck_players.py tests/test_facade_azul.py tests/test_facade_ask_ahead.py tests/test_facade_random_api.py -m "not gpu" -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the benchmarked kernel itself (azul_selfplay2_kernel: azul_selfplay_kernels.hpp on azul_selfplay2.hpp, unmodified) under the lockstep emulation, UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_LIB=libsimt_selfplay2_ubsan.so timeout 1500 python -m pytest tests/test_hostcheck_selfplay2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same, ASan + UBSan (fibers announced to ASan with __sanitizer_start/finish_switch_fiber)"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_LIB=libsimt_selfplay2_asan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_selfplay2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## env side of the persistent policy rollout kernel (azul_env2.hpp, unmodified) under the lockstep emulation, UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_ENV_LIB=libsimt_env2_ubsan.so timeout 1500 python -m pytest tests/test_hostcheck_env2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same, ASan + UBSan"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_ENV_LIB=libsimt_env2_asan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_env2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the persistent policy rollout kernels THEMSELVES (azul_rollout2.hpp / azul_policy.hpp, unmodified: workgroups of 8 emulated waves, MFMA + buffer loads + s_barrier emulated), UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_ROLLOUT_LIB=libsimt_rollout2_ubsan.so timeout 1500 python -m pytest tests/test_hostcheck_rollout2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same, ASan + UBSan"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_ROLLOUT_LIB=libsimt_rollout2_asan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_rollout2.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the A2C gradient kernel, the ActorCritic forward + head kernel, the ring selection and returns kernels (azul_learner.hpp / azul_policy.hpp / azul_selfplay_kernels.hpp, unmodified) against torch autograd / direct models, UBSan"
  LD_PRELOAD="$UBSAN" AZUL_SIMT_LEARNER_LIB=libsimt_learner_ubsan.so timeout 1500 python -m pytest tests/test_hostcheck_learner.py -q -p no:cacheprovider 2>&1 | tail -4
  echo "## the same, ASan + UBSan"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0 AZUL_SIMT_LEARNER_LIB=libsimt_learner_asan.so \
    timeout 1500 python -m pytest tests/test_hostcheck_learner.py -q -p no:cacheprovider 2>&1 | tail -4
} | tee "$LOG"
