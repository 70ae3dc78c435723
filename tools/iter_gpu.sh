#!/bin/bash
# One development iteration on the GPU box: parity tests of the self-play kernel, a short bench, SQ counters.
#   gpurun -- bash tools/iter_gpu.sh <tag> [pmc]
TAG=${1:-it}; R=$PWD/gpurun_out/$TAG; mkdir -p $R; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_selfplay.py "tests/test_full_size_configs.py::test_selfplay_32768_games_is_shard_invariant_and_matches_the_oracle" -m gpu -x -q > $R/pytest.log 2>&1
echo "pytest rc=$?" >> $R/pytest.log; tail -3 $R/pytest.log
grep -q "rc=0" $R/pytest.log || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/bench.json 2> $R/bench.err || { tail -5 $R/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$R/bench.json").read().strip().splitlines()[-1])
print("BENCH %.1f M/s  launch %.4f ms  gates: %s | %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["parity_gate"][:12], d["parity_gate_after_timed_region"][:12]))
PY
if [ "$2" = "pmc" ]; then bash tools/pmc_quick.sh $TAG/pmc 2>&1 | tail -16; fi
