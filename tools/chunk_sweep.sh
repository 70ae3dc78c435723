#!/bin/bash
# Moves per launch of the headline kernel at 4096 games (development helper): the launch lasts as long as its slowest wave, and the spread of
# the waves' total times shrinks relative to the launch with its length.   gpurun -- bash tools/chunk_sweep.sh
for T in 128 256 512 1024 2048 4096; do
  K=$(( 10240 / T )); [ $K -lt 3 ] && K=3
  python3 bench.py --games 4096 --chunk $T --steps $K --warmup 2 --no-cpu-baseline --no-extras --sustained 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('moves per launch %5d (%3d launches): %.3f G env steps/s, launch %.4f ms = %.4f us per move, %s' % ($T, $K, d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms']*1e3/$T, d['parity_gate_after_timed_region'][:2]))"
done
