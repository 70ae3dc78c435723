#!/bin/bash
# A/B builds of libazulhip.so on the headline kernel at several batch sizes (development helper): tools/ab_games.sh "4096 8192" a.so b.so
GAMES=$1; shift
L=azul_deep_reinforcement_learning_amd/libazulhip.so
cp $L /tmp/orig.so
for rep in 1 2 3; do
for f in "$@"; do
  cp $f $L
  for G in $GAMES; do
    echo -n "$f games $G: "; python3 bench.py --games $G --steps 8 --warmup 2 --no-cpu-baseline --no-extras --sustained 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  launch %.4f ms  %s  resident waves/SIMD %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['parity_gate_after_timed_region'][:2], d['roofline']['kernel_resources'].get('resident_waves_per_simd')))"
  done
done; done
cp /tmp/orig.so $L
