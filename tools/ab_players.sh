#!/bin/bash
# A/B builds of libazulhip.so with tools/players_bench.py (development helper; timing experiments): tools/ab_players.sh a.so b.so ...
L=azul_deep_reinforcement_learning_amd/libazulhip.so
cp $L /tmp/orig.so
for rep in 1 2; do
for f in "$@"; do
  cp $f $L
  echo "$f:"; python3 tools/players_bench.py --launches 12 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('   P=%d  launch %.4f ms  %.3f G/s' % (d['players'], d['avg_launch_ms'], d['env_steps_per_s_kernel'] / 1e9))"
done; done
cp /tmp/orig.so $L
