#!/bin/bash
# A/B two builds of libazulhip.so with bench.py (development helper): tools/ab_bench.sh a.so b.so
L=azul_deep_reinforcement_learning_amd/libazulhip.so
cp $L /tmp/orig.so
for rep in 1 2 3; do
for f in "$@"; do
  cp $f $L
  echo -n "$f: "; python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  launch %.4f ms  %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['parity_gate'][:2]))"
done; done
cp /tmp/orig.so $L
