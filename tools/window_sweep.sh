#!/bin/bash
# Development helper (GPU box): the policy rollout kernel (configs[2], flat self-play of one net) for windows of 8 .. 256 moves per launch.
cd "$(dirname "$0")/.."
for w in 8 16 32 64 128 256; do
  n=$((640 / w)); [ $n -lt 3 ] && n=3
  python bench_policy.py --window $w --windows $n --no-compare 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('window %4d x %3d: %.1f M env steps/s, %s' % ($w, $n, d['value']/1e6, {k: d[k] for k in d if 'ms' in k}))"
done
