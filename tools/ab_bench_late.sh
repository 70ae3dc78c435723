#!/bin/bash
# Development helper (GPU box): bench.py's own `sustained` block means for each library build given (swapped into the package in place).
set -e
cd "$(dirname "$0")/.."
cp azul_deep_reinforcement_learning_amd/libazulhip.so /tmp/libazulhip_keep.so
for lib in "$@"; do
  cp "$lib" azul_deep_reinforcement_learning_amd/libazulhip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
s = d['sustained']
print(sys.argv[1], 'value %.4g  ms_per_step %.4f  sustained blocks' % (d['value'], d['ms_per_step']), ['%.4f' % b['mean_launch_ms'] for b in s['blocks']], 'value_sustained %.4g' % d['value_sustained'],
      (s.get('never_ending_games') or {}).get('launch_ms_with'), (s.get('never_ending_games') or {}).get('launch_ms_replaced'), flush=True)
PY
done
cp /tmp/libazulhip_keep.so azul_deep_reinforcement_learning_amd/libazulhip.so
