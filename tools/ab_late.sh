#!/bin/bash
# Development helper (GPU box): tools/ab_late.py for each library build given (swapped into the package in place; the last one stays).
set -e
cd "$(dirname "$0")/.."
cp azul_deep_reinforcement_learning_amd/libazulhip.so /tmp/libazulhip_keep.so
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" azul_deep_reinforcement_learning_amd/libazulhip.so
    python tools/ab_late.py "$lib"
  done
done
cp /tmp/libazulhip_keep.so azul_deep_reinforcement_learning_amd/libazulhip.so
