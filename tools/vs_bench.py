#!/usr/bin/env python3
"""Development helper: bench.py's policy_vs_policy line alone (the window kernel with a network opponent).  Usage: tools/vs_bench.py [games]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

print(json.dumps(bench.policy_vs_policy(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, windows=int(sys.argv[2]) if len(sys.argv) > 2 else 40), indent=1))
