#!/usr/bin/env python3
"""sha256 over the product's device / host source (azul_deep_reinforcement_learning_amd/csrc/* and include/azul_hip.h, names and contents,
sorted): what a tracked raw log of a soak / sanitizer / profile run cites, so that "ran on these kernels" can be checked against the tree.
    python tools/provenance.py        prints the 16-hex-digit prefix"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    h = hashlib.sha256()
    base = os.path.join(ROOT, "azul_deep_reinforcement_learning_amd", "csrc")
    files = [os.path.join(base, f) for f in sorted(os.listdir(base)) if f.endswith((".hpp", ".hip", ".h"))] + [os.path.join(ROOT, "include", "azul_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
