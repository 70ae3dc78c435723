#!/usr/bin/env python3
"""Static ISA statistics of the self-play kernels (development helper): instruction counts by pipe, registers, LDS, scratch.
Usage: tools/isa_stats.py [kernel-name-substring ...]   (compiles csrc/azul_kernels.hip with -S for gfx950 into gpurun_out/).
       tools/isa_stats.py --mix OUT.json [kernel-name-substring]   the opcode histogram of the kernel's MOVE LOOP (the region between the
           header and the latch of its largest loop: the common move and the rare paths that branch back into it) -- the static input of
           the issue-side roofline (tools/summarize_profile.py prices the classes the PMC counters do not separate with it)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(ROOT, "gpurun_out", "azul_kernels.s")
os.makedirs(os.path.dirname(asm), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-Wno-unused-value", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm,
                       os.path.join(ROOT, "azul_deep_reinforcement_learning_amd", "csrc", "azul_kernels.hip")], stderr=subprocess.DEVNULL)
s = open(asm).read()


def move_loop_mix(kernel):
    """Opcode counts of the region [header, latch] of the kernel's largest loop."""
    name = [n for n in re.findall(r"^(_Z\w+):\s*; @", s, flags=re.M) if kernel in n][0]
    body = re.search(r"^" + re.escape(name) + r":(.*?)^\.Lfunc_end", s, re.S | re.M).group(1)
    labels, ins = {}, []
    for ln in body.split("\n"):
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t.startswith("."):
            ins.append(t.split(";")[0].strip())
    best = None
    for i, t in enumerate(ins):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", t)
        if m and labels[m.group(2)] <= i and (best is None or i - labels[m.group(2)] > best[0]):
            best = (i - labels[m.group(2)], labels[m.group(2)], i)
    region = ins[best[1]:best[2] + 1]
    c = collections.Counter(t.split()[0] for t in region)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    quarter_mul = sum(v for k, v in c.items() if re.match(r"v_mul_(hi|lo)_[ui]32", k))
    return {"kernel": name, "instructions_in_kernel": len(ins), "move_loop": {"first": best[1], "last": best[2], "instructions": len(region)},
            "valu": valu, "salu": sum(v for k, v in c.items() if k.startswith("s_")), "lds": sum(v for k, v in c.items() if k.startswith("ds_")),
            "vmem": sum(v for k, v in c.items() if k.startswith(("global_", "flat_", "buffer_"))),
            "quarter_rate_int32_multiplies": quarter_mul, "quarter_rate_int32_multiply_share_of_valu": quarter_mul / max(valu, 1),
            "opcodes": dict(sorted(c.items(), key=lambda kv: -kv[1]))}


if "--mix" in sys.argv:
    i = sys.argv.index("--mix")
    out = sys.argv[i + 1]
    kern = sys.argv[i + 2] if len(sys.argv) > i + 2 else "azul_selfplay2_kernelILb1ELi1ELb1ELb0"
    mix = move_loop_mix(kern)
    mix["build"] = "hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-sched-strategy=max-ilp (the shipped flags), csrc sha256 " + subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "tools", "provenance.py")], text=True).strip()
    json.dump(mix, open(out, "w"), indent=1)
    print("%s: %d instructions in the move loop, %d VALU, quarter-rate int32 multiplies %.2f %% of them" % (
        out, mix["move_loop"]["instructions"], mix["valu"], 100 * mix["quarter_rate_int32_multiply_share_of_valu"]))
    sys.exit(0)

want = sys.argv[1:] or ["azul_selfplay2_kernelILb1ELi1E", "azul_selfplay_kernelILb1ELi1E"]
meta = s[s.index("amdhsa.kernels"):]
for name in re.findall(r"^(_Z\w+):\s*; @", s, flags=re.M):
    if not any(w in name for w in want):
        continue
    body = re.search(r"^" + re.escape(name) + r":(.*?)^\.Lfunc_end", s, re.S | re.M).group(1)
    ins = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((".", ";")) and not ln.strip().endswith(":")]
    c = collections.Counter(ln.split()[0] for ln in ins)
    pipe = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%s\n   instructions %d: s_ %d  v_ %d  ds_ %d  global_/flat_ %d  scratch_ %d  branches %d" % (
        name, len(ins), pipe("s_"), pipe("v_"), pipe("ds_"), pipe("global_") + pipe("flat_"), pipe("scratch_"),
        sum(v for k, v in c.items() if k.startswith("s_cbranch") or k == "s_branch")))
    j = meta.index(name)
    blk = meta[max(0, j - 2500):j + 1500]
    vals = {}
    for key in (".sgpr_count", ".vgpr_count", ".agpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size", ".vgpr_spill_count",
                ".sgpr_spill_count"):
        mm = re.findall(re.escape(key) + r":\s+(\d+)", blk)
        vals[key] = mm[-1] if mm else "?"
    print("   " + "  ".join("%s %s" % (k[1:], v) for k, v in vals.items()))
    if "-v" in sys.argv:
        for k, v in c.most_common(40):
            print("      %5d %s" % (v, k))
