#!/bin/bash
# Runs on the GPU box (via gpurun): the training loop (policy rollout + A2C update per window) under rocprofv3 -- kernel-trace stats
# and, in separate passes, the matrix-pipe counters of the rollout and gradient kernels.
# Usage: tools/profile_train.sh <tag> <name>   -> gpurun_out/<tag>/..., summary in profiles/<name>_train_kernel_stats.csv and
#                                                 profiles/<name>_mfma_counters.json
set -u
TAG=${1:-trainprof}; NAME=${2:-roundX}
R=$PWD/gpurun_out/$TAG
mkdir -p $R profiles
export TMPDIR=/tmp
TRAIN="python3 bench_policy.py --train --windows 200"
SHORT="python3 bench_policy.py --train --windows 6"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $TRAIN > $R/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $R/pmc_mfma -- $SHORT > $R/pmc_mfma.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_grbm -- $SHORT > $R/pmc_grbm.log 2>&1
python3 - <<PY
import collections, csv, glob, json, shutil
R, NAME = "$R", "$NAME"
st = glob.glob(R + "/stats/*/*_kernel_stats.csv")
if st: shutil.copyfile(st[0], "profiles/%s_train_kernel_stats.csv" % NAME)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_mfma", "pmc_grbm"):
    for f in glob.glob(R + "/" + sub + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "rollout" in k or "a2c_grad" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -- python3 bench_policy.py --train --windows 6  (GRBM_GUI_ACTIVE in a separate pass)",
       "note": "means per launch; SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x number of v_mfma_f32_16x16x4_f32 issued, summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        e["mfma_busy_cycles_per_simd"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
        e["gui_active_cycles_per_xcd"] = e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_busy_fraction_of_gui_active"] = e["mfma_busy_cycles_per_simd"] / e["gui_active_cycles_per_xcd"]
    out[k] = e
json.dump(out, open("profiles/%s_mfma_counters.json" % NAME, "w"), indent=1)
print(json.dumps({k: v.get("mfma_busy_fraction_of_gui_active") for k, v in out.items() if isinstance(v, dict)}))
PY
cp profiles/${NAME}_mfma_counters.json profiles/${NAME}_train_kernel_stats.csv $R/ 2>/dev/null
grep metric $R/stats.log | cut -c1-300
echo done > $R/DONE
