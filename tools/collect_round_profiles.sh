#!/bin/bash
# Copy the summaries of gpurun_out/<tag>/ (tools/round_profiles.sh a + b) into profiles/<name>_* and regenerate NUMBERS.md:
#   bash tools/collect_round_profiles.sh r4_final4 round4
set -eu
T=${1:?tag}; N=${2:?name}; R=gpurun_out/$T
# inputs of the issue-side roofline first (summarize_profile.py reads them from profiles/): the occupancy sweep of THIS run, and the static
# opcode mix of the headline kernel's move loop as built from the tree's csrc
cp $R/games_sweep.txt profiles/${N}_games_sweep.txt
python3 tools/isa_stats.py --mix profiles/${N}_isa_mix.json > /dev/null
python3 tools/summarize_profile.py $T $N > /dev/null
cp $R/segment_shares.txt profiles/${N}_segment_shares.txt
cp $R/policy_rollout_phases.txt profiles/${N}_policy_rollout_phases.txt
cp $R/${T}_mfma_counters.json profiles/${N}_mfma_counters.json
cp $R/${T}_train_kernel_stats.csv profiles/${N}_train_kernel_stats.csv
cp $R/policy_bench.json profiles/${N}_policy_bench.json
cp "$(ls $R/policy_stats/*/*_kernel_stats.csv | head -1)" profiles/${N}_policy_kernel_stats.csv
for k in players_kernel players_displays2p1_kernel; do
  cp $R/${T}_${k}_kernel_stats.csv profiles/${N}_${k}_kernel_stats.csv
  cp $R/${T}_${k}_pmc.json profiles/${N}_${k}_pmc.json
done
if [ -f $R/policy_vs_policy_bench.json ]; then
  cp $R/policy_vs_policy_bench.json profiles/${N}_policy_vs_policy_bench.json
  cp "$(ls $R/vs_stats/*/*_kernel_stats.csv | head -1)" profiles/${N}_policy_vs_policy_kernel_stats.csv
  cp "$(ls $R/c1_stats/*/*_kernel_stats.csv | head -1)" profiles/${N}_c1_pack_kernel_stats.csv
  python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(list)
for sub in ("vs_pmc", "vs_grbm"):
    for f in glob.glob("$R/" + sub + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "rollout2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"kernel": "azul_policy_rollout2_kernel<LID, 2> (network opponent)", "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -- python3 tools/vs_bench.py 4096 6 (GRBM_GUI_ACTIVE in a separate pass)",
       "means_per_launch": m, "launches": len(agg.get("SQ_WAVES", []))}
if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
    out["matrix_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    out["note"] = "SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs / (1024 x GRBM_GUI_ACTIVE / 8 XCDs)"
json.dump(out, open("profiles/${N}_policy_vs_policy_mfma_counters.json", "w"), indent=1)
PY
fi
cp $R/learning_curve.txt profiles/${N}_learning_curve.txt
for s in selfplay rollout players; do cp $R/${s}_soak_raw.txt profiles/${N}_${s}_soak_raw.txt; done
python3 tools/numbers_table.py > NUMBERS.md
echo "profiles/${N}_* <- $R (csrc sha256 $(cat $R/csrc_sha256.txt))"
