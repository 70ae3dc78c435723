#!/bin/bash
# End-of-round evidence in two gpurun calls (everything lands under gpurun_out/<tag>/; the builder then copies the summaries into profiles/):
#   gpurun -- bash tools/round_profiles.sh a r4_final     headline kernel: bench + rocprofv3 stats + PMC passes + calibration, games sweep, segment shares
#   gpurun -- bash tools/round_profiles.sh b r4_final     policy / training / players profiles, learning curve, the two soaks
set -u
PART=${1:-a}; TAG=${2:-rX_final}
R=$PWD/gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
python3 tools/provenance.py > $R/csrc_sha256.txt
if [ "$PART" = a ]; then
  bash tools/profile_gpu.sh $TAG > $R/profile_gpu.log 2>&1
  bash tools/games_sweep.sh > $R/games_sweep.txt 2> $R/games_sweep.err
  { echo "# csrc sha256 $(cat $R/csrc_sha256.txt); diagnostic build (-DAZ_PROFILE_SEGMENTS): shares only, never quote its run time"
    timeout -k 10 300 python3 tools/segment_profile.py 2 0 8 2>&1 | grep -v amdgpu.ids
    timeout -k 10 300 python3 tools/segment_profile.py 3 0 8 2>&1 | grep -v amdgpu.ids
    timeout -k 10 300 python3 tools/segment_profile.py 4 0 8 2>&1 | grep -v amdgpu.ids
    timeout -k 10 300 python3 tools/segment_profile.py 4 7 8 2>&1 | grep -v amdgpu.ids; } > $R/segment_shares.txt
else
  { echo "# csrc sha256 $(cat $R/csrc_sha256.txt); diagnostic build (-DAZ_PROFILE_SEGMENTS)"
    timeout -k 10 300 python3 tools/rollout_profile.py 2>&1 | grep -v amdgpu.ids
    timeout -k 10 300 python3 tools/rollout_profile.py --opponent 2>&1 | grep -v amdgpu.ids
    echo "---- network opponent: the agent pass and the reply rounds summed per agent step (the sub-phase lines are the agent pass alone) ----"
    timeout -k 10 300 python3 tools/rollout_profile.py --net 2>&1 | grep -v "amdgpu.ids\|Linear(\|^)"; } > $R/policy_rollout_phases.txt
  # GameRunner(opponent=Agent) inside the window kernel: bench line, kernel stats, matrix-pipe counters
  timeout -k 10 300 python3 tools/vs_bench.py 2>/dev/null > $R/policy_vs_policy_bench.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/vs_stats -- python3 tools/vs_bench.py > $R/vs_stats.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $R/vs_pmc -- python3 tools/vs_bench.py 4096 6 > $R/vs_pmc.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/vs_grbm -- python3 tools/vs_bench.py 4096 6 > $R/vs_grbm.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/c1_stats -- python3 tools/c1_pack_profile.py > $R/c1_stats.log 2>&1
  bash tools/profile_train.sh ${TAG}_train ${TAG} > $R/profile_train.log 2>&1
  cp gpurun_out/${TAG}_train/${TAG}_mfma_counters.json gpurun_out/${TAG}_train/${TAG}_train_kernel_stats.csv $R/ 2>/dev/null
  timeout -k 10 300 python3 bench_policy.py 2>/dev/null | tail -1 > $R/policy_bench.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/policy_stats -- python3 bench_policy.py > $R/policy_stats.log 2>&1
  bash tools/profile_players.sh ${TAG}_players ${TAG}_players_kernel --games 4096 --chunk 256 --launches 12 > /dev/null 2>&1
  bash tools/profile_players.sh ${TAG}_players_wide ${TAG}_players_displays2p1_kernel --games 4096 --chunk 256 --launches 12 --ext 7 > /dev/null 2>&1
  cp gpurun_out/${TAG}_players/${TAG}_players_kernel_* gpurun_out/${TAG}_players_wide/${TAG}_players_displays2p1_kernel_* $R/ 2>/dev/null
  timeout -k 10 300 python3 tools/learn_check.py 3000 2>&1 | grep -v amdgpu.ids > $R/learning_curve.txt
  timeout -k 10 400 python3 tools/soak.py 2>&1 | grep -v amdgpu.ids > $R/selfplay_soak_raw.txt
  timeout -k 10 600 python3 tools/soak_rollout.py 300 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|print(" > $R/rollout_soak_raw.txt
  timeout -k 10 600 python3 tools/soak_rollout.py 150 4096 net 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|print(" >> $R/rollout_soak_raw.txt
  timeout -k 10 600 python3 tools/soak_players.py 2>&1 | grep -v amdgpu.ids > $R/players_soak_raw.txt
fi
echo done $PART > $R/DONE_$PART
ls $R
