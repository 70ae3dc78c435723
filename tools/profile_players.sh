#!/bin/bash
# Runs on the GPU box (via gpurun): the 3- / 4-player self-play kernel under rocprofv3 -- kernel-trace stats and, in separate passes,
# the SQ instruction counters.  Usage: tools/profile_players.sh <tag> <name> [extra players_bench.py args]
#   -> gpurun_out/<tag>/..., summaries profiles/<name>_kernel_stats.csv and profiles/<name>_pmc.json
set -u
TAG=${1:-playersprof}; NAME=${2:-roundX_players}; shift 2
R=$PWD/gpurun_out/$TAG
mkdir -p $R profiles
export TMPDIR=/tmp
CMD="python3 tools/players_bench.py $*"
timeout -k 10 200 python3 tools/players_bench.py "$@" > $R/bench.jsonl 2> $R/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $CMD > $R/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/pmc_sq -- $CMD > $R/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $R/pmc_sq2 -- $CMD > $R/pmc_sq2.log 2>&1
python3 - "$R" "$NAME" "$CMD" <<'PY'
import collections, csv, glob, json, shutil, sys
R, NAME, CMD = sys.argv[1:4]
st = glob.glob(R + "/stats/*/*_kernel_stats.csv")
if st: shutil.copyfile(st[0], "profiles/%s_kernel_stats.csv" % NAME)
runs = [json.loads(l) for l in open(R + "/bench.jsonl") if l.startswith("{")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(R + "/" + sub + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "selfplay" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "rocprofv3 --pmc <SQ counters, two passes> -- " + CMD, "runs": runs,
       "note": "means per launch; per_game_move = counter / (games x moves per launch); SQ_*_CYCLES / WAIT / ACTIVE counters are in quanta of 4 cycles"}
gm = runs[0]["games"] * runs[0]["moves_per_launch"] if runs else 1
for k, d in agg.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["per_game_move"] = {c: v / gm for c, v in e.items() if c.startswith("SQ_")}
    out[k] = e
json.dump(out, open("profiles/%s_pmc.json" % NAME, "w"), indent=1)
print(json.dumps({k: v["per_game_move"] for k, v in out.items() if isinstance(v, dict) and "per_game_move" in v}, indent=1))
PY
cp profiles/${NAME}_pmc.json profiles/${NAME}_kernel_stats.csv $R/ 2>/dev/null
cat $R/bench.jsonl
echo done > $R/DONE
