#!/bin/bash
# quick SQ-counter pass for kernel iteration: tools/pmc_quick.sh <tag>
TAG=${1:-q}; R=$PWD/gpurun_out/$TAG; mkdir -p $R; export TMPDIR=/tmp
BENCH="python3 bench.py --steps 8 --warmup 1 --chunk 64 --no-cpu-baseline --no-extras"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/pmc_sq -- $BENCH > $R/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --output-format csv -d $R/pmc_sq2 -- $BENCH > $R/pmc_sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
for sub in ("pmc_sq","pmc_sq2"):
    f=glob.glob("$R/"+sub+"/*/*_counter_collection.csv")[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "selfplay" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print("%-22s per game-move %10.2f"%(k, sum(v)/len(v)/(4096*64)))
PY
