#!/bin/bash
TAG=r3q; R=$PWD/gpurun_out/$TAG; mkdir -p $R; export TMPDIR=/tmp
BENCH="python3 bench.py --steps 8 --warmup 1 --chunk 64 --no-cpu-baseline --no-extras"
run() { timeout -k 10 200 rocprofv3 --pmc $2 --output-format csv -d $R/$1 -- $BENCH > $R/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"
run b "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run c "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_CYCLES"
python3 - <<PY
import csv,glob,collections
for sub in ("a","b","c"):
    fs=glob.glob("$R/"+sub+"/*/*_counter_collection.csv")
    if not fs: print(sub,"no file"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "selfplay" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print("%-30s per game-move %12.3f"%(k, sum(v)/len(v)/(4096*64)))
PY
