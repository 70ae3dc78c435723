#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + separate PMC passes.
# Usage: tools/profile_gpu.sh <tag>     -> gpurun_out/<tag>/...
set -u
TAG=${1:-prof}
R=$PWD/gpurun_out/$TAG
mkdir -p $R
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sustained 0"   # the driver's command without its secondary parts (every step = one launch = 512 moves x 4096 games)
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $R/bench.json 2> $R/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- $BENCH > $R/stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- $BENCH > $R/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- $BENCH > $R/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/pmc_sq -- $BENCH > $R/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $R/pmc_sq2 -- $BENCH > $R/pmc_sq2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_grbm -- $BENCH > $R/pmc_grbm.log 2>&1
# the VALU instructions by class (what the issue-side roofline prices: tools/summarize_profile.py)
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $R/pmc_sq3 -- $BENCH > $R/pmc_sq3.log 2>&1
# counter calibration in this path's access widths
if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/calib tools/calib_kernels.hip > $R/calib_build.log 2>&1; then
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/calib_fetch -- $R/calib > $R/calib_fetch.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/calib_write -- $R/calib > $R/calib_write.log 2>&1
  rm -f $R/calib
fi
echo done > $R/DONE
cat $R/bench.json
