#!/bin/bash
# tools/pmc_run.sh <tag> -- <command...> : rocprofv3 PMC passes (separate runs, counters only with --kernel-trace)
# Usage on the GPU box (inside gpurun): tools/pmc_run.sh r01 python bench.py --steps 20 --warmup 3 --no-cpu-baseline
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
find $OUT -name "*counter_collection.csv" | head
