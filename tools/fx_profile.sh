#!/bin/bash
# tools/fx_profile.sh <tag> : rocprofv3 kernel stats + HBM traffic counters (separate passes) for the two effect kernels.
# Run on the GPU box inside gpurun; writes gpurun_out/fx_<tag>/.
TAG=$1
REPO=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/fx_$TAG
mkdir -p $OUT
CMD="python $REPO/tools/fx_scale.py pingpong 4096 65536 reverb 4096 16384"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- $CMD > $OUT/pmc_$C.log 2>&1 || echo "pmc $C failed"
done
find $OUT -name "*.csv" | head -20
