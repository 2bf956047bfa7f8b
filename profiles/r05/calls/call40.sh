#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pp -o pp -- python3 $GRAFT_REPO_ROOT/tools/preprocess_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pp_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt_pp -name '*kernel_stats.csv' | head -1)
head -8 $f >> gpurun_out/pp_bench.log
cat gpurun_out/pp_bench.log | tail -14
