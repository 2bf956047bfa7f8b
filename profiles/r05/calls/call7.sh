#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in p2 h2; do
  MVAL_CONV=$m MVAL_STREAMS=1 python tools/op_times.py 128 resnet50 > gpurun_out/op_times_r50_${m}_128.log 2>&1
  MVAL_CONV=$m MVAL_STREAMS=1 python tools/op_times.py 8 resnet50 > gpurun_out/op_times_r50_${m}_8.log 2>&1
done
head -50 gpurun_out/op_times_r50_p2_128.log
