#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call06.log
rm -f $L
timeout 2400 python -m pytest tests/test_gpu_train.py -q -m gpu 2>&1 | tail -40 >> $L
cat $L
