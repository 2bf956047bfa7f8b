#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_distributed.py -q -m gpu -x > gpurun_out/train_tests_free_full.log 2>&1
echo "rc $?" >> gpurun_out/train_tests_free_full.log
grep -a -E "passed|failed|rc " gpurun_out/train_tests_free_full.log | tail -5
