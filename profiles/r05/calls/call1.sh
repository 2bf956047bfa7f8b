#!/bin/bash
# round-5 GPU call 1: new parity tests + census reports + counter listing + baseline profile collection with the C3/C4 PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|_EA|hbm|TCC_REQ|TCC_HIT|TCC_MISS" | head -150) > gpurun_out/counters_list.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -s -k "full_size_gradients or wide_gamma or slack_guard or revalidates or switches or golden or properties" > gpurun_out/call1_tests_train.log 2>&1
echo "train rc $?" >> gpurun_out/call1_tests_train.log
timeout 900 python -m pytest tests/test_gpu_p2.py tests/test_gpu_models.py -x -q -m gpu -s -k "census" > gpurun_out/call1_tests_census.log 2>&1
echo "census rc $?" >> gpurun_out/call1_tests_census.log
timeout 1500 bash tools/collect_profiles.sh r5a > gpurun_out/call1_collect.log 2>&1
tail -5 gpurun_out/call1_tests_train.log gpurun_out/call1_tests_census.log
