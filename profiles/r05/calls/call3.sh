#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for t in "" nores "" nores; do
  echo "=== variant '$t'" >> gpurun_out/bneck_nores.log
  MVAL_LIB_TAG=$t python tools/p2_bneck.py time 128 30 >> gpurun_out/bneck_nores.log 2>&1
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'])" >> gpurun_out/bneck_nores.log 2>&1
done
python tools/train_op_times.py 128 > gpurun_out/train_op_times_r5a.log 2>&1
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -s -k "wide_gamma or slack_guard or revalidates" > gpurun_out/call3_tests.log 2>&1
echo "rc $?" >> gpurun_out/call3_tests.log
cat gpurun_out/bneck_nores.log; tail -5 gpurun_out/call3_tests.log
