#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(tools/micro/bin/p2_loop 512 24; tools/micro/bin/p2_loop 512 19; tools/micro/bin/p2_loop 512 16) > gpurun_out/p2_loop.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu > gpurun_out/call5_tests.log 2>&1
echo "rc $?" >> gpurun_out/call5_tests.log
for f in 1 0 1 0; do
  MVAL_TRAIN_FOLD=$f python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold $f c3', d['ms_per_step'])" >> gpurun_out/fold_ab.log 2>&1
done
cat gpurun_out/p2_loop.log; tail -5 gpurun_out/call5_tests.log; cat gpurun_out/fold_ab.log
