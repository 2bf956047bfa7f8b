#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_train.py -q -m gpu -x > gpurun_out/call9_train.log 2>&1
echo "rc $?" >> gpurun_out/call9_train.log
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_p2.py -q -m gpu > gpurun_out/call9_b.log 2>&1
echo "rc $?" >> gpurun_out/call9_b.log
rm -f gpurun_out/c1_ab.log
for m in p2 h2; do
  MVAL_CONV=$m python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m c1x16', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
done
for f in 1 0 1 0; do
  MVAL_TRAIN_DGRAD_PARITY_P2=$f python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parity_p2 $f c3', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
done
MVAL_CONV=p2 MVAL_STREAMS=1 python tools/op_times.py 128 resnet50 > gpurun_out/op_times_r50_p2_128.log 2>&1
tail -6 gpurun_out/call9_train.log; tail -6 gpurun_out/call9_b.log; cat gpurun_out/c1_ab.log; head -16 gpurun_out/op_times_r50_p2_128.log
