#!/bin/bash
# round 6, call 5: BatchNorm apply inside the consumer conv (forward half): bit-identity test, training suite, C3 A/B on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call05.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "bn_in_conv" 2>&1 | tail -15 >> $L
for r in 1 2 3; do
for v in 1 0; do
  MVAL_TRAIN_BN_IN_CONV=$v python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_in_conv $v c3', d['ms_per_step'])" >> $L 2>&1
done
done
for v in 1 0; do
  MVAL_TRAIN_BN_IN_CONV=$v MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_in_conv $v c3 one stream', d['ms_per_step'])" >> $L 2>&1
done
for v in 1 0; do
  echo "=== train_op_times bn_in_conv $v" >> $L
  MVAL_TRAIN_BN_IN_CONV=$v python tools/train_op_times.py 2>/dev/null | head -14 >> $L
done
timeout 2400 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -6 >> $L
cat $L
