#!/bin/bash
# round 6, call 18: lane mapping of the INZ staging (a pixel's channel blocks on consecutive lanes = the product, against a block's pixels on
# consecutive lanes = libmval_hip_map0.so): bit-identity tests, C3 A/B, per-operator table
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call18.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "bn_in_conv or golden or lanes" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -5 >> $L
for r in 1 2 3; do
for t in "" map0; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
done
for t in "" map0; do
  MVAL_LIB_TAG=$t MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3 one stream', d['ms_per_step'])" >> $L 2>&1
  echo "=== train_op_times variant '$t'" >> $L
  MVAL_LIB_TAG=$t python tools/train_op_times.py 2>/dev/null | head -12 >> $L
done
cat $L
