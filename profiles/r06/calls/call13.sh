#!/bin/bash
# round 6, call 13: weight gradient with its operands requested TWO tiles ahead (WB_PF 2, the product) against one (libmval_hip_pf1.so)
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call13.log
rm -f $L
timeout 1500 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "wgrad or golden or lanes or bn_in_conv or full_size" 2>&1 | tail -4 >> $L
for r in 1 2 3; do
for t in "" pf1; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
done
for t in "" pf1; do
  MVAL_LIB_TAG=$t MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3 one stream', d['ms_per_step'])" >> $L 2>&1
  echo "=== train_op_times variant '$t'" >> $L
  MVAL_LIB_TAG=$t python tools/train_op_times.py 2>/dev/null | head -12 >> $L
done
cat $L
