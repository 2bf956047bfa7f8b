#!/bin/bash
# round 6, call 16: the slab reductions of a backward segment as one launch (MVAL_TRAIN_WGRAD_DEFER): tests, C3 A/B, then the training and
# distributed suites
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call16.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "batched_slab or bn_in_conv" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -6 >> $L
for r in 1 2 3; do
for v in 1 0; do
  MVAL_TRAIN_WGRAD_BATCH=$v python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad batch $v c3', d['ms_per_step'])" >> $L 2>&1
done
done
for v in 1 0; do
  MVAL_TRAIN_WGRAD_BATCH=$v MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad batch $v c3 one stream', d['ms_per_step'])" >> $L 2>&1
done
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | grep -a -E "passed|failed|Error" | tail -4 >> $L
cat $L
