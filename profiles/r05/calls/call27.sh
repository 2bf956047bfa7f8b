#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/train_lanes.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "lanes" 2>&1 | tail -15 >> $L
for r in 1 2; do
for f in 1 0; do
  MVAL_TRAIN_LANES=$f python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_TRAIN_LANES=$f c3', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
