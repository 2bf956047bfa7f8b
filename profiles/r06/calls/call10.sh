#!/bin/bash
# round 6, call 10: the data-gradient epilogue sums extended to producers WITH residuals (mask from the kept bits, scatter in the apply pass)
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call10.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s -k "bn_in_conv" 2>&1 | grep -a -E "bn bwd|passed|failed|Error|assert" | tail -15 >> $L
for r in 1 2 3; do
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  MVAL_TRAIN_BN_IN_CONV=$1 MVAL_TRAIN_BN_BWD_IN_DGRAD=$2 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd $1 bwd $2 c3', d['ms_per_step'], d['config'].get('bn_in_conv'))" >> $L 2>&1
done
done
MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('both c3 one stream', d['ms_per_step'])" >> $L 2>&1
echo "=== train_op_times (default)" >> $L
python tools/train_op_times.py 2>/dev/null | head -14 >> $L
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -8 >> $L
cat $L
