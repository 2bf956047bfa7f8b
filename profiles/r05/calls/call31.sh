#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/wgrad_th8.log
rm -f $L
MVAL_LIB_TAG=th8 timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "golden or switches or wgrad or lanes" 2>&1 | tail -3 >> $L
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "repacks or golden" 2>&1 | tail -3 >> $L
for r in 1 2; do
for t in "" th8; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
done
for t in "" th8; do
  echo "== $t" >> $L
  MVAL_LIB_TAG=$t python tools/train_op_times.py 2>&1 | grep -E "families|k3s1 +(32->32|64->64|128->128|256->256) " >> $L
done
cat $L
