#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/apply_u.log
for t in "" u2 u3 "" u2 u3; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U variant \"$t\" c3', d['ms_per_step'])" >> gpurun_out/apply_u.log 2>&1
done
MVAL_LIB_TAG=u2 timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k "golden or switches or bn_train_ops" 2>&1 | tail -2 >> gpurun_out/apply_u.log
cat gpurun_out/apply_u.log
