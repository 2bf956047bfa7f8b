#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/w32_nt2_step.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x -k "conv_vs_float64 or batch or nan" 2>&1 | tail -2 >> $L
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_train.py -q -m gpu -k "w48" 2>&1 | tail -2 >> $L
MVAL_LIB_TAG=x3 timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x -k "conv_vs_float64 or batch or nan" 2>&1 | tail -2 >> $L
MVAL_LIB_TAG=x3 timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "golden and w32" 2>&1 | tail -2 >> $L
for r in 1 2 3; do
for t in "" x1 x2 x3; do
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c2', d['ms_per_step'])" >> $L 2>&1
done
done
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product c4', d['ms_per_step'])" >> $L 2>&1
cat $L
