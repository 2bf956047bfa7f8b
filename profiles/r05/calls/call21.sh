#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/nt2_more.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x 2>&1 | tail -2 >> $L
timeout 1200 python -m pytest tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -2 >> $L
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product c2', d['ms_per_step'])" >> $L 2>&1
for t in z64 z192; do
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c2', d['ms_per_step'])" >> $L 2>&1
done
for t in "" y8 z256; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> $L 2>&1
done
for t in "" y16 y48; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
done
MVAL_LIB_TAG=y48 timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "golden or switches or exact" 2>&1 | tail -2 >> $L
cat $L
