#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/dma.log
MVAL_LIB_TAG=dma timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x -k "conv_vs_float64 or batch or nan" 2>&1 | tail -4 >> gpurun_out/dma.log
MVAL_LIB_TAG=dma timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "golden and w32" 2>&1 | tail -3 >> gpurun_out/dma.log
for t in "" dma; do
  echo "=== variant '$t'" >> gpurun_out/dma.log
  MVAL_LIB_TAG=$t python tools/p2_sweep.py time 128 50 2>&1 | grep -E "k3 s1" >> gpurun_out/dma.log
done
for t in "" dma "" dma; do
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c2', d['ms_per_step'])" >> gpurun_out/dma.log 2>&1
done
cat gpurun_out/dma.log
