#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/gamma_diag.py -8 > gpurun_out/gamma_diag.log 2>&1
python tools/gamma_diag.py 0 >> gpurun_out/gamma_diag.log 2>&1
python tools/gamma_diag.py -4 >> gpurun_out/gamma_diag.log 2>&1
for t in "" xf2 xf3; do
  echo "=== variant '$t'" >> gpurun_out/xf_sweep.log
  MVAL_LIB_TAG=$t python tools/p2_sweep.py time 128 50 >> gpurun_out/xf_sweep.log 2>&1
done
for t in "" xf2 xf3 "" xf2 xf3; do
  echo "=== variant '$t'" >> gpurun_out/xf_bench.log
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/xf_bench.log 2>&1
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'])" >> gpurun_out/xf_bench.log 2>&1
done
cat gpurun_out/gamma_diag.log; cat gpurun_out/xf_bench.log
