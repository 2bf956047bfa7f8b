#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/wgs_per_cu.log
rm -f $L
for r in 1 2; do
for w in 0 1 2 3; do
  MVAL_P2_WGS=$w MVAL_LIB_TAG=tune python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_P2_WGS=$w c2', d['ms_per_step'])" >> $L 2>&1
done
for w in 0 1; do
  MVAL_P2_WGS=$w MVAL_LIB_TAG=tune python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_P2_WGS=$w c4', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
