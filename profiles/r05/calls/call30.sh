#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/lanes_occupancy.log
rm -f $L
for r in 1 2; do
for t in "" ps512 ps384; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
for w in 0 1 2; do
  MVAL_P2_WGS=$w MVAL_LIB_TAG=tune python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tune MVAL_P2_WGS=$w c3', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
