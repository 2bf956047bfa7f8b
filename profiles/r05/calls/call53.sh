#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/lane_map.log
rm -f $L
for r in 1 2 3; do
for t in "" lm2 lm3; do
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c2', d['ms_per_step'])" >> $L 2>&1
done
done
for t in "" lm2 lm3; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> $L 2>&1
done
cat $L
