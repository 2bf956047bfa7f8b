#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/w48_nt3_step.log
rm -f $L
for r in 1 2 3; do
for t in "" n2 n10 n6 n14; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
