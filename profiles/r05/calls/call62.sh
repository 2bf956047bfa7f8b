#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/p2_nt_res2.log
rm -f $L
for r in 1 2 3; do
for t in "" ra rs; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> $L 2>&1
done
done
for t in "" ra rs; do
  MVAL_LIB_TAG=$t MVAL_CONV=p2 python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c1x16', d['ms_per_step'])" >> $L 2>&1
done
cat $L
