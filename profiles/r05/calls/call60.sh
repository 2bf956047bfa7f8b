#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/bn_nt_res.log
rm -f $L
for r in 1 2 3; do
for t in "" nr; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
