#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/nrs.log
for t in "" nrs8 nrs6 "" nrs8 nrs6; do
  MVAL_LIB_TAG=$t python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c2', d['ms_per_step'])" >> gpurun_out/nrs.log 2>&1
done
for t in "" nrs8 nrs6; do
  echo "== $t" >> gpurun_out/nrs.log
  MVAL_LIB_TAG=$t python tools/p2_sweep.py time 128 50 2>&1 | grep -E "k3 s1" >> gpurun_out/nrs.log
done
cat gpurun_out/nrs.log
