#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/step_level_knobs.log
rm -f $L
c2() { MVAL_LIB_TAG=$1 python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 c2', d['ms_per_step'])" >> $L 2>&1; }
c3() { MVAL_LIB_TAG=$1 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 c3', d['ms_per_step'])" >> $L 2>&1; }
for r in 1 2; do
  c2 "" "product"
  MVAL_P2_TILE=8,0,0 c2 tune "tune ms=8"
  c2 w3 "P2_W3"
  c3 "" "product"
  c3 e3 "P2_EPI3_W3"
done
cat $L
