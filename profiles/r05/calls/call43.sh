#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/graph_replay.log
rm -f $L
for r in 1 2; do
for g in "" 1 0; do
  MVAL_GRAPH=$g python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_GRAPH=\"$g\" c2', d['ms_per_step'])" >> $L 2>&1
done
for g in "" 1; do
  MVAL_GRAPH=$g python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_GRAPH=\"$g\" c4', d['ms_per_step'])" >> $L 2>&1
done
done
cat $L
