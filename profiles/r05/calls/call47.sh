#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/train_graph.log
rm -f $L
for r in 1 2 3; do
  MVAL_TRAIN_LANES=3 MVAL_TRAIN_GRAPH=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes 3 eager c3', d['ms_per_step'])" >> $L 2>&1
  MVAL_TRAIN_LANES=2 MVAL_TRAIN_GRAPH=1 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes 2 graph c3', d['ms_per_step'])" >> $L 2>&1
  MVAL_TRAIN_LANES=2 MVAL_TRAIN_GRAPH=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes 2 eager c3', d['ms_per_step'])" >> $L 2>&1
done
cat $L
