#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/full_gpu_tests_graph.log 2>&1
echo "rc $?" >> gpurun_out/full_gpu_tests_graph.log
grep -a -E "passed|failed|rc " gpurun_out/full_gpu_tests_graph.log | tail -3
L=gpurun_out/graph_default.log
rm -f $L
for w in c2 c4 c1 c1x16 c5; do
for g in "" 0; do
  MVAL_GRAPH=$g python bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MVAL_GRAPH=\"$g\" $w', d['ms_per_step'])" >> $L 2>&1
done
done
python bench.py --with-input --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 with input', d['ms_per_step'])" >> $L 2>&1
cat $L
