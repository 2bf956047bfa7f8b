#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r50_lanes.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_train.py -q -m gpu -k "r50 or resnet or replay_a_graph" 2>&1 | grep -a -E "passed|failed|rror" | tail -4 >> $L
for r in 1 2 3; do
  python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1x16 lanes+graph', d['ms_per_step'])" >> $L 2>&1
  MVAL_STREAMS=1 MVAL_GRAPH=0 python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1x16 one stream eager (as before)', d['ms_per_step'])" >> $L 2>&1
  MVAL_GRAPH=0 python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1x16 lanes eager', d['ms_per_step'])" >> $L 2>&1
done
python bench.py --workload c1 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 200 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1', d['ms_per_step'])" >> $L 2>&1
MVAL_STREAMS=1 python bench.py --workload c1 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 200 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c1 one stream', d['ms_per_step'])" >> $L 2>&1
cat $L
