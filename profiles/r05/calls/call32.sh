#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/pool_host.log
rm -f $L
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slice c4', d['ms_per_step'])" >> $L 2>&1
python bench.py --workload c4 --pool 4000 --no-cpu-baseline --no-rooflines --warmup 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool 4000: ms per batch', d['ms_per_step']/500)" >> $L 2>&1
python bench.py --workload c4 --pool 4000 --no-overlap --no-cpu-baseline --no-rooflines --warmup 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool 4000 --no-overlap: ms per batch', d['ms_per_step']/500)" >> $L 2>&1
python -m cProfile -s cumtime bench.py --workload c4 --pool 2000 --no-cpu-baseline --no-rooflines --warmup 0 2>/dev/null | head -60 >> $L
cat $L
