#!/bin/bash
mkdir -p gpurun_out/pool_head
export TMPDIR=/tmp
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slice c4', d['ms_per_step'])" > gpurun_out/pool_head/pool.log 2>&1
python3 bench.py --workload c4 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | grep -a '^{' | tail -1 > gpurun_out/pool_head/bench_c4_pool50000.json
python3 bench.py --workload c5 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | grep -a '^{' | tail -1 > gpurun_out/pool_head/bench_c5_pool50000.json
python -c "
import json
for w in ('c4','c5'):
    d=json.load(open('gpurun_out/pool_head/bench_%s_pool50000.json'%w)); print(w,'pool 50000:', d['ms_per_step']/1000,'s', d['value'], d['config'].get('network_launch','')[:20])" >> gpurun_out/pool_head/pool.log
cat gpurun_out/pool_head/pool.log
