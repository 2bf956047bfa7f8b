#!/bin/bash
mkdir -p gpurun_out/pool_r5b
export TMPDIR=/tmp
L=gpurun_out/pool_r5b/pool_after_pmc.log
rm -f $L
sl() { python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 slice c4', d['ms_per_step'])" >> $L 2>&1; }
sl "fresh box:"
python3 bench.py --workload c4 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | tail -1 > gpurun_out/pool_r5b/bench_c4_pool50000.json
python3 bench.py --workload c5 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | tail -1 > gpurun_out/pool_r5b/bench_c5_pool50000.json
sl "after the pool passes:"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_x -o x -- python3 bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1
sl "after a --pmc pass:"
sleep 20
sl "20 s later:"
python -c "
import json
for w in ('c4','c5'):
    d=json.load(open('gpurun_out/pool_r5b/bench_%s_pool50000.json'%w)); print(w,'pool 50000:', d['ms_per_step']/1000,'s', d['value'])" >> $L
cat $L
