#!/bin/bash
# round 6, call 9: the whole GPU suite at the round's code, the driver's command, then the BASELINE-size pool passes in a call of their own
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call09.log
rm -f $L
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -8 >> $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> $L
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_driver_cmd.stdout 2> gpurun_out/r6/bench_driver_cmd.stderr ) 2>> $L
wc -c gpurun_out/r6/bench_driver_cmd.stdout >> $L
tail -c 4200 gpurun_out/r6/bench_driver_cmd.stdout >> $L
cp bench_detail.json gpurun_out/r6/bench_driver_cmd_detail.json
python bench.py --workload c4 --pool 50000 --no-cpu-baseline --warmup 0 --detail-out gpurun_out/r6/bench_c4_pool50000_detail.json 2>/dev/null | tail -1 > gpurun_out/r6/bench_c4_pool50000.json
python bench.py --workload c5 --pool 50000 --no-cpu-baseline --warmup 0 --detail-out gpurun_out/r6/bench_c5_pool50000_detail.json 2>/dev/null | tail -1 > gpurun_out/r6/bench_c5_pool50000.json
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 slice on this box', d['ms_per_step'])" >> $L 2>&1
cat gpurun_out/r6/bench_c4_pool50000.json gpurun_out/r6/bench_c5_pool50000.json | cut -c1-400 >> $L
cat $L
