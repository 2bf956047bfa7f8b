#!/bin/bash
# round 6, call 19: final code again (INZ lane mapping per chunk count): whole GPU suite + driver command
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call19.log
rm -f $L
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -a -E "passed|failed|error" | tail -5 >> $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a smoke >> $L
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6/bench_driver_cmd_final.json
cp bench_detail.json gpurun_out/r6/bench_driver_cmd_final_detail.json
python3 -c "
import json
s = open('gpurun_out/r6/bench_driver_cmd_final.json').read().strip()
d = json.loads(s)
print('line', len(s), 'chars; c2', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', round(d['cpu_baseline']['value'], 1), 'exact', d['exact_modes'], 'c3', d['companions']['c3']['ms_per_step'], 'c4', d['companions']['c4']['ms_per_step'], 'c2_with_input', d['companions']['c2_with_input']['ms_per_step'])" >> $L
python3 bench.py --workload c3 --steps 20 --detail-out gpurun_out/r6/bench_c3_final_detail.json 2>/dev/null | tail -1 > gpurun_out/r6/bench_c3_final.json
cat $L
