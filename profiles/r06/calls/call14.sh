#!/bin/bash
# round 6, call 14: the whole GPU suite at the final code (the pass count captured with grep: the RCCL banner had pushed it out of call 12's tail)
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call14.log
rm -f $L
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -a -E "passed|failed|error" | tail -5 >> $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a smoke >> $L
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "
import sys, json
s = sys.stdin.read().strip()
d = json.loads(s)
print('line', len(s), 'chars; c2', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', round(d['cpu_baseline']['value'], 1), 'exact', d['exact_modes'], 'c3', d['companions']['c3']['ms_per_step'], 'c4', d['companions']['c4']['ms_per_step'])" >> $L
cat $L
