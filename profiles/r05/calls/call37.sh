#!/bin/bash
mkdir -p gpurun_out/final
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/final/full_gpu_tests.log 2>&1
echo "rc $?" >> gpurun_out/final/full_gpu_tests.log
python3 bench.py --workload c3 --steps 20 2>/dev/null | grep -a '^{' | tail -1 > gpurun_out/final/bench_c3_r5.json
python3 bench.py 2>/dev/null | grep -a '^{' | tail -1 > gpurun_out/final/bench_c2_r5.json
grep -a -E "passed|failed|rc " gpurun_out/final/full_gpu_tests.log | tail -3
python -c "
import json
d=json.load(open('gpurun_out/final/bench_c2_r5.json')); print('c2', d['ms_per_step'], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['companions'].items()})
d=json.load(open('gpurun_out/final/bench_c3_r5.json')); print('c3', d['ms_per_step'], d['config']['training_passes'][:40])"
