#!/bin/bash
mkdir -p gpurun_out/final2
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/final2/full_gpu_tests.log 2>&1
echo "rc $?" >> gpurun_out/final2/full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final2/smoke.log 2>&1
python3 bench.py 2>/dev/null | grep -a '^{' | tail -1 > gpurun_out/final2/bench_c2_head.json
grep -a -E "passed|failed|rc " gpurun_out/final2/full_gpu_tests.log | tail -3; tail -1 gpurun_out/final2/smoke.log
python -c "
import json
d=json.load(open('gpurun_out/final2/bench_c2_head.json')); print('c2', d['ms_per_step'], d['value'], d['roofline']['frac'], {k:(v.get('ms_per_step') if isinstance(v,dict) else None) for k,v in d['companions'].items()})"
