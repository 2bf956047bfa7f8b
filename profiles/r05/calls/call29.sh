#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1
echo "rc $?" >> gpurun_out/full_gpu_tests.log
python bench.py --workload c3 --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > gpurun_out/bench_c3_lanes.json
tail -5 gpurun_out/full_gpu_tests.log
python -c "
import json; d=json.load(open('gpurun_out/bench_c3_lanes.json')); print(d['ms_per_step'], d['config'].get('training_passes','')[:60]); print([ (f['kernel'][:30], f['frac'], f['seconds_in_kernel_per_step']) for f in [d['roofline']]+d['roofline'].get('other_kernels',[])])"
