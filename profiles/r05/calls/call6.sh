#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_p2.py -x -q -m gpu -k "transposed or stride2 or plan_structure" > gpurun_out/call6_a.log 2>&1
echo "rc $?" >> gpurun_out/call6_a.log
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_p2.py tests/test_gpu_hotpath.py -q -m gpu > gpurun_out/call6_b.log 2>&1
echo "rc $?" >> gpurun_out/call6_b.log
for m in p2 h2; do
  MVAL_CONV=$m python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m c1x16', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
  MVAL_CONV=$m python bench.py --workload c1 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m c1', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
done
tail -30 gpurun_out/call6_a.log; tail -15 gpurun_out/call6_b.log; cat gpurun_out/c1_ab.log
