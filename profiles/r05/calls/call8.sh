#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_p2.py tests/test_gpu_hotpath.py -q -m gpu > gpurun_out/call8_b.log 2>&1
echo "rc $?" >> gpurun_out/call8_b.log
rm -f gpurun_out/c1_ab.log
for m in p2 h2; do
  MVAL_CONV=$m python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m c1x16', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
done
python bench.py --workload c1 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default c1', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
python bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'])" >> gpurun_out/c1_ab.log 2>&1
MVAL_CONV=p2 MVAL_STREAMS=1 python tools/op_times.py 128 resnet50 > gpurun_out/op_times_r50_p2_128.log 2>&1
tail -8 gpurun_out/call8_b.log; cat gpurun_out/c1_ab.log; head -30 gpurun_out/op_times_r50_p2_128.log
