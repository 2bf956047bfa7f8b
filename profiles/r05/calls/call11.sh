#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
MVAL_CONV=p2 MVAL_STREAMS=1 python tools/op_times.py 128 resnet50 > gpurun_out/op_times_r50_p2_128.log 2>&1
grep -E "deconv|forward" gpurun_out/op_times_r50_p2_128.log
timeout 600 python -m pytest tests/test_gpu_p2.py -q -m gpu -k "transposed or stride2" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -k "all_gradients or golden" 2>&1 | tail -2
for m in p2 h2; do
  MVAL_CONV=$m python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m c1x16', d['ms_per_step'])"
done
