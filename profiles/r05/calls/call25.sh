#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r50_nt2.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x 2>&1 | tail -2 >> $L
for t in r3 r6; do
  MVAL_LIB_TAG=$t timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "r50 or resnet" 2>&1 | tail -2 >> $L
done
for r in 1 2 3; do
for t in "" r1 r2 r3 r6; do
  MVAL_LIB_TAG=$t MVAL_CONV=p2 python bench.py --workload c1x16 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c1x16', d['ms_per_step'])" >> $L 2>&1
done
done
python tools/shape_fuzz.py >> gpurun_out/shape_fuzz_nt.log 2>&1
cat $L; tail -30 gpurun_out/shape_fuzz_nt.log
