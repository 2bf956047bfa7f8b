#!/bin/bash
# round 6, call 4: step-level UPPER BOUND of BatchNorm-in-conv for the residual-free layers (measurement library libmval_hip_abl.so:
# MVAL_TRAIN_ABL bit 0 = their forward applies skipped, bit 1 = their backward reductions skipped; numerics are garbage by design),
# the product-loop transfer-count test and the product-loop rate
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call04.log
rm -f $L
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "no_host_copy" 2>&1 | tail -4 >> $L
for r in 1 2; do
for abl in 0 1 2 3; do
  MVAL_LIB_TAG=abl MVAL_TRAIN_ABL=$abl MVAL_TRAIN_SLACK_CHECK=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>gpurun_out/r6/abl_err.txt | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('abl $abl c3', d['ms_per_step'])" >> $L 2>&1
  grep -a "ablate" gpurun_out/r6/abl_err.txt | head -1 >> $L
done
done
for abl in 0 1 2 3; do
  MVAL_LIB_TAG=abl MVAL_TRAIN_ABL=$abl MVAL_TRAIN_LANES=0 MVAL_TRAIN_SLACK_CHECK=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('abl $abl c3 one stream', d['ms_per_step'])" >> $L 2>&1
done
python tools/score_bench.py --product-loop 512 2>/dev/null | tail -1 >> $L
python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 bench (device-resident frames)', d['ms_per_step'], d['value'])" >> $L 2>&1
for shp in "64 64 32 32 3 1" "128 128 16 16 3 1" "256 256 8 8 3 1" "32 64 64 64 3 2"; do
  for t in stold stnew; do
    echo "=== stamps $t $shp" >> $L
    MVAL_LIB_TAG=$t python tools/p2_stamps.py $shp 2>&1 | grep -v amdgpu.ids >> $L
  done
done
cat $L
