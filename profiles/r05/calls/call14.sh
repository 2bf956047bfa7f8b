#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/w48_tiles.log
for t in "" ms7 "" ms7; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> gpurun_out/w48_tiles.log 2>&1
done
for t in "" ms7; do
  echo "== $t" >> gpurun_out/w48_tiles.log
  MVAL_LIB_TAG=$t MVAL_STREAMS=1 python tools/op_times.py 64 hrnet_w48 2>&1 | grep -E "12x9|forward" | head -8 >> gpurun_out/w48_tiles.log
done
MVAL_LIB_TAG=ms7 timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "w48" 2>&1 | tail -2 >> gpurun_out/w48_tiles.log
cat gpurun_out/w48_tiles.log
