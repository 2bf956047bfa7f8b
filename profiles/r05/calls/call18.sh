#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/w48_nt3.log
rm -f $L
for t in n14 n7; do
  echo "== tests $t" >> $L
  MVAL_LIB_TAG=$t timeout 900 python -m pytest tests/test_gpu_p2.py -q -m gpu -x -k "conv_vs_float64 or batch or nan" 2>&1 | tail -3 >> $L
  MVAL_LIB_TAG=$t timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "w48" 2>&1 | tail -3 >> $L
done
for t in "" n1 n8 n2 n4; do
  echo "== $t" >> $L
  MVAL_LIB_TAG=$t MVAL_STREAMS=1 python tools/op_times.py 64 hrnet_w48 2>&1 | grep -E "forward|k3s1 +(48->48|96->96|192->192|256->48) " >> $L
done
for t in "" n14 n7 "" n14 n7; do
  MVAL_LIB_TAG=$t python bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c4', d['ms_per_step'])" >> $L 2>&1
done
cat $L
