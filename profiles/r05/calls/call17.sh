#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/micro/bin/limb_atomics > gpurun_out/limb_atomics.log 2>&1
rm -f gpurun_out/wgrad_slabs.log
for t in "" ps512 ps384 ps256 nostore ""; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> gpurun_out/wgrad_slabs.log 2>&1
done
for t in "" ps512 ps384 ps256 nostore; do
  echo "== $t" >> gpurun_out/wgrad_slabs.log
  MVAL_LIB_TAG=$t python tools/train_op_times.py 2>&1 | grep -E "families|k3s1 +(32->32|64->64|128->128|256->256) " >> gpurun_out/wgrad_slabs.log
done
cat gpurun_out/limb_atomics.log gpurun_out/wgrad_slabs.log
