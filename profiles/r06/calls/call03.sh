#!/bin/bash
# round 6, call 3: the main loop's issue order / prefetch depth in the product kernels (conv_p2.hip, conv_block_p2.hip): parity, then
# per-operator times and step times of the variant libraries ("" = product defaults; r5k = rounds 3-5's knobs; o1 = order only;
# xw = ring depths only; prio = s_setprio only; o2 = P2_ORDER 2)
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call03.log
rm -f $L
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_p2.py -q -m gpu -x 2>&1 | tail -4 >> $L
for t in "" r5k o1 xw prio o2; do
  [ -n "$t" ] && [ ! -f multi_view_active_learning_amd/csrc/libmval_hip_$t.so ] && continue
  echo "=== variant '$t' op_times w32" >> $L
  MVAL_LIB_TAG=$t python tools/op_times.py 128 hrnet_w32 2>/dev/null | head -14 >> $L
done
for r in 1 2; do
for t in "" r5k o1 xw prio o2; do
  [ -n "$t" ] && [ ! -f multi_view_active_learning_amd/csrc/libmval_hip_$t.so ] && continue
  for w in c2 c4; do
    MVAL_LIB_TAG=$t python bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" $w', d['ms_per_step'])" >> $L 2>&1
  done
done
done
for t in "" r5k; do
  MVAL_LIB_TAG=$t python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$t\" c3', d['ms_per_step'])" >> $L 2>&1
done
cat $L
