#!/bin/bash
# round 6, call 15 (item 2b, measured instead of argued): (i) every op's slab reduction reading its slabs from HBM instead of the Infinity
# Cache (MVAL_WGRAD_SLAB_ROT=48: 48 slab regions per lane walked op by op -- the traffic a reduction deferred to the end of a segment has);
# (ii) no slab reductions at all (measurement build, MVAL_TRAIN_ABL=4): the upper bound of batching the 293 launches; (iii) lanes soak
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call15.log
rm -f $L
for r in 1 2 3; do
  python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product c3', d['ms_per_step'])" >> $L 2>&1
  MVAL_WGRAD_SLAB_ROT=48 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab regions rotated (48 per lane) c3', d['ms_per_step'])" >> $L 2>&1
  MVAL_LIB_TAG=abl MVAL_TRAIN_ABL=4 MVAL_TRAIN_SLACK_CHECK=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no slab reductions (upper bound) c3', d['ms_per_step'])" >> $L 2>&1
done
for v in "" 48; do
  MVAL_WGRAD_SLAB_ROT=${v:-1} MVAL_TRAIN_LANES=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream, slab rot ${v:-1}: c3', d['ms_per_step'])" >> $L 2>&1
done
MVAL_LIB_TAG=abl MVAL_TRAIN_ABL=4 MVAL_TRAIN_LANES=0 MVAL_TRAIN_SLACK_CHECK=0 python bench.py --workload c3 --no-cpu-baseline --no-rooflines --steps 20 --detail-out '' 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream, no slab reductions: c3', d['ms_per_step'])" >> $L 2>&1
python tools/lanes_soak.py 30 hrnet_w32 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
cat $L
