#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/gpurun_out/pp_lds.log
rm -f $L
timeout 600 python -m pytest tests/ -q -m gpu -k "preprocess or prepare or input" 2>&1 | grep -a -E "passed|failed|Error" >> $L
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pp -o pp -- python3 $GRAFT_REPO_ROOT/tools/preprocess_bench.py 2>/dev/null | grep -a "resize_views\|heat" >> $L
f=$(find /tmp/kt_pp -name '*kernel_stats.csv' | head -1)
head -6 $f | cut -c1-160 >> $L
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  python bench.py --with-input --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>&1 | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 with input', d['ms_per_step'])" >> $L 2>&1
done
cat $L
