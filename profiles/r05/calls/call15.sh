#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/call15_all.log 2>&1
echo "rc $?" >> gpurun_out/call15_all.log
tail -5 gpurun_out/call15_all.log
for w in c4 c2 c1x16; do
python bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'])"
done
