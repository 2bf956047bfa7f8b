#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/with_input.log
rm -f $L
for r in 1 2; do
for w in c2 c4; do
  python bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w resident', d['ms_per_step'])" >> $L 2>&1
  python bench.py --workload $w --with-input --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 100 2>&1 | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w with input', d['ms_per_step'], d['input_inclusive']['host_to_device_GBps'], d.get('parity_sample'))" >> $L 2>&1
done
done
timeout 600 python -m pytest tests/ -q -m gpu -k "with_input or preprocess or prepare" 2>&1 | grep -a -E "passed|failed" >> $L
cat $L
