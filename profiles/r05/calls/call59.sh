#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/final3
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/final3/full_gpu_tests.log 2>&1
echo "rc $?" >> gpurun_out/final3/full_gpu_tests.log
grep -a -E "passed|failed|rc " gpurun_out/final3/full_gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
