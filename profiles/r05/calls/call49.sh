#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "lanes or golden or full_size_prop" 2>&1 | grep -a -E "passed|failed|rror" | tail -3
python bench.py --workload c3 --no-cpu-baseline --no-rooflines --no-companions --steps 20 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'])"
