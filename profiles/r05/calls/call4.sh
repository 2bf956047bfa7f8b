#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/micro/bin/p2_loop 512 > gpurun_out/p2_loop.log 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/call4_tests.log 2>&1
echo "rc $?" >> gpurun_out/call4_tests.log
cat gpurun_out/p2_loop.log; tail -5 gpurun_out/call4_tests.log
