#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/call10_all.log 2>&1
echo "rc $?" >> gpurun_out/call10_all.log
tail -12 gpurun_out/call10_all.log
