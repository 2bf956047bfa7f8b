#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/dual_plan_probe.py hrnet_w32 128 256 256 100 > gpurun_out/dual_plan.log 2>&1
python tools/dual_plan_probe.py hrnet_w48 64 384 288 60 >> gpurun_out/dual_plan.log 2>&1
grep -a "plan" gpurun_out/dual_plan.log
