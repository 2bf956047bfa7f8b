#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/lanes_soak.py 40 hrnet_w32 > gpurun_out/lanes_soak.log 2>&1; echo "rc $?" >> gpurun_out/lanes_soak.log
python tools/lanes_soak.py 20 hrnet_w48 >> gpurun_out/lanes_soak.log 2>&1; echo "rc $?" >> gpurun_out/lanes_soak.log
grep -a "repeats\|one-stream\|DIFFERS\|rc \|Error" gpurun_out/lanes_soak.log
