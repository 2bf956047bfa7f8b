#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/w48_stamps.log
rm -f $L
MVAL_LIB_TAG=stamp python tools/p2_stamps.py 48 48 96 72 3 1 64 >> $L 2>&1
MVAL_LIB_TAG=stamp python tools/p2_stamps.py 96 96 48 36 3 1 64 >> $L 2>&1
MVAL_LIB_TAG=stamp python tools/p2_stamps.py 64 64 32 32 3 1 128 >> $L 2>&1
cat $L
