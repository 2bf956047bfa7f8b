#!/bin/bash
# round 6, call 2: the rest of the rehearsal tests + the conv main-loop skeleton's issue-order / prefetch-depth variants
export TMPDIR=/tmp
mkdir -p gpurun_out/r6 tools/micro/bin
L=gpurun_out/r6/call02.log
rm -f $L
hipcc --offload-arch=gfx950 -O3 tools/micro/p2_loop.hip -o tools/micro/bin/p2_loop 2>&1 | tail -3 >> $L
for x in 24 19; do tools/micro/bin/p2_loop 512 $x v >> $L 2>&1; done
timeout 1500 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "rehearsal" 2>&1 | tail -15 >> $L
cat $L
