#!/bin/bash
# round 6, call 8: the round's measurement set (tools/collect_profiles.sh r6: bench lines + details of every workload, kernel stats of
# C2 / C3 / C4, FETCH / WRITE / SQ / L2 counter passes) and the BASELINE-size pool passes
export TMPDIR=/tmp
mkdir -p gpurun_out
POOL50K=1 bash tools/collect_profiles.sh r6 > gpurun_out/collect_r6.log 2>&1
tail -40 gpurun_out/collect_r6.log
