#!/bin/bash
# round 6, call 1: the new GPU tests (2-rank DDP training step, bench.py N>1 rehearsal, unconditional MPJPE) + the driver's command
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call01.log
rm -f $L
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_models.py -q -m gpu -x -s -k "two_rank or rehearsal or c2_full_size or rccl_world_1" 2>&1 | tail -25 >> $L
echo "--- driver command" >> $L
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_driver_cmd.stdout 2> gpurun_out/r6/bench_driver_cmd.stderr ) 2>> $L
wc -c gpurun_out/r6/bench_driver_cmd.stdout >> $L
tail -c 4200 gpurun_out/r6/bench_driver_cmd.stdout >> $L
cp bench_detail.json gpurun_out/r6/bench_c2_detail_call01.json
tail -5 gpurun_out/r6/bench_driver_cmd.stderr >> $L
cat $L
