#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_distributed.py -q -m gpu -k "graph or epilogue or rccl or ranks" 2>&1 | grep -a -E "passed|failed|rror" | tail -3
# the N > 1 launch shape on one GPU: torchrun world 1 with the process group up (watchdog thread alive) while the plan captures
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --rccl-world-1 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 50 2>/dev/null | grep -a '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun world 1 c2', d['ms_per_step'], d['config']['network_launch'][:25])"
