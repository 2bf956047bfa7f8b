#!/bin/bash
# round 6, call 12: last check after the bench.py label split and the pre-summed state fix: training + distributed tests, driver command
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
L=gpurun_out/r6/call12.log
rm -f $L
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_distributed.py -q -m gpu 2>&1 | tail -4 >> $L
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/bench_driver_cmd_last.json 2> gpurun_out/r6/bench_driver_cmd.stderr ) 2>> $L
wc -c gpurun_out/r6/bench_driver_cmd_last.json >> $L
cp bench_detail.json gpurun_out/r6/bench_driver_cmd_last_detail.json
python3 - >> $L <<'PY'
import json
d = json.loads(open("gpurun_out/r6/bench_driver_cmd_last.json").read().strip().splitlines()[-1])
print("parsed:", d["ms_per_step"], d["value"], d["roofline"], d["cpu_baseline"]["value"], d["exact_modes"], d["companions"])
det = json.load(open("gpurun_out/r6/bench_driver_cmd_last_detail.json"))
print("detail kernel:", det["roofline"]["kernel"], "| note:", det["roofline"].get("kernel_note", "")[:80], "| others:", [f["kernel"] for f in det["roofline"]["other_kernels"]][:12])
PY
cat $L
