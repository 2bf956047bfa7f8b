#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/pool_sustained.log
rm -f $L gpurun_out/smi_pool.csv gpurun_out/smi_pool.csv.stop
python tools/smi_trace.py gpurun_out/smi_pool.csv 2.0 400 &
SP=$!
sleep 4
for p in 4000 16000 32000; do
  python bench.py --workload c4 --pool $p --no-cpu-baseline --no-rooflines --warmup 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool $p: ms per batch', d['ms_per_step']/($p/8))" >> $L 2>&1
done
touch gpurun_out/smi_pool.csv.stop
wait $SP
cat $L
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/smi_pool.csv')))
h=rows[0]; print(h)
for r in rows[1::6]: print(r)
P
