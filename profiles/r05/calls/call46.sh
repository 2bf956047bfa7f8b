#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/train_graph_dbg.log
for d in 2 3; do
  echo "== MVAL_DBG_FREE=$d" >> gpurun_out/train_graph_dbg.log
  MVAL_DBG_FREE=$d timeout 300 python tools/run/tg_debug.py 2>&1 | grep -a "step\|ok\|core\|rror" >> gpurun_out/train_graph_dbg.log
  echo "rc ${PIPESTATUS[0]}" >> gpurun_out/train_graph_dbg.log
done
cat gpurun_out/train_graph_dbg.log | cut -c1-200
