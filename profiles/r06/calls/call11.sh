#!/bin/bash
# round 6, call 11 (last): the whole GPU suite and the driver's command at the final code; the C3 part of the collection again
# (kernel stats + counter passes + bench line: the data-gradient epilogue sums now also cover producers with residuals)
export TMPDIR=/tmp
mkdir -p gpurun_out/r6 gpurun_out/prof_r6b
L=gpurun_out/r6/call11.log
rm -f $L
out=gpurun_out/prof_r6b
tag=r6
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -4 >> $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $L
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_r6.json 2> gpurun_out/r6/bench_driver_cmd.stderr ) 2>> $L
cp bench_detail.json $out/bench_driver_cmd_r6_detail.json
wc -c $out/bench_driver_cmd_r6.json >> $L
python3 bench.py --workload c3 --steps 20 --detail-out $out/bench_c3_${tag}_detail.json 2>/dev/null | tail -1 > $out/bench_c3_$tag.json
export MVAL_STREAMS=1 MVAL_TRAIN_LANES=0 MVAL_GRAPH=0 MVAL_TRAIN_SLACK_CHECK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3 -o c3 -- python3 bench.py --workload c3 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 5 2>/dev/null | tail -1 > $out/bench_c3_${tag}_under_rocprof.json
k3=$(find $out/kt_c3 -name '*kernel_stats.csv' | head -1)
cp $k3 $out/bench_c3_kernel_stats_$tag.csv
SQC="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
w=c3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc $SQC --output-format csv -d $out/pmc_sq_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 2 --warmup 1 > /dev/null 2>&1
python3 tools/pmc_summary.py --all $k3 $(find $out/pmc_fetch_$w -name '*counter_collection.csv' | head -1) $(find $out/pmc_write_$w -name '*counter_collection.csv' | head -1) $(find $out/pmc_sq_$w -name '*counter_collection.csv' | head -1) > $out/bench_${w}_${tag}_summary.json
rm -rf $out/kt_c3 $out/pmc_fetch_$w $out/pmc_write_$w $out/pmc_sq_$w
ls -la $out >> $L
cat $L
