#!/bin/bash
# Collect the per-round measurement set on the GPU box (run from the repo root through gpurun):
#   bench lines of every workload, rocprofv3 kernel stats of C2 / C3 / C4, and the FETCH_SIZE / WRITE_SIZE / SQ counter passes of
#   each of the three (separate --pmc runs; the program itself directly after `--`), summarised by tools/pmc_summary.py into
#   bench_c{2,3,4}_<tag>_summary.json.  usage: [POOL50K=1] [PMC_C34=0] tools/collect_profiles.sh <tag>
tag=${1:-v0}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for w in c1 c1x16 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 20 --detail-out $out/bench_${w}_${tag}_detail.json 2>/dev/null | tail -1 > $out/bench_${w}_$tag.json
done
python3 bench.py --workload c3 --steps 20 --detail-out $out/bench_c3_${tag}_detail.json 2>/dev/null | tail -1 > $out/bench_c3_$tag.json   # (with its cpu_baseline)
python3 bench.py --detail-out $out/bench_c2_${tag}_detail.json 2>/dev/null | tail -1 > $out/bench_c2_$tag.json                            # (the driver's command: headline + exact_modes + companions)
export MVAL_STREAMS=1
export MVAL_TRAIN_LANES=0   # (likewise the training passes: one stream, so that a kernel's traced duration is its own)
export MVAL_GRAPH=0         # (eager launches under the profiler: the traces and counter passes of rounds 1-5 were all taken that way)
export MVAL_TRAIN_SLACK_CHECK=0   # (the training plan's one-off bound-slack probe -- 481 measurement launches on step 0 -- stays out of the traces)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 30 2>/dev/null | tail -1 > $out/bench_c2_${tag}_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3 -o c3 -- python3 bench.py --workload c3 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 5 2>/dev/null | tail -1 > $out/bench_c3_${tag}_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c4 -o c4 -- python3 bench.py --workload c4 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --steps 20 2>/dev/null | tail -1 > $out/bench_c4_${tag}_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 3 > /dev/null 2>&1
ks=$(find $out/kt_c2 -name '*kernel_stats.csv' | head -1)
k3=$(find $out/kt_c3 -name '*kernel_stats.csv' | head -1)
fe=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1)
wr=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
sq=$(find $out/pmc_sq -name '*counter_collection.csv' | head -1)
cp $ks $out/bench_c2_kernel_stats_$tag.csv
cp $k3 $out/bench_c3_kernel_stats_$tag.csv
k4=$(find $out/kt_c4 -name '*kernel_stats.csv' | head -1)
cp $k4 $out/bench_c4_kernel_stats_$tag.csv
l2=$(find $out/pmc_l2 -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $ks $fe $wr $sq $l2 > $out/bench_c2_${tag}_summary.json
rm -rf $out/kt_c2 $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_l2
if [ "${PMC_C34:-1}" != "0" ]; then   # the same three counter passes for the training step and the HRNet-W48 slice (every kernel >= 0.3 % of the trace)
  SQC="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  for w in c3 c4; do
    st=3; [ $w = c4 ] && st=5
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps $st --warmup 1 > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps $st --warmup 1 > /dev/null 2>&1
    rocprofv3 --pmc $SQC --output-format csv -d $out/pmc_sq_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 2 --warmup 1 > /dev/null 2>&1
    kk=$k3; [ $w = c4 ] && kk=$k4
    python3 tools/pmc_summary.py --all $kk $(find $out/pmc_fetch_$w -name '*counter_collection.csv' | head -1) $(find $out/pmc_write_$w -name '*counter_collection.csv' | head -1) \
        $(find $out/pmc_sq_$w -name '*counter_collection.csv' | head -1) > $out/bench_${w}_${tag}_summary.json
    rm -rf $out/pmc_fetch_$w $out/pmc_write_$w $out/pmc_sq_$w
  done
fi
rm -rf $out/kt_c3 $out/kt_c4
if [ -n "$POOL50K" ]; then   # the BASELINE-size pool passes (about 140 s each): unedited bench lines
  python3 bench.py --workload c4 --pool 50000 --no-cpu-baseline --warmup 0 --detail-out $out/bench_c4_pool50000_detail.json 2>/dev/null | tail -1 > $out/bench_c4_pool50000.json
  python3 bench.py --workload c5 --pool 50000 --no-cpu-baseline --warmup 0 --detail-out $out/bench_c5_pool50000_detail.json 2>/dev/null | tail -1 > $out/bench_c5_pool50000.json
fi
ls -la $out
