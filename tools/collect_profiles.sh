#!/bin/bash
# Collect the per-round measurement set on the GPU box (run from the repo root through gpurun):
#   bench lines of every workload, rocprofv3 kernel stats of C2 / C3, and the FETCH_SIZE / WRITE_SIZE counter passes of
#   C2 (separate --pmc runs), summarised by tools/pmc_summary.py.  usage: [POOL50K=1] tools/collect_profiles.sh <tag>
tag=${1:-v0}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for w in c1 c1x16 c4 c5; do
  python3 bench.py --workload $w --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > $out/bench_${w}_$tag.json
done
python3 bench.py --workload c3 --steps 20 2>/dev/null | tail -1 > $out/bench_c3_$tag.json   # (with its cpu_baseline)
python3 bench.py 2>/dev/null | tail -1 > $out/bench_c2_$tag.json                            # (the driver's command: headline + exact_modes + companions)
export MVAL_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c2 -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 30 2>/dev/null | tail -1 > $out/bench_c2_${tag}_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c3 -o c3 -- python3 bench.py --workload c3 --no-cpu-baseline --steps 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_c4 -o c4 -- python3 bench.py --workload c4 --no-cpu-baseline --steps 20 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_sq -o c2 -- python3 bench.py --no-cpu-baseline --no-exact-modes --no-companions --steps 3 > /dev/null 2>&1
ks=$(find $out/kt_c2 -name '*kernel_stats.csv' | head -1)
k3=$(find $out/kt_c3 -name '*kernel_stats.csv' | head -1)
fe=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1)
wr=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
sq=$(find $out/pmc_sq -name '*counter_collection.csv' | head -1)
cp $ks $out/bench_c2_kernel_stats_$tag.csv
cp $k3 $out/bench_c3_kernel_stats_$tag.csv
k4=$(find $out/kt_c4 -name '*kernel_stats.csv' | head -1)
cp $k4 $out/bench_c4_kernel_stats_$tag.csv
python3 tools/pmc_summary.py $ks $fe $wr $sq > $out/bench_c2_${tag}_summary.json
rm -rf $out/kt_c2 $out/kt_c3 $out/kt_c4 $out/pmc_fetch $out/pmc_write $out/pmc_sq
if [ -n "$POOL50K" ]; then   # the BASELINE-size pool passes (about 140 s each): unedited bench lines
  python3 bench.py --workload c4 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | tail -1 > $out/bench_c4_pool50000.json
  python3 bench.py --workload c5 --pool 50000 --no-cpu-baseline --warmup 0 2>/dev/null | tail -1 > $out/bench_c5_pool50000.json
fi
ls -la $out
