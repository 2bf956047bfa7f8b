#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 1 0; do
cd /tmp; rm -rf /tmp/kt_c2
MVAL_GRAPH=$g rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 12 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt_c2 -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$g" <<'P' >> gpurun_out/c2_stream_occupancy.log
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
stem=[e for e in ev if 'conv_stem_p2_kernel' in e[2]]
print('MVAL_GRAPH=%s: kernels traced %d, forwards %d' % (sys.argv[2], len(ev), len(stem)))
if len(stem) >= 10:
    t0=stem[-9][0]; t1=stem[-1][0]; n=8
    win=[e for e in ev if e[0]>=t0 and e[0]<t1]
    pts=[]
    for s,e,_ in win: pts.append((s,1)); pts.append((min(e,t1),-1))
    pts.sort(); cur=0; last=t0; hist={}
    for t,d in pts:
        if t>last: hist[cur]=hist.get(cur,0)+(t-last); last=t
        cur+=d
    tot=sum(hist.values())
    print('  step span ms %.3f, kernels per step %.1f, sum of kernel durations per step ms %.3f' % ((t1-t0)/n/1e6, len(win)/n, sum(e-s for s,e,_ in win)/n/1e6))
    print('  concurrency share:', {k: round(v/tot,4) for k,v in sorted(hist.items())})
P
done
cat gpurun_out/c2_stream_occupancy.log
