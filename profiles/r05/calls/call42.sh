#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
export MVAL_TRAIN_SLACK_CHECK=0
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_l -o l -- python3 $GRAFT_REPO_ROOT/bench.py --workload c3 --no-cpu-baseline --no-exact-modes --no-companions --no-rooflines --min-timed-seconds 0 --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/kt_l -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'P' > gpurun_out/c3_lane_occupancy.log
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Stream_Id',0) or 0)) for r in rows]
ev.sort()
# last 4 steps: find adam_step_kernel occurrences
adam=[e for e in ev if 'adam_step' in e[2]]
print('adam steps', len(adam))
t0=adam[-4][1]; t1=adam[-1][1]
win=[e for e in ev if e[0]>=t0 and e[1]<=t1]
span=(t1-t0)/3
print('step span ms', span/1e6, 'kernels per step', len(win)/3)
# union coverage and concurrency histogram
pts=[]
for s,e,_,_ in win: pts.append((s,1)); pts.append((e,-1))
pts.sort()
cur=0; last=t0; cov=0; hist={}
for t,d in pts:
    if t>last:
        hist[cur]=hist.get(cur,0)+(t-last)
        if cur>0: cov+=t-last
        last=t
    cur+=d
tot=sum(hist.values())
print('covered fraction', cov/(t1-t0))
for k in sorted(hist): print('concurrency',k, round(hist[k]/tot,4))
busy=sum(e-s for s,e,_,_ in win)/3
print('sum of kernel durations per step ms', busy/1e6)
# idle gaps by preceding kernel
P
cat gpurun_out/c3_lane_occupancy.log
