#!/bin/bash
ROOT=/root/repo; OUT=$ROOT/gpurun_out/prof_tick; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --steps 20 --no-cpu-baseline --legs window_tick > $OUT/log 2>&1
cd $ROOT
python - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_tick/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
mc=glob.glob('/root/repo/gpurun_out/prof_tick/t/**/*memory_copy_trace.csv',recursive=True)
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-28:]) for r in rows]
if mc:
    for r in csv.DictReader(open(mc[0])): ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')+' '+r.get('Bytes','')))
ev.sort()
# find last tick: locate last k_lm_sort
idx=[i for i,e in enumerate(ev) if 'k_lm_sort' in e[2]]
i0=idx[-1]
start=i0
while start>0 and ev[start][0]-ev[start-1][1] < 300000: start-=1
t0=ev[start][0]
for e in ev[start:start+60]:
    print(f"{(e[0]-t0)/1e3:9.1f} {(e[1]-e[0])/1e3:8.1f}  {e[2]}")
PY
