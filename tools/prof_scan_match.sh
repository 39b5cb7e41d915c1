#!/bin/bash
# Runs ON THE GPU BOX: kernel + copy timeline of ONE configs[2] scan-match frame (lvf_scan_match).  usage: tools/prof_scan_match.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$ROOT/gpurun_out/prof_sm; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -o t -- python $ROOT/tools/run_scan_match.py 10 > $OUT/log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/t/**/*kernel_trace.csv', recursive=True)[0]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-34:]) for r in csv.DictReader(open(f))]
for m in glob.glob(out + '/t/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(m)): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') ))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_sm_step' in e[2]]
# frames: a frame has several k_sm_step; take the last ~40 events before the end
last = idx[-1]
start = last
while start > 0 and ev[start][0] - ev[start - 1][1] < 150000: start -= 1
t0 = ev[start][0]
busy = 0
for e in ev[start:last + 3]:
    busy += e[1] - e[0]
    print("%9.1f %8.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
print("span %.1f us, busy %.1f us" % ((ev[min(last + 2, len(ev) - 1)][1] - t0) / 1e3, busy / 1e3))
PY
tail -2 $OUT/log
