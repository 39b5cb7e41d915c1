#!/bin/bash
# Runs ON THE GPU BOX: the HIP API calls of one window tick, in order (which copies / launches a tick makes).  usage: tools/prof_tick_api.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$ROOT/gpurun_out/prof_tick_api; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --hip-runtime-trace --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --legs window_tick > $OUT/log 2>&1
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/t/**/*hip_api_trace.csv', recursive=True)
if not f: print("no hip api trace", glob.glob(sys.argv[1] + '/t/**/*', recursive=True)[:20]); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Function'] for r in rows]
# last tick: from the last-but-one hipStreamSynchronize-terminated group containing k_window... approximate: print the last 120 calls
t0 = int(rows[-140]['Start_Timestamp']) if len(rows) > 140 else int(rows[0]['Start_Timestamp'])
for r in rows[-140:]:
    print("%9.1f %7.1f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Function']))
PY
