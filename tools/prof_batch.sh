#!/bin/bash
# usage (on the GPU box): tools/prof_batch.sh <tag> <W>  -> gpurun_out/prof_<tag>/batch_stats.txt
TAG="${1:-rXX}"; W="${2:-8}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch_trace" -o b -- python $ROOT/tools/run_batch.py 30 $W > "$OUT/batch.log" 2>&1
cd "$ROOT"
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
f = glob.glob(out + "/batch_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("  calls    total_us     avg_us   pct  kernel")
for r in rows[:24]:
    print(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e3:11.1f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):5.1f}  {r['Name'][:90]}")
PY
tail -4 "$OUT/batch.log"
find "$OUT" -name '*.csv' -size +4M -delete
