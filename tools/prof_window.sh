#!/bin/bash
# kernel-trace stats of N LM iterations of the configs[3] window; usage: tools/prof_window.sh <tag> [iters]
TAG="${1:-w}"; IT="${2:-30}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o win -- python $ROOT/tools/run_full_window.py $IT $LVF_WINDOW_ARGS > "$OUT/trace.log" 2>&1
cd "$ROOT"; python tools/prof_summary.py "$OUT" 2>/dev/null | head -${3:-30}
tail -1 "$OUT/trace.log"
find "$OUT" -name '*.csv' -size +8M -delete
