#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): ONE rocprofv3 --kernel-trace --stats pass over chosen bench legs, summarised as text.
# usage: tools/prof_legs.sh <tag> <comma list of bench legs>     output: gpurun_out/prof_<tag>/legs_summary.txt
set -u
TAG="${1:-rXX}"; LEGS="${2:-icp,scan_match_frame,relocalize_8_candidates,map_maintenance}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/legs_trace" -o legs -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs $LEGS > "$OUT/legs_trace.log" 2>&1
cd "$ROOT"
{ echo "## bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs $LEGS"; python - "$OUT/legs_trace" <<'PY'
import sys
sys.path.insert(0, "tools")
import prof_summary
prof_summary.kernel_stats(sys.argv[1])
PY
} > "$OUT/legs_summary.txt" 2>&1
grep -h '^{' "$OUT/legs_trace.log" | tail -1 > "$OUT/legs_bench_under_rocprof.json"
find "$OUT" -name '*.csv' -size +6M -delete
head -30 "$OUT/legs_summary.txt"
