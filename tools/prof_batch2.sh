#!/bin/bash
# usage (on the GPU box): tools/prof_batch2.sh <tag> <W...>  -> gpurun_out/prof_<tag>/batch<W>_stats.txt (kernel trace stats of tools/run_batch.py 30 W)
TAG="${1:-rXX}"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
for W in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch${W}_trace" -o b -- python $ROOT/tools/run_batch.py 30 $W > "$OUT/batch${W}.log" 2>&1
  python - "$OUT/batch${W}_trace" <<'PY' > "$OUT/batch${W}_stats.txt"
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("  calls    total_us     avg_us   pct  kernel")
for r in rows[:20]:
    print(f"{int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e3:11.1f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):5.1f}  {r['Name'][:90]}")
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
seen = set()
for r in csv.DictReader(open(t[0])):
    n = r["Kernel_Name"]
    if n in seen or "lvf::k_" not in n: continue
    seen.add(n)
    print(f"# {n[:60]:60s} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')} agpr {r.get('Accum_VGPR_Count','?')} sgpr {r.get('SGPR_Count','?')} scratch {r.get('Scratch_Size','?')}")
PY
  grep -v "^[EWI]2026" "$OUT/batch${W}.log" | tail -2 >> "$OUT/batch${W}_stats.txt"
  cat "$OUT/batch${W}_stats.txt"
done
find "$OUT" -name '*.csv' -size +4M -delete
