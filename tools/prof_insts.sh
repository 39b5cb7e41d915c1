#!/bin/bash
# usage (on the GPU box): tools/prof_insts.sh <tag>  -> gpurun_out/prof_<tag>/insts.txt : dynamic instruction mix per kernel and wave
TAG="${1:-rXX}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i + 1))
  timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/insts_$i" -o p -- python $ROOT/tools/run_batch.py 6 1 > "$OUT/insts_$i.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/insts_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(out + "/insts.txt", "w") as fo:
    for k in sorted(acc):
        if not k.startswith("lvf::"): continue
        line = k + ": " + "  ".join(f"{c}={acc[k][c] / max(cnt[k][c], 1):.0f}" for c in names)
        print(line); fo.write(line + "\n")
PY
find "$OUT" -name '*.csv' -size +4M -delete
