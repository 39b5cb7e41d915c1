#!/bin/bash
# Runs ON THE GPU BOX: HBM counter bytes of the BATCHED launch chain per window-iteration (bench.py's roofline_batched.traffic).
# FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots), each over
# `tools/run_batch.py <iters> <W> batchonly` = ONE batched solve of W configs[3] windows, <iters> LM iterations each (tolerances off).
# usage: tools/prof_batch_pmc.sh <tag> [W=64] [iters=10]   -> gpurun_out/prof_<tag>/pmc_batch.json (copy to profiles/pmc_batch_latest.json)
set -u
TAG="${1:-rXX}"; W="${2:-64}"; IT="${3:-10}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmcb_$C" -o b -- python $ROOT/tools/run_batch.py $IT $W batchonly > "$OUT/pmcb_$C.log" 2>&1
done
cd "$ROOT"
python - "$OUT" "$TAG" "$W" "$IT" <<'PY' > "$OUT/pmc_batch.json"
import sys, glob, csv, json, collections
out, tag, W, IT = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + f"/pmcb_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "lvf::" not in n or r["Counter_Name"] != c:
            continue
        tot[n][c] += float(r["Counter_Value"]); calls[n][c] += 1
wi = float(W * IT)                                   # window-iterations of the one batched solve
by = {}
for n in tot:
    # KiB -> bytes; FETCH_SIZE x2: the gfx950 half-count of wide reads (MI355X_MICROARCH.md, HBM section)
    b = (2.0 * tot[n].get("FETCH_SIZE", 0.0) + tot[n].get("WRITE_SIZE", 0.0)) * 1024.0 / wi
    by[n.split("(")[0].replace("lvf::", "")] = {"bytes_per_window_iteration": b, "fetch_kib_raw_total": tot[n].get("FETCH_SIZE", 0.0), "write_kib_total": tot[n].get("WRITE_SIZE", 0.0),
                                                 "dispatches": calls[n].get("FETCH_SIZE", 0)}
total = sum(v["bytes_per_window_iteration"] for v in by.values())
top = dict(sorted(((k, v["bytes_per_window_iteration"]) for k, v in by.items()), key=lambda kv: -kv[1])[:12])
print(json.dumps({"tag": tag, "windows": W, "iterations": IT, "bytes_per_window_iteration": total, "by_kernel_bytes_per_window_iteration": top,
                  "units": "bytes of HBM traffic per window per LM iteration = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every lvf:: kernel of one batched solve / (windows x iterations); "
                           "includes the chain's one-off launches (first zeroing, table set-up)", "detail": by}, indent=1))
PY
python -c "import json;d=json.load(open('$OUT/pmc_batch.json'));print('bytes per window-iteration: %.1f MB' % (d['bytes_per_window_iteration']/1e6)); [print('  %-28s %8.2f MB' % (k, v/1e6)) for k,v in d['by_kernel_bytes_per_window_iteration'].items()]"
find "$OUT" -name '*.csv' -size +8M -delete
