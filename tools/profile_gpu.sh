#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats of the default bench + separate PMC passes (FETCH_SIZE / WRITE_SIZE
# cannot share a pass on gfx950: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2 — MI355X_MICROARCH.md §rocprofv3 PMC slots).
# usage: tools/profile_gpu.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG="${1:-rXX}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# the headline (configs[3] LM iterations) + the K1 and batched-windows legs under the kernel trace
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --legs pose_only_K1,batched_windows_8"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
PMCB="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs pose_only_K1"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $PMCB > "$OUT/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $PMCB > "$OUT/pmc_write.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o bench -- $PMCB > "$OUT/pmc_l2.log" 2>&1
# the matrix-core counters of the same command (band Schur complement, Cholesky tile products)
PMCM="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o bench -- $PMCM > "$OUT/pmc_mfma.log" 2>&1
cd "$ROOT"
python tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
python tools/prof_summary.py "$OUT" --json > "$OUT/pmc.json" 2>/dev/null
tail -3 "$OUT/trace.log"
cat "$OUT/summary.txt"
# keep the merged payload small: the raw per-dispatch CSVs can be large
find "$OUT" -name '*.csv' -size +8M -delete
