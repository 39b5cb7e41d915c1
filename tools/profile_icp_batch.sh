#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the ICP half of the metric and the batched regime under the profiler.
#   trace  : rocprofv3 --kernel-trace --stats of bench.py --legs icp,scan_match_frame,relocalize_8_candidates
#   PMC    : separate passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT/MISS | SQ_INSTS_VALU, SQ_WAVES, SQ_BUSY_CYCLES) of the icp leg
#   batch  : kernel trace of tools/run_batch.py 30 <W> for W in 8, 64
# usage: tools/profile_icp_batch.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG="${1:-rXX}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ICP="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs icp,scan_match_frame,relocalize_8_candidates"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/icp_trace" -o icp -- $ICP > "$OUT/icp_trace.log" 2>&1
ICP1="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs icp,scan_match_frame"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/icp_pmc_fetch" -o icp -- $ICP1 > "$OUT/icp_pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/icp_pmc_write" -o icp -- $ICP1 > "$OUT/icp_pmc_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/icp_pmc_l2" -o icp -- $ICP1 > "$OUT/icp_pmc_l2.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/icp_pmc_valu" -o icp -- $ICP1 > "$OUT/icp_pmc_valu.log" 2>&1
for W in 8 64; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/batch${W}_trace" -o b -- python $ROOT/tools/run_batch.py 30 $W > "$OUT/batch${W}.log" 2>&1
done
# instruction counts of the single-window chain (the honest bound of k_lin_visual: fp64 VALU issue)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/win_pmc_valu" -o w -- python $ROOT/tools/run_batch.py 10 1 > "$OUT/win_pmc_valu.log" 2>&1
cd "$ROOT"
python tools/prof_icp_summary.py "$OUT" > "$OUT/icp_batch_summary.txt" 2>&1
python tools/prof_icp_summary.py "$OUT" --json > "$OUT/pmc_icp.json" 2>/dev/null
cat "$OUT/icp_batch_summary.txt"
grep -h '^{' "$OUT/icp_trace.log" | tail -1 > "$OUT/icp_bench_under_rocprof.json"
find "$OUT" -name '*.csv' -size +6M -delete
