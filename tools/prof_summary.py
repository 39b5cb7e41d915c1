#!/usr/bin/env python
"""Summarise a tools/profile_gpu.sh output directory: per-kernel time statistics (rocprofv3 --kernel-trace --stats)
and per-kernel average PMC counters (separate --pmc passes), as text for committing under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name, n=100):
    return name if len(name) <= n else name[:n - 3] + "..."


def kernel_stats(d):
    f = find(d, "kernel_stats.csv")
    if not f:
        print("# no kernel_stats.csv under", d)
        return
    print("# rocprofv3 --kernel-trace --stats   (ns)")
    print(f"{'calls':>7} {'total_ns':>14} {'avg_ns':>12} {'min_ns':>10} {'max_ns':>10} {'pct':>6}  kernel")
    for r in csv.DictReader(open(f)):
        print(f"{int(r['Calls']):7d} {int(float(r['TotalDurationNs'])):14d} {float(r['AverageNs']):12.1f} {int(float(r['MinNs'])):10d} "
              f"{int(float(r['MaxNs'])):10d} {float(r['Percentage']):6.2f}  {short(r['Name'])}")
    f = find(d, "kernel_trace.csv")
    if f:
        print("\n# resources (first dispatch of each kernel): grid, workgroup, lds_bytes, vgpr, accum_vgpr, sgpr, scratch")
        seen = set()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if n in seen:
                continue
            seen.add(n)
            print(f"{r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9} {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5} {r.get('LDS_Block_Size', '?'):>7} "
                  f"{r.get('VGPR_Count', '?'):>4} {r.get('Accum_VGPR_Count', '?'):>4} {r.get('SGPR_Count', '?'):>4} {r.get('Scratch_Size', '?'):>5}  {short(n)}")


def pmc(d, label):
    f = find(d, "counter_collection.csv")
    if not f:
        print(f"# {label}: no counter_collection.csv under {d}")
        return
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"\n# PMC pass: {label}   (average counter value per dispatch; FETCH_SIZE/WRITE_SIZE are in KiB)")
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        print("  " + "  ".join(f"{c}={v[0] / v[1]:.1f} (n={v[1]})" for c, v in sorted(cs.items())) + "  " + short(k, 90))


def pmc_json(root, tag):
    """per-kernel average counters of all PMC passes -> dict for profiles/pmc_latest.json (read by bench.py's roofline.traffic)"""
    out = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_mfma"):
        f = find(os.path.join(root, sub), "counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, cs in acc.items():
            for c, v in cs.items():
                out.setdefault(k, {})[c] = v[0] / v[1]
    return {"tag": tag, "units": "average per dispatch; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-counts wide reads 2x on gfx950)", "kernels": out}


if __name__ == "__main__":
    root = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "--json":
        import json
        print(json.dumps(pmc_json(root, os.path.basename(root.rstrip("/"))), indent=1))
        sys.exit(0)
    kernel_stats(os.path.join(root, "trace"))
    for sub, label in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_l2", "TCC_HIT_sum TCC_MISS_sum"),
                       ("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE")):
        if os.path.isdir(os.path.join(root, sub)):
            pmc(os.path.join(root, sub), label)
    if os.path.isdir(os.path.join(root, "win_trace")):
        print("\n## tools/run_full_window.py 30  (configs[3]: 50 KF / 10 k landmarks, one LM iteration per step)")
        kernel_stats(os.path.join(root, "win_trace"))
        pmc(os.path.join(root, "win_pmc_mfma"), "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE (MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * 4 SIMDs * CUs))")
