#!/usr/bin/env python
"""Summarise a tools/profile_icp_batch.sh output directory as text for profiles/: per-kernel time statistics of the ICP legs and of
the batched windows, and per-kernel average PMC counters of the ICP legs (separate --pmc passes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import kernel_stats, pmc, find   # noqa: E402

PASSES = (("icp_pmc_fetch", "FETCH_SIZE"), ("icp_pmc_write", "WRITE_SIZE"), ("icp_pmc_l2", "TCC_HIT_sum TCC_MISS_sum"),
          ("icp_pmc_valu", "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"), ("win_pmc_valu", "single window: SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"))


def as_json(root):
    out = {}
    for sub, _ in PASSES:
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
    return {"tag": os.path.basename(root.rstrip("/")), "units": "average per dispatch; FETCH_SIZE/WRITE_SIZE in KiB", "kernels": out}


if __name__ == "__main__":
    root = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "--json":
        print(json.dumps(as_json(root), indent=1))
        sys.exit(0)
    print("## bench.py --legs icp,scan_match_frame,relocalize_8_candidates")
    kernel_stats(os.path.join(root, "icp_trace"))
    for sub, label in PASSES:
        if os.path.isdir(os.path.join(root, sub)):
            pmc(os.path.join(root, sub), label)
    for W in (8, 64):
        d = os.path.join(root, f"batch{W}_trace")
        if os.path.isdir(d):
            print(f"\n## tools/run_batch.py 30 {W}")
            kernel_stats(d)
            log = os.path.join(root, f"batch{W}.log")
            if os.path.exists(log):
                print("".join("# " + l for l in open(log).readlines() if not l[:5] in ("E2026", "W2026", "I2026"))[-600:])
