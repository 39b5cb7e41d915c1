#!/usr/bin/env python
"""Dump the per-kernel statistics of a rocprofv3 rocpd database (..._results.db) as text — the same numbers
`rocprofv3 --kernel-trace --stats` prints — for committing under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# rocprofv3 --kernel-trace --stats  (durations in microseconds)", f"# source: {db}",
             f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"{calls:7d} {tot:12.3f} {avg:10.3f} {pct:6.2f}  {name}")
    lines.append("\n# resources (first dispatch of each kernel): grid, workgroup, lds_bytes, vgpr, sgpr, scratch")
    seen = set()
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels order by id"):
        if r[0] in seen:
            continue
        seen.add(r[0])
        lines.append(f"{r[1]:9d} {r[2]:5d} {r[3]:7d} {r[4]:4d} {r[5]:4d} {r[6]:5d}  {r[0][:110]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
