#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a text table (what `--stats` prints as CSV)."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
        "max(lds_size), max(scratch_size), max(workgroup_x), max(grid_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in ns)",
             f"{'calls':>6} {'total_ns':>14} {'avg_ns':>12} {'min_ns':>12} {'max_ns':>12} {'pct':>6} {'vgpr':>5} {'sgpr':>5} {'lds':>6} {'scratch':>7} {'wg':>5} {'grid':>10}  name"]
    for name, calls, tot, avg, mn, mx, vg, sg, lds, scr, wg, grid in rows:
        lines.append(f"{calls:>6} {tot:>14} {avg:>12.0f} {mn:>12} {mx:>12} {100.0*tot/total:>6.2f} {vg or 0:>5} {sg or 0:>5} {lds or 0:>6} {scr or 0:>7} {wg or 0:>5} {grid or 0:>10}  {name}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
