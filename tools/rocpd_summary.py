#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace as text: per-kernel calls, total,
average, min, max duration and share -- the same table `--stats` prints -- so it can be committed
under profiles/.   usage: rocpd_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in microseconds)",
             f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>6s} "
             f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s}"]
    for r in rows:
        lines.append(f"{r[0][:70]:70s} {r[1]:6d} {r[2] / 1e3:12.1f} {r[3] / 1e3:12.1f} {r[4] / 1e3:12.1f} {r[5] / 1e3:12.1f} "
                     f"{100 * r[2] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:7d}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
