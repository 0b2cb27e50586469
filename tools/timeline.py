#!/usr/bin/env python3
"""Timeline of one EM iteration from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):
every dispatch between two consecutive solver launches of a chosen iteration with its start offset, duration and the
idle gap before it -- what the fixed per-iteration cost (everything that is not solver / post kernel) consists of.

  timeline.py <kernel_trace.csv> [iteration (default 6)]
"""
import csv, re, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
want = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"<.*", "", n)
    return n.replace("stm::", "").replace("void ", "")[:34]


# an iteration starts at the first solver dispatch after a post-kernel dispatch
its, cur, seen_post = [], [], True
for r in rows:
    n = r["Kernel_Name"]
    if "solver_kernel" in n and seen_post:
        if cur:
            its.append(cur)
        cur, seen_post = [], False
    if "post_kernel" in n or "post_any_kernel" in n or "post_big2_kernel" in n:
        seen_post = True
    cur.append(r)
its.append(cur)
it = its[want]
t0 = int(it[0]["Start_Timestamp"])
nxt = int(its[want + 1][0]["Start_Timestamp"]) if want + 1 < len(its) else None
prev_end = t0
busy = gaps = 0.0
print(f"EM iteration {want}: {len(it)} dispatches")
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  kernel")
for r in it:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {short(r['Kernel_Name'])}")
    busy += (e - s) / 1e3
    gaps += max(gap, 0.0)
    prev_end = max(prev_end, e)
if nxt:
    tail = (nxt - prev_end) / 1e3
    big = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in it if "solver_kernel" in r["Kernel_Name"] or "post_kernel" in r["Kernel_Name"] or "post_big" in r["Kernel_Name"])
    print(f"iteration period {(nxt - t0) / 1e3:.1f} us: solver + post {big:.1f}, other kernels {busy - big:.1f}, idle between kernels {gaps:.1f}, idle before the next solver {tail:.1f}")
