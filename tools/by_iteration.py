#!/usr/bin/env python3
"""Group rocprofv3 per-dispatch records of a `bench.py --steps K --warmup W` run by EM iteration.

The post kernel is launched exactly once per EM iteration, the solver once per LDS-occupancy class of the
longest-first document order (several dispatches per iteration): dispatches are walked in order and an
iteration starts with the first solver dispatch behind a post-kernel dispatch (the beta_ss pass that follows the
K <= 64 post kernel belongs to the iteration of that post kernel).  The first W iterations are the warm-up (EM iterations
0..W-1 of a fit that is then reset), the next K the timed EM iterations 0..K-1.

  by_iteration.py trace  <kernel_trace.csv> W K          -> per-iteration kernel durations (ms), grouped 0 / 1-3 / 4 / 5+
  by_iteration.py pmc    <counter_collection.csv> W K    -> per-group sums of every counter, per kernel
"""
import collections, csv, sys

mode, path, W, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = list(csv.DictReader(open(path)))


def kname(r):
    n = r["Kernel_Name"]
    if "solver_kernel" in n:
        return "solver"
    if "post_any_kernel" in n or "post_kernel" in n or "post_big2_kernel" in n:
        return "post"
    if "beta_ss_part" in n or "beta_ss_reduce_kernel" in n:
        return "betass"      # the word-major beta_ss pass behind the K <= 64 post kernel (round 3)
    return None


def group_of(it):
    return "it0" if it == 0 else "it1-3" if it <= 3 else "it4" if it == 4 else "it5+"


if mode == "trace":
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    its, cur, seen_post = [], collections.defaultdict(float), False
    for r in rows:
        k = kname(r)
        if not k:
            continue
        if k == "solver" and seen_post:
            its.append(cur); cur, seen_post = collections.defaultdict(float), False
        cur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cur[k + "_n"] += 1
        if k == "post":
            seen_post = True
    its.append(cur)
    timed = its[W:W + K]
    print(f"{len(its)} EM iterations in the trace ({W} warm-up, {len(timed)} timed); kernel time per EM iteration, ms")
    print(f"{'EM it':>6s} {'solver':>9s} {'(launches)':>10s} {'post':>9s} {'beta_ss':>9s}")
    for i, t in enumerate(timed):
        print(f"{i:6d} {t['solver']:9.3f} {int(t['solver_n']):10d} {t['post']:9.3f} {t['betass']:9.3f}")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, t in enumerate(timed):
        for k in ("solver", "post", "betass"):
            agg[group_of(i)][k].append(t[k])
    print("groups (mean ms per EM iteration):")
    for g in ("it0", "it1-3", "it4", "it5+"):
        if g in agg:
            print(f"  {g:6s} solver {sum(agg[g]['solver']) / len(agg[g]['solver']):8.3f}   post {sum(agg[g]['post']) / len(agg[g]['post']):8.3f}   beta_ss {sum(agg[g]['betass']) / len(agg[g]['betass']):8.3f}   ({len(agg[g]['post'])} iterations)")
    for k in ("solver", "post", "betass"):
        v = [t[k] for t in timed]
        print(f"  all    {k} mean {sum(v) / len(v):.3f} ms over the {len(v)} timed iterations")
else:
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # one record per (dispatch, counter): walk the dispatches in order
    by_disp = collections.OrderedDict()
    for r in rows:
        k = kname(r)
        if not k:
            continue
        by_disp.setdefault(int(r["Dispatch_Id"]), (k, {}))[1][r["Counter_Name"]] = by_disp.get(int(r["Dispatch_Id"]), (k, {}))[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    it, agg, nits, seen_post = 0, collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter(), False
    for d, (k, cs) in by_disp.items():
        if k == "solver" and seen_post:
            it, seen_post = it + 1, False
        if k == "post":
            seen_post = True
        if W <= it < W + K:
            g = group_of(it - W)
            for c, v in cs.items():
                agg[(g, k)][c] += v
            if k == "post":
                nits[g] += 1
    for (g, k), cs in sorted(agg.items()):
        print(f"{g} {k}  (summed over {nits[g]} EM iterations; divide for per-iteration figures)")
        for c, v in sorted(cs.items()):
            print(f"   {c:30s} {v:18.0f}   per iteration {v / max(nits[g], 1):16.0f}")
        wc, wa = cs.get("SQ_WAVE_CYCLES"), cs.get("SQ_WAIT_ANY")
        if wc and wa:
            print(f"   -> parked on s_waitcnt {100 * wa / wc:.1f} % of wave-cycles, issuing {100 * cs.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f} %, "
                  f"issue-stalled {100 * cs.get('SQ_WAIT_INST_ANY', 0) / wc:.1f} %")
