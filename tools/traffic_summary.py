#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, separate passes, KB) -> HBM bytes per E-step per kernel
(summed over the kernel's dispatches of a one-step bench run: the solver is launched once per LDS
occupancy class, everything else once).

MI355X_MICROARCH.md (HBM section): hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; on gfx950 FETCH_SIZE
under-reports coalesced reads by a pattern-dependent factor (exactly 2x for 16-B/lane streams) and has to
be calibrated on a known byte count in the same access pattern.  Calibration kernel: covariance_kernel,
which streams eta exactly once with 8-B/lane coalesced loads (N * (K-1) * 8 bytes per dispatch).

usage: traffic_summary.py fetch.csv write.csv out.json [N K V words]
"""
import collections, csv, json, sys


def per_kernel(path):
    agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = k.split("<")[0]
        agg[k] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: agg[k] for k in agg}, {k: len(cnt[k]) for k in agg}


(fetch, ndisp), (write, _) = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
K = int(sys.argv[5]) if len(sys.argv) > 5 else 50
# since round 2 the resident loop launches covariance_kernel with mu = nullptr (eta^T eta for the packed moments): it streams
# eta once per dispatch
known = 1.0 * N * (K - 1) * 8 * max(ndisp.get("stm::covariance_kernel", 1), 1)
cal = known / (fetch.get("stm::covariance_kernel", 0.0) * 1024) if fetch.get("stm::covariance_kernel") else None
out = {"_workload": {"docs": N, "vocab": int(sys.argv[6]) if len(sys.argv) > 6 else 10000, "topics": K,
                     "words": int(sys.argv[7]) if len(sys.argv) > 7 else 150},
       "_units": "bytes per E-step (all dispatches of the kernel in one EM iteration)", "_fetch_calibration": cal,
       "_note": "FETCH_SIZE*1024*calibration + WRITE_SIZE*1024; calibration = known bytes of covariance_kernel / its FETCH_SIZE",
       "_source": "EM iteration 0 ONLY: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --steps 1 --warmup 0 --cpu-sample 0` (tools/profile_r03.sh)"}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, 0.0) * 1024 * (cal or 1.0)
    w = write.get(k, 0.0) * 1024
    out[k] = f + w
    out[k + "#raw"] = {"FETCH_SIZE_KB": fetch.get(k, 0.0), "WRITE_SIZE_KB": write.get(k, 0.0), "dispatches": ndisp.get(k, 0)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if not k.startswith("_") and not k.endswith("#raw"):
        print(f"{k:45s} {v / 1e6:12.2f} MB/E-step   raw {out[k + '#raw']}")
print("fetch calibration factor:", cal)
