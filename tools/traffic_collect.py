#!/usr/bin/env python3
"""HBM traffic of every E-step kernel PER EM ITERATION of a bench.py run, into profiles/hbm_traffic.json's format.

  traffic_collect.py <out.json> <fetch counter_collection.csv> <write counter_collection.csv> docs vocab topics words levels "<source text>"

The two CSVs are separate rocprofv3 passes (`--kernel-trace --pmc FETCH_SIZE`, `--kernel-trace --pmc WRITE_SIZE`: TCC counters, they do
not fit one pass) of the SAME `bench.py --config X --steps S --warmup 0 --cpu-sample 0 --late-sample 0` command.  Dispatches are
walked in order; an EM iteration starts with the first solver dispatch behind a post-kernel dispatch (tools/by_iteration.py).
MI355X_MICROARCH.md (HBM section): bytes = (FETCH_SIZE + WRITE_SIZE) x 1024, and on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of a wide coalesced streaming read (16 B/lane, global_load and LDS-DMA alike -- what the solver's and the post kernels' beta row
fetches are): FETCH_SIZE is doubled, as the guide prescribes.  Cross-check stored with the entry: covariance_kernel streams eta
exactly once per dispatch when K - 1 <= 64 (N (K-1) 8 bytes), which gives 1.989 on configs[1] and [4]; beyond 64 columns its
blocks re-read column slices through the L2 and the check does not apply.
The output holds ONE entry (this workload); tools/traffic_merge.py folds entries into profiles/hbm_traffic.json."""
import collections, csv, json, sys

out_path, fpath, wpath = sys.argv[1:4]
docs, vocab, topics, words, levels = (int(a) for a in sys.argv[4:9])
source = sys.argv[9] if len(sys.argv) > 9 else ""


def short(name):
    k = name.split("(")[0].replace("void ", "")
    return k.split("<")[0]


def per_iteration(path):
    rows = list(csv.DictReader(open(path)))
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        d = int(r["Dispatch_Id"])
        k = short(r["Kernel_Name"])
        by.setdefault(d, [k, 0.0])[1] += float(r["Counter_Value"])
    its, cur, seen_post = [], collections.defaultdict(float), False
    ndisp = collections.Counter()
    for d, (k, v) in by.items():
        if "solver_kernel" in k and seen_post:
            its.append(cur); cur, seen_post = collections.defaultdict(float), False
        if "post_kernel" in k or "post_any_kernel" in k or "post_big2_kernel" in k:
            seen_post = True
        cur[k] += v
        ndisp[k] += 1
    its.append(cur)
    return its, ndisp


fetch, ndisp = per_iteration(fpath)
write, _ = per_iteration(wpath)
n_it = min(len(fetch), len(write))
cov_kb = sum(it.get("stm::covariance_kernel", 0.0) for it in fetch[:n_it])
cov_n = ndisp.get("stm::covariance_kernel", 0)
check = (1.0 * docs * (topics - 1) * 8 * cov_n) / (cov_kb * 1024) if (cov_kb and topics - 1 <= 64) else None
cal = 2.0
entry = {"_workload": {"docs": docs, "vocab": vocab, "topics": topics, "words": words, "levels": levels},
         "_units": "bytes per EM iteration (all dispatches of the kernel in that iteration)", "_fetch_calibration": cal,
         "_fetch_check_covariance_kernel": check,
         "_note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE = half the bytes of 16-B/lane streams); check = known bytes of covariance_kernel / its FETCH_SIZE bytes",
         "_source": source, "iterations": {}}
for i in range(n_it):
    ks = sorted(set(fetch[i]) | set(write[i]))
    entry["iterations"][str(i)] = {k: fetch[i].get(k, 0.0) * 1024 * (cal or 1.0) + write[i].get(k, 0.0) * 1024 for k in ks if k.startswith("stm::")}
    entry["iterations"][str(i)]["#raw_kb"] = {k: [fetch[i].get(k, 0.0), write[i].get(k, 0.0)] for k in ks if k.startswith("stm::")}
json.dump(entry, open(out_path, "w"), indent=1)
print(f"{n_it} EM iterations; fetch factor {cal} (covariance_kernel check: {check})")
for i in sorted({0, 1, min(7, n_it - 1), n_it - 1}):
    it = entry["iterations"][str(i)]
    print(f"  EM it {i}: " + ", ".join(f"{k.replace('stm::', '')} {v / 1e9:.3f} GB" for k, v in it.items() if not k.startswith("#") and v > 5e7))
