#!/usr/bin/env python3
"""What each E-step kernel is bound by on the COMPUTE side, from the SQ counters (one entry of profiles/compute_counters.json).

  compute_collect.py <out.json> <pass A counter_collection.csv> <pass B counter_collection.csv> docs vocab topics words levels W K "<source>"

Pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
Pass B: SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM
of the same `bench.py <args> --steps K --warmup W --cpu-sample 0 --late-sample 0` command (separate rocprofv3 --kernel-trace --pmc runs: the
two sets do not fit one pass).  Dispatches are walked in order and grouped by EM iteration as tools/by_iteration.py does; the entry
holds, per kernel, the MEAN PER EM ITERATION over the K timed iterations of
  * valu_insts, mfma_f64_insts, mfma_busy_cycles, lds_insts, lds_bank_conflict_cycles (wave-level instruction / cycle counts)
  * launch_ms (End - Start of the dispatches in pass B: with counters on, a few % above the untraced time)
  * fp64_pipe_busy = (4 (valu_insts - mfma_f64_insts) + mfma_busy_cycles) / (launch_ms x 2.4 GHz x 1024 SIMDs): a wave64 vector
    instruction holds its SIMD's one vector pipe for 4 cycles, v_mfma_f64_16x16x4_f64 for 64 -- and nothing else issues on that SIMD
    meanwhile (tools/microbench/fp64_pipe.hip; the shader clock under these kernels is 2.39-2.42 GHz: tools/slot_gaps.py)
  * mfma_tflops = mfma_f64_insts x 2048 flop / launch time
  * waves_parked / waves_issuing / waves_issue_stalled = SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (pass A)
tools/compute_merge.py folds entries into profiles/compute_counters.json; bench.py attaches the entry of its workload to roofline.kernels.*"""
import collections, csv, json, sys

out_path, apath, bpath = sys.argv[1:4]
docs, vocab, topics, words, levels, W, K = (int(a) for a in sys.argv[4:11])
source = sys.argv[11] if len(sys.argv) > 11 else ""
CLOCK_GHZ, SIMDS = 2.4, 1024


def kname(n):
    if "solver_kernel" in n:
        return "solver"
    if "post_any_kernel" in n or "post_kernel" in n or "post_big2_kernel" in n:
        return "post"
    if "beta_ss_part" in n or "beta_ss_reduce_kernel" in n:
        return "betass"
    return None


def per_kernel(path):
    rows = list(csv.DictReader(open(path)))
    by = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        k = kname(r["Kernel_Name"])
        if not k:
            continue
        d = by.setdefault(int(r["Dispatch_Id"]), dict(k=k, c=collections.defaultdict(float), ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    it, seen_post, agg, nits = 0, False, collections.defaultdict(lambda: collections.defaultdict(float)), 0
    for d in by.values():
        if d["k"] == "solver" and seen_post:
            it, seen_post = it + 1, False
        if d["k"] == "post":
            seen_post = True
        if W <= it < W + K:
            for c, v in d["c"].items():
                agg[d["k"]][c] += v
            agg[d["k"]]["_ns"] += d["ns"]
            if d["k"] == "post":
                nits += 1
    return {k: {c: v / max(nits, 1) for c, v in cs.items()} for k, cs in agg.items()}, nits


A, na = per_kernel(apath)
B, nb = per_kernel(bpath)
entry = {"_workload": {"docs": docs, "vocab": vocab, "topics": topics, "words": words, "levels": levels},
         "_units": "mean per EM iteration over the timed iterations (all dispatches of the kernel in an iteration)",
         "_iterations": {"pass_a": na, "pass_b": nb, "warmup": W}, "_clock_ghz": CLOCK_GHZ, "_simds": SIMDS, "_source": source, "kernels": {}}
for k in sorted(set(A) | set(B)):
    a, b = A.get(k, {}), B.get(k, {})
    ms = b.get("_ns", a.get("_ns", 0.0)) / 1e6
    valu, mfma, mbusy = b.get("SQ_INSTS_VALU", 0.0), b.get("SQ_INSTS_VALU_MFMA_F64", 0.0), b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    simd_cycles = ms * 1e-3 * CLOCK_GHZ * 1e9 * SIMDS
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    entry["kernels"][k] = {
        "launch_ms": ms, "valu_insts": valu, "mfma_f64_insts": mfma, "mfma_busy_cycles": mbusy, "lds_insts": b.get("SQ_INSTS_LDS", 0.0),
        "lds_bank_conflict_cycles": b.get("SQ_LDS_BANK_CONFLICT", 0.0), "salu_insts": b.get("SQ_INSTS_SALU", 0.0),
        "valu_insts_per_doc": valu / docs, "mfma_f64_insts_per_doc": mfma / docs,
        "fp64_pipe_busy": (4.0 * (valu - mfma) + mbusy) / simd_cycles if simd_cycles else None,
        "mfma_tflops": mfma * 2048.0 / (ms * 1e-3) / 1e12 if ms else None,
        "waves_parked": a.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
        "waves_issuing": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
        "waves_issue_stalled": a.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
    }
json.dump(entry, open(out_path, "w"), indent=1)
for k, v in entry["kernels"].items():
    print(k, {q: (round(x, 4) if isinstance(x, float) else x) for q, x in v.items() if q in ("launch_ms", "fp64_pipe_busy", "mfma_tflops", "waves_parked", "waves_issuing", "valu_insts_per_doc", "mfma_f64_insts_per_doc")})
