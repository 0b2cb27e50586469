#!/usr/bin/env python3
"""Re-run a case saved by tools/fuzz_parity.py (gpurun_out/fuzz_case_<n>.npz) with and without the solver's line-search
shortcuts (STM_DEBUG_FLAGS=6: every evaluation scipy makes is made) and list the documents that differ from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import stm_oracle
from strutopy_amd.engine import estep_host
stm_oracle.build()
g = np.load(sys.argv[1])
asp = g["aspect"] if g["aspect"].size else None
args = (g["indptr"], g["indices"], g["counts"], g["beta"], g["mu"], g["eta"], g["siginv"], float(g["sigent"]))
o = stm_oracle.estep(*args, aspect=asp, nthreads=0)
for flags in ("0", "6"):
    os.environ["STM_DEBUG_FLAGS"] = flags
    d = estep_host(*args, aspect=asp, testing=True)   # (STM_DEBUG_FLAGS: the -DSTM_TESTING build)
    bad = np.nonzero((d["nit"] != o["nit"]) | (d["status"] != o["status"]) | (d["pd_path"] != o["pd_path"]))[0]
    print(f"STM_DEBUG_FLAGS={flags}: {len(bad)} documents differ; nfev mean gpu {d['nfev'].mean():.1f} oracle {o['nfev'].mean():.1f}")
    de = np.max(np.abs(d["eta"] - o["eta"]), axis=1)
    for i in np.argsort(-de)[:3]:
        print(f"   largest |eta diff|: doc {i} {de[i]:.2e}  words {int(g['indptr'][i + 1] - g['indptr'][i])}  N_d {g['counts'][g['indptr'][i]:g['indptr'][i + 1]].sum():.0f}  nit {d['nit'][i]} / {o['nit'][i]}  nfev {d['nfev'][i]} / {o['nfev'][i]}")
    for i in bad[:10]:
        print(f"   doc {i}: nit {d['nit'][i]} / {o['nit'][i]}  status {d['status'][i]} / {o['status'][i]}  nfev {d['nfev'][i]} / {o['nfev'][i]}  "
              f"pd {d['pd_path'][i]} / {o['pd_path'][i]}  |eta diff| {np.max(np.abs(d['eta'][i] - o['eta'][i])):.2e}  (gpu / oracle)")
