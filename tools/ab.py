#!/usr/bin/env python3
"""A/B runs of bench.py under different environment settings: tools/ab.py "<bench args>" "VAR=v VAR2=w" "VAR=x" ...
prints mean solver / post (+ beta_ss pass) kernel ms and ms per EM iteration of every setting ("-" = no variables)."""
import json, os, subprocess, sys
args = sys.argv[1].split()
for setting in sys.argv[2:]:
    env = dict(os.environ)
    if setting != "-":
        for kv in setting.split():
            k, v = kv.split("=", 1)
            env[k] = v
    out = subprocess.run([sys.executable, "bench.py", "--cpu-sample", "0", "--late-sample", "0", *args], env=env, capture_output=True, text=True)
    try:
        d = json.loads([l for l in out.stdout.split("\n") if l.startswith('{"metric')][-1])
        ps = d["per_step"]
        print(f"{setting:50s} ms/step {d['ms_per_step']:8.3f}  solver {sum(p['solver_kernel_ms'] for p in ps) / len(ps):7.3f}  post {sum(p['post_kernel_ms'] for p in ps) / len(ps):7.3f}  docs/s {d['value']:.4g}", flush=True)
    except Exception as e:
        print(setting, "FAILED", e, out.stderr[-400:], flush=True)
