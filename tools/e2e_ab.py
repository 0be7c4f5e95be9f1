#!/usr/bin/env python3
"""Same-box A/B of two builds of the `cfrk` CLI, end to end: a configs[1]-sized FASTA (10 M x 150 bp, 1.63 GB) through
tools/_bin/cfrk_prev (old; build it from the commit to compare with) and cfrk_amd/cfrk (new), alternating, four rounds:
process wall clock and the phases `--timing` reports.  usage (GPU box, repo root): python tools/e2e_ab.py"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import bench_e2e
fa = "/tmp/e2e_ab.fasta"
bench_e2e.write_fasta(fa, 10_000_000, 150, 10_000_000)
env = dict(os.environ); env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "cfrk_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
for rnd in range(4):
    for name, exe in (("old", os.path.join(ROOT, "tools/_bin/cfrk_prev")), ("new", os.path.join(ROOT, "cfrk_amd/cfrk"))):
        t0 = time.perf_counter()
        r = subprocess.run([exe, fa, "/tmp/e2e_ab.cfrk", "15", "64", "--global", "--canonical", "--timing"], capture_output=True, text=True, env=env)
        w = time.perf_counter() - t0
        t = [json.loads(l[len("cfrk-timing "):]) for l in r.stderr.splitlines() if l.startswith("cfrk-timing ")][0]
        brief = {k2: v for k2, v in t.items() if k2.endswith("_s")}
        print(f"round {rnd} {name}: process wall {w:.3f} s  {brief}", flush=True)
os.remove(fa)
