#!/usr/bin/env python3
"""Regression sweep on the GPU box: for every k in a range, both strand modes and three genome
regimes, the digest of the partitioned path must equal the digest of the general HBM-table path
(CFRK_FORCE_HASH) on the same device-resident synthetic reads.
usage: ksweep_digest.py [kmin kmax [reads]]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cfrk_amd  # noqa: E402

kmin = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
L = 150
ctx = cfrk_amd.Context(0)
d = ctx.alloc(R * (L + 1))
bad = 0
for glen in (20_000, 3_000_000, 2_000_000_000):
    ctx.synth_reads_device(0, R, L, glen, d)
    t0 = time.time()
    for k in range(kmin, kmax + 1):
        for flags in (0, cfrk_amd.CFRK_CANONICAL):
            dg = []
            for fh in (0, cfrk_amd.CFRK_FORCE_HASH):
                g = cfrk_amd.GlobalCounter(ctx, k, flags | fh, R * (L - k + 1) if glen > 1e9 else 0)
                g.add_device(d, R * (L + 1))
                dg.append(g.digest())
            if dg[0] != dg[1]:
                bad += 1
                print("MISMATCH glen=%d k=%d flags=%d %s %s" % (glen, k, flags, dg[0], dg[1]), flush=True)
    print("glen=%d: k=%d..%d done in %.1f s, mismatches so far %d" % (glen, kmin, kmax, time.time() - t0, bad), flush=True)
ctx.free(d)
sys.exit(1 if bad else 0)
