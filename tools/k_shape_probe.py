#!/usr/bin/env python3
"""One add of R synthetic reads (length L, genome G) at several k with the same capacity hint: kernel
time of the add, passes, and what spilled (cfrk_debug_msp_info).  usage: k_shape_probe.py R L G hint k [k ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cfrk_amd  # noqa: E402

R, L, G, hint = (int(x) for x in sys.argv[1:5])
ctx = cfrk_amd.Context(0)
nN = R * (L + 1)
d = ctx.alloc(nN + 64)
ctx.synth_reads_device(0, R, L, G, d)
ctx.sync()
for k in (int(x) for x in sys.argv[5:]):
    for it in range(2):
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, hint)
        g.add_device(d, nN)
        ctx.sync()
        ms = g.last_add_ms()
        info = g.msp_info()
        dg = g.digest()
        passes = g.last_add_passes()
        del g
    print(f"k={k} add {ms:.2f} ms passes={passes} distinct={dg[0]} sum={dg[1]} info={info}", flush=True)
