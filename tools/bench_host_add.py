#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point: cfrk_global_add on reads that live in
pageable host memory (layout check, staging through the pinned double buffer, H2D, counting) + finish.
usage (GPU box): tools/bench_host_add.py [reads [k]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cfrk_amd  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
L = 150
ctx = cfrk_amd.Context(0)
nN = R * (L + 1)
d = ctx.alloc(nN + 64)
ctx.synth_reads_device(0, R, L, R, d)
ctx.sync()
data = np.empty(nN, np.int8)
ctx.d2h(data, d)
ctx.free(d)
start = np.arange(R, dtype=np.int64) * (L + 1)
length = np.full(R, L, np.int32)
for tables in (False, True):
    best = None
    for it in range(4):
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, R + 1024)
        t0 = time.perf_counter()
        if tables:
            g.add(data, start, length)
        else:
            g.add(data)
        n = g.finish()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        del g
    print(f"{R} reads x {L} bp, k={k}, {'with' if tables else 'without'} start/length tables: "
          f"{best * 1e3:.1f} ms per add + finish = {nN / best / 1e9:.1f} GB/s, {R * (L - k + 1) / best / 1e9:.1f} G k-mers/s, {n} distinct", flush=True)

# the kmer_main drop-in with host buffers (pageable, touched before): H2D of the chunk, kernel, D2H of the rows
import ctypes as C  # noqa: E402
from cfrk_amd.lib import _ptr  # noqa: E402
for nS, kk in ((8192, 8), (1_000_000, 4), (200_000, 6)):
    sub = data[:nS * (L + 1)]
    st = start[:nS]
    ln = length[:nS]
    freq = np.zeros(nS * 4 ** kk, np.int32)
    best = None
    for it in range(4):
        t0 = time.perf_counter()
        ctx.check(ctx._L.cfrk_per_read_dense(ctx._h, _ptr(sub), _ptr(st), _ptr(ln), len(sub), nS, kk,
                                             cfrk_amd.CFRK_COMPAT, _ptr(freq)), "cfrk_per_read_dense")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    nbytes = len(sub) + 12 * nS + freq.nbytes
    print(f"per-read dense, {nS} reads, k={kk}: {best * 1e3:.1f} ms per call, {nbytes / best / 1e9:.1f} GB/s over PCIe "
          f"({freq.nbytes / 1e9:.2f} GB of rows)", flush=True)
