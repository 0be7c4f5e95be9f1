#!/usr/bin/env python3
"""Per-read dense path (the kmer_main drop-in) on device-resident chunks: time and algorithmic GB/s.
Algorithmic bytes per call = nN + 12*nS + 4*nS*4^k (codes + start/length tables + the row matrix written once)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cfrk_amd

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()          # a real stream: handle 0 (the default stream) would make the
torch.cuda.set_stream(stream)         # context create its own, invisible to torch events
ctx = cfrk_amd.Context(0, stream.cuda_stream)
L = 150
out = []
for nS, k in [(8192, 2), (8192, 4), (8192, 6), (8192, 8), (1_000_000, 2), (1_000_000, 4), (1_000_000, 6)]:
    nN = nS * (L + 1)
    d = torch.empty(nN + 64, dtype=torch.int8, device=dev)
    st = torch.empty(nS, dtype=torch.int64, device=dev)
    ln = torch.empty(nS, dtype=torch.int32, device=dev)
    fr = torch.empty(nS * 4 ** k, dtype=torch.int32, device=dev)
    ctx.synth_reads_device(0, nS, L, max(nS, 1000), d.data_ptr(), st.data_ptr(), ln.data_ptr())
    f = ctx._L.cfrk_per_read_dense_device
    args = (ctx._h, C.c_void_p(d.data_ptr()), C.c_void_p(st.data_ptr()), C.c_void_p(ln.data_ptr()), nN, nS, k,
            cfrk_amd.CFRK_COMPAT, C.c_void_p(fr.data_ptr()))
    for _ in range(3):
        ctx.check(f(*args), "dense")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ctx.check(f(*args), "dense")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b_alg = nN + 12 * nS + 4 * nS * 4 ** k
    out.append({"reads": nS, "k": k, "ms": round(ms, 4), "windows_per_s": nS * (L - 1) / (ms * 1e-3),
                "alg_GBps": b_alg / (ms * 1e-3) / 1e9, "frac_8TBps": b_alg / (ms * 1e-3) / 8e12})
    print(json.dumps(out[-1]), flush=True)
