#!/usr/bin/env python3
"""Time the two library calls around the multi-GPU exchange on ONE GPU: export by leaf for `world`
owners, and the owner's merge of `world` received lists (here: the same list `world` times, as in a
weak-scaling run where every rank counts reads of the same genome).  usage: merge_emul.py [world [reads [k]]]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import cfrk_amd  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
L = 150
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
ctx = cfrk_amd.Context(0, stream.cuda_stream)
octx = cfrk_amd.Context(0, stream.cuda_stream)
d = torch.empty(R * (L + 1), dtype=torch.int8, device=dev)
ctx.synth_reads_device(0, R, L, R, d.data_ptr())
hint = R + 1024
flags = cfrk_amd.CFRK_CANONICAL
for it in range(3):
    g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
    g.add_device(d.data_ptr(), R * (L + 1))
    ctx.sync()
    lpp = g.leaves_per_part(world)
    keys = torch.empty(hint, dtype=torch.int64, device=dev)
    hi = torch.empty(hint, dtype=torch.int64, device=dev) if k > 32 else None
    cnt = torch.empty(hint, dtype=torch.int32, device=dev)
    lc = torch.empty(world * lpp, dtype=torch.int32, device=dev)
    t0 = time.perf_counter()
    pc = g.export_leaves_device(keys.data_ptr(), cnt.data_ptr(), hint, world, lc.data_ptr(), hi.data_ptr() if hi is not None else 0)
    ctx.sync()
    t1 = time.perf_counter()
    n0 = int(pc[0])
    rkeys = keys[:n0].repeat(world)
    rhi = hi[:n0].repeat(world) if hi is not None else None
    rcnt = cnt[:n0].repeat(world)
    rlc = lc[:lpp].repeat(world)
    torch.cuda.synchronize()
    og = cfrk_amd.GlobalCounter(octx, k, flags, hint // world + 1024)
    t2 = time.perf_counter()
    og.merge_leaves_device(rkeys.data_ptr(), rcnt.data_ptr(), [n0] * world, rlc.data_ptr(), rhi.data_ptr() if rhi is not None else 0)
    octx.sync()
    t3 = time.perf_counter()
    dg = og.digest()
    print("world=%d k=%d: export %.2f ms (%d entries, part 0: %d), owner merge of %d x %d entries %.2f ms, distinct at owner %d, sum %d"
          % (world, k, (t1 - t0) * 1e3, sum(int(x) for x in pc), n0, world, n0, (t3 - t2) * 1e3, dg[0], dg[1]), flush=True)
