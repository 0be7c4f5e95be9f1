#!/usr/bin/env python3
"""One rank's shard of a strong-scaling step (default: rank 0 of 8 of configs[3]) through the pipelined export: device
time of the partition and of every dedupe + pack group, the owner's groups on what this one rank sent (a quick A/B
probe for kernel variants; tools/scale_emul.py is the full rehearsal).  usage: dd_probe.py [world [reads [k [groups]]]]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cfrk_amd
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
G = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L = 150
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = cfrk_amd.Context(0, stream.cuda_stream); octx = cfrk_amd.Context(0, stream.cuda_stream)
Rl = R // world; nN = Rl * (L + 1)
d = torch.empty(nN + 64, dtype=torch.int8, device=dev)
ctx.synth_reads_device(0, Rl, L, R, d.data_ptr()); ctx.sync()
lpp = -(-65536 // world)
seg_cap = int((2.5 * Rl + 0.3 * R) * (2 if k > 32 else 1) / (G * world) * 1.25) + lpp + 4096     # (k > 32: two rows per record)
buf = torch.empty((G, world, seg_cap, 2), dtype=torch.int64, device=dev)
flags = cfrk_amd.CFRK_CANONICAL
best = None
for it in range(4):
    g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, R + 1024)
    g.add_device(d.data_ptr(), nN)
    g.export_runs_async(buf.data_ptr(), seg_cap, world, G)
    rows = [g.export_runs_wait(gi) for gi in range(G)]
    cur = (g.last_add_ms(), [g.runs_group_ms(gi) for gi in range(G)])
    if best is None or cur[1][-1] < best[1][-1]:
        best = cur
# the owner on `world` copies of this rank's segment 0 (same sizes as a real owner's input)
own = None
for it in range(3):
    og = cfrk_amd.GlobalCounter(octx, k, flags, R // world + 1024)
    cum = []
    for gi in range(G):
        seg = buf[gi, 0, :rows[gi][0]]
        recv = torch.cat([seg] * world)
        og.merge_runs_group_device(recv.data_ptr(), [rows[gi][0]] * world, gi, G)
        cum.append(og.last_add_ms())
    cur = [cum[0]] + [cum[i] - cum[i - 1] for i in range(1, G)]
    if own is None or sum(cur) < sum(own):
        own = cur
print(json.dumps({"world": world, "partition_ms": best[0], "dedupe_pack_ms": best[1][-1] - best[0], "group_end_ms": best[1],
                  "owner_group_ms_on_copies_of_one_rank": own, "rows_total": sum(sum(r) for r in rows)}))
