#!/usr/bin/env python3
"""Per-rank critical path of a STRONG-scaling step (BASELINE configs[3]: the same R reads split over
`world` ranks) rehearsed on ONE GPU, wire time excluded.  Every rank's shard is run in turn:
  runs exchange (default): partition + deduplicate (CFRK_RUNS_ONLY add) and packed export; then owner 0
      receives its segment from every rank and counts its leaves (cfrk_global_merge_runs_device);
  --leaf: the counted-list exchange of round 1 for comparison (count, export by leaf, LDS merge).
Prints per phase the slowest rank's time, the bytes a rank puts on the wire, and checks that owner 0's
sum of counts is its share of the job.   usage: scale_emul.py [world [reads [k]]] [--leaf] [--L=n] [--glen=n] [--weak]
--weak: `reads` is what EVERY rank holds (the job has world x reads reads: BASELINE configs[4] is 8 x 125 M x 250 bp).
--link-gbs=G: WIRE MODEL (no multi-GPU box exists to measure one): every rank sends 1/(world-1) of its bytes to each peer over
that peer's own xGMI link (point-to-point topology: up to 7 links per GPU), G GB/s per link and direction (default 120 =
0.8 x the 153 GB/s the links are specified at), plus --wire-lat-us (default 30: two collectives' launch + one host read of
the sizes).  `step_ms_model` = critical path + modelled wire, nothing overlapped (what the code does today)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import cfrk_amd  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
leaf_mode = "--leaf" in sys.argv
weak = "--weak" in sys.argv
world = int(pos[0]) if len(pos) > 0 else 8
R = int(pos[1]) if len(pos) > 1 else 100_000_000
k = int(pos[2]) if len(pos) > 2 else 31
L = int(opt.get("L", 150))
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = cfrk_amd.Context(0, stream.cuda_stream)
octx = cfrk_amd.Context(0, stream.cuda_stream)
flags = cfrk_amd.CFRK_CANONICAL
G = int(opt.get("glen", R))
hint = G + 1024
Rl = R if weak else R // world
nN = Rl * (L + 1)
d = torch.empty(nN + 64, dtype=torch.int8, device=dev)

t_count, t_export, wire = [], [], []
segs, rows0 = [], []
buf = None if leaf_mode else torch.empty((max(1 << 20, int(min(14 * Rl, 2.5 * Rl + 0.3 * G)) * (2 if k > 32 else 1) + (1 << 17)), 2), dtype=torch.int64, device=dev)   # the send buffer lives across steps
keys = cnt = lc = None
for r in range(world):
    ctx.synth_reads_device(r * Rl, Rl, L, G, d.data_ptr())
    ctx.sync()
    best = None
    for it in range(2):                                   # second run: pools and code objects are warm
        g = cfrk_amd.GlobalCounter(ctx, k, flags | (0 if leaf_mode else cfrk_amd.CFRK_RUNS_ONLY), hint)
        t0 = time.perf_counter()
        g.add_device(d.data_ptr(), nN)
        ctx.sync()
        t1 = time.perf_counter()
        if leaf_mode:
            lpp = g.leaves_per_part(world)
            if keys is None:
                keys = torch.empty(hint, dtype=torch.int64, device=dev)
                khi = torch.empty(hint, dtype=torch.int64, device=dev) if k > 32 else None
                cnt = torch.empty(hint, dtype=torch.int32, device=dev)
                lc = torch.empty(world * lpp, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            pc = g.export_leaves_device(keys.data_ptr(), cnt.data_ptr(), hint, world, lc.data_ptr(), khi.data_ptr() if k > 32 else 0)
            ctx.sync()
            t2 = time.perf_counter()
            out = (keys[:pc[0]].clone(), cnt[:pc[0]].clone(), lc[:lpp].clone(), pc[0], khi[:pc[0]].clone() if k > 32 else None)
            w = (20 if k > 32 else 12) * (sum(pc) - pc[r]) + 4 * lpp * (world - 1)
        else:
            pr = g.export_runs_device(buf.data_ptr(), buf.shape[0], world)
            ctx.sync()
            t2 = time.perf_counter()
            out = (buf[:pr[0]].clone(), pr[0])
            w = 16 * (sum(pr) - pr[r])
        best = (g.last_add_ms(), (t1 - t0) * 1e3, (t2 - t1) * 1e3, w, out)
    t_count.append(best[:2]); t_export.append(best[2]); wire.append(best[3]); segs.append(best[4])
    del g

og = None
t_owner = None
for it in range(2):
    og = cfrk_amd.GlobalCounter(octx, k, flags, hint // world + 1024)
    if leaf_mode:
        rkeys = torch.cat([s[0] for s in segs]); rcnt = torch.cat([s[1] for s in segs]); rlc = torch.cat([s[2] for s in segs])
        rhi = torch.cat([s[4] for s in segs]) if k > 32 else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        og.merge_leaves_device(rkeys.data_ptr(), rcnt.data_ptr(), [int(s[3]) for s in segs], rlc.data_ptr(), rhi.data_ptr() if k > 32 else 0)
    else:
        packed = torch.cat([s[0] for s in segs])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        og.merge_runs_device(packed.data_ptr(), [int(s[1]) for s in segs])
    octx.sync()
    t_owner = ((time.perf_counter() - t0) * 1e3, og.last_add_ms() if not leaf_mode else None)
dg = og.digest()
res = {
    "what": "%s-scaling step rehearsed on one GPU (%s exchange), wire time excluded" % ("weak" if weak else "strong", "leaf" if leaf_mode else "runs"),
    "world": world, "reads": R * world if weak else R, "read_len": L, "genome": G, "k": k,
    "rank_count_kernels_ms_max": max(t[0] for t in t_count), "rank_count_wall_ms_max": max(t[1] for t in t_count),
    "rank_export_ms_max": max(t_export), "owner_merge_wall_ms": t_owner[0], "owner_kernels_ms": t_owner[1],
    "critical_path_ms": max(t[1] for t in t_count) + max(t_export) + t_owner[0],
    "wire_bytes_per_rank_max": max(wire), "owner0_distinct": dg[0], "owner0_sum": dg[1],
}
link = float(opt.get("link-gbs", 120.0))
lat = float(opt.get("wire-lat-us", 30.0))
if world > 1:
    wire_ms = max(wire) / (world - 1) / (link * 1e9) * 1e3 + lat * 1e-3
    res["wire_model"] = {"link_GBps": link, "links_used": world - 1, "latency_us": lat, "wire_ms": wire_ms,
                         "assumption": "point-to-point xGMI, one link per peer, all links in parallel; NOT measured"}
    res["step_ms_model"] = res["critical_path_ms"] + wire_ms
print(json.dumps(res), flush=True)
