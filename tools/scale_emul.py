#!/usr/bin/env python3
"""Per-rank critical path of a STRONG-scaling step (BASELINE configs[3]: the same R reads split over `world` ranks)
REHEARSED ON ONE GPU -- no multi-GPU box has been available to any round, so the wire is MODELLED, never measured.

Every rank's shard is run in turn through the code a rank really runs; owner 0 then counts what the ranks sent it.

  pipelined (default, one-word keys): begin(CFRK_RUNS_ONLY | CFRK_RUNS_DEFER) -> add_device -> export_runs_async ->
      export_runs_wait per group; owner: merge_runs_group_device per group (cfrk_amd/sharded.py:
      exchange_by_runs_pipelined).  Device times are HIP events inside the library (start of the add -> end of every
      group; the owner's kernels per group); the step is then SCHEDULED the way the code overlaps it:
          compute stream:  [P1 P2][dedupe+pack g0][dedupe+pack g1]...[owner g0][owner g1]...   (all enqueued up front)
          comm stream:     per group: host wakes on the group's event (host_sync) -> size all-to-all (coll_lat) -> host
                           reads the sizes (host_sync) -> payload all-to-all (coll_lat + bytes / link)
          owner g starts when its payload has arrived AND the compute stream is free (after the last dedupe group / owner g-1)
      charged: --coll-lat-us per collective (default 60, two collectives per group), the MEASURED host wake-up latency
      per host wait (two per group), the measured host time of the two cfrk_global_begin calls at the start of a step.
  --classic: the one-shot runs exchange of rounds 2-4 (dedupe in place, sizes + plan + gather, ONE all-to-all, layout +
      scatter + leaf kernel), nothing overlapped: rank + export + wire + owner.
  --leaf: the counted-list exchange of round 1 for comparison.
usage: scale_emul.py [world [reads [k]]] [--groups=G] [--classic] [--leaf] [--L=n] [--glen=n] [--weak]
                     [--link-gbs=120] [--coll-lat-us=60]
--weak: `reads` is what EVERY rank holds.   --link-gbs: one xGMI link per peer and direction (point-to-point topology: up
to 7 links per GPU); 120 = 0.8 x the 153 GB/s the links are specified at."""
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
classic = "--classic" in sys.argv or leaf_mode
weak = "--weak" in sys.argv
world = int(pos[0]) if len(pos) > 0 else 8
R = int(pos[1]) if len(pos) > 1 else 100_000_000
k = int(pos[2]) if len(pos) > 2 else 31
L = int(opt.get("L", 150))
G_GROUPS = int(opt.get("groups", 2))
link = float(opt.get("link-gbs", 120.0))
coll_us = float(opt.get("coll-lat-us", 60.0))
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
base = {"world": world, "reads": R * world if weak else R, "read_len": L, "genome": G, "k": k,
        "scaling": "weak" if weak else "strong"}


def host_sync_latency_us():
    """what a host wait costs beyond the work it waits for: enqueue a trivial kernel + event, wait, read the clock"""
    x = torch.zeros(64, device=dev)
    best = 1e9
    for _ in range(50):
        torch.cuda.synchronize()
        e = torch.cuda.Event()
        t0 = time.perf_counter()
        x.add_(1)
        e.record(stream)
        e.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6)
    return best


if not classic:
    lpp = -(-65536 // world)
    rows_est = int(min(14 * Rl, 2.5 * Rl + 0.3 * G)) * (2 if k > 32 else 1) + (1 << 17)   # (k > 32: two rows per record)
    seg_cap = int(rows_est / (G_GROUPS * world) * 1.25) + (lpp + G_GROUPS - 1) // G_GROUPS + 4096
    buf = torch.empty((G_GROUPS, world, seg_cap, 2), dtype=torch.int64, device=dev)
    sync_us = host_sync_latency_us()
    ranks = []                      # per rank: dict(add_ms, group_end_ms[], rows[g][p], begin_add_host_ms)
    segs = [[None] * world for _ in range(G_GROUPS)]
    for r in range(world):
        ctx.synth_reads_device(r * Rl, Rl, L, G, d.data_ptr())
        ctx.sync()
        best = None
        for it in range(3):                                   # later runs: pools and code objects are warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, hint)
            t1 = time.perf_counter()
            g.add_device(d.data_ptr(), nN)
            g.export_runs_async(buf.data_ptr(), seg_cap, world, G_GROUPS)
            t2 = time.perf_counter()                          # everything of the rank is enqueued
            rows = [g.export_runs_wait(gi) for gi in range(G_GROUPS)]
            t3 = time.perf_counter()
            cur = {"begin_host_ms": (t1 - t0) * 1e3, "enqueue_host_ms": (t2 - t1) * 1e3, "wall_ms": (t3 - t0) * 1e3,
                   "partition_ms": g.last_add_ms(), "group_end_ms": [g.runs_group_ms(gi) for gi in range(G_GROUPS)], "rows": rows}
            if best is None or cur["group_end_ms"][-1] < best["group_end_ms"][-1]:
                best = cur
            del g
        # (the buffer holds the LAST run's segments: its row counts go with them -- a leaf whose record table is nearly
        #  full deduplicates or not depending on the order its records arrive in, so runs may differ by a few rows)
        best = dict(best, rows=rows)
        ranks.append(best)
        for gi in range(G_GROUPS):
            segs[gi][r] = buf[gi, 0, :rows[gi][0]].clone()      # what owner 0 receives from this rank
    # owner 0 (the ranks' buffers and pools are given back first: a configs[4]-sized shard fills most of the device)
    del buf, d
    ctx.close()
    torch.cuda.empty_cache()
    recvs = [torch.cat(segs[gi]) for gi in range(G_GROUPS)]
    recv_rows = [[int(s.shape[0]) for s in segs[gi]] for gi in range(G_GROUPS)]
    own = None
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        og = cfrk_amd.GlobalCounter(octx, k, flags, hint // world + 1024)
        t1 = time.perf_counter()
        cum = []
        for gi in range(G_GROUPS):
            og.merge_runs_group_device(recvs[gi].data_ptr(), recv_rows[gi], gi, G_GROUPS)
            cum.append(og.last_add_ms())                      # (synchronises: this is the measurement, not the pipeline)
        cur = {"begin_host_ms": (t1 - t0) * 1e3, "group_ms": [cum[0]] + [cum[i] - cum[i - 1] for i in range(1, G_GROUPS)]}
        if own is None or sum(cur["group_ms"]) < sum(own["group_ms"]):
            own = cur
    try:
        dg = og.digest()
    except cfrk_amd.CfrkError as e:
        print("owner digest failed:", e, og.msp_info(), "recv rows", recv_rows, file=sys.stderr, flush=True)
        raise

    def schedule(me, rk):
        """step time of one rank (ms) under the overlap the code implements; returns (step, detail)"""
        t = rk["begin_host_ms"] + own["begin_host_ms"]        # both jobs begin before anything is enqueued
        dd_end = [t + x for x in rk["group_end_ms"]]
        comm_free, own_end, detail = 0.0, dd_end[-1], []
        for gi in range(G_GROUPS):
            send_max = max(rk["rows"][gi][p] for p in range(world) if p != me) if world > 1 else 0   # (a link per peer: the largest segment decides)
            wire_ms = send_max * 16 / (link * 1e9) * 1e3
            sizes_known = dd_end[gi] + sync_us * 1e-3
            sz_done = max(sizes_known, comm_free) + (coll_us + sync_us) * 1e-3
            pay_done = sz_done + coll_us * 1e-3 + wire_ms
            comm_free = pay_done
            start = max(pay_done + 0.010, own_end)
            own_end = start + own["group_ms"][gi]
            detail.append({"dedupe_pack_end_ms": dd_end[gi], "payload_arrived_ms": pay_done, "wire_ms": wire_ms,
                           "owner_start_ms": start, "owner_end_ms": own_end, "owner_waited_for_wire_ms": max(0.0, pay_done + 0.010 - (dd_end[-1] if gi == 0 else detail[gi - 1]["owner_end_ms"]))})
        return own_end, detail
    steps = [schedule(i, rk) for i, rk in enumerate(ranks)]
    worst = max(range(world), key=lambda i: steps[i][0])
    rk = ranks[worst]
    res = dict(base)
    res.update({
        "what": "strong-scaling step REHEARSED on one GPU: pipelined runs exchange in %d groups; device times measured, wire MODELLED" % G_GROUPS,
        "groups": G_GROUPS,
        "rank_partition_ms_max": max(x["partition_ms"] for x in ranks),
        "rank_dedupe_pack_ms_max": max(x["group_end_ms"][-1] - x["partition_ms"] for x in ranks),
        "rank_gpu_ms_max": max(x["group_end_ms"][-1] for x in ranks),
        "rank_enqueue_host_ms_max": max(x["enqueue_host_ms"] for x in ranks),
        "owner_group_ms": own["group_ms"], "owner_gpu_ms": sum(own["group_ms"]),
        "begin_host_ms": rk["begin_host_ms"] + own["begin_host_ms"],
        "host_sync_latency_us_measured": sync_us,
        "gpu_work_per_rank_ms": max(x["group_end_ms"][-1] for x in ranks) + sum(own["group_ms"]),
        "wire_bytes_per_rank_max": max(16 * sum(sum(x["rows"][gi][p] for p in range(world) if p != i) for gi in range(G_GROUPS)) for i, x in enumerate(ranks)),
        "owner0_distinct": dg[0], "owner0_sum": dg[1],
        "wire_model": {"link_GBps": link, "links_used": world - 1, "collective_latency_us": coll_us, "collectives_per_group": 2,
                       "host_waits_per_group": 2,
                       "assumption": "point-to-point xGMI, one link per peer, all links in parallel; NOT measured"},
        "schedule_of_the_slowest_rank": steps[worst][1],
        "step_ms_model": steps[worst][0],
    })
    print(json.dumps(res), flush=True)
    sys.exit(0)

# ---------------------------------------------------------------------------------- one-shot forms (rounds 1-4)
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
res = dict(base)
res.update({
    "what": "%s-scaling step rehearsed on one GPU (one-shot %s exchange), nothing overlapped; wire MODELLED" % ("weak" if weak else "strong", "leaf" if leaf_mode else "runs"),
    "rank_count_kernels_ms_max": max(t[0] for t in t_count), "rank_count_wall_ms_max": max(t[1] for t in t_count),
    "rank_export_ms_max": max(t_export), "owner_merge_wall_ms": t_owner[0], "owner_kernels_ms": t_owner[1],
    "critical_path_ms": max(t[1] for t in t_count) + max(t_export) + t_owner[0],
    "wire_bytes_per_rank_max": max(wire), "owner0_distinct": dg[0], "owner0_sum": dg[1],
})
if world > 1:
    # two collectives (sizes, payload) and the host read of the sizes between them (`recv.cpu().tolist()`)
    sync_us = host_sync_latency_us()
    wire_ms = max(wire) / (world - 1) / (link * 1e9) * 1e3 + (2 * coll_us + sync_us) * 1e-3
    res["wire_model"] = {"link_GBps": link, "links_used": world - 1, "collective_latency_us": coll_us, "collectives": 2,
                         "host_sync_latency_us_measured": sync_us, "wire_ms": wire_ms,
                         "assumption": "point-to-point xGMI, one link per peer, all links in parallel; NOT measured"}
    res["step_ms_model"] = res["critical_path_ms"] + wire_ms
print(json.dumps(res), flush=True)
