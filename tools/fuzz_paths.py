#!/usr/bin/env python3
"""Randomised cross-check of the counting paths on the GPU, at sizes the CPU oracle would take
minutes for: for random (k, read shape, genome size, strand flag, capacity hint, memory budget)
the partitioned / radix path must give the digest of the general HBM-table path (whose agreement
with the oracle the parity tests pin key by key) and sum(count) must equal the number of valid
k-mers; for 16 <= k <= 64 the runs exchange over 2..5 emulated ranks must give the same digest too, and so must its
pipelined form (round 5: deferred add, 1..8 groups, 2..8 ranks, both key widths).
usage (GPU box): tools/fuzz_paths.py [seconds [seed]]      prints one line per case, exits 1 on a mismatch"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cfrk_amd  # noqa: E402

budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = cfrk_amd.Context(0)
octx = cfrk_amd.Context(0)
M64 = (1 << 64) - 1


def merge(ds):
    d = s = w = x = 0
    for a in ds:
        d = (d + a[0]) & M64; s = (s + a[1]) & M64; w = (w + a[2]) & M64; x ^= a[3]
    return (d, s, w, x)


t_end = time.time() + budget_s
case = 0
bad = 0
while time.time() < t_end:
    case += 1
    k = int(rng.choice([int(rng.integers(1, 65)), 15, 16, 31, 32, 33, 43, 44, 63, 64]))
    L = int(rng.choice([40, 75, 100, 150, 250, 300]))
    if L < k:
        L = k + int(rng.integers(0, 60))
    R = int(10 ** rng.uniform(4.5, 6.5))
    uniform = bool(rng.random() < 0.15)
    G = 0 if uniform else int(10 ** rng.uniform(2.5, 7.5))
    canon = bool(rng.random() < 0.7)
    flags = cfrk_amd.CFRK_CANONICAL if canon else 0
    nk = R * (L - k + 1)
    distinct_guess = nk if uniform else min(2 * G, nk)
    hint = int(max(1024, distinct_guess * float(rng.choice([1.1, 1.1, 3.0, 40.0]))))
    hint = min(hint, 600_000_000)
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    seeds = dict(seedG=int(rng.integers(1, 1 << 30)), seedR=int(rng.integers(1, 1 << 30)), seedS=int(rng.integers(1, 1 << 30)))
    ctx.synth_reads_device(0, R, L, G, d, uniform=uniform, **seeds)
    ctx.sync()
    tag = f"case {case}: k={k} L={L} R={R} G={'uniform' if uniform else G} canon={int(canon)} hint={hint}"
    try:
        gh = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_FORCE_HASH, hint)
        gh.add_device(d, nN)
        want = gh.digest()
        del gh
        g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
        shared = k >= 16 and rng.random() < 0.4      # leaves shared by record whatever the hint (msp.hip / msp2.hip)
        chunked = k >= 16 and rng.random() < 0.4      # counted in chunks, leaf streams sized from the first one (round 3)
        dbg = (cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS if shared else 0) | (cfrk_amd.lib.CFRK_DEBUG_SMALL_PIPELINE if chunked else 0)
        if dbg:
            g.set_debug_flags(dbg)
        forced = rng.random() < 0.25 and k >= 16
        if forced:                                   # several leaf-subset passes
            g.set_mem_budget(int(nN * float(rng.choice([4, 8, 14]))) + (64 << 20))
        g.add_device(d, nN)
        got = g.digest()
        passes = g.last_add_passes()
        if forced:
            g.set_mem_budget(0)
        if dbg:
            g.set_debug_flags(0)
        del g
        ok = got == want and got[1] == nk
        line = f"{tag} passes={passes}{' shared' if shared else ''}{' chunked' if chunked else ''} distinct={got[0]} {'ok' if ok else 'MISMATCH ' + str(got) + ' != ' + str(want)}"
        if ok and 16 <= k <= 64 and rng.random() < 0.5:
            world = int(rng.integers(2, 6))
            Rl = [R * r // world for r in range(world + 1)]
            sends = []
            fell_back = False
            for r in range(world):
                gr = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY, hint)
                nsh = (Rl[r + 1] - Rl[r]) * (L + 1)
                dsh = ctx.alloc(nsh + 64)                        # (the rank's reads, generated again: an aligned buffer of its own)
                ctx.synth_reads_device(Rl[r], Rl[r + 1] - Rl[r], L, G, dsh, uniform=uniform, **seeds)
                ctx.sync()
                gr.add_device(dsh, nsh)
                ctx.free(dsh)
                cap = (2 if k > 32 else 1) * 4 * (Rl[r + 1] - Rl[r]) * max(2, (L - k + 1) // 4) + (1 << 18)   # (k > 32: two rows per record)
                buf = ctx.alloc(cap * 16)
                try:
                    rows = gr.export_runs_device(buf, cap, world)
                except cfrk_amd.CfrkError as e:
                    if e.code != -4:
                        raise
                    fell_back = True
                    ctx.free(buf)
                    break
                host = np.empty((sum(rows), 2), np.uint64)
                ctx.d2h(host, buf)
                ctx.free(buf)
                sends.append((host, rows))
                del gr
            if not fell_back:
                digs = []
                for owner in range(world):
                    segs = [h[sum(rw[:owner]):sum(rw[:owner + 1])] for h, rw in sends]
                    allb = np.concatenate(segs)
                    buf = ctx.alloc(max(len(allb), 1) * 16)
                    ctx.h2d(buf, allb)
                    og = cfrk_amd.GlobalCounter(octx, k, flags, hint // world + 1024)
                    og.merge_runs_device(buf, [len(x) for x in segs])
                    digs.append(og.digest())
                    del og
                    ctx.free(buf)
                ok2 = merge(digs) == want
                line += f" | runs exchange x{world}: {'ok' if ok2 else 'MISMATCH ' + str(merge(digs))}"
                ok = ok and ok2
            else:
                line += " | runs exchange: refused (spill), skipped"
        if ok and 16 <= k <= 64 and rng.random() < 0.5:
            # the PIPELINED runs exchange (round 5): deferred add, fused dedupe + pack per group, owners read the lists in place
            world = int(rng.integers(2, 9))
            ngroups = int(rng.choice([1, 2, 3, 8]))
            Rl = [R * r // world for r in range(world + 1)]
            # rows of a rank: at most its records (about one per four bases and two read ends per read), two rows each for k > 32
            rows_est = (2 if k > 32 else 1) * ((R // world + 1) * ((L + 1) // 4 + 2)) + (1 << 16)
            lpp = (65536 + world - 1) // world
            seg_cap = min(3 * rows_est // (ngroups * world) + lpp + 4096, 0x7FFFFFF)     # (three times an even share)
            per_rank, refused = [], None
            for r in range(world):
                gr = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, hint)
                nsh = (Rl[r + 1] - Rl[r]) * (L + 1)
                dsh = ctx.alloc(nsh + 64)
                ctx.synth_reads_device(Rl[r], Rl[r + 1] - Rl[r], L, G, dsh, uniform=uniform, **seeds)
                ctx.sync()
                buf = ctx.alloc(ngroups * world * seg_cap * 16)
                try:
                    gr.add_device(dsh, nsh)
                    gr.export_runs_async(buf, seg_cap, world, ngroups)
                    groups = []
                    for gi in range(ngroups):
                        rows = gr.export_runs_wait(gi)
                        host = np.empty((world * seg_cap, 2), np.uint64)
                        ctx.d2h(host, buf + gi * world * seg_cap * 16)
                        groups.append((rows, [host[o * seg_cap:o * seg_cap + rows[o]].copy() for o in range(world)]))
                        del host
                    per_rank.append(groups)
                except cfrk_amd.CfrkError as e:
                    if e.code not in (-4, -9, -11):
                        raise
                    refused = e.code
                ctx.sync()
                ctx.free(buf)
                ctx.free(dsh)
                del gr
                if refused is not None:
                    break
            if refused is None:
                digs = []
                for owner in range(world):
                    og = cfrk_amd.GlobalCounter(octx, k, flags, hint // world + 1024)
                    bufs = []
                    for gi in range(ngroups):
                        segs = [per_rank[r][gi][1][owner] for r in range(world)]
                        allb = np.concatenate(segs)
                        buf = ctx.alloc(max(len(allb), 1) * 16)
                        ctx.h2d(buf, allb)
                        og.merge_runs_group_device(buf, [len(x) for x in segs], gi, ngroups)
                        bufs.append(buf)
                    digs.append(og.digest())
                    del og
                    for b in bufs:
                        ctx.free(b)
                ok3 = merge(digs) == want
                line += f" | pipelined x{world}/{ngroups}: {'ok' if ok3 else 'MISMATCH ' + str(merge(digs))}"
                ok = ok and ok3
            else:
                line += f" | pipelined x{world}/{ngroups}: refused ({refused}), skipped"
    except cfrk_amd.CfrkError as e:
        ok = False
        line = f"{tag} ERROR {e}"
    ctx.free(d)
    print(line, flush=True)
    if not ok:
        bad += 1
print(f"{case} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
