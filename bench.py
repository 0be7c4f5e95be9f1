#!/usr/bin/env python3
"""bench.py -- k-mers/sec of the global counting hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): R synthetic 150 bp
reads generated on device (SURVEY.md 8d generator), k=31, canonical; inputs resident in HBM
before the timed region.  A step = one full pass: clear the table, count every k-mer of the
resident batch, (N>1: exchange owner segments over RCCL and merge), sync.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 31] [--L 150]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--reads", type=int, default=100_000_000)
    p.add_argument("--L", type=int, default=150)
    p.add_argument("--k", type=int, default=31)
    p.add_argument("--glen", type=int, default=0, help="genome length (default: = reads)")
    p.add_argument("--no-canonical", action="store_true")
    p.add_argument("--cpu-reads", type=int, default=2_000_000,
                   help="reads of the same generator timed on the host cores (0: skip)")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo: rehearsal mode -- collectives staged through host memory, every "
                        "rank may sit on the same GPU (--same-gpu)")
    p.add_argument("--same-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal)")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak (default): every GPU gets --reads reads of the same genome (the job "
                        "grows with N); strong (BASELINE configs[3]): the --reads reads are split "
                        "over the GPUs")
    p.add_argument("--owner-hash", action="store_true",
                   help="force the generic hash-owner exchange + HBM-table merge")
    return p.parse_args()


def cpu_baseline(args, glen):
    """the oracle (a CPU restatement; the reference has no CPU path) on a bounded sample"""
    from tests import oracle_lib as orc
    R = min(args.cpu_reads, args.reads)
    threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
    data, _, _ = orc.synth_reads(0, R, args.L, glen)
    flags = 0 if args.no_canonical else orc.ORC_CANONICAL
    t0 = time.perf_counter()
    lo, hi, cnt = orc.global_count(data, args.k, flags, threads=threads)
    dt = time.perf_counter() - t0
    kmers = R * (args.L - args.k + 1)
    return {"value": kmers / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": f"first {R} reads of the same generator ({kmers} k-mers, {dt:.2f} s, "
                      f"{len(lo)} distinct), oracle/cfrk_oracle.c orc_global_count_mt"}, (lo, hi, cnt)


def measured_traffic(R, L, k, canonical):
    """HBM bytes per launch from the committed PMC measurement of this exact workload
    (profiles/*/traffic_*.json, produced by tools/traffic.sh); None when there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        w = t.get("workload", {})
        if (w.get("reads"), w.get("read_len"), w.get("k"), w.get("canonical")) == (R, L, k, canonical):
            return t.get("hbm_bytes_per_launch"), os.path.relpath(f, ROOT)
    return None, None


def main():
    args = parse()
    # the host driver only supports dmabuf IPC (RCCL / CUDA-tensor sharing across processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import cfrk_amd
    from cfrk_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    wire = "cpu" if args.dist_backend == "gloo" else None
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    L, k = args.L, args.k
    glen = args.glen or args.reads
    R = args.reads * world if args.scaling == "weak" else args.reads   # total reads of the job
    flags = 0 if args.no_canonical else cfrk_amd.CFRK_CANONICAL
    r0, r1 = sharded.shard_range(R, rank, world)       # strong scaling: the R reads are split
    Rl = r1 - r0
    nN = Rl * (L + 1)

    stream = torch.cuda.current_stream().cuda_stream
    ctx = cfrk_amd.Context(local_rank, stream)
    # struct-read buffers, resident in HBM (torch owns the memory; the library gets pointers)
    d_data = torch.empty(nN + 64, dtype=torch.int8, device=dev)
    d_start = torch.empty(Rl, dtype=torch.int64, device=dev)
    d_length = torch.empty(Rl, dtype=torch.int32, device=dev)
    ctx.synth_reads_device(r0, Rl, L, glen, d_data.data_ptr(), d_start.data_ptr(), d_length.data_ptr())
    torch.cuda.synchronize()

    hint = min(glen, R * max(L - k + 1, 0)) + 1024
    owner_ctx = cfrk_amd.Context(local_rank, stream) if world > 1 else None

    class Engine:
        def __init__(self):
            self.g = None
            self.bufs = None
            self.lbufs = None

        def export_leaves(self, parts):
            lpp = self.g.leaves_per_part(parts)
            if self.lbufs is None:
                cap = hint           # distinct keys of a shard never exceed the hint
                self.lbufs = (torch.empty(cap, dtype=torch.int64, device=dev),
                              torch.empty(cap, dtype=torch.int64, device=dev) if k > 32 else None,
                              torch.empty(cap, dtype=torch.int32, device=dev),
                              torch.empty(parts * lpp, dtype=torch.int32, device=dev))
            keys, hi, cnt, lc = self.lbufs
            try:
                pc = self.g.export_leaves_device(keys.data_ptr(), cnt.data_ptr(), keys.numel(), parts,
                                                 lc.data_ptr(), hi.data_ptr() if hi is not None else 0)
            except cfrk_amd.CfrkError as e:
                if e.code != -4:
                    raise
                return None
            return keys, hi, cnt, pc, lc

        def export_parts(self, parts):
            if self.bufs is None:
                cap = hint
                self.bufs = (torch.empty(cap, dtype=torch.int64, device=dev),
                             torch.empty(cap, dtype=torch.int64, device=dev) if k > 32 else None,
                             torch.empty(cap, dtype=torch.int32, device=dev))
            lo, hi, cnt = self.bufs
            pc = self.g.export_device(lo.data_ptr(), hi.data_ptr() if hi is not None else 0, cnt.data_ptr(),
                                      lo.numel(), parts)
            return lo, hi, cnt, pc

    eng = Engine()
    kernel_ms = []
    phase_s = [0.0, 0.0]      # N > 1: seconds until the exchange has returned / spent in the owner merge

    def step():
        eng.g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
        eng.g.add_device(d_data.data_ptr(), nN)
        if world == 1:
            ctx.sync()
            return eng.g
        og = cfrk_amd.GlobalCounter(owner_ctx, k, flags, hint // world + 1024)
        ta = time.perf_counter()
        got = None if args.owner_hash else sharded.exchange_by_leaf(eng, world, dev, wire)
        if got is not None:       # per-leaf lists, added in LDS on the owner
            rkeys, rhi, rcnt, recv_l, rlc = got
            tb = time.perf_counter()
            og.merge_leaves_device(rkeys.data_ptr(), rcnt.data_ptr(), recv_l, rlc.data_ptr(),
                                   rhi.data_ptr() if rhi is not None else 0)
            owner_ctx.sync()
            phase_s[0] += tb - ta
            phase_s[1] += time.perf_counter() - tb
            return og
        else:                     # generic: owner = hash(key), HBM-table merge
            rlo, rhi, rcnt = sharded.exchange_by_owner(eng, world, dev, wire)
            tb = time.perf_counter()      # the exchange ends with a stream sync: counting is done too
            og.merge_device(rlo.data_ptr(), rhi.data_ptr() if rhi is not None else 0, rcnt.data_ptr(), rlo.numel())
            owner_ctx.sync()
            phase_s[0] += tb - ta
            phase_s[1] += time.perf_counter() - tb
            return og

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    phase_s[0] = phase_s[1] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final = step()
        kernel_ms.append(eng.g.last_add_ms())
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if wire else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    info = eng.g.msp_info() if os.environ.get("CFRK_BENCH_INFO") else None
    digest = final.digest()
    if world > 1:
        digest = sharded.merge_digests(digest, "cpu" if wire else dev)

    kmers_total = R * max(L - k + 1, 0)
    D = digest[0]
    ok = digest[1] == kmers_total
    if rank == 0:
        S = 12 if k <= 32 else 20
        # SURVEY 8d algorithmic bytes per launch (per GPU): reads incl. terminators + the
        # start/length tables + every occupied slot written once and read once
        b_alg = Rl * (L + 1) + 12 * Rl + 2 * (D // world if world > 1 else D) * S
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = b_alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = (measured_traffic(R, L, k, bool(flags)) if world == 1 else (None, None))
        out = {
            "metric": "k-mers/sec", "value": kmers_total * args.steps / dt, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{R} synthetic {L} bp reads, k={k}, "
                                   f"{'canonical' if flags else 'forward'}, genome {glen} "
                                   + (("(BASELINE.json configs[2])" if world == 1 else
                                       "(BASELINE.json configs[3])" if args.scaling == "strong" else
                                       f"(configs[2] per GPU x{world}: weak scaling)")
                                      if (args.reads, L, k) == (100_000_000, 150, 31) and flags else "(off-config run)"),
                       "reads": R, "read_len": L, "k": k, "parallelism": f"read-shard x{world}"
                       + (" + owner all-to-all" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "counting kernels of cfrk_global_add_device (HIP events)",
                         "kernel_ms": avg_ms, "algorithmic_bytes": b_alg},
            "distinct": D, "sum_count_ok": ok,
            "digest": [f"{x:016x}" for x in digest],
        }
        if world > 1:
            # rank 0's view of one step: counting kernels (HIP events), everything up to the
            # return of the all-to-all (count + export + exchange), owner-side merge
            out["step_breakdown_ms"] = {"count_kernels": avg_ms,
                                        "count_export_exchange": phase_s[0] / args.steps * 1e3,
                                        "owner_merge": phase_s[1] / args.steps * 1e3}
        if info:
            out["msp_info"] = info
        if world == 1 and args.cpu_reads > 0:
            cb, _ = cpu_baseline(args, glen)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"sum(count) {digest[1]} != {kmers_total}")


if __name__ == "__main__":
    main()
