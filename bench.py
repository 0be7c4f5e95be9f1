#!/usr/bin/env python3
"""bench.py -- k-mers/sec of the global counting hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): R synthetic 150 bp
reads generated on device (SURVEY.md 8d generator), k=31, canonical; inputs resident in HBM
before the timed region.  A step = one full pass: clear the table, count every k-mer of the
resident batch, (N>1: exchange owner segments over RCCL and merge), sync.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|k63|c5] [--reads R] ...

`--gpus N` (N > 1) from a bare shell starts the N ranks itself: the parent spawns one child per GPU
BEFORE it imports torch or touches HIP and only waits for them (replaces the pthread fan-out of
/root/reference/src/main.cu:277-295).  Under torchrun (RANK / WORLD_SIZE already set) the process is
a rank.  N > 1 reports BASELINE configs[3] as written -- the SAME reads split over the GPUs (strong
scaling) -- as the headline, with the weak-scaling measurement under the key "weak".
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (reads, L, k, glen); reads of "c5" is per GPU (configs[4] is 10^9 reads on 8 GPUs)
CONFIGS = {
    "c3": (100_000_000, 150, 31, 0),          # BASELINE.json configs[2] / configs[3]
    "c2": (10_000_000, 150, 15, 0),           # configs[1]
    "k63": (100_000_000, 150, 63, 0),         # two-word keys on the C3 read shape
    "c5": (125_000_000, 250, 63, 1_000_000_000),   # configs[4]: one GPU's share
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    p.add_argument("--reads", type=int, default=0, help="override the config's read count")
    p.add_argument("--L", type=int, default=0)
    p.add_argument("--k", type=int, default=0)
    p.add_argument("--glen", type=int, default=0, help="genome length (default: = reads)")
    p.add_argument("--uniform", action="store_true", help="uniform random reads (all-distinct variant)")
    p.add_argument("--no-canonical", action="store_true")
    p.add_argument("--cpu-reads", type=int, default=10_000_000,
                   help="reads of the same generator timed on the host cores (0: skip)")
    p.add_argument("--e2e-reads", type=int, default=10_000_000,
                   help="N = 1 only, outside the timed region: reads of a FASTA FILE written with the same generator and "
                        "counted by the `cfrk` CLI (parse + H2D + count + export + write) -> the `end_to_end` object; "
                        "0 or --cpu-reads 0: skip")
    p.add_argument("--cpu-threads", type=int, default=0)
    p.add_argument("--cpu-runs", type=int, default=3, help="runs of the cpu_baseline sample (the median is reported)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo: rehearsal mode -- collectives staged through host memory, every "
                        "rank may sit on the same GPU (--same-gpu)")
    p.add_argument("--same-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal)")
    p.add_argument("--dist-at-one", action="store_true",
                   help="with --gpus 1: a one-rank process group, and the step goes through the exchange (this rank owns "
                        "every leaf) -- on a one-GPU box the only way the collectives execute over RCCL (tests)")
    p.add_argument("--scaling", default="", choices=["", "weak", "strong", "both"],
                   help="N > 1 only.  strong (BASELINE configs[3]): the reads are split over the GPUs; "
                        "weak: every GPU gets the config's reads; both (default): strong is the "
                        "headline and the weak measurement is reported under \"weak\"")
    p.add_argument("--owner-hash", action="store_true",
                   help="force the generic hash-owner exchange + HBM-table merge")
    p.add_argument("--debug-runs-budget", default="", metavar="RANK:BYTES",
                   help="(tests) cap rank RANK's record buffers at BYTES during its CFRK_RUNS_ONLY add")
    p.add_argument("--pipeline-groups", type=int, default=2,
                   help="N > 1, runs exchange, k <= 32: groups of leaves of the PIPELINED form (dedupe of group g+1 and the "
                        "owner's counting of group g-1 run beside the wire of group g); 0: the one-shot exchange")
    p.add_argument("--exchange", default="auto", choices=["auto", "runs", "leaf", "owner"],
                   help="N > 1: what the ranks exchange.  runs (auto for 16 <= k <= 64): deduplicated "
                        "runs, counted by the leaf's owner; leaf: counted per-leaf lists; owner: counted keys")
    a = p.parse_args()
    reads, L, k, glen = CONFIGS[a.config]
    # configs[4] is defined per GPU (125 M reads each: gpus x 125 M reads in all, whatever --scaling
    # says); every other config names the job's total
    a.per_gpu = a.config == "c5" and not a.reads
    a.reads = a.reads or reads
    a.L = a.L or L
    a.k = a.k or k
    a.glen = a.glen or glen
    return a


# ----------------------------------------------------------------------------------- launcher
def launch(args):
    """`bench.py --gpus N` from a bare shell: one child per GPU.  The parent makes no GPU call
    (it never imports torch or the library) and never re-execs; it relays the exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus),
                    "LOCAL_WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in alive:               # one rank failed: the others would wait for it forever
                    q.terminate()
    return rc


# --------------------------------------------------------------------------------- cpu baseline
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, glen, gpu_digest_of_prefix=None):
    """the oracle (a CPU restatement; the reference has no CPU path) on a bounded sample.
    gpu_digest_of_prefix(R) -> digest of the same R reads counted by the HIP path (outside the
    timed region): the oracle's counts are not thrown away but compared, so every default run is
    also a parity check at 10^7 reads on the box it runs on (`prefix_parity`)."""
    from tests import oracle_lib as orc
    R = min(args.cpu_reads, args.reads)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = args.cpu_threads or avail
    data, _, _ = orc.synth_reads(0, R, args.L, glen, uniform=args.uniform)
    flags = 0 if args.no_canonical else orc.ORC_CANONICAL
    # median of three runs of the same sample (BASELINE.md 3.4 asks for a median; ~8 s each on the GPU box's host)
    dts = []
    for _ in range(max(1, args.cpu_runs)):
        t0 = time.perf_counter()
        lo, hi, cnt = orc.global_count(data, args.k, flags, threads=threads)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    kmers = R * (args.L - args.k + 1)
    out = {"value": kmers / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
           "cpu_model": cpu_model(), "host_cpus_online": os.cpu_count(), "host_cpus_usable": avail,
           "runs_s": [round(x, 2) for x in dts],
           "sample": f"first {R} reads of the same generator ({kmers} k-mers, median of {len(dts)} runs: {dt:.2f} s, "
                     f"{len(lo)} distinct), oracle/cfrk_oracle.c, {threads} threads"}
    parity = None
    if gpu_digest_of_prefix is not None:
        want = tuple(int(x) for x in orc.digest(lo, hi, cnt))
        got = tuple(int(x) for x in gpu_digest_of_prefix(R))
        parity = got == want
        out["prefix_digest_oracle"] = [f"{x:016x}" for x in want]
        out["prefix_digest_gpu"] = [f"{x:016x}" for x in got]
    return out, parity


def measured_traffic(R, L, k, canonical, glen):
    """HBM bytes per launch from the committed PMC measurement of this exact workload
    (profiles/*/traffic_*.json, produced by tools/traffic.sh); None when there is none."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic_*.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        w = t.get("workload", {})
        if (w.get("reads"), w.get("read_len"), w.get("k"), w.get("canonical")) == (R, L, k, canonical) and \
                w.get("genome", glen) == glen:
            return t.get("hbm_bytes_per_launch"), os.path.relpath(f, ROOT) + (f" @ {t['git']}" if t.get("git") else " @ (commit not recorded)")
    return None, None


def config_label(args, world, scaling):
    base = (args.reads, args.L, args.k, args.glen or args.reads, args.uniform, args.no_canonical)
    for name, (reads, L, k, glen) in CONFIGS.items():
        if base == (reads, L, k, glen or reads, False, False):
            if name == "c3":
                if world == 1:
                    return "BASELINE.json configs[2]"
                return "BASELINE.json configs[3]" if scaling == "strong" else f"configs[2] per GPU x{world}: weak scaling"
            if name == "c2":
                return "BASELINE.json configs[1]" + ("" if world == 1 else f", {scaling} scaling")
            if name == "c5":
                return (f"BASELINE.json configs[4]: {world}/8 of it, 125 M reads per GPU" if world != 8
                        else "BASELINE.json configs[4]")
            return "C3 read shape at k=63"
    return "off-config run"


# ----------------------------------------------------------------------------------------- rank
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch(args))
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
    # --dist-at-one: a ONE-rank process group, and the step takes the exchange all the same (this rank owns every leaf):
    # on a one-GPU box this is the only way the collectives run over RCCL at all (tests)
    multi = world > 1 or args.dist_at_one
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    wire = "cpu" if args.dist_backend == "gloo" else None
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank)

    L, k = args.L, args.k
    flags = 0 if args.no_canonical else cfrk_amd.CFRK_CANONICAL
    stream = torch.cuda.current_stream().cuda_stream
    ctx = cfrk_amd.Context(local_rank, stream)
    owner_ctx = cfrk_amd.Context(local_rank, stream) if multi else None

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(scaling):
        """W warm-up + K timed steps of the job under `scaling`; returns rank 0's result dict"""
        # total reads of the job (a per-GPU config has the same job under both scalings)
        R = args.reads * world if (scaling == "weak" or args.per_gpu) else args.reads
        glen = args.glen or args.reads
        r0, r1 = sharded.shard_range(R, rank, world)       # contiguous read ranges
        Rl = r1 - r0
        nN = Rl * (L + 1)
        # struct-read buffers, resident in HBM (torch owns the memory; the library gets pointers)
        d_data = torch.empty(nN + 64, dtype=torch.int8, device=dev)
        d_start = torch.empty(Rl, dtype=torch.int64, device=dev)
        d_length = torch.empty(Rl, dtype=torch.int32, device=dev)
        ctx.synth_reads_device(r0, Rl, L, glen, d_data.data_ptr(), d_start.data_ptr(), d_length.data_ptr(),
                               uniform=args.uniform)
        torch.cuda.synchronize()
        kmers_total = R * max(L - k + 1, 0)
        hint = (kmers_total if args.uniform else min(glen, kmers_total)) + 1024

        class Engine:
            def __init__(self):
                self.g = None
                self.bufs = None
                self.lbufs = None
                self.runs_refused = False
                self.deferred = False

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

            def export_runs(self, parts):
                if self.runs_refused:      # the CFRK_RUNS_ONLY add itself was refused (shard needs several passes)
                    return None
                if getattr(self, "rbuf", None) is None:
                    # rows of 16 bytes: distinct runs (<= the leaf streams) + truncated runs + headers
                    # (truncated runs: two per read; distinct complete runs: ~2 strands x 2/(W+1) per genome base,
                    #  never more than the shard's super-k-mers; a too small buffer is doubled below)
                    rows = int(min(14 * Rl, 2.5 * Rl + 0.3 * glen)) * (2 if k > 32 else 1) + (1 << 17)   # (k > 32: two rows per record)
                    self.rbuf = torch.empty((max(1 << 20, rows), 2), dtype=torch.int64, device=dev)
                for _ in range(4):
                    try:
                        pr = self.g.export_runs_device(self.rbuf.data_ptr(), self.rbuf.shape[0], parts)
                        return self.rbuf, pr
                    except cfrk_amd.CfrkError as e:
                        if e.code == -9:          # buffer too small: double it
                            self.rbuf = torch.empty((2 * self.rbuf.shape[0], 2), dtype=torch.int64, device=dev)
                            continue
                        if e.code != -4:
                            raise
                        if self.deferred:         # a CFRK_RUNS_DEFER add that overflowed a region: add again, the settling way
                            self.deferred = False
                            self.g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY, hint)
                            try:
                                self.g.add_device(d_data.data_ptr(), nN)
                            except cfrk_amd.CfrkError as e2:
                                if e2.code != cfrk_amd.CFRK_ERR_RUNS_REFUSED:
                                    raise
                                return None
                            continue
                        return None
                return None

            # -- pipelined runs exchange (sharded.exchange_by_runs_pipelined) --
            def export_runs_pipelined_begin(self, parts, groups):
                if self.runs_refused:
                    return None
                lpp = self.g.leaves_per_part(parts)
                if getattr(self, "pbuf", None) is None or self.pbuf.shape[:2] != (groups, parts):
                    # rows of a rank as for the one-shot form, spread over groups x parts segments, a quarter to spare
                    # (a segment that runs out is reported by the wait: the step takes the one-shot form and the next
                    #  step gets twice the room)
                    rows = int(min(14 * Rl, 2.5 * Rl + 0.3 * glen)) * (2 if k > 32 else 1) + (1 << 17)   # (k > 32: two rows per record)
                    cap = int(rows / (groups * parts) * 1.25 * getattr(self, "pgrow", 1)) + (lpp + groups - 1) // groups + 4096
                    self.pbuf = torch.empty((groups, parts, cap, 2), dtype=torch.int64, device=dev)
                try:
                    self.g.export_runs_async(self.pbuf.data_ptr(), self.pbuf.shape[2], parts, groups)
                except cfrk_amd.CfrkError as e:
                    if e.code != -4:
                        raise
                    return None
                return self.pbuf

            def export_runs_pipelined_wait(self, grp):
                try:
                    return self.g.export_runs_wait(grp)
                except cfrk_amd.CfrkError as e:
                    if e.code == -9:             # a segment ran out of room: more next time
                        self.pgrow = 2 * getattr(self, "pgrow", 1)
                        self.pbuf = None
                    elif e.code != -4:
                        raise
                    used["pipelined_refused"] = str(e)
                    return None

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

        exch = "owner" if args.owner_hash else args.exchange
        if exch == "auto":
            # ranks ship deduplicated runs (read ends as 16-bit notes), the owners count -- for a job of fixed
            # size (strong scaling) and for full-depth shards alike: rehearsed at N = 8, a C3-sized shard
            # ships 0.65 GB of runs against 1.2 GB of counted k-mers and skips its leaf kernel, configs[4]'s
            # shard 4.1 GB against 17.5 GB.  A rank that cannot export runs makes all take the
            # leaf exchange (counted per-leaf lists), k <= 15 always does.
            exch = "runs" if 16 <= k <= 64 else "leaf"
        used = {"exchange": exch, "wire_bytes": 0}

        def step():
            runs = multi and exch == "runs"
            piped = runs and args.pipeline_groups > 0
            eng.deferred = piped
            eng.g = cfrk_amd.GlobalCounter(ctx, k, flags | (cfrk_amd.CFRK_RUNS_ONLY if runs else 0)
                                           | (cfrk_amd.CFRK_RUNS_DEFER if piped else 0), hint)
            if os.environ.get("CFRK_DEBUG_FLAGS"):       # test switches; bits >= 0x100 are the timing ablations, which only
                eng.g.set_debug_flags(int(os.environ["CFRK_DEBUG_FLAGS"], 0))   # an ablation build accepts (tools/ablate.sh)
            for kv in filter(None, os.environ.get("CFRK_BENCH_PARAMS", "").split(",")):   # sizing experiments
                eng.g.set_debug_param(int(kv.split(":")[0]), float(kv.split(":")[1]))
            budget = runs and args.debug_runs_budget and int(args.debug_runs_budget.split(":")[0]) == rank
            if budget:                                   # (tests) this rank's shard does not fit one pass
                eng.g.set_mem_budget(int(args.debug_runs_budget.split(":")[1]))
            eng.runs_refused = False
            try:
                eng.g.add_device(d_data.data_ptr(), nN)
            except cfrk_amd.CfrkError as e:
                # a CFRK_RUNS_ONLY job must fit device memory in one pass (CFRK_ERR_RUNS_REFUSED -- and only that
                # code: a genuine call-sequence error is not turned into a vote): this rank votes "no runs" in the
                # size exchange and every rank takes the leaf exchange together
                if not (runs and e.code == cfrk_amd.CFRK_ERR_RUNS_REFUSED):
                    raise
                eng.runs_refused = True
                used["runs_refused"] = str(e)
            finally:
                if budget:
                    eng.g.set_mem_budget(0)
            if not multi:
                ctx.sync()
                return eng.g
            og = cfrk_amd.GlobalCounter(owner_ctx, k, flags, hint // world + 1024)
            ta = time.perf_counter()
            if piped:
                # groups of leaves: dedupe(g+1) || wire(g) || owner(g-1); nothing but the comm side ever waits
                kept, wb = sharded.exchange_by_runs_pipelined(
                    eng, lambda recv, rows, grp, ngrp: og.merge_runs_group_device(recv.data_ptr(), rows, grp, ngrp),
                    world, dev, wire, args.pipeline_groups)
                if kept is not None:
                    tb = time.perf_counter()
                    owner_ctx.sync()
                    phase_s[0] += tb - ta
                    phase_s[1] += time.perf_counter() - tb
                    used["exchange"] = f"runs, pipelined in {args.pipeline_groups} groups"
                    used["wire_bytes"] = wb
                    used["owner_ms"] = og.last_add_ms()
                    return og
                # some rank could not deliver group `wb`: everybody takes the one-shot form, on a fresh owner job
                used["pipelined_fell_back_at_group"] = wb
                og = cfrk_amd.GlobalCounter(owner_ctx, k, flags, hint // world + 1024)
            if runs:
                got = sharded.exchange_by_runs(eng, world, dev, wire)
                if got is not None:   # the owner expands and counts the runs of its leaves
                    packed, recv_rows = got
                    tb = time.perf_counter()
                    og.merge_runs_device(packed.data_ptr(), recv_rows)
                    owner_ctx.sync()
                    phase_s[0] += tb - ta
                    phase_s[1] += time.perf_counter() - tb
                    used["wire_bytes"] = 16 * (sum(recv_rows) - recv_rows[rank])
                    used["owner_ms"] = og.last_add_ms()
                    return og
                # some rank could not export runs: count the shard after all, then exchange counts
                used["exchange"] = "leaf (runs export refused)"
                eng.g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
                eng.g.add_device(d_data.data_ptr(), nN)
            got = None if exch == "owner" else sharded.exchange_by_leaf(eng, world, dev, wire)
            if got is not None:       # per-leaf lists, added in LDS on the owner
                rkeys, rhi, rcnt, recv_l, rlc = got
                tb = time.perf_counter()
                og.merge_leaves_device(rkeys.data_ptr(), rcnt.data_ptr(), recv_l, rlc.data_ptr(),
                                       rhi.data_ptr() if rhi is not None else 0)
                owner_ctx.sync()
            else:                     # generic: owner = hash(key), HBM-table merge
                rlo, rhi, rcnt = sharded.exchange_by_owner(eng, world, dev, wire)
                tb = time.perf_counter()      # the exchange ends with a stream sync: counting is done too
                og.merge_device(rlo.data_ptr(), rhi.data_ptr() if rhi is not None else 0, rcnt.data_ptr(), rlo.numel())
                owner_ctx.sync()
            phase_s[0] += tb - ta
            phase_s[1] += time.perf_counter() - tb
            return og

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
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if wire else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        info = eng.g.msp_info() if os.environ.get("CFRK_BENCH_INFO") else None
        digest = final.digest()
        if multi:
            digest = sharded.merge_digests(digest, "cpu" if wire else dev)
        del d_data, d_start, d_length, eng
        torch.cuda.empty_cache()

        D = digest[0]
        ok = digest[1] == kmers_total
        S = 12 if k <= 32 else 20
        Dl = D // world if world > 1 else D
        # SURVEY 8d algorithmic bytes per launch (per GPU): reads incl. terminators + the
        # start/length tables + every occupied slot written once and read once
        b_alg = Rl * (L + 1) + 12 * Rl + 2 * Dl * S
        # ... and the bytes the TIMED kernels cannot avoid: they read the code buffer (never
        # start/length: terminators delimit the reads) and write every result entry once; the
        # export read happens after the timed region
        b_timed = Rl * (L + 1) + Dl * S
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = b_alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = (measured_traffic(R, L, k, bool(flags), glen) if world == 1 and not args.uniform
                                else (None, None))
        out = {
            "metric": "k-mers/sec", "value": kmers_total * args.steps / dt, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            # (N = 1 is the first point of the strong-scaling curve of configs[3]: the job is fixed;
            #  a per-GPU config -- configs[4]: 125 M reads on every GPU -- grows with N whatever was asked)
            "scaling": "weak" if (args.per_gpu and world > 1) else scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{R} synthetic {L} bp reads, k={k}, "
                                   f"{'canonical' if flags else 'forward'}, "
                                   + ("uniform random reads " if args.uniform else f"genome {glen} ")
                                   + f"({config_label(args, world, scaling)})",
                       "reads": R, "read_len": L, "k": k, "parallelism": f"read-shard x{world}"
                       + (" + owner all-to-all" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_kind": ("committed PMC measurement of this workload (tools/traffic.sh, file named in "
                                          "traffic_source): NOT taken during this run" if traffic else None),
                         "kernel": "counting kernels of cfrk_global_add_device (HIP events on the context stream)",
                         "kernel_ms": avg_ms, "algorithmic_bytes": b_alg,
                         "timed_bytes": b_timed, "frac_timed_bytes": b_timed / (avg_ms * 1e-3) / 8e12,
                         "frac_of_copy_ceiling_6.29TBs": achieved / 6290.0},
            # device memory of the job: the library's buffers (level-1 records, leaf streams, result
            # list, table) + the resident reads
            "device_bytes": {"library": ctx.device_bytes(), "reads": int(nN + 12 * Rl)},
            "distinct": D, "sum_count_ok": ok,
            "digest": [f"{x:016x}" for x in digest],
        }
        if multi:
            # rank 0's view of one step: counting kernels (HIP events), everything up to the
            # return of the all-to-all (count + export + exchange), owner-side merge
            out["step_breakdown_ms"] = {"count_kernels": avg_ms,
                                        "count_export_exchange": phase_s[0] / args.steps * 1e3,
                                        "owner_merge": phase_s[1] / args.steps * 1e3}
            out["exchange"] = used
        if info:
            out["msp_info"] = info
        return out, ok, kmers_total

    scaling = args.scaling or "both"
    parity = None
    if world == 1:
        out, ok, total = measure("strong")
    elif scaling == "both" and not args.per_gpu:
        out, ok, total = measure("strong")
        # The headline is safe from here on: if the weak pass (N x the per-GPU memory) dies or
        # hangs, the watchdog / the handler below still print the strong-scaling line.
        _PENDING["line"] = dict(out, weak={"error": "weak-scaling pass did not finish"}, cpu_baseline=None)
        _PENDING["ok"] = ok
        try:
            w, wok, _ = measure("weak")
            ok = ok and wok
            out["weak"] = {key: w[key] for key in ("value", "ms_per_step", "config", "distinct", "sum_count_ok",
                                                   "digest", "step_breakdown_ms", "exchange")}
        except Exception as e:            # noqa: BLE001 -- reported in the line, the headline stands
            out["weak"] = {"error": f"{type(e).__name__}: {e}"}
    else:
        out, ok, total = measure("strong" if scaling == "both" else scaling)
    _PENDING["line"] = None

    if rank == 0:
        if world == 1 and not multi and args.cpu_reads > 0:
            def gpu_digest_of_prefix(Rp):
                nNp = Rp * (L + 1)
                d = torch.empty(nNp + 64, dtype=torch.int8, device=dev)
                ctx.synth_reads_device(0, Rp, L, args.glen or args.reads, d.data_ptr(), uniform=args.uniform)
                g = cfrk_amd.GlobalCounter(ctx, k, flags, min(args.glen or args.reads, Rp * max(L - k + 1, 0)) + 1024)
                g.add_device(d.data_ptr(), nNp)
                return g.digest()
            out["cpu_baseline"], parity = cpu_baseline(args, args.glen or args.reads, gpu_digest_of_prefix)
            out["prefix_parity"] = parity
            if args.e2e_reads > 0:
                # the product end to end on a FILE (SURVEY 8d "what is timed"; /root/reference/src/main.cu:259-268):
                # reported beside the headline, never as `value`
                from tools import bench_e2e
                out["end_to_end"] = bench_e2e.measure(min(args.e2e_reads, args.reads), L, k, 0, ctx=ctx, quick=True)   # (genome = reads: configs[1]'s shape)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    # (only a timing ablation -- bits >= 0x100, accepted by an ablation build of the library alone -- may miscount)
    if not ok and not (int(os.environ.get("CFRK_DEBUG_FLAGS", "0"), 0) & ~0xFF):
        raise SystemExit(f"sum(count) != {total}")
    if parity is False:
        raise SystemExit("prefix_parity false: the HIP path and the oracle disagree on the cpu_baseline sample")


# what rank 0 prints if the process has to give up after the headline measurement finished
_PENDING = {"line": None, "ok": True}


def _watchdog(seconds):
    """Hard limit for one rank: a rank stuck in a collective (the first RCCL run is the driver's)
    must not hang the job.  Dumps every thread's stack, prints a finished headline if there is one,
    and leaves with os._exit -- no destructors, no process-group teardown that could hang again."""
    import faulthandler
    import threading

    def fire():
        r = os.environ.get("RANK", "0")
        sys.stderr.write(f"[bench rank {r}] hard timeout after {seconds} s; stacks follow\n")
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        line = _PENDING["line"]
        if line is not None and r == "0":
            print(json.dumps(line), flush=True)
            os._exit(0 if _PENDING["ok"] else 1)
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def _rank_entry():
    """main() with every failure reported per rank on stderr (rank id, traceback, the library's
    last error text is in the exception) and a non-hanging exit."""
    limit = float(os.environ.get("CFRK_BENCH_TIMEOUT", "1500"))
    wd = _watchdog(limit) if limit > 0 else None
    try:
        main()
    except SystemExit:
        raise
    except BaseException:             # noqa: BLE001
        import traceback
        r, w = os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1")
        sys.stderr.write(f"[bench rank {r}/{w}] failed:\n{traceback.format_exc()}\n")
        sys.stderr.flush()
        line = _PENDING["line"]
        if line is not None and r == "0":
            print(json.dumps(line), flush=True)
        # other ranks may be inside a collective with this one: leave without teardown
        os._exit(1)
    finally:
        if wd is not None:
            wd.cancel()


if __name__ == "__main__":
    _rank_entry()
