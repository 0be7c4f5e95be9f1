"""N>1 path on CPU: 2 ranks over gloo.  The counting engine is the oracle (the checker); what is
under test is cfrk_amd/sharded.py: read-range sharding, the owner all-to-all with ragged
segments, the count-add merge and the cross-rank digest."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, L, G, K = 3000, 100, 20000, 31


class OracleEngine:
    """duck-typed stand-in for the HIP engine (tests only)"""

    def __init__(self, data, k):
        from tests import oracle_lib as orc
        self.orc = orc
        self.lo, _, self.cnt = orc.global_count(data, k, orc.ORC_CANONICAL)
        self.merged = {}

    def export_parts(self, parts):
        own = np.array([self.orc.splitmix64(int(x)) % parts for x in self.lo], np.int64)
        order = np.argsort(own, kind="stable")
        pc = [int((own == p).sum()) for p in range(parts)]
        lo = torch.from_numpy(self.lo[order].view(np.int64).copy())
        cnt = torch.from_numpy(self.cnt[order].astype(np.uint32).view(np.int32).copy())
        return lo, None, cnt, pc

    def export_leaves(self, parts):
        return None                     # the oracle has no leaf form: exercises the fallback vote

    NL = 16                             # "leaves" of the stand-in: splitmix(key) % NL

    def export_leaves_fake(self, parts):
        """leaf form of the same table: part = leaf % parts, lists ordered by (part, leaf)"""
        leaf = np.array([self.orc.splitmix64(int(x)) % self.NL for x in self.lo], np.int64)
        lpp = self.NL // parts
        order = np.lexsort((self.lo, leaf // parts, leaf % parts))
        pc = [int((leaf % parts == p).sum()) for p in range(parts)]
        lc = np.zeros((parts, lpp), np.int32)
        for lf in leaf:
            lc[lf % parts, lf // parts] += 1
        lo = torch.from_numpy(self.lo[order].view(np.int64).copy())
        cnt = torch.from_numpy(self.cnt[order].astype(np.uint32).view(np.int32).copy())
        return lo, None, cnt, pc, torch.from_numpy(lc.reshape(-1).copy())

    def merge(self, lo, hi, cnt):
        for k_, c in zip(lo.numpy().view(np.uint64), cnt.numpy().view(np.uint32)):
            self.merged[int(k_)] = self.merged.get(int(k_), 0) + int(c)

    def digest(self):
        keys = np.array(sorted(self.merged), np.uint64)
        cnts = np.array([self.merged[int(x)] for x in keys], np.uint64)
        return self.orc.digest(keys, None, cnts)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cfrk_amd import sharded
    from tests import oracle_lib as orc
    r0, r1 = sharded.shard_range(R, rank, world)
    data, _, _ = orc.synth_reads(r0, r1 - r0, L, G)
    eng = OracleEngine(data, K)
    assert sharded.exchange_by_leaf(eng, world, torch.device("cpu")) is None   # all ranks agree
    # one rank cannot export by leaf: every rank must fall back
    eng.export_leaves = (lambda parts: None) if rank == 1 else eng.export_leaves_fake
    assert sharded.exchange_by_leaf(eng, world, torch.device("cpu")) is None
    # all can: segment sizes and the vote travel in one collective
    eng.export_leaves = eng.export_leaves_fake
    rkeys, rhi_, rcnt_, recv_l, rlc = sharded.exchange_by_leaf(eng, world, torch.device("cpu"))
    assert rhi_ is None and sum(recv_l) == len(rkeys) == len(rcnt_) == int(rlc.sum())
    assert all(orc.splitmix64(int(x)) % OracleEngine.NL % world == rank for x in rkeys.numpy().view(np.uint64))
    # runs exchange (strong scaling): two collectives, ragged packed segments; rows carry their
    # (source, destination, index) so that the receiver can tell where everything came from
    def fake_runs(parts, me=rank):
        rows = [3 + 2 * me + p for p in range(parts)]
        packed = torch.tensor([[me * 1000 + p, i] for p in range(parts) for i in range(rows[p])], dtype=torch.int64)
        return packed, rows
    eng.export_runs = (lambda parts: None) if rank == 0 else fake_runs
    assert sharded.exchange_by_runs(eng, world, torch.device("cpu")) is None          # one rank cannot: nobody does
    eng.export_runs = fake_runs
    packed, recv_rows = sharded.exchange_by_runs(eng, world, torch.device("cpu"))
    assert recv_rows == [3 + 2 * src + rank for src in range(world)]
    want_rows = [[src * 1000 + rank, i] for src in range(world) for i in range(3 + 2 * src + rank)]
    assert packed.tolist() == want_rows
    eng.export_leaves = lambda parts: None
    rlo, rhi, rcnt = sharded.exchange_by_owner(eng, world, torch.device("cpu"))
    eng.merge(rlo, rhi, rcnt)
    # every received key must be owned by this rank
    assert all(orc.splitmix64(int(x)) % world == rank for x in rlo.numpy().view(np.uint64))
    total = sharded.merge_digests(eng.digest(), torch.device("cpu"))
    if rank == 0:
        q.put(total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_owner_exchange_matches_single_process():
    from tests import oracle_lib as orc
    data, _, _ = orc.synth_reads(0, R, L, G)
    lo, hi, cnt = orc.global_count(data, K, orc.ORC_CANONICAL)
    want = orc.digest(lo, hi, cnt)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tuple(got) == tuple(want)


def test_shard_range_covers_everything_once():
    from cfrk_amd import sharded
    for total in (0, 1, 7, 100_000_000):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
