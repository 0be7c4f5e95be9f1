"""Read-sharded global counting over N GPUs with one key-owner exchange.

The reference's "multi GPU" fan-out (/root/reference/src/main.cu:208-230,277-284) runs every
pthread on the same device and has no merge step; this replaces it.  Reads are range-partitioned
(rank r counts reads [r*R/N, (r+1)*R/N)); each rank then splits its table into N segments by
owner(key) = mix(key) % N, one all-to-all (RCCL over xGMI; gloo in the CPU tests) moves
segment j to rank j, and the owner adds the received (key, count) pairs.  Afterwards rank j
holds the final counts of exactly the keys it owns: a reduce-scatter by key.  An element-wise
ncclReduce of the raw tables would be wrong because slot positions depend on insertion order.

The counting engine is passed in (duck-typed):
    engine.export_parts(parts) -> (lo, hi_or_None, cnt, part_counts)   torch tensors + list
    engine.merge(lo, hi_or_None, cnt)                                  add pairs into the owner table

Three exchange forms, from the cheapest for a job of FIXED size split over the ranks to the most
general: exchange_by_runs (ranks only partition and deduplicate, owners count: per-rank work and
bytes on the wire shrink with N), exchange_by_leaf (counted per-leaf lists, LDS merge on the owner),
exchange_by_owner (counted keys by owner hash, HBM-table merge).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """contiguous range of units for `rank` (GPU g gets reads [g*R/G, (g+1)*R/G))"""
    return (total * rank) // world, (total * (rank + 1)) // world


def exchange_by_owner(engine, world, device, wire_device=None):
    """all-to-all of the owner segments; returns (lo, hi, cnt) received by this rank.

    Keys travel as int64 bit patterns, counts as int32 (no unsigned collectives needed).
    wire_device: where the collective runs (default: `device`, i.e. RCCL on GPU tensors);
    "cpu" stages through host memory so that the same code runs over gloo."""
    wire = torch.device(wire_device) if wire_device is not None else torch.device(device)
    lo, hi, cnt, part_counts = engine.export_parts(world)
    send = torch.tensor(part_counts, dtype=torch.int64, device=wire)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    send_l = [int(x) for x in part_counts]
    recv_l = [int(x) for x in recv.cpu().tolist()]
    n_send, n_recv = sum(send_l), sum(recv_l)

    def a2a(t):
        src = t[:n_send].contiguous().to(wire)
        out = torch.empty(n_recv, dtype=t.dtype, device=wire)
        dist.all_to_all_single(out, src, recv_l, send_l)
        return out.to(device)

    rlo = a2a(lo.view(torch.int64))
    rhi = a2a(hi.view(torch.int64)) if hi is not None else None
    rcnt = a2a(cnt.view(torch.int32))
    _fence(device)
    return rlo, rhi, rcnt


def exchange_by_leaf(engine, world, device, wire_device=None):
    """Owner exchange at leaf granularity (the partitioned path's result form): returns
    (keys, keys_hi_or_None, counts, recv_counts, leaf_counts) for engine-side `merge_leaves`, or
    None when some rank cannot export by leaf -- every rank then takes the generic
    exchange_by_owner path.  Two collectives: the segment sizes with every rank's vote, then ONE
    payload of 8-byte words per destination: [per-leaf counts | keys | high key words | counts].

    engine.export_leaves(parts) -> (keys, keys_hi_or_None, counts, part_counts, leaf_counts) or None."""
    wire = torch.device(wire_device) if wire_device is not None else torch.device(device)
    exp = engine.export_leaves(world)
    # one small all-to-all carries the segment sizes AND every rank's "I can export by leaf" vote
    # (each rank tells every other): one collective and one host sync instead of two
    part_counts = exp[3] if exp is not None else [0] * world
    mine = 1 if exp is not None else 0
    send = torch.tensor([[int(c), mine] for c in part_counts], dtype=torch.int64, device=wire)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    got = recv.cpu().tolist()
    if mine == 0 or min(int(g[1]) for g in got) == 0:
        return None
    keys, hi, cnt, part_counts, leaf_counts = exp
    send_l = [int(x) for x in part_counts]
    recv_l = [int(g[0]) for g in got]
    lc = leaf_counts.view(torch.int32).reshape(world, -1)                  # [world][leaves_per_part]
    lpp = lc.shape[1]
    lw = (lpp + 1) // 2                                                   # words of a segment's leaf counts

    def words(n):                                                         # words of a segment with n entries
        return lw + n * (2 if hi is not None else 1) + (n + 1) // 2

    def pad32(t):                                                         # int32 vector -> whole int64 words (a copy:
        return torch.cat([t, t.new_zeros(t.numel() % 2)]).view(torch.int64)   # a slice may start on an odd element)

    k64, c32 = keys.view(torch.int64), cnt.view(torch.int32)
    h64 = hi.view(torch.int64) if hi is not None else None
    parts, off = [], 0
    for p, n in enumerate(send_l):
        parts.append(pad32(lc[p]))
        parts.append(k64[off:off + n])
        if h64 is not None:
            parts.append(h64[off:off + n])
        parts.append(pad32(c32[off:off + n]))
        off += n
    src = torch.cat(parts).to(wire)
    out = torch.empty(sum(words(n) for n in recv_l), dtype=torch.int64, device=wire)
    dist.all_to_all_single(out, src, [words(n) for n in recv_l], [words(n) for n in send_l])
    out = out.to(device)
    rk, rh, rc, rl, off = [], [], [], [], 0
    for n in recv_l:
        rl.append(out[off:off + lw].view(torch.int32)[:lpp]); off += lw
        rk.append(out[off:off + n]); off += n
        if h64 is not None:
            rh.append(out[off:off + n]); off += n
        rc.append(out[off:off + (n + 1) // 2].view(torch.int32)[:n]); off += (n + 1) // 2
    rkeys = torch.cat(rk)
    rhi = torch.cat(rh) if h64 is not None else None
    rcnt = torch.cat(rc).contiguous()
    rlc = torch.cat(rl).contiguous()
    _fence(device)
    return rkeys, rhi, rcnt, recv_l, rlc


def exchange_by_runs(engine, world, device, wire_device=None):
    """Strong-scaling exchange: every rank has only PARTITIONED and DEDUPLICATED its shard
    (CFRK_RUNS_ONLY) and ships each leaf's distinct runs with multiplicities plus its truncated runs
    to the leaf's owner, which expands and counts them once.  Two collectives: the segment sizes
    (with every rank's "I can export runs" vote riding along) and one packed payload of 16-byte
    rows (segment = header with the per-leaf sizes + records).  Returns (packed, recv_rows) for
    engine-side `merge_runs`, or None when some rank cannot export runs.

    engine.export_runs(parts) -> (packed int64 tensor [rows, 2], part_rows) or None."""
    wire = torch.device(wire_device) if wire_device is not None else torch.device(device)
    exp = engine.export_runs(world)
    part_rows = exp[1] if exp is not None else [0] * world
    mine = 1 if exp is not None else 0
    send = torch.tensor([[int(c), mine] for c in part_rows], dtype=torch.int64, device=wire)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    got = recv.cpu().tolist()
    if mine == 0 or min(int(g[1]) for g in got) == 0:
        return None
    packed = exp[0]
    send_l = [int(x) for x in part_rows]
    recv_l = [int(g[0]) for g in got]
    src = packed[:sum(send_l)].contiguous().to(wire)
    out = torch.empty((sum(recv_l), 2), dtype=torch.int64, device=wire)
    dist.all_to_all_single(out, src, recv_l, send_l)
    out = out.to(device)
    _fence(device)
    return out, recv_l


def exchange_by_runs_pipelined(engine, owner_merge, world, device, wire_device=None, groups=2):
    """The runs exchange cut into `groups` groups of leaves so that nothing waits for anything it does not need
    (VERDICT r4 item 1):

        compute stream:  [partition] [dedupe + pack g0] [dedupe + pack g1] ... [owner counts g0] [owner counts g1] ...
        comm stream:                                    [sizes g0][payload g0]  [sizes g1][payload g1]

    Everything of the compute stream is ENQUEUED before the first wait: the rank's kernels run back to back, the
    wire of group g runs under the deduplication of group g+1 and under the owner's work on group g-1.  Per group the
    host waits twice, for things that are long done unless the wire is the bottleneck: for the group's event (its
    segments are complete, their sizes sit in pinned memory) and for the size all-to-all -- both on the comm side, never
    for the compute stream as a whole.  No header parsing, no layout pass and no copy on the owner: its leaf kernel
    reads the received segments in place.

    engine.export_runs_pipelined_begin(parts, groups) -> send buffer (int64 tensor [groups, parts, seg_cap, 2]) or None
        (this rank cannot: a refused add, leaves shared by sub-value); enqueues the packing of every group and returns at once.
    engine.export_runs_pipelined_wait(g) -> rows per owner segment of group g, or None (a segment ran out of room, the
        add overflowed a region).
    owner_merge(recv, recv_rows, g, groups): enqueue the owner's counting of group g from `recv` (rows in rank order).
    Returns (list of the received buffers -- keep them until the owner context is synchronised --, bytes received), or
    (None, g) when some rank could not deliver group g: every rank sees the same votes and leaves together; the caller
    takes exchange_by_runs (g > 0: on a FRESH owner job -- groups before g have been merged into this one)."""
    wire = torch.device(wire_device) if wire_device is not None else torch.device(device)
    on_gpu = torch.device(device).type == "cuda"
    buf = engine.export_runs_pipelined_begin(world, groups)
    comm = torch.cuda.Stream(torch.device(device)) if on_gpu else None
    compute = torch.cuda.current_stream(torch.device(device)) if on_gpu else None

    class _Null:
        def __enter__(self): return self
        def __exit__(self, *a): return False
    on_comm = (lambda: torch.cuda.stream(comm)) if on_gpu else _Null
    kept, wire_bytes = [], 0
    for g in range(groups):
        rows = engine.export_runs_pipelined_wait(g) if buf is not None else None       # host wait 1: group g's event
        mine = 1 if rows is not None else 0
        rows = rows if rows is not None else [0] * world
        with on_comm():
            send = torch.tensor([[int(c), mine] for c in rows], dtype=torch.int64, device=wire)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send)
            got = recv.cpu().tolist()                                                   # host wait 2: the comm stream only
        if mine == 0 or min(int(x[1]) for x in got) == 0:
            if on_gpu:
                comm.synchronize()
            return None, g
        send_l = [int(x) for x in rows]
        recv_l = [int(x[0]) for x in got]
        with on_comm():
            if wire.type == "cpu":              # rehearsal over gloo: staged through host memory
                src = torch.cat([buf[g, p, :send_l[p]] for p in range(world)]).to(wire)
                out = torch.empty((sum(recv_l), 2), dtype=torch.int64, device=wire)
                dist.all_to_all_single(out, src, recv_l, send_l)
                out = out.to(device)
            else:                               # RCCL: the used rows of every segment straight from the send buffer
                out = torch.empty((sum(recv_l), 2), dtype=torch.int64, device=wire)
                dist.all_to_all(list(out.split(recv_l)), [buf[g, p, :send_l[p]] for p in range(world)])
            if on_gpu:
                done = torch.cuda.Event()
                done.record(comm)
        if on_gpu:
            compute.wait_event(done)            # the owner's kernels of this group start behind the payload, nothing else waits
        owner_merge(out, recv_l, g, groups)
        kept.append(out)
        wire_bytes += 16 * (sum(recv_l) - recv_l[dist.get_rank()])
    return kept, wire_bytes


def _fence(device):
    """The collectives are ordered on torch's current stream; the counting library launches on
    its own HIP stream.  Drain torch's stream before handing the received buffers over."""
    if torch.device(device).type == "cuda":
        torch.cuda.current_stream(torch.device(device)).synchronize()


def merge_digests(local, device):
    """combine per-rank digests (owners hold disjoint key sets): sums mod 2^64 and xor"""
    world = dist.get_world_size()
    device = torch.device(device)
    mine = torch.tensor([_to_i64(x) for x in local], dtype=torch.int64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    d = s = w = x = 0
    for v in allv:
        a = [int(t) & 0xFFFFFFFFFFFFFFFF for t in v.cpu().tolist()]
        d = (d + a[0]) & 0xFFFFFFFFFFFFFFFF
        s = (s + a[1]) & 0xFFFFFFFFFFFFFFFF
        w = (w + a[2]) & 0xFFFFFFFFFFFFFFFF
        x ^= a[3]
    return (d, s, w, x)


def _to_i64(u):
    u &= 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u
