"""GPU: 32-bit counts SATURATE at CFRK_COUNT_MAX = 2^32 - 2 in every path and the job says so
(CFRK_ERR_COUNT_OVERFLOW); single-key floods take the exact slow path instead of overflowing an LDS count or a
record's 26-bit multiplicity; unknown debug bits are refused (the timing ablations exist in an ablation build only).

The reference's rows are `int` and simply wrap (/root/reference/src/tipos.h:28, src/kmer_kernel.cu:66); the product
holds the count at 0xFFFFFFFE and reports it.  Pre-counted pairs through cfrk_global_merge_device make 2^32 cheap
to reach.
"""
import numpy as np
import pytest

from . import oracle_lib as orc
from . import refsem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import cfrk_amd
    c = cfrk_amd.Context(0)
    yield c
    c.close()


def _merge(ctx, g, lo, hi, cnt):
    lo = np.ascontiguousarray(lo, np.uint64)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    d_lo, d_cnt = ctx.alloc(lo.nbytes), ctx.alloc(cnt.nbytes)
    ctx.h2d(d_lo, lo); ctx.h2d(d_cnt, cnt)
    d_hi = 0
    if hi is not None:
        hi = np.ascontiguousarray(hi, np.uint64)
        d_hi = ctx.alloc(hi.nbytes)
        ctx.h2d(d_hi, hi)
    g.merge_device(d_lo, d_hi, d_cnt, len(lo))
    ctx.sync()
    ctx.free(d_lo); ctx.free(d_cnt)
    if d_hi:
        ctx.free(d_hi)


@pytest.mark.parametrize("k", [31, 63])
def test_merged_counts_saturate_and_the_job_reports_it(ctx, k):
    """a homopolymer k-mer (key 0) merged three times with 0x60000000 occurrences each reaches 2^32 - 2: its count is
    HELD at CFRK_COUNT_MAX (not wrapped to 0x20000000), every other key is exact, and finish / digest / export return
    CFRK_ERR_COUNT_OVERFLOW while still delivering the (saturated) result"""
    import cfrk_amd
    two = k > 32
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 1024)
    keys = np.array([0, 5, 77, 0, 0, 5], np.uint64)
    his = np.array([0, 1, 2, 0, 0, 1], np.uint64) if two else None
    cnts = np.array([0x60000000, 7, 0xFFFFFFF0, 0x60000000, 0x60000000, 8], np.uint32)
    for i in range(0, 6, 2):                               # three merges of two pairs each
        _merge(ctx, g, keys[i:i + 2], his[i:i + 2] if two else None, cnts[i:i + 2])
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.finish()
    assert e.value.code == cfrk_amd.CFRK_ERR_COUNT_OVERFLOW
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.digest()
    assert e.value.code == cfrk_amd.CFRK_ERR_COUNT_OVERFLOW
    lo, hi, cnt = g.export(allow_saturated=True)
    got = {(int(a), int(b)): int(c) for a, b, c in zip(lo, hi, cnt)}
    assert got == {(0, 0): cfrk_amd.CFRK_COUNT_MAX, (5, 1 if two else 0): 15, (77, 2 if two else 0): 0xFFFFFFF0}
    # the digest of the saturated result is that of the list above
    want = orc.digest(lo, hi if two else np.zeros_like(lo), cnt.astype(np.uint64), two_word=two)
    assert g.digest(allow_saturated=True) == tuple(int(x) for x in want)
    # ... and a job that stays below the maximum is not flagged: 0xFFFFFFFD is an ordinary count
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 1024)
    _merge(ctx, g, keys[:1], his[:1] if two else None, np.array([0xFFFFFFF0], np.uint32))
    _merge(ctx, g, keys[:1], his[:1] if two else None, np.array([0xD], np.uint32))
    lo, hi, cnt = g.export()
    assert int(cnt[0]) == 0xFFFFFFFD


def test_saturation_in_the_lds_merge_of_per_leaf_lists(ctx):
    """the owner's LDS merge of per-leaf lists (cfrk_global_merge_leaves_device) adds counts nothing bounds: two
    ranks that each hold 0xC0000000 occurrences of one key give CFRK_COUNT_MAX, not 0x80000000"""
    import cfrk_amd
    k = 31
    rng = np.random.default_rng(5)
    reads = [rng.integers(0, 4, 200).astype(np.int8) for _ in range(64)]
    data, start, length = refsem.flatten(reads)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 1 << 16)
    g.add(data, start, length)
    n = g.finish()
    parts = 1
    lpp = g.leaves_per_part(parts)
    d_keys, d_cnt, d_lc = ctx.alloc(8 * n), ctx.alloc(4 * n), ctx.alloc(4 * lpp)
    pc = g.export_leaves_device(d_keys, d_cnt, n, parts, d_lc)
    assert pc == [n]
    cnt = np.empty(n, np.uint32)
    ctx.d2h(cnt, d_cnt)
    cnt[:] = 1
    cnt[0] = 0xC0000000
    ctx.h2d(d_cnt, cnt)
    # "two ranks" with the same lists: the owner (parts = 2 would need interleaved leaves: merge the one list twice
    # through two calls of a one-part merge is not allowed on a non-empty job, so build a two-part input by hand)
    keys = np.empty(n, np.uint64); lc = np.empty(lpp, np.uint32)
    ctx.d2h(keys, d_keys); ctx.d2h(lc, d_lc)
    lpp2 = g.leaves_per_part(2)
    # owner 0 of a two-part job owns the even leaves: take the even leaves' segments, twice (rank 0 and rank 1)
    off = np.concatenate([[0], np.cumsum(lc)]).astype(np.int64)
    ev = [i for i in range(0, lpp, 2)]
    seg_k = np.concatenate([keys[off[i]:off[i + 1]] for i in ev]) if ev else keys[:0]
    seg_c = np.concatenate([cnt[off[i]:off[i + 1]] for i in ev]) if ev else cnt[:0]
    seg_l = np.array([lc[i] for i in ev] + [0] * (lpp2 - len(ev)), np.uint32)
    k2 = np.concatenate([seg_k, seg_k]); c2 = np.concatenate([seg_c, seg_c]); l2 = np.concatenate([seg_l, seg_l])
    d_k2, d_c2, d_l2 = ctx.alloc(max(8, k2.nbytes)), ctx.alloc(max(4, c2.nbytes)), ctx.alloc(l2.nbytes)
    ctx.h2d(d_k2, k2); ctx.h2d(d_c2, c2); ctx.h2d(d_l2, l2)
    og = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 1 << 16)
    og.merge_leaves_device(d_k2, d_c2, [len(seg_k), len(seg_k)], d_l2)
    lo, hi, c = og.export(allow_saturated=True)
    got = dict(zip((int(x) for x in lo), (int(x) for x in c)))
    want = {}
    for kk, cc in zip(seg_k, seg_c):
        want[int(kk)] = min(2 * int(cc), cfrk_amd.CFRK_COUNT_MAX)
    assert got == want
    heavy = int(keys[0]) in want                             # (the heavy key sits in an even leaf or not)
    if heavy:
        with pytest.raises(cfrk_amd.CfrkError) as e:
            og.finish()
        assert e.value.code == cfrk_amd.CFRK_ERR_COUNT_OVERFLOW
    for p in (d_keys, d_cnt, d_lc, d_k2, d_c2, d_l2):
        ctx.free(p)


@pytest.mark.parametrize("k,L", [(31, 150), (63, 250), (15, 150)])
def test_homopolymer_flood_is_counted_exactly(ctx, k, L):
    """300 000 poly-A reads: ONE canonical k-mer, every record in one leaf -- at k = 31 that leaf holds 3.6e7 records,
    more than HUGE_LEAF (2^25): the LDS leaf kernel leaves it alone (a run's multiplicity is kept in 26 bits there) and
    the huge-leaf kernel counts it k-mer by k-mer through the saturating HBM table.  Exact result: one key, R (L-k+1)
    occurrences (+ a sprinkling of other reads so that ordinary leaves are counted beside it)."""
    import cfrk_amd
    R = 300_000
    rng = np.random.default_rng(k)
    other = rng.integers(0, 4, (2000, L)).astype(np.int8)
    block = np.zeros((R + 2000, L + 1), np.int8)
    block[:, L] = -1
    block[R:, :L] = other
    data = block.reshape(-1)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 1 << 20)
    g.add(data)
    lo, hi, cnt = g.export()
    wlo, whi, wcnt = orc.global_count(np.ascontiguousarray(block[R - 3:].reshape(-1)), k, orc.ORC_CANONICAL)
    # the oracle counted 3 poly-A reads + the others: scale the poly-A key (0) up to R reads
    want = {(int(a), int(b)): int(c) for a, b, c in zip(wlo, whi, wcnt)}
    assert want[(0, 0)] == 3 * (L - k + 1)
    want[(0, 0)] = R * (L - k + 1)
    got = {(int(a), int(b)): int(c) for a, b, c in zip(lo, hi, cnt)}
    assert got == want


def test_unknown_debug_bits_are_refused(ctx):
    """cfrk_debug_set_flags accepts the eight documented test switches; the timing ablations (bits >= 0x100: kernels
    that skip a phase and count WRONG) are compiled into an ablation build only -- the product refuses them"""
    import cfrk_amd
    g = cfrk_amd.GlobalCounter(ctx, 31, 0, 1024)
    g.set_debug_flags(0xFF)
    g.set_debug_flags(0)
    for bad in (0x100, 0x800, 0x1000, 0x2000, 0x4000, 0x10000, 0x100000, 0x8000, 0x80000000):
        with pytest.raises(cfrk_amd.CfrkError) as e:
            g.set_debug_flags(bad)
        assert e.value.code == -1
