"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference goldens.

Bit-exact everywhere: this is integer work.  Run with `pytest -m gpu` on an MI355X.
"""
import os

import numpy as np
import pytest

from . import oracle_lib as orc
from . import refsem
from .conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import cfrk_amd
    c = cfrk_amd.Context(0)
    yield c
    c.close()


def _random_reads(rng, n, lo, hi, p_invalid=0.02):
    reads = []
    for L in rng.integers(lo, hi, n):
        r = rng.integers(0, 4, int(L)).astype(np.int8)
        if p_invalid:
            r[rng.random(int(L)) < p_invalid] = -1
        reads.append(r)
    return reads


# ------------------------------------------------------------------ per-read dense (kmer_main)

@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_golden_k2_byte_exact(ctx, derived_fasta, name):
    """reference test/test.sh:13-19 against the reference's own goldens"""
    import cfrk_amd
    raw = open(derived_fasta[name], "rb").read()
    reads = refsem.remainder_chunk(refsem.read_fasta_compat(raw), 8192)
    data, start, length = refsem.flatten(reads)
    rd = cfrk_amd.Read(data, length, start)
    cfrk_amd.kmer_main(rd, len(data), len(length), 2, 0)
    got = orc.format_cfrk(rd.Freq, 2)           # checker-side formatter (src/main.cu:26-62)
    want = open(os.path.join(GOLDEN, f"out-{name}.cfrk"), "rb").read()
    assert got == want


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 10])
@pytest.mark.parametrize("compat", [True, False])
def test_dense_vs_oracle(ctx, k, compat):
    import cfrk_amd
    rng = np.random.default_rng(100 + k)
    n = 300 if k <= 8 else 20
    reads = _random_reads(rng, n, 1, 400)
    reads[3] = np.zeros(0, np.int8)                       # empty read
    reads[4] = np.array([2], np.int8)                     # single base
    reads[5] = rng.integers(0, 4, 3000).astype(np.int8)   # > 1025: compat truncates at 1024 windows
    reads[6] = np.full(50, -1, np.int8)                   # all invalid
    reads[0][:k] = -1                                     # read 0 spills to Freq[-1] in compat: dropped
    data, start, length = refsem.flatten(reads)
    flags = cfrk_amd.CFRK_COMPAT if compat else 0
    got = ctx.per_read_dense(data, start, length, k, flags)
    want = orc.per_read_dense(data, start, length, k, orc.ORC_COMPAT if compat else 0)
    assert (got == want).all()


@pytest.mark.parametrize("k,nreads", [(13, 4), (14, 2), (15, 1), (12, 3)])
@pytest.mark.parametrize("compat", [True, False])
def test_dense_float_index_matches_the_reference_arithmetic(ctx, k, nreads, compat):
    """CFRK_FLOAT_INDEX: the window index accumulated through float exactly like ComputeIndex
    (/root/reference/src/kmer_kernel.cu:38) -- for k = 13..15 that is NOT the exact integer (most
    windows land in a neighbouring bin, an all-T window carries into the next row), and the HIP path
    reproduces it bin for bin against the oracle's ORC_FLOAT_INDEX emulation; for k <= 12 the flag
    changes nothing"""
    import cfrk_amd
    rng = np.random.default_rng(1300 + k)
    reads = [rng.integers(0, 4, int(L)).astype(np.int8) for L in rng.integers(200, 1500, nreads)]
    reads[0][50:50 + k + 3] = 3                           # all-T windows: the float index rounds up to 4^k
    reads[-1][-(k + 2):] = 3                              # ... in the last read too (the reference writes out of bounds there)
    reads[0][120] = -1                                    # an invalid base
    data, start, length = refsem.flatten(reads)
    flags = (cfrk_amd.CFRK_COMPAT if compat else 0) | cfrk_amd.CFRK_FLOAT_INDEX
    got = ctx.per_read_dense(data, start, length, k, flags)
    want = orc.per_read_dense(data, start, length, k, (orc.ORC_COMPAT if compat else 0) | orc.ORC_FLOAT_INDEX)
    assert (got == want).all()
    exact = orc.per_read_dense(data, start, length, k, orc.ORC_COMPAT if compat else 0)
    if k >= 13:
        assert (want != exact).any()                      # the reference really is wrong there
    else:
        assert (want == exact).all()


def test_dense_single_read_and_empty_batch(ctx):
    import cfrk_amd
    data, start, length = refsem.flatten([np.array([0, 1, 2, 3, 0, 1], np.int8)])
    got = ctx.per_read_dense(data, start, length, 2, cfrk_amd.CFRK_COMPAT)
    assert (got == orc.per_read_dense(data, start, length, 2, orc.ORC_COMPAT)).all()
    e = ctx.per_read_dense(np.zeros(0, np.int8), np.zeros(0, np.int64), np.zeros(0, np.int32), 3)
    assert e.shape == (0, 64)


def test_dense_chunk_of_8192_reads_k4(ctx):
    """the reference's default chunk (src/main.cu:235) at k=4"""
    import cfrk_amd
    rng = np.random.default_rng(7)
    reads = _random_reads(rng, 8192, 140, 160, 0.005)
    data, start, length = refsem.flatten(reads)
    got = ctx.per_read_dense(data, start, length, 4, cfrk_amd.CFRK_COMPAT)
    want = orc.per_read_dense(data, start, length, 4, orc.ORC_COMPAT)
    assert (got == want).all()


def test_dense_device_resident_variant(ctx):
    """cfrk_per_read_dense_device: same result with every buffer already on the device"""
    import ctypes as C
    import cfrk_amd
    rng = np.random.default_rng(9)
    reads = _random_reads(rng, 500, 1, 300)
    data, start, length = refsem.flatten(reads)
    for k, compat in ((3, True), (6, False), (8, True)):
        want = orc.per_read_dense(data, start, length, k, orc.ORC_COMPAT if compat else 0)
        dd, ds, dl = ctx.alloc(len(data) + 64), ctx.alloc(len(start) * 8), ctx.alloc(len(length) * 4)
        df = ctx.alloc(want.size * 4)
        ctx.h2d(dd, data); ctx.h2d(ds, start); ctx.h2d(dl, length)
        ctx.check(ctx._L.cfrk_per_read_dense_device(ctx._h, C.c_void_p(dd), C.c_void_p(ds), C.c_void_p(dl),
                                                    len(data), len(length), k,
                                                    cfrk_amd.CFRK_COMPAT if compat else 0, C.c_void_p(df)),
                  "cfrk_per_read_dense_device")
        got = np.empty(want.size, np.int32)
        ctx.d2h(got, df)
        assert (got.reshape(want.shape) == want).all()
        for p in (dd, ds, dl, df):
            ctx.free(p)


def test_dense_errors(ctx):
    import cfrk_amd
    data, start, length = refsem.flatten([np.array([0, 1, 2, 3], np.int8)])
    for k in (0, 16, -1):
        with pytest.raises(cfrk_amd.CfrkError) as e:
            ctx.per_read_dense(data, start, length, k)
        assert e.value.code == -1


# ------------------------------------------------------------------ global counting

def _cmp_global(ctx, data, k, canonical, hint=0, start=None, length=None, force_hash=False, dbg=0):
    import cfrk_amd
    flags = (cfrk_amd.CFRK_CANONICAL if canonical else 0) | (cfrk_amd.CFRK_FORCE_HASH if force_hash else 0)
    g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
    g.set_debug_flags(dbg)
    try:
        g.add(data, start, length)
        lo, hi, cnt = g.export()
    finally:
        g.set_debug_flags(0)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
    assert len(lo) == len(wlo)
    assert (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert g.digest() == orc.digest(wlo, whi, wcnt, two_word=k > 32)
    return g


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 38, 40, 41, 43, 44, 45, 46, 47, 48, 55, 62, 63, 64])
@pytest.mark.parametrize("canonical", [False, True])
@pytest.mark.parametrize("force_hash", [False, True])
def test_global_vs_oracle(ctx, k, canonical, force_hash):
    rng = np.random.default_rng(200 + k)
    reads = _random_reads(rng, 400, 1, 300)
    reads.append(np.full(200, 3, np.int8))      # poly-T: the all-ones key at k=32
    reads.append(np.full(200, 0, np.int8))      # poly-A
    reads.append(np.zeros(0, np.int8))
    data, start, length = refsem.flatten(reads)
    _cmp_global(ctx, data, k, canonical, start=start, length=length, force_hash=force_hash)


@pytest.mark.parametrize("k", [33, 34, 41, 47, 48, 55, 63, 64])
@pytest.mark.parametrize("canonical", [False, True])
def test_two_word_keys_with_the_large_leaf_tables_on_small_inputs(ctx, k, canonical):
    """jobs that announce few distinct k-mers per leaf take the small-leaf instantiation of the two-word leaf kernel (a
    1024-slot k-mer table, two workgroups per CU: round 5) -- every small test above does; CFRK_DEBUG_NO_SMALL_LEAVES keeps
    the 4096-slot instantiation covered at these sizes (a second input: deep reads of one amplicon, both ways)."""
    import cfrk_amd
    rng = np.random.default_rng(900 + k)
    reads = _random_reads(rng, 400, 1, 300)
    reads.append(np.full(200, 0, np.int8))
    data, start, length = refsem.flatten(reads)
    _cmp_global(ctx, data, k, canonical, start=start, length=length, dbg=cfrk_amd.lib.CFRK_DEBUG_NO_SMALL_LEAVES)
    amp = rng.integers(0, 4, 30_000).astype(np.int8)
    reads = [amp[i:i + 250].copy() for i in rng.integers(0, 30_000 - 250, 3000)]
    data, start, length = refsem.flatten(reads)
    _cmp_global(ctx, data, k, canonical, start=start, length=length)
    _cmp_global(ctx, data, k, canonical, start=start, length=length, dbg=cfrk_amd.lib.CFRK_DEBUG_NO_SMALL_LEAVES)
    if k in (33, 48, 63):
        # the mid-size instantiation (2048-slot table, two workgroups per CU): what a job takes that announces more than 512
        # table slots per leaf -- one key subset up to 2048 slots per leaf, two from the start beyond (C3's hint)
        _cmp_global(ctx, data, k, canonical, start=start, length=length, hint=40_000_000)
        _cmp_global(ctx, data, k, canonical, start=start, length=length, hint=100_000_000)


@pytest.mark.parametrize("canonical", [False, True])
def test_k16_on_the_partitioned_path_when_the_radix_path_is_switched_off(ctx, canonical):
    """k = 16 takes radix.hip for batches up to ~4e9 bases (round 5) and msp.hip (windows of four k-mers) beyond:
    CFRK_DEBUG_NO_RADIX16 keeps the small-input coverage of the latter"""
    import cfrk_amd
    rng = np.random.default_rng(216)
    reads = _random_reads(rng, 3000, 1, 300)
    reads.append(np.full(200, 0, np.int8))
    data, start, length = refsem.flatten(reads)
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    g = cfrk_amd.GlobalCounter(ctx, 16, flags, 1 << 20)
    g.set_debug_flags(cfrk_amd.CFRK_DEBUG_NO_RADIX16)
    try:
        g.add(data, start, length)
        lo, hi, cnt = g.export()
        assert g.msp_info()["l2_records"] > 0                 # (the partitioned path's records: it really ran)
    finally:
        g.set_debug_flags(0)
    wlo, whi, wcnt = orc.global_count(data, 16, orc.ORC_CANONICAL if canonical else 0)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    g2 = cfrk_amd.GlobalCounter(ctx, 16, flags, 1 << 20)
    g2.add(data, start, length)
    assert g2.msp_info()["l2_records"] == 0 and g2.digest() == orc.digest(wlo, whi, wcnt)   # (radix.hip: no records)


def test_global_ragged_tail_sizes(ctx):
    """buffer lengths around the 32-byte chunk / 2 KiB tile edges of the packed front end"""
    rng = np.random.default_rng(5)
    for n in (1, 15, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 2079, 2080, 4096 + 31, 70000):
        data = rng.integers(0, 4, n).astype(np.int8)
        data[rng.random(n) < 0.01] = -1
        data[-1] = -1
        for k in (1, 7, 12, 16, 22, 31, 32, 33, 64):
            _cmp_global(ctx, data, k, True)
            _cmp_global(ctx, data, k, False, force_hash=True)


def test_global_multiple_adds_accumulate(ctx):
    import cfrk_amd
    d1, _, _ = orc.synth_reads(0, 3000, 150, 20000)
    d2, _, _ = orc.synth_reads(3000, 2000, 150, 20000)
    g = cfrk_amd.GlobalCounter(ctx, 31, cfrk_amd.CFRK_CANONICAL, 50000)
    g.add(d1)
    g.add(d2)
    lo, hi, cnt = g.export()
    wlo, whi, wcnt = orc.global_count(np.concatenate([d1, d2]), 31, orc.ORC_CANONICAL)
    assert (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert int(cnt.sum()) == 5000 * 120


def test_synth_device_matches_oracle_generator(ctx):
    R, L, G = 1000, 150, 5000
    want, wstart, wlen = orc.synth_reads(17, R, L, G)
    d = ctx.alloc(R * (L + 1)); s = ctx.alloc(R * 8); l = ctx.alloc(R * 4)
    ctx.synth_reads_device(17, R, L, G, d, s, l)
    got = np.empty(R * (L + 1), np.int8); gs = np.empty(R, np.int64); gl = np.empty(R, np.int32)
    ctx.d2h(got, d); ctx.d2h(gs, s); ctx.d2h(gl, l)
    assert (got == want).all() and (gs == wstart).all() and (gl == wlen).all()
    want_u, _, _ = orc.synth_reads(0, 200, 100, 0, uniform=True)
    ctx.synth_reads_device(0, 200, 100, 0, d, uniform=True)
    got_u = np.empty(200 * 101, np.int8); ctx.d2h(got_u, d)
    assert (got_u == want_u).all()
    for p in (d, s, l):
        ctx.free(p)


def test_c2_shape_1m_reads_k15_full_equality(ctx):
    """BASELINE config 2 shape (150 bp, k=15, canonical) at 1 M reads: full (key,count) equality"""
    import cfrk_amd
    R, L, G = 1_000_000, 150, 1_000_000
    d = ctx.alloc(R * (L + 1))
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, 15, cfrk_amd.CFRK_CANONICAL, 2_000_000)
    g.add_device(d, R * (L + 1))
    lo, hi, cnt = g.export()
    host = np.empty(R * (L + 1), np.int8)
    ctx.d2h(host, d)
    wlo, _, wcnt = orc.global_count(host, 15, orc.ORC_CANONICAL, threads=8)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    ctx.free(d)


def test_c2_full_size_10m_reads_k15_entry_for_entry(ctx):
    """BASELINE configs[1] AS WRITTEN: 10 M synthetic 150 bp reads (genome 10^7), k=15, canonical,
    one GPU -- every (key, count) of the HIP path against the oracle's threaded counter (a diff,
    as /root/reference/test/test.sh:13-19 does it, not a checksum), plus the digest both ways."""
    import cfrk_amd
    R, L, G, k = 10_000_000, 150, 10_000_000, 15
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
    g.add_device(d, nN)
    lo, hi, cnt = g.export()
    gd = g.digest()
    host = np.empty(nN, np.int8)
    ctx.d2h(host, d)
    ctx.free(d)
    # the device generator is the oracle's generator (read 0 and the last read, byte for byte)
    w0, _, _ = orc.synth_reads(0, 1, L, G)
    w1, _, _ = orc.synth_reads(R - 1, 1, L, G)
    assert (host[:L + 1] == w0).all() and (host[-(L + 1):] == w1).all()
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    wlo, whi, wcnt = orc.global_count(host, k, orc.ORC_CANONICAL, threads=threads)
    assert int(wcnt.sum()) == R * (L - k + 1)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert gd == orc.digest(wlo, whi, wcnt)


def test_c3_shape_properties_k31(ctx):
    """config-3 shape (k=31, 150 bp, canonical) at 4 M reads: size-independent properties --
    sum(count) == R*(L-k+1) exactly (no invalid bases), counting reads twice doubles every
    count (linearity), distinct <= 2*Glen windows of the genome."""
    import cfrk_amd
    R, L, G, k = 4_000_000, 150, 4_000_000, 31
    nN = R * (L + 1)
    d = ctx.alloc(nN)
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
    g.add_device(d, nN)
    d1 = g.digest()
    assert d1[1] == R * (L - k + 1)
    assert d1[0] <= G - k + 1
    g.add_device(d, nN)
    d2 = g.digest()
    assert d2[0] == d1[0] and d2[1] == 2 * d1[1]
    assert d2[2] == (2 * d1[2]) % (1 << 64)
    # SURVEY 8d "parity at scale": full (key, count) equality on the first 10^6 reads
    g2 = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
    n1 = 1_000_000 * (L + 1)
    g2.add_device(d, n1)
    host = np.empty(n1, np.int8)
    ctx.d2h(host, d)
    lo, hi, cnt = g2.export()
    wlo, whi, wcnt = orc.global_count(host, k, orc.ORC_CANONICAL, threads=8)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert g2.digest() == orc.digest(wlo, whi, wcnt)
    ctx.free(d)


def test_c3_generator_prefix_1m_reads_full_equality(ctx):
    """the first 10^6 reads of configs[2]'s own generator (genome 10^8: ~1.5x coverage, nearly
    every k-mer distinct -- the opposite regime of the test above), full equality"""
    import cfrk_amd
    R, L, G, k = 1_000_000, 150, 100_000_000, 31
    nN = R * (L + 1)
    d = ctx.alloc(nN)
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, R * (L - k + 1))
    g.add_device(d, nN)
    lo, hi, cnt = g.export()
    host = np.empty(nN, np.int8)
    ctx.d2h(host, d)
    ctx.free(d)
    wlo, whi, wcnt = orc.global_count(host, k, orc.ORC_CANONICAL, threads=8)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert int(wcnt.sum()) == R * (L - k + 1)


@pytest.mark.parametrize("variant", ["high_collision", "all_distinct"])
def test_c5_shape_1m_reads_250bp_k63_full_equality(ctx, variant):
    """BASELINE configs[4]'s read shape (250 bp, k=63, two-word keys) at 10^6 reads, both stress
    variants of SURVEY 8d: Glen = 10^6 (every key hit ~190 times, both strands) and uniform random
    reads (every k-mer distinct: D = K = 1.88e8) -- full (key, count) equality with the oracle"""
    import cfrk_amd
    R, L, k = 1_000_000, 250, 63
    G = 1_000_000
    uniform = variant == "all_distinct"
    nN = R * (L + 1)
    K = R * (L - k + 1)
    d = ctx.alloc(nN)
    ctx.synth_reads_device(0, R, L, 0 if uniform else G, d, uniform=uniform)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, K if uniform else 2 * G)
    g.add_device(d, nN)
    dg = g.digest()
    assert dg[1] == K
    lo, hi, cnt = g.export()
    host = np.empty(nN, np.int8)
    ctx.d2h(host, d)
    ctx.free(d)
    del g
    wlo, whi, wcnt = orc.global_count(host, k, orc.ORC_CANONICAL, threads=8)
    del host
    assert len(lo) == len(wlo)
    assert (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert dg == orc.digest(wlo, whi, wcnt, two_word=True)
    if uniform:
        assert len(lo) > 0.999 * K
    else:
        assert len(lo) <= 2 * G and int(wcnt.max()) > 100


@pytest.mark.parametrize("k", [2, 5, 8, 10])
def test_global_equals_column_sum_of_per_read_dense_on_gpu(ctx, k):
    """Ties the benchmarked global path to kmer_main's semantics ON THE GPU: the column sum of
    cfrk_per_read_dense's native rows (the guarded ComputeFreq, src/kmer_kernel.cu:52-70) equals
    cfrk_global_* (forward strand) on ragged reads with invalid bases."""
    import cfrk_amd
    rng = np.random.default_rng(900 + k)
    reads = _random_reads(rng, 150 if k == 10 else 600, 1, 400, 0.02)
    reads.append(np.zeros(0, np.int8))
    reads.append(np.full(300, 3, np.int8))
    data, start, length = refsem.flatten(reads)
    dense = ctx.per_read_dense(data, start, length, k, 0).astype(np.int64).sum(axis=0)
    g = cfrk_amd.GlobalCounter(ctx, k, 0, 4 ** k)
    g.add(data, start, length)
    lo, hi, cnt = g.export()
    col = np.zeros(4 ** k, np.int64)
    col[lo.astype(np.int64)] = cnt.astype(np.int64)
    assert (hi == 0).all() and (col == dense).all()
    assert int(dense.sum()) > 0


def _oracle_full_digest(name):
    """tests/golden/oracle_digest_<name>_full.json: written by tools/oracle_full_digest.py from the ORACLE's
    bounded-memory count (orc_synth_digest) of the whole config"""
    import json
    rec = json.load(open(os.path.join(GOLDEN, f"oracle_digest_{name}_full.json")))
    assert rec["counter"].startswith("oracle/cfrk_oracle.c") and rec["occurrences"] == rec["expected_occurrences"]
    return rec["digest_hex"]


def _bench_line(args, timeout=600):
    """run bench.py as a child process (it starts its own ranks for --gpus N) and parse its line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_matches_the_one_gpu_digest():
    """`bench.py --gpus 2` from a bare shell (no torchrun): the parent starts the ranks itself;
    strong scaling of the same 2 M reads gives the 1-GPU digest (rehearsal: both ranks on
    cuda:0, collectives over gloo)"""
    common = ["--steps", "1", "--warmup", "0", "--reads", "2000000", "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    two = _bench_line(["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--scaling", "strong"] + common)
    assert one["sum_count_ok"] and two["sum_count_ok"]
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["digest"] == one["digest"]
    both = _bench_line(["--gpus", "2", "--same-gpu", "--dist-backend", "gloo"] + common)
    assert both["scaling"] == "strong" and both["digest"] == one["digest"]
    assert both["weak"]["config"]["reads"] == 4_000_000 and both["weak"]["sum_count_ok"]


def test_bench_rank_whose_runs_only_add_is_refused_makes_all_ranks_take_the_leaf_exchange():
    """a rank whose shard does not fit one pass cannot run a CFRK_RUNS_ONLY job (CFRK_ERR_NOMEM / CFRK_ERR_STATE
    from the add itself, not from the export): it votes "no runs" and BOTH ranks count their shards and exchange
    per-leaf lists -- nobody is left waiting in the all-to-all.  Forced on rank 1 with a 50 MB budget."""
    common = ["--steps", "1", "--warmup", "0", "--reads", "2000000", "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    two = _bench_line(["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--scaling", "strong",
                       "--debug-runs-budget", "1:50000000"] + common)
    assert two["sum_count_ok"] and two["digest"] == one["digest"]
    assert two["exchange"]["exchange"].startswith("leaf")


def test_bench_under_torchrun_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 2 ...` (RANK / WORLD_SIZE / MASTER_* come from the launcher): rank 0 prints the ONE line,
    same digest as one GPU (rehearsal: both ranks on cuda:0, collectives over gloo)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "0", "--reads", "2000000", "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--same-gpu", "--dist-backend", "gloo"] + common,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["sum_count_ok"]
    assert two["digest"] == one["digest"] and two["exchange"]["exchange"].startswith("runs")


def test_bench_two_ranks_take_the_pipelined_runs_exchange_and_the_one_shot_form_on_request():
    """strong scaling at k = 31: the default is the PIPELINED exchange (two groups of leaves, CFRK_RUNS_DEFER add,
    owner reads the lists in place); `--pipeline-groups 0` keeps the one-shot form; four groups work as well -- all
    give the 1-GPU digest (rehearsal: both ranks on cuda:0, collectives over gloo)"""
    common = ["--steps", "2", "--warmup", "1", "--reads", "2000000", "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    rehearsal = ["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--scaling", "strong"]
    two = _bench_line(rehearsal + common)
    assert two["sum_count_ok"] and two["digest"] == one["digest"]
    assert two["exchange"]["exchange"] == "runs, pipelined in 2 groups" and two["exchange"]["wire_bytes"] > 0
    classic = _bench_line(rehearsal + ["--pipeline-groups", "0"] + common)
    assert classic["digest"] == one["digest"] and classic["exchange"]["exchange"] == "runs"
    four = _bench_line(rehearsal + ["--pipeline-groups", "4"] + common)
    assert four["digest"] == one["digest"] and four["exchange"]["exchange"] == "runs, pipelined in 4 groups"
    # the wire carries the same runs either way (the pipelined segments have a few header rows more)
    assert abs(four["exchange"]["wire_bytes"] - classic["exchange"]["wire_bytes"]) < 0.05 * classic["exchange"]["wire_bytes"]


def test_bench_two_ranks_exchange_runs_of_two_word_keys():
    """the same rehearsal at k = 63: strong scaling takes the runs exchange of msp2.hip -- pipelined since round 5,
    the one-shot form of round 3 on request -- and gives the 1-GPU digest"""
    common = ["--steps", "1", "--warmup", "0", "--reads", "2000000", "--k", "63", "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    rehearsal = ["--gpus", "2", "--same-gpu", "--dist-backend", "gloo", "--scaling", "strong"]
    two = _bench_line(rehearsal + common)
    assert one["sum_count_ok"] and two["sum_count_ok"] and two["digest"] == one["digest"]
    assert two["exchange"]["exchange"] == "runs, pipelined in 2 groups" and two["exchange"]["wire_bytes"] > 0
    classic = _bench_line(rehearsal + ["--pipeline-groups", "0"] + common)
    assert classic["digest"] == one["digest"] and classic["exchange"]["exchange"] == "runs"


@pytest.mark.parametrize("k", [31, 63, 15])
def test_bench_one_rank_process_group_over_rccl_executes_every_exchange(k):
    """What a one-GPU box can execute of RCCL: `bench.py --gpus 1 --dist-at-one --dist-backend nccl` -- a ONE-rank
    process group (backend "nccl" IS RCCL) and a step that goes through the exchange all the same (the rank sends
    everything to itself and owns every leaf): init_process_group with device_id, the size all-to-all, the payload
    `dist.all_to_all` on segment views of the send buffer (pipelined form), the ragged `all_to_all_single` of the
    one-shot, leaf and key forms, `all_reduce` / `all_gather` of the timing and the digests, barrier, comm-stream
    events -- all on device tensors, none staged through the host.  The digest equals the plain one-GPU run's."""
    import subprocess
    import sys
    # (RCCL itself must be able to start here -- a one-rank all_reduce in a process of its own; a box where it cannot is
    #  no evidence about this repo's calls)
    probe = ("import os, torch, torch.distributed as d; os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29537', RANK='0', "
             "WORLD_SIZE='1'); torch.cuda.set_device(0); d.init_process_group('nccl', device_id=torch.device('cuda', 0)); "
             "t = torch.ones(4, device='cuda'); d.all_reduce(t); torch.cuda.synchronize(); d.destroy_process_group(); print('ok')")
    pr = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300)
    if pr.returncode != 0 or "ok" not in pr.stdout:
        pytest.skip("RCCL does not start on this box: " + pr.stderr[-300:])
    common = ["--steps", "2", "--warmup", "1", "--reads", "2000000", "--k", str(k), "--cpu-reads", "0"]
    one = _bench_line(["--gpus", "1"] + common)
    rccl = ["--gpus", "1", "--dist-at-one", "--dist-backend", "nccl"]
    forms = [([], "runs, pipelined in 2 groups"), (["--pipeline-groups", "0"], "runs"), (["--exchange", "leaf"], "leaf"),
             (["--exchange", "owner"], "owner")] if k >= 16 else [([], "leaf"), (["--exchange", "owner"], "owner")]
    for extra, name in forms:
        got = _bench_line(rccl + extra + common)
        assert got["sum_count_ok"] and got["digest"] == one["digest"], (name, got.get("exchange"))
        assert got["exchange"]["exchange"] == name, got["exchange"]


def test_bench_two_ranks_over_rccl_on_two_gpus():
    """`bench.py --gpus 2` on two REAL devices over RCCL (backend "nccl"): skipped on the one-GPU boxes the
    suite usually runs on -- the collectives of cfrk_amd/sharded.py are otherwise only exercised over gloo"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    for k in ("31", "63"):
        common = ["--steps", "1", "--warmup", "0", "--reads", "2000000", "--k", k, "--cpu-reads", "0"]
        one = _bench_line(["--gpus", "1"] + common)
        two = _bench_line(["--gpus", "2", "--dist-backend", "nccl", "--scaling", "strong"] + common)
        assert one["sum_count_ok"] and two["sum_count_ok"] and two["digest"] == one["digest"]


@pytest.mark.parametrize("k", [15, 21, 31, 63])
def test_c5_high_collision_variant_small_genome(ctx, k):
    """SURVEY 8d's high-collision variant at reduced size: a tiny genome, so every key is hit
    tens of thousands of times (one hot slot per key, every leaf table nearly empty) -- full
    export compared with the oracle, and the partitioned path against the HBM-table path"""
    import cfrk_amd
    R, L, G = 300_000, 150, 1500
    data, _, _ = orc.synth_reads(0, R, L, G)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL, threads=8 if k <= 32 else 0)
    assert int(wcnt.max()) > 10_000
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 100_000)
    g.add(data)
    lo, hi, cnt = g.export()
    assert (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()
    gh = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 100_000)
    gh.add(data)
    assert gh.digest() == g.digest() == orc.digest(wlo, whi, wcnt, two_word=k > 32)


@pytest.mark.parametrize("k", [33, 63])
def test_two_word_single_amplicon_overflows_its_level1_regions_and_is_laid_out_again(ctx, k):
    """300 k copies of ONE 150-base read: every record lands in one of ~9 level-1 bins, twice what
    their regions hold.  The first partition kernel's cursors keep counting, the host lays the level
    out again with the exact sizes and reruns it (as msp.hip does for 16 <= k <= 32): nothing is
    counted through the HBM table, and the counts are the oracle's."""
    import cfrk_amd
    rng = np.random.default_rng(5)
    read = np.append(rng.integers(0, 4, 150).astype(np.int8), np.int8(-1))
    R = 300_000
    data = np.tile(read, R)
    want = orc.global_count(data, k, orc.ORC_CANONICAL)
    assert len(want[0]) == 150 - k + 1 and int(want[2].min()) == R
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 100_000)
    g.add(data)
    lo, hi, cnt = g.export()
    assert (lo == want[0]).all() and (hi == want[1]).all() and (cnt.astype(np.uint64) == want[2]).all()
    info = g.msp_info()
    assert info["spilled_records"] == 0 and info["spilled_kmers"] == 0


@pytest.mark.parametrize("k", [9, 13, 14, 15, 16])
def test_radix_single_amplicon_overflows_its_level1_regions_and_is_laid_out_again(ctx, k):
    """the same for 8 <= k <= 16 (radix.hip): 800 000 copies of one read put every key into ~140 of
    the first level's regions' bins, more than their fixed stride holds; the level is laid out again
    with exact sizes instead of counting the excess through the HBM table.  For k >= 13 the batch is small
    enough for the host to choose leaves of packed 16-bit counters (idx = 14; 15 at k = 16), and every leaf
    that holds a key then has 800 000 elements or more -- above 2^16: the leaf kernel's two-pass mode with
    32-bit counters (mode 3; ADVICE r4 asked for k = 13 / 14 cases that reach it)"""
    import cfrk_amd
    rng = np.random.default_rng(6)
    read = np.append(rng.integers(0, 4, 150).astype(np.int8), np.int8(-1))
    R = 800_000
    data = np.tile(read, R)
    want = orc.global_count(data, k, orc.ORC_CANONICAL, threads=8)
    assert int(want[2].min()) >= R
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 100_000)
    g.add(data)
    lo, hi, cnt = g.export()
    assert (lo == want[0]).all() and (cnt.astype(np.uint64) == want[2]).all()
    gh = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 100_000)
    gh.add(data)
    assert gh.digest() == g.digest()


def test_export_partition_and_merge_roundtrip(ctx):
    """key-owner partitioned export, then count-add merge into another table"""
    import cfrk_amd
    data, _, _ = orc.synth_reads(0, 5000, 150, 30000)
    for k in (31, 47):
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 100000)
        g.add(data)
        want = g.digest()
        n = want[0]
        lo, hi, cn = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 4)
        parts = g.export_device(lo, hi, cn, n, parts=3)
        assert sum(parts) == n and min(parts) > 0
        klo = np.empty(n, np.uint64); ctx.d2h(klo, lo)
        ctx2 = cfrk_amd.Context(0)
        g2 = cfrk_amd.GlobalCounter(ctx2, k, cfrk_amd.CFRK_CANONICAL, 100000)
        off = 0
        for p in parts:                      # merge segment by segment, twice the first one
            g2.merge_device(lo + off * 8, hi + off * 8, cn + off * 4, p)
            off += p
        assert g2.digest() == want
        g2.merge_device(lo, hi, cn, n)
        d2 = g2.digest()
        assert d2[0] == n and d2[1] == 2 * want[1]
        ctx2.close()
        for p in (lo, hi, cn):
            ctx.free(p)


@pytest.mark.parametrize("k,shared", [(31, False), (63, False), (63, True), (40, True)])
def test_leaf_export_and_lds_merge_two_emulated_ranks(ctx, k, shared):
    """SURVEY 8e with leaf owners: two shards counted on their own contexts, per-leaf export with
    owner(leaf) = leaf % 2, then each owner adds both incoming lists of its leaves in LDS; the
    union of the two owners' results equals the single-context result.  k = 63: two-word keys;
    shared: the ranks' leaves are shared by record (their lists sit in one segment per sub-value)."""
    import cfrk_amd
    R, L, G = 40000, 150, 200000
    two = k > 32
    whole, _, _ = orc.synth_reads(0, R, L, G)
    g0 = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
    g0.add(whole)
    want = g0.digest()
    ranks = [cfrk_amd.Context(0), cfrk_amd.Context(0)]
    parts = 2
    exports = []
    for r, c in enumerate(ranks):
        data, _, _ = orc.synth_reads(r * R // 2, R // 2, L, G)
        g = cfrk_amd.GlobalCounter(c, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
        if shared:
            g.set_debug_flags(cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS)
        g.add(data)
        g.set_debug_flags(0)
        n = g.finish()
        lpp = g.leaves_per_part(parts)
        dk, dh, dc, dl = c.alloc(n * 8 + 8), c.alloc(n * 8 + 8), c.alloc(n * 4 + 4), c.alloc(parts * lpp * 4)
        pc = g.export_leaves_device(dk, dc, n, parts, dl, dh if two else 0)
        assert sum(pc) == n
        keys = np.empty(n, np.uint64); hi = np.zeros(n, np.uint64); cnt = np.empty(n, np.uint32)
        lc = np.empty(parts * lpp, np.uint32)
        c.d2h(keys, dk); c.d2h(cnt, dc); c.d2h(lc, dl)
        if two:
            c.d2h(hi, dh)
        assert int(lc[:lpp].sum()) == pc[0] and int(lc[lpp:].sum()) == pc[1]
        exports.append((keys, hi, cnt, lc, pc, lpp))
    total = [0, 0, 0, 0]
    for owner in range(parts):                      # what the all-to-all would deliver to `owner`
        rk, rh, rc_, rl, rn = [], [], [], [], []
        for keys, hi, cnt, lc, pc, lpp in exports:
            o = sum(pc[:owner])
            rk.append(keys[o:o + pc[owner]]); rh.append(hi[o:o + pc[owner]]); rc_.append(cnt[o:o + pc[owner]])
            rl.append(lc[owner * lpp:(owner + 1) * lpp]); rn.append(pc[owner])
        rk, rh, rc_, rl = np.concatenate(rk), np.concatenate(rh), np.concatenate(rc_), np.concatenate(rl)
        oc = cfrk_amd.Context(0)
        og = cfrk_amd.GlobalCounter(oc, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
        dk, dh, dc, dl = (oc.alloc(len(rk) * 8 + 8), oc.alloc(len(rk) * 8 + 8), oc.alloc(len(rk) * 4 + 4),
                          oc.alloc(len(rl) * 4))
        oc.h2d(dk, rk); oc.h2d(dh, rh); oc.h2d(dc, rc_); oc.h2d(dl, rl)
        og.merge_leaves_device(dk, dc, rn, dl, dh if two else 0)
        d = og.digest()
        total = [(total[0] + d[0]) & (2**64 - 1), (total[1] + d[1]) & (2**64 - 1),
                 (total[2] + d[2]) & (2**64 - 1), total[3] ^ d[3]]
        oc.close()
    assert tuple(total) == want
    # a job that spilled / is not in leaf form refuses the leaf export
    gh = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 2 * G)
    gh.add(whole)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        gh.export_leaves_device(0, 0, 0, 2, 0)
    assert e.value.code == -4
    if shared:
        # the same context, now a one-word job: its list has one segment per leaf again
        c = ranks[0]
        data, _, _ = orc.synth_reads(0, R // 2, L, G)
        g = cfrk_amd.GlobalCounter(c, 31, cfrk_amd.CFRK_CANONICAL, 2 * G)
        g.add(data)
        n = g.finish()
        lpp = g.leaves_per_part(parts)
        dk, dc, dl = c.alloc(n * 8 + 8), c.alloc(n * 4 + 4), c.alloc(parts * lpp * 4)
        pc = g.export_leaves_device(dk, dc, n, parts, dl, 0)
        cnt = np.empty(n, np.uint32)
        c.d2h(cnt, dc)
        assert sum(pc) == n and int(cnt.astype(np.uint64).sum()) == g.digest()[1]
    for c in ranks:
        c.close()


def test_global_errors(ctx):
    import cfrk_amd
    with pytest.raises(cfrk_amd.CfrkError) as e:
        cfrk_amd.GlobalCounter(ctx, 65)
    assert e.value.code == -1
    data, start, length = refsem.flatten([np.array([0, 1, 2, 3], np.int8), np.array([1, 1], np.int8)])
    g = cfrk_amd.GlobalCounter(ctx, 3)
    bad = start.copy(); bad[1] += 1
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.add(data, bad, length)
    assert e.value.code == -5
    noterm = data.copy(); noterm[4] = 0
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.add(noterm, start, length)
    assert e.value.code == -5
    # a corrupt table whose entries are consistent with each other but negative (the threaded
    # layout check compares a piece's first read with a predecessor another thread validates,
    # nS > 2^20 makes several pieces) is refused before any byte of data[] is looked at
    nS = (1 << 20) + 4096
    ln = np.full(nS, 3, np.int32)
    st = np.arange(nS, dtype=np.int64) * 4
    dat = np.tile(np.array([0, 1, 2, -1], np.int8), nS)
    cut = nS // 2
    st2 = st.copy(); st2[cut - 1:] -= (1 << 40)           # start[cut-1] < 0 and start[cut] matches it
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.add(dat, st2, ln)
    assert e.value.code == -5
    ln2 = ln.copy(); ln2[cut] = -7
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.add(dat, st, ln2)
    assert e.value.code == -5
    # table overflow is reported, not silently wrong
    rnd = np.random.default_rng(0).integers(0, 4, 200000).astype(np.int8)
    g = cfrk_amd.GlobalCounter(ctx, 31, cfrk_amd.CFRK_FORCE_HASH, 64)
    g.add(rnd)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.finish()
    assert e.value.code == -6


# ------------------------------------------------------------------ minimizer-partitioned path

@pytest.mark.parametrize("k", [16, 19, 24, 27, 31, 32])
@pytest.mark.parametrize("canonical", [False, True])
def test_pipelined_partition_matches_oracle(ctx, k, canonical):
    """the chunked pipeline (partition kernel, then second-level kernel, chunk after chunk on one
    level-1 buffer) forced onto a small input: every (key, count) against the oracle, with ragged
    reads, invalid bases and low-complexity stretches crossing the chunk boundaries"""
    import cfrk_amd
    rng = np.random.default_rng(900 + k)
    reads = _random_reads(rng, 3000, 1, 400, 0.01)
    reads.append(np.zeros(3000, np.int8))                    # poly-A across several tiles
    reads.append(np.tile(np.array([0, 1, 2, 3, 3, 2], np.int8), 800))
    genome = rng.integers(0, 4, 20000).astype(np.int8)       # deep coverage: complete runs repeat
    for _ in range(2000):
        a = int(rng.integers(0, len(genome) - 150))
        reads.append(genome[a:a + 150])
    data, start, length = refsem.flatten(reads)
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    g = cfrk_amd.GlobalCounter(ctx, k, flags, 0)
    g.set_debug_flags(cfrk_amd.CFRK_DEBUG_SMALL_PIPELINE)
    g.add(data, start, length)
    lo, hi, cnt = g.export()
    g.set_debug_flags(0)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
    assert len(lo) == len(wlo)
    assert (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    assert g.digest() == orc.digest(wlo, whi, wcnt)


@pytest.mark.parametrize("k", [33, 40, 47, 63, 64])
@pytest.mark.parametrize("canonical", [False, True])
def test_chunked_two_word_path_matches_oracle(ctx, k, canonical):
    """the two-word path (33 <= k <= 64) counted in chunks with leaf streams sized from the first chunk
    (forced onto a small input): every (key, count) against the oracle"""
    import cfrk_amd
    rng = np.random.default_rng(950 + k)
    reads = _random_reads(rng, 2500, 1, 500, 0.01)
    reads.append(np.zeros(3000, np.int8))
    genome = rng.integers(0, 4, 20000).astype(np.int8)
    for _ in range(2000):
        a = int(rng.integers(0, len(genome) - 250))
        reads.append(genome[a:a + 250])
    data, start, length = refsem.flatten(reads)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL if canonical else 0, 0)
    g.set_debug_flags(cfrk_amd.CFRK_DEBUG_SMALL_PIPELINE)
    g.add(data, start, length)
    lo, hi, cnt = g.export()
    g.set_debug_flags(0)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
    assert len(lo) == len(wlo)
    assert (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()


def test_pipelined_and_one_chunk_paths_agree_at_20m_reads(ctx):
    """a batch large enough to pipeline by itself (20 M x 150 bp, k = 31 and 25): digest equal to the
    one-chunk path's (CFRK_DEBUG_NO_PIPELINE) and sum(count) exact"""
    import cfrk_amd
    R, L, G = 20_000_000, 150, 20_000_000
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(0, R, L, G, d)
    for k in (31, 25):
        got = []
        for dbg in (0, cfrk_amd.CFRK_DEBUG_NO_PIPELINE):
            g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, G + 1024)
            g.set_debug_flags(dbg)
            g.add_device(d, nN)
            got.append(g.digest())
            g.set_debug_flags(0)
        assert got[0] == got[1]
        assert got[0][1] == R * (L - k + 1)
    ctx.free(d)


def test_c3_full_size_result_keeps_its_leaf_index_and_parks_nothing(ctx):
    """BASELINE configs[2] at full size (100 M x 150 bp, k = 31): the result is the known digest, no record was
    parked or spilled (the leaf streams sized from the first chunk hold every leaf), and every leaf has ONE
    segment in the result list, so the export by leaf -- what a multi-GPU job of this size takes -- works
    (until round 3 an unlucky probe chain split a handful of the 65 536 leaves by key and the export refused)"""
    import cfrk_amd
    R, L, k, G = 100_000_000, 150, 31, 100_000_000
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, G + 1024)
    g.add_device(d, nN)
    info = g.msp_info()
    assert info["spilled_records"] == 0 and info["spilled_kmers"] == 0 and info["list_entries"] == 99_999_970
    assert g.last_add_passes() == 1
    world = 8
    lpp = g.leaves_per_part(world)
    cap = G + 1024
    keys, cnt, lc = ctx.alloc(cap * 8), ctx.alloc(cap * 4), ctx.alloc(world * lpp * 4)
    pc = g.export_leaves_device(keys, cnt, cap, world, lc, 0)
    assert sum(pc) == 99_999_970 and min(pc) > 0.12 * 99_999_970
    dg = g.digest()
    assert dg[0] == 99_999_970 and dg[1] == R * (L - k + 1)
    # the digest of the ORACLE's count of the same 10^8 reads (tools/oracle_full_digest.py c3, run once on the GPU
    # box's host cores; BASELINE.md section 4: C3 - C5 are gated on oracle digest equality, as the reference's own
    # test diffs against an independent result, test/test.sh:13-19) -- not a digest this GPU path produced
    assert [f"{x:016x}" for x in dg] == _oracle_full_digest("c3")
    for p in (keys, cnt, lc, d):
        ctx.free(p)


def test_c5_shard_full_size_against_the_oracle_digest(ctx):
    """one GPU's share of BASELINE configs[4] at full size (125 M x 250 bp, k = 63, genome 10^9: ~10^9 distinct
    two-word keys, leaves shared by record, counted in chunks): digest equal to the ORACLE's count of the same
    reads (tools/oracle_full_digest.py c5shard), one pass, nothing spilled"""
    import cfrk_amd
    R, L, k, G = 125_000_000, 250, 63, 1_000_000_000
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(0, R, L, G, d)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, min(G, R * (L - k + 1)) + 1024)
    g.add_device(d, nN)
    dg = g.digest()
    info = g.msp_info()
    assert g.last_add_passes() == 1 and info["spilled_records"] == 0 and info["spilled_kmers"] == 0
    assert dg[1] == R * (L - k + 1)
    assert [f"{x:016x}" for x in dg] == _oracle_full_digest("c5shard")
    del g
    ctx.free(d)


def test_chunked_batch_sizes_leaf_streams_from_first_chunk_and_survives_a_lumpy_batch(ctx):
    """a batch large enough to be counted in chunks by itself (4 M x 150 bp): (a) uniform coverage of a
    4 Mb genome -- the leaf streams are sized from the first chunk's records; (b) the same reads from
    a 30 kb genome -- a few thousand leaves take everything, the measured sizes are far too small, the
    batch starts over on the one-chunk path with an exact layout.  Digest equal to the HBM-table path."""
    import cfrk_amd
    R, L, k = 4_000_000, 150, 31
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    for G in (4_000_000, 30_000):
        ctx.synth_reads_device(0, R, L, G, d)
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
        g.add_device(d, nN)
        got = g.digest()
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 2 * G)
        g.add_device(d, nN)
        assert got == g.digest()
        assert got[1] == R * (L - k + 1)
    ctx.free(d)


def test_chunked_batch_whose_first_chunk_is_no_measure_of_the_rest(ctx):
    """the leaf streams of a chunked batch are sized from its FIRST chunk.  Here the first quarter of
    a 4 M-read batch is invalid bytes only (an N-rich prefix: no records at all), so the streams are sized for
    nothing and every leaf overflows in the later chunks.  The cursors count on over all chunks, the streams are
    laid out exactly -- in a buffer that grows to what the batch really holds -- and the chunks run again:
    same digest as the HBM-table path, exact sum, nothing counted through the spill path."""
    import cfrk_amd
    R, L, k, G = 4_000_000, 150, 31, 4_000_000
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(0, R, L, G, d)
    bad = R // 4
    ctx.h2d(d, np.full(bad * (L + 1), -1, np.int8))
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 2 * G)
    g.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP_CHUNKS, 4)
    try:
        g.add_device(d, nN)
        got = g.digest()
        info = g.msp_info()
    finally:
        g.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP_CHUNKS, 0)
    assert info["spilled_records"] == 0 and info["spilled_kmers"] == 0
    gh = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 2 * G)
    gh.add_device(d, nN)
    assert got == gh.digest()
    assert got[1] == (R - bad) * (L - k + 1)
    ctx.free(d)


def test_pipelined_overflow_falls_back_to_exact_layout(ctx):
    """deep coverage of a tiny genome through the pipeline: the leaf streams overflow by far more
    than the parking buffer takes, the batch starts over on the one-chunk path and is laid out exactly"""
    import cfrk_amd
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 1500).astype(np.int8)
    reads = []
    for _ in range(30000):
        a = int(rng.integers(0, len(genome) - 120))
        reads.append(genome[a:a + 120])
    data, start, length = refsem.flatten(reads)
    g = cfrk_amd.GlobalCounter(ctx, 31, cfrk_amd.CFRK_CANONICAL, 0)
    g.set_debug_flags(cfrk_amd.CFRK_DEBUG_SMALL_PIPELINE)
    g.add(data, start, length)
    lo, hi, cnt = g.export()
    g.set_debug_flags(0)
    wlo, whi, wcnt = orc.global_count(data, 31, orc.ORC_CANONICAL)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()


def test_msp_low_complexity_and_many_invalid(ctx):
    """homopolymers / short tandem repeats (runs longer than one record, one minimizer for
    thousands of k-mers) and reads riddled with N"""
    rng = np.random.default_rng(11)
    reads = [np.zeros(5000, np.int8), np.full(5000, 3, np.int8),
             np.tile(np.array([0, 1], np.int8), 3000), np.tile(np.array([0, 1, 2, 3, 3, 2], np.int8), 1500)]
    reads += _random_reads(rng, 200, 20, 400, 0.05)
    data, _, _ = refsem.flatten(reads)
    for k in (3, 9, 15, 16, 18, 21, 24, 27, 31, 32, 33, 50, 63):
        for canonical in (False, True):
            _cmp_global(ctx, data, k, canonical)


def test_msp_leaf_table_overflow_spills_to_hbm_table(ctx):
    """all-distinct input with more distinct k-mers per leaf (5500) than an LDS table holds: the
    leaf is counted in several passes over subsets of its key space (earlier: the excess went to
    the HBM table) and the result is exact"""
    import cfrk_amd
    data, _, _ = orc.synth_reads(0, 3_000_000, 150, 0, uniform=True)   # 3.6e8 k-mers, ~all distinct
    g = cfrk_amd.GlobalCounter(ctx, 31, cfrk_amd.CFRK_CANONICAL, 400_000_000)
    g.add(data)
    assert g.msp_info()["spilled_kmers"] == 0
    d = g.digest()
    assert d[1] == 3_000_000 * 120
    g2 = cfrk_amd.GlobalCounter(ctx, 31, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 400_000_000)
    g2.add(data)
    assert g2.digest() == d


@pytest.mark.parametrize("k", [28, 31, 32])
def test_msp_record_table_overflow_second_chance(ctx, k):
    """k >= 28 deduplicates a leaf's complete runs in a 1024-slot LDS table; a leaf with more
    distinct runs takes a second pass with a table over the whole LDS pool, whose entries replace
    the head of the leaf's stream in HBM.  The debug switch sends EVERY leaf down that path
    (k < 28 always deduplicates that way: covered by every other test of those k)."""
    import cfrk_amd
    rng = np.random.default_rng(900 + k)
    genome = rng.integers(0, 4, 30_000).astype(np.int8)
    reads = []
    for _ in range(6000):                                   # ~30x coverage, both strands
        p = int(rng.integers(0, len(genome) - 150))
        r = genome[p:p + 150].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.int8)
        if rng.random() < 0.1:
            r[int(rng.integers(0, 150))] = -1
        reads.append(r)
    data, _, _ = refsem.flatten(reads)
    for canonical in (False, True):
        flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
        g = cfrk_amd.GlobalCounter(ctx, k, flags, 0)
        g.set_debug_flags(cfrk_amd.CFRK_DEBUG_FORCE_RT_OVERFLOW)
        try:
            g.add(data)
        finally:
            g.set_debug_flags(0)
        assert g.msp_info()["spilled_kmers"] == 0
        lo, hi, cnt = g.export()
        wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
        assert len(lo) == len(wlo) and (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()


@pytest.mark.parametrize("k", [21, 24, 27, 28, 31, 32, 33, 48, 63])
def test_partition_kernel_direct_append_path(ctx, k):
    """the first partition kernel keeps a wave's runs in registers (4..8 trips of 64); runs beyond
    that are built and appended one by one.  Real inputs never get there; the debug switch cuts
    the capacity to one trip so that most records of this input do."""
    import cfrk_amd
    rng = np.random.default_rng(1200 + k)
    reads = _random_reads(rng, 3000, 20, 400, 0.01)
    data, _, _ = refsem.flatten(reads)
    for canonical in (False, True):
        flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
        g = cfrk_amd.GlobalCounter(ctx, k, flags, 0)
        g.set_debug_flags(cfrk_amd.CFRK_DEBUG_SMALL_WAVE_CAP)
        try:
            g.add(data)
        finally:
            g.set_debug_flags(0)
        lo, hi, cnt = g.export()
        wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
        assert len(lo) == len(wlo) and (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()


def test_msp_then_merge_and_second_add(ctx):
    """the leaf-output list is folded into the table when a later add / merge needs it"""
    import cfrk_amd
    d1, _, _ = orc.synth_reads(0, 4000, 150, 30000)
    d2, _, _ = orc.synth_reads(4000, 4000, 150, 30000)
    g = cfrk_amd.GlobalCounter(ctx, 25, cfrk_amd.CFRK_CANONICAL, 100000)
    g.add(d1)
    a = g.digest()
    g.add(d2)
    lo, hi, cnt = g.export()
    wlo, whi, wcnt = orc.global_count(np.concatenate([d1, d2]), 25, orc.ORC_CANONICAL)
    assert (lo == wlo).all() and (cnt.astype(np.uint64) == wcnt).all()
    w1 = orc.global_count(d1, 25, orc.ORC_CANONICAL)
    assert a == orc.digest(*w1)


@pytest.mark.parametrize("k", [31, 21, 63, 40])
def test_partitioned_path_counts_in_passes_when_memory_is_short(k):
    """a batch whose record buffers exceed the memory budget is counted in several passes over
    ranges of the input (the passes' per-leaf lists are added in LDS at the end): same result as
    one pass.  The budget is lowered step by step until the library needs more than one pass
    (its fixed buffers depend on tuning constants this test should not know)."""
    import cfrk_amd
    c = cfrk_amd.Context(0)
    data, _, _ = orc.synth_reads(0, 200_000, 150, 300_000)
    d_data = c.alloc(len(data))
    c.h2d(d_data, data)
    g = cfrk_amd.GlobalCounter(c, k, cfrk_amd.CFRK_CANONICAL, 400_000)
    g.add_device(d_data, len(data))
    assert g.last_add_passes() == 1
    one = g.digest()
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL)
    assert one == orc.digest(wlo, whi, wcnt, two_word=k > 32)
    budget, seen = 6 << 30, 1
    while True:
        g = cfrk_amd.GlobalCounter(c, k, cfrk_amd.CFRK_CANONICAL, 400_000)
        g.set_mem_budget(budget)
        g.add_device(d_data, len(data))
        seen = g.last_add_passes()
        if seen != 1:
            break
        # (still one pass: as the budget shrinks, possibly with the leaf streams sized from a counting pass)
        assert g.digest() == one
        budget = budget * 15 // 16
    assert seen >= 2, "the budget fell below one minimal pass without a multi-pass add in between"
    lo, hi, cnt = g.export()
    g.set_mem_budget(0)
    assert (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()
    # a budget below one minimal pass: the general HBM-table path takes over, still exact
    g = cfrk_amd.GlobalCounter(c, k, cfrk_amd.CFRK_CANONICAL, 400_000)
    g.set_mem_budget(1 << 20)
    g.add_device(d_data, len(data))
    assert g.last_add_passes() == 0 and g.digest() == one
    g.set_mem_budget(0)
    c.free(d_data)
    c.close()


# ------------------------------------------------------------------ the cfrk command

def _cli():
    import subprocess
    from .conftest import ROOT
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cfrk_amd", "host")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "cfrk_amd", "cfrk")


@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_cli_reproduces_reference_test(derived_fasta, tmp_path, name):
    """reference test/test.sh:13-19: `cfrk seqN.fasta out.cfrk 2 12 8192`, diff against the golden"""
    import subprocess
    out = tmp_path / "out.cfrk"
    subprocess.check_call([_cli(), derived_fasta[name], str(out), "2", "12", "8192"])
    assert out.read_bytes() == open(os.path.join(GOLDEN, f"out-{name}.cfrk"), "rb").read()


def test_cli_chunking_quirks_and_modes(tmp_path):
    import subprocess
    cli = _cli()
    fa = tmp_path / "r.fasta"
    raw = b"".join(b">r%d\n%s\n" % (i, s) for i, s in enumerate([b"ACGTACGT", b"TTTTGGGG", b"ACNNAC", b"GGGGGGGG", b"CATCATCAT"]))
    fa.write_bytes(raw)
    out = tmp_path / "o.cfrk"
    # only the remainder chunk reaches the file (src/main.cu:303-305)
    subprocess.check_call([cli, str(fa), str(out), "3", "12", "2"])
    assert out.read_bytes() == refsem.reference_cfrk_bytes(raw, 3, 2)
    # reads % chunkSize == 0 -> empty file
    subprocess.check_call([cli, str(fa), str(out), "3", "12", "5"])
    assert out.read_bytes() == b""
    # argc == 5: the 4th positional is nt, chunk stays 8192
    subprocess.check_call([cli, str(fa), str(out), "3", "2"])
    assert out.read_bytes() == refsem.reference_cfrk_bytes(raw, 3, 8192)
    # --all-chunks: every chunk, each counted on its own (spill stays inside a chunk)
    subprocess.check_call([cli, str(fa), str(out), "2", "12", "2", "--all-chunks"])
    reads = refsem.read_fasta_compat(raw)
    rows = []
    for c in range(0, 5, 2):
        d, s, l = refsem.flatten(reads[c:c + 2])
        rows.append(orc.format_cfrk(orc.per_read_dense(d, s, l, 2, orc.ORC_COMPAT), 2))
    assert out.read_bytes() == b"\n".join(rows)
    # --global --canonical: sparse "key:count" lines
    subprocess.check_call([cli, str(fa), str(out), "5", "--global", "--canonical"])
    data, _, _ = refsem.flatten([refsem._CODE[np.frombuffer(s, np.uint8)] for s in
                                 [b"ACGTACGT", b"TTTTGGGG", b"ACNNAC", b"GGGGGGGG", b"CATCATCAT"]])
    lo, _, cnt = orc.global_count(data, 5, orc.ORC_CANONICAL)
    assert out.read_bytes() == b"".join(b"%d:%d\n" % (int(a), int(b)) for a, b in zip(lo, cnt))
    # usage
    r = subprocess.run([cli, str(fa)], capture_output=True)
    assert r.returncode == 1 and r.stdout.startswith(b"Usage: ./cfrk")


def _read_glb1(raw):
    """parse the CFRKGLB1 binary form (cfrk_host.h) -> k, flags, lo, hi, counts"""
    assert raw[:8] == b"CFRKGLB1"
    k, flags = np.frombuffer(raw, "<u4", 2, 8)
    n, total = np.frombuffer(raw, "<u8", 2, 16)
    two = bool(flags & 2)
    dt = np.dtype([("hi", "<u8"), ("lo", "<u8"), ("c", "<u4")] if two else [("lo", "<u8"), ("c", "<u4")])
    assert dt.itemsize == (20 if two else 12) and len(raw) == 32 + int(n) * dt.itemsize
    rec = np.frombuffer(raw, dt, int(n), 32)
    assert int(rec["c"].astype(np.uint64).sum()) == int(total)
    return int(k), int(flags), rec["lo"].copy(), (rec["hi"].copy() if two else np.zeros(int(n), np.uint64)), rec["c"].copy()


@pytest.mark.parametrize("k", [15, 31, 33, 47, 63, 64])
@pytest.mark.parametrize("canonical", [False, True])
def test_cli_global_sparse_text_and_binary_for_every_k(ctx, tmp_path, k, canonical):
    """`cfrk in.fasta out k --global [--canonical] [--binary]` for one- and two-word keys: the text
    ("key:count" / "hi:lo:count") and the binary form both hold exactly what cfrk_global_export
    returns for the same reads -- and what the oracle counts"""
    import subprocess
    import cfrk_amd
    cli = _cli()
    rng = np.random.default_rng(300 + k)
    genome = rng.integers(0, 4, 6000)
    seqs = []
    for _ in range(1500):
        L = int(rng.integers(10, 220))
        a = int(rng.integers(0, len(genome) - L))
        r = genome[a:a + L]
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        sq = "".join("ACGT"[c] for c in r)
        if L > 40 and rng.random() < 0.1:
            sq = sq[:L // 3] + "N" + sq[L // 3 + 1:]
        seqs.append(sq)
    fa = tmp_path / "g.fasta"
    fa.write_text("".join(f">r{i}\n{s}\n" for i, s in enumerate(seqs)))
    reads = [refsem._CODE[np.frombuffer(s.encode(), np.uint8)] for s in seqs]
    data, start, length = refsem.flatten(reads)
    g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL if canonical else 0, 0)
    g.add(data, start, length)
    lo, hi, cnt = g.export()
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
    assert len(lo) == len(wlo) and (lo == wlo).all() and (hi == whi).all() and (cnt.astype(np.uint64) == wcnt).all()
    extra = ["--canonical"] if canonical else []
    txt, binf = tmp_path / "o.txt", tmp_path / "o.bin"
    subprocess.check_call([cli, str(fa), str(txt), str(k), "--global"] + extra)
    subprocess.check_call([cli, str(fa), str(binf), str(k), "--global", "--binary"] + extra)
    if k <= 32:
        want = b"".join(b"%d:%d\n" % (int(a), int(c)) for a, c in zip(lo, cnt))
    else:
        want = b"".join(b"%d:%d:%d\n" % (int(h), int(a), int(c)) for h, a, c in zip(hi, lo, cnt))
    assert txt.read_bytes() == want
    bk, bflags, blo, bhi, bcnt = _read_glb1(binf.read_bytes())
    assert bk == k and (bflags & 1) == int(canonical) and bool(bflags & 2) == (k > 32)
    assert (blo == lo).all() and (bhi == hi).all() and (bcnt == cnt).all()
    # ascending by (hi, lo)
    key = [(int(h) << 64) | int(a) for h, a in zip(bhi, blo)]
    assert key == sorted(key) and len(set(key)) == len(key)


def test_cli_chunksize_is_narrowed_to_ushort_like_the_reference(tmp_path):
    """src/main.cu:110,160: chunkSize > 65535 wraps inside SelectChunk*: with chunkSize 65537 and
    70 000 reads nChunk = 1 and the chunk that reaches the file starts at read (65537 mod 65536) * 1
    = 1, not at read 65537; 65536 narrows to 0, i.e. the chunk starts at read 0"""
    import subprocess
    cli = _cli()
    rng = np.random.default_rng(11)
    n = 70_000
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, int(L))) for L in rng.integers(8, 14, n)]
    raw = "".join(f">r{i}\n{s}\n" for i, s in enumerate(seqs)).encode()
    fa = tmp_path / "many.fasta"
    fa.write_bytes(raw)
    out = tmp_path / "o.cfrk"
    for cs in (65537, 65536, 69999):
        subprocess.check_call([cli, str(fa), str(out), "2", "12", str(cs)])
        want = refsem.reference_cfrk_bytes(raw, 2, cs)
        assert out.read_bytes() == want
        assert want.count(b"\n") + 1 == n - cs
    reads = refsem.read_fasta_compat(raw)
    assert refsem.remainder_chunk(reads, 65537)[0] is reads[1]
    assert refsem.remainder_chunk(reads, 65536)[0] is reads[0]


def test_cli_pipeline_over_two_contexts_per_device_and_batch_mode(tmp_path):
    """--all-chunks through the double-buffered pipeline (two contexts per device, `--gpus 2` rehearsed
    on one device) gives the bytes of the sequential definition; --batch N counts <prefix>_<i>.fasta
    into <out>_<i>.cfrk like the Swift/T loop (swift/cfrk.swf:15-20)"""
    import subprocess
    cli = _cli()
    rng = np.random.default_rng(12)
    files = []
    for f in range(3):
        seqs = ["".join("ACGTN"[c] for c in rng.choice(5, int(L), p=[.245, .245, .245, .245, .02]))
                for L in rng.integers(30, 200, 700 + 100 * f)]
        raw = "".join(f">r{i}\n{s}\n" for i, s in enumerate(seqs)).encode()
        (tmp_path / f"ds_{f}.fasta").write_bytes(raw)
        files.append(raw)

    def want_all(raw, k, cs):
        reads = refsem.read_fasta_compat(raw)
        rows = []
        for c in range(0, len(reads), cs):
            d, s, l = refsem.flatten(reads[c:c + cs])
            rows.append(orc.format_cfrk(orc.per_read_dense(d, s, l, k, orc.ORC_COMPAT), k))
        return b"\n".join(rows)

    out = tmp_path / "o.cfrk"
    for extra in ([], ["--gpus", "2", "--same-device"]):
        subprocess.check_call([cli, str(tmp_path / "ds_0.fasta"), str(out), "3", "4", "64", "--all-chunks"] + extra)
        assert out.read_bytes() == want_all(files[0], 3, 64)
    subprocess.check_call([cli, "--batch", "3", str(tmp_path / "ds"), str(tmp_path / "res"), "2", "4", "256",
                           "--gpus", "2", "--same-device"])
    for f in range(3):
        assert (tmp_path / f"res_{f}.cfrk").read_bytes() == refsem.reference_cfrk_bytes(files[f], 2, 256)
    subprocess.check_call([cli, "--batch", "2", str(tmp_path / "ds"), str(tmp_path / "all"), "2", "4", "256", "--all-chunks"])
    for f in range(2):
        assert (tmp_path / f"all_{f}.cfrk").read_bytes() == want_all(files[f], 2, 256)


@pytest.mark.parametrize("k,canonical", [(31, True), (28, True), (32, True), (31, False), (29, False),
                                         (63, True), (33, True), (48, True), (64, True), (47, False)])
def test_truncated_runs_anchored_to_their_complete_twin_give_the_same_counts(ctx, k, canonical):
    """The leaf kernel notes a read-end run with the complete run it is a prefix (or, on the other
    strand, a suffix) of instead of inserting its k-mers one by one.  Deep coverage with invalid
    bases, short reads (every run cut on both sides) and repeats: identical to the oracle, and to
    the same kernel with the shortcut switched off (CFRK_DEBUG_NO_ANCHORS)."""
    import cfrk_amd
    rng = np.random.default_rng(77 + k)
    G = 40_000
    genome = rng.integers(0, 4, G).astype(np.int8)
    genome[5000:5400] = genome[1000:1400]              # a repeat: the same k-mers at two loci
    genome[9000:9060] = 0                              # homopolymer
    reads = []
    for _ in range(60_000):
        L = int(rng.choice([150, 150, 150, 100, 60, k, k + 3, 20]))
        p = int(rng.integers(0, G - L))
        r = genome[p:p + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1].copy()
        if rng.random() < 0.05:
            r[int(rng.integers(0, L))] = -1
        reads.append(r)
    data, start, length = refsem.flatten(reads)
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    want = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0, threads=8)
    digests = []
    for dbg in (0, cfrk_amd.lib.CFRK_DEBUG_NO_ANCHORS):
        g = cfrk_amd.GlobalCounter(ctx, k, flags, 4 * G)
        g.set_debug_flags(dbg)
        g.add(data, start, length)
        lo, hi, cnt = g.export()
        g.set_debug_flags(0)
        assert len(lo) == len(want[0])
        assert (lo == want[0]).all() and (hi == want[1]).all() and (cnt.astype(np.uint64) == want[2]).all()
        if k <= 32:
            assert g.msp_info()["spilled_kmers"] == 0
        digests.append(g.digest())
    assert digests[0] == digests[1] == orc.digest(*want, two_word=k > 32)


@pytest.mark.parametrize("k,canonical,world,G", [(31, True, 2, 30_000), (31, True, 8, 30_000), (21, True, 4, 30_000),
                                                 (28, False, 3, 30_000), (31, True, 4, 1_200), (16, True, 2, 5_000_000)])
def test_runs_exchange_emulated_ranks_equals_one_gpu_and_oracle(ctx, k, canonical, world, G):
    """Strong-scaling exchange with `world` emulated ranks on one GPU: every rank only
    partitions and deduplicates its read range (CFRK_RUNS_ONLY) and exports per-leaf runs packed per
    owner; every owner counts the runs of its leaves.  The union of the owners' results equals the
    oracle's count of ALL reads, key by key."""
    import cfrk_amd
    # G = 1200: everything in a few hundred leaves, far beyond their fixed stride (exact re-layout on the
    # ranks); G = 5e6: hardly any run repeats (nothing to deduplicate)
    R, L = 24_000, 150
    data, _, _ = orc.synth_reads(0, R, L, G)
    data = data.copy()
    data[::1013] = -1
    data.reshape(R, L + 1)[:, L] = -1
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    sends = []                                        # per rank: (packed rows as uint64 [rows, 2], part_rows)
    for r in range(world):
        r0, r1 = R * r // world, R * (r + 1) // world
        shard = data[r0 * (L + 1):r1 * (L + 1)]
        g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY, 2 * G)
        g.add(shard)
        with pytest.raises(cfrk_amd.CfrkError):       # a job that holds runs has no counts
            g.digest()
        cap = 1 << 20
        d = ctx.alloc(cap * 16)
        rows = g.export_runs_device(d, cap, world)
        host = np.empty((sum(rows), 2), np.uint64)
        ctx.d2h(host, d)
        ctx.free(d)
        sends.append((host, rows))
    merged = {}
    total_rows = 0
    for owner in range(world):
        segs, recv = [], []
        for host, rows in sends:
            a = sum(rows[:owner])
            segs.append(host[a:a + rows[owner]])
            recv.append(rows[owner])
        buf = np.concatenate(segs)
        total_rows += len(buf)
        d = ctx.alloc(max(len(buf), 1) * 16)
        ctx.h2d(d, buf)
        og = cfrk_amd.GlobalCounter(ctx, k, flags, 2 * G)
        og.merge_runs_device(d, recv)
        lo, hi, cnt = og.export()
        ctx.free(d)
        for key, c in zip(lo, cnt):
            assert int(key) not in merged            # owners hold disjoint key sets
            merged[int(key)] = int(c)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0, threads=4)
    assert len(merged) == len(wlo)
    assert all(merged[int(a)] == int(b) for a, b in zip(wlo, wcnt))
    # deduplication really happened on the ranks: far fewer records than super-k-mers (8 per read at least)
    lpp = (65536 + world - 1) // world
    # ... and read ends travel as 2-byte notes: before they alone were 2 rows per read
    if G <= 30_000:
        assert total_rows - world * world * ((lpp * 12 + 15) // 16) < 3 * R


@pytest.mark.parametrize("k,canonical,world,G,dbg", [
    (63, True, 2, 30_000, 0), (40, True, 8, 30_000, 0), (33, False, 3, 30_000, 0), (64, True, 4, 1_200, 0),
    (47, True, 2, 5_000_000, 0), (55, True, 4, 30_000, "subsets"), (63, True, 2, 30_000, "rt_overflow"),
    (44, True, 3, 30_000, "chunked"), (63, False, 2, 1_200, "subsets"), (63, True, 2, 1_200, "few_notes"),
    (40, True, 4, 30_000, "few_notes")])
def test_runs_exchange_two_word_keys_emulated_ranks_equal_the_oracle(ctx, k, canonical, world, G, dbg):
    """The runs exchange for 33 <= k <= 64 (32-byte records travel as two rows): `world`
    emulated ranks partition and deduplicate their read range, the owners count what they receive;
    the union of the owners' results equals the oracle's count of ALL reads, key by key.  Variants:
    leaves shared by record on ranks and owners, a record table that overflows (no deduplication,
    multiplicity 1), ranks that partition in chunks."""
    import cfrk_amd
    R, L = 16_000, 150
    data, _, _ = orc.synth_reads(0, R, L, G)
    data = data.copy()
    data[::1013] = -1
    data.reshape(R, L + 1)[:, L] = -1
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    # ("few_notes": only the first 16 runs of a leaf's list can be named by a note -- in production 2048 -- and the
    #  partition kernel's direct-append path runs; G = 1200 puts hundreds of distinct runs into each of a few leaves)
    dflags = {0: 0, "subsets": cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS, "rt_overflow": cfrk_amd.lib.CFRK_DEBUG_FORCE_RT_OVERFLOW,
              "chunked": cfrk_amd.lib.CFRK_DEBUG_SMALL_PIPELINE, "few_notes": cfrk_amd.lib.CFRK_DEBUG_SMALL_WAVE_CAP}[dbg]
    sends = []
    for r in range(world):
        r0, r1 = R * r // world, R * (r + 1) // world
        shard = data[r0 * (L + 1):r1 * (L + 1)]
        g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY, 2 * G)
        g.set_debug_flags(dflags)
        g.add(shard)
        with pytest.raises(cfrk_amd.CfrkError):       # a job that holds runs has no counts
            g.digest()
        with pytest.raises(cfrk_amd.CfrkError):       # ... and takes one add
            g.add(shard)
        cap = 1 << 20
        d = ctx.alloc(cap * 16)
        with pytest.raises(cfrk_amd.CfrkError) as ei: # a buffer that is too small is reported, not overrun
            g.export_runs_device(d, 1000, world)
        assert ei.value.code == -9
        rows = g.export_runs_device(d, cap, world)
        host = np.empty((sum(rows), 2), np.uint64)
        ctx.d2h(host, d)
        ctx.free(d)
        g.set_debug_flags(0)
        sends.append((host, rows))
    merged = {}
    total_rows = 0
    for owner in range(world):
        segs, recv = [], []
        for host, rows in sends:
            a = sum(rows[:owner])
            segs.append(host[a:a + rows[owner]])
            recv.append(rows[owner])
        buf = np.concatenate(segs)
        total_rows += len(buf)
        d = ctx.alloc(max(len(buf), 1) * 16)
        ctx.h2d(d, buf)
        og = cfrk_amd.GlobalCounter(ctx, k, flags, 2 * G)
        og.set_debug_flags(dflags & ~cfrk_amd.lib.CFRK_DEBUG_SMALL_PIPELINE)
        og.merge_runs_device(d, recv)
        lo, hi, cnt = og.export()
        og.set_debug_flags(0)
        ctx.free(d)
        for a, b, c in zip(lo, hi, cnt):
            key = (int(b) << 64) | int(a)
            assert key not in merged                  # owners hold disjoint key sets
            merged[key] = int(c)
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0, threads=4)
    assert len(merged) == len(wlo)
    assert all(merged[(int(b) << 64) | int(a)] == int(c) for a, b, c in zip(wlo, whi, wcnt))
    # deduplication really happened on the ranks, and read ends travel as notes: far fewer rows than the
    # two per super-k-mer (>= 8 per read at these k) that undeduplicated records would take
    lpp = (65536 + world - 1) // world
    if G <= 30_000 and dbg not in ("rt_overflow", "few_notes"):
        assert total_rows - world * world * ((lpp * 12 + 15) // 16) < 8 * R


def test_runs_exchange_notes_outnumber_the_rows_that_carry_them(ctx):
    """300 k reads of a 1200-base genome on two emulated ranks: 600 k read ends travel as 75 k rows of
    notes, so an owner's leaf streams hold several times more records than the rows it received (the
    streams are sized from the headers, not from the rows); result = the oracle's, key by key"""
    import cfrk_amd
    R, L, k, world = 300_000, 150, 31, 2
    data, _, _ = orc.synth_reads(0, R, L, 1_200)
    sends = []
    for r in range(world):
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_RUNS_ONLY, 10_000)
        g.add(data[(R * r // world) * (L + 1):(R * (r + 1) // world) * (L + 1)])
        cap = 1 << 20
        d = ctx.alloc(cap * 16)
        rows = g.export_runs_device(d, cap, world)
        host = np.empty((sum(rows), 2), np.uint64)
        ctx.d2h(host, d)
        ctx.free(d)
        sends.append((host, rows))
    lpp = 65536 // world
    assert sum(sum(rows) for _, rows in sends) < R           # 2 R read ends and the headers in fewer than R rows
    keys, cnts = [], []
    for owner in range(world):
        segs = [host[sum(rows[:owner]):sum(rows[:owner + 1])] for host, rows in sends]
        buf = np.concatenate(segs)
        d = ctx.alloc(len(buf) * 16)
        ctx.h2d(d, buf)
        og = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 10_000)
        og.merge_runs_device(d, [len(x) for x in segs])
        lo, hi, cnt = og.export()
        ctx.free(d)
        keys.append(lo); cnts.append(cnt)
    lo = np.concatenate(keys); cnt = np.concatenate(cnts)
    order = np.argsort(lo)
    wlo, _, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL, threads=8)
    assert len(lo) == len(wlo) and (lo[order] == wlo).all() and (cnt[order].astype(np.uint64) == wcnt).all()


@pytest.mark.parametrize("K", [31, 63])
def test_runs_exchange_call_sequence_and_malformed_messages(ctx, K):
    """error behaviour of the runs exchange (one-word and two-word records): CFRK_RUNS_ONLY outside 16 <= k <= 64, a second add,
    counts asked of a job that holds runs, a too small send buffer (the needed size is reported and the
    export can be repeated), an owner that is not fresh, and a received segment whose header does not
    add up to its rows"""
    import cfrk_amd
    data, _, _ = orc.synth_reads(0, 4000, 150, 20_000)
    for k, fl in ((15, 0), (65, 0), (K, cfrk_amd.CFRK_FORCE_HASH)):
        with pytest.raises(cfrk_amd.CfrkError) as e:
            cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_RUNS_ONLY | fl, 1000)
        assert e.value.code == -1
    rm = 2 if K > 32 else 1                                    # rows per record
    g = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_RUNS_ONLY, 100_000)
    g.add(data)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.add(data)
    assert e.value.code == -4
    for call in (g.finish, g.digest, g.export):
        with pytest.raises(cfrk_amd.CfrkError) as e:
            call()
        assert e.value.code == -4
    d = ctx.alloc(1 << 24)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.export_runs_device(d, 1000, 2)
    assert e.value.code == -9 and "rows" in str(e.value)
    rows = g.export_runs_device(d, (1 << 24) // 16, 2)       # ... and with room it goes through
    host = np.empty((sum(rows), 2), np.uint64)
    ctx.d2h(host, d)
    # a counting job cannot export runs, a runs job cannot be the owner
    gc = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL, 100_000)
    gc.add(data)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        gc.export_runs_device(d, (1 << 24) // 16, 2)
    assert e.value.code == -4
    with pytest.raises(cfrk_amd.CfrkError) as e:              # not fresh: it already counted something
        gc.merge_runs_device(d, [rows[0], rows[0]])
    assert e.value.code == -4
    # owner 0 gets rank 0's segment twice (two "ranks"): fine; with a row count that contradicts the header: refused
    seg0 = host[:rows[0]]
    buf = np.concatenate([seg0, seg0])
    ctx.h2d(d, buf)
    og = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL, 100_000)
    og.merge_runs_device(d, [rows[0], rows[0]])
    lo, hi, cnt = og.export()
    wlo, _, wcnt = orc.global_count(data, K, orc.ORC_CANONICAL)
    assert 0 < len(lo) < len(wlo) and int(cnt.sum()) % 2 == 0        # owner 0's leaves only, every count doubled
    og2 = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL, 100_000)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        og2.merge_runs_device(d, [rows[0] - 1, rows[0] + 1])
    assert e.value.code == -1
    assert og2.finish() == 0                                           # nothing of the refused message stays
    # notes (16-bit stand-ins for truncated runs: position of the twin << 5 | n-1, after a leaf's records):
    # one that points outside its leaf's list is not followed out of the segment, and a header that
    # announces notes for a leaf without runs is refused
    lpp = 65536 // 2
    hdr = seg0.reshape(-1).view(np.uint32)[:3 * lpp].reshape(lpp, 3).copy()
    rows_of = rm * (hdr[:, 0].astype(np.int64) + hdr[:, 1]) + (hdr[:, 2].astype(np.int64) + 7) // 8
    first = np.concatenate([[0], np.cumsum(rows_of)[:-1]]) + (lpp * 12 + 15) // 16
    ll = int(np.argmax(hdr[:, 2] > 0))
    assert hdr[ll, 2] > 0 and hdr[:, 2].sum() > 0.5 * (hdr[:, 1].sum() + hdr[:, 2].sum())   # most read ends are notes
    bad = seg0.copy()
    bad.reshape(-1).view(np.uint16)[8 * int(first[ll] + rm * (int(hdr[ll, 0]) + int(hdr[ll, 1])))] = (1023 << 5) | 31
    ctx.h2d(d, np.concatenate([bad, seg0]))
    og3 = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL, 100_000)
    og3.merge_runs_device(d, [rows[0], rows[0]])
    assert og3.finish() > 0
    bad = seg0.copy()
    bh = bad.reshape(-1).view(np.uint32)
    empty = int(np.argmax((hdr[:, 0] == 0) & (hdr[:, 2] == 0)))
    one = int(np.argmax(hdr[:, 2] % 8 == 1))                 # a leaf whose last note has a row of its own
    assert hdr[empty, 0] == 0 and hdr[one, 2] % 8 == 1
    bh[3 * one + 2] -= 1                                      # (the rows still add up)
    bh[3 * empty + 2] = 1
    ctx.h2d(d, np.concatenate([bad, seg0]))
    og4 = cfrk_amd.GlobalCounter(ctx, K, cfrk_amd.CFRK_CANONICAL, 100_000)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        og4.merge_runs_device(d, [rows[0], rows[0]])
    assert e.value.code == -1
    ctx.free(d)


def test_cli_global_over_several_devices_equals_one_device(tmp_path):
    """`cfrk ... --global --canonical --gpus N`: reads range-partitioned over N devices (rehearsed on one
    with --same-device), runs exchange staged through host memory, owners' lists merged: the same
    bytes as the single-device run"""
    import subprocess
    cli = _cli()
    rng = np.random.default_rng(21)
    G = 20_000
    genome = rng.integers(0, 4, G)
    seqs = []
    for _ in range(6000):
        L = int(rng.integers(20, 200))
        p = int(rng.integers(0, G - L))
        r = genome[p:p + L]
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        sq = "".join("ACGT"[c] for c in r)
        if rng.random() < 0.05:
            sq = sq[:L // 2] + "N" + sq[L // 2 + 1:]
        seqs.append(sq)
    fa = tmp_path / "g.fasta"
    fa.write_text("".join(f">r{i}\n{s}\n" for i, s in enumerate(seqs)))
    one = tmp_path / "one.cfrk"
    for k, extra in ((31, ["--canonical"]), (21, []), (63, ["--canonical"]), (40, ["--binary"])):
        subprocess.check_call([cli, str(fa), str(one), str(k), "--global"] + extra)
        for n in (2, 3):
            many = tmp_path / f"many{n}.cfrk"
            subprocess.check_call([cli, str(fa), str(many), str(k), "--global", "--gpus", str(n), "--same-device"] + extra)
            assert many.read_bytes() == one.read_bytes()
        assert one.stat().st_size > 100_000


def test_cli_global_over_two_real_devices(tmp_path):
    """the same without --same-device: hipMemcpyPeerAsync / hipDeviceEnablePeerAccess between two real
    devices (cfrk_memcpy_peer).  Skipped on a one-GPU box."""
    import subprocess
    import cfrk_amd
    if cfrk_amd.device_count() < 2:
        pytest.skip("needs two devices")
    cli = _cli()
    rng = np.random.default_rng(22)
    genome = rng.integers(0, 4, 20_000)
    seqs = []
    for _ in range(6000):
        L = int(rng.integers(20, 200))
        a = int(rng.integers(0, len(genome) - L))
        seqs.append("".join("ACGT"[c] for c in genome[a:a + L]))
    fa = tmp_path / "g.fasta"
    fa.write_text("".join(f">r{i}\n{s}\n" for i, s in enumerate(seqs)))
    for k in (31, 63):
        one, two = tmp_path / "one.cfrk", tmp_path / "two.cfrk"
        subprocess.check_call([cli, str(fa), str(one), str(k), "--global", "--canonical"])
        subprocess.check_call([cli, str(fa), str(two), str(k), "--global", "--canonical", "--gpus", "2"])
        assert two.read_bytes() == one.read_bytes()


# src/tipos.h:5,8-10,23-30 -- the reference's types, declared (the same restatement as
# tests/test_integration_stub_cpu.py)
_TIPOS_H = """
#ifndef _tipos_h
#define _tipos_h
#define POW(k) (1U << 2*(k))
typedef unsigned short ushort;
typedef long int lint;
typedef unsigned int uint;
struct read { char *data; int *length; lint *start; int *Freq; struct read *next; };
#endif
"""

# A caller in the shape of src/main.cu:233-305 (written for this test, nothing copied): parse the FASTA with
# the reference's ingest rules, cut it into chunks of chunkSize reads with chunk-relative start[] as
# SelectChunk / SelectChunkRemain do (main.cu:110-206), call kmer_main(&chunk, nN, nS, k, device) for every
# full chunk and for the remainder exactly as main.cu:222,294,300 do, and print what PrintFreq prints
# (main.cu:26-62; the file is opened with "w" for the chunks and again for the remainder, main.cu:303-305).
_MAIN_SHAPED_CALLER = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "tipos.h"
#include "cfrk_host.h"
void kmer_main(struct read *rd, lint nN, lint nS, int k, ushort device);   /* src/kmer.cuh:6 */

static void make_chunk(const cfrk_batch &b, lint first, lint count, struct read *rd, lint *nN)
{
   rd->data = (char *)(b.data + b.start[first]);
   rd->length = (int *)(b.length + first);
   rd->start = (lint *)malloc(sizeof(lint) * (count ? count : 1));
   lint len = 0;
   for (lint i = 0; i < count; i++) { rd->start[i] = len; len += b.length[first + i] + 1; }
   rd->Freq = NULL; rd->next = NULL;
   *nN = len;
}

static void print_freq(const char *path, struct read *chunks, const lint *nS, int n, int k)
{
   FILE *out = fopen(path, "w");
   lint fourk = POW(k);
   int first = 1;
   for (int j = 0; j < n; j++)
      for (lint i = 0; i < nS[j]; i++) {
         if (!first) fputc('\n', out);
         first = 0;
         for (lint c = 0; c < fourk; c++) fprintf(out, "%ld:%d ", c, chunks[j].Freq[i * fourk + c]);
      }
   fclose(out);
}

int main(int argc, char **argv)
{
   if (argc < 4) return 1;
   int k = atoi(argv[3]);
   lint chunkSize = argc == 6 ? atoi(argv[5]) : 8192;
   ushort device = 0;
   cfrk_batch b;
   if (cfrk_host_read_fasta(argv[1], CFRK_INGEST_COMPAT, &b)) return 2;
   lint gnS = b.nS;
   int nChunk = (int)(gnS / chunkSize);
   std::vector<struct read> chunk(nChunk ? nChunk : 1);
   std::vector<lint> nS(nChunk ? nChunk : 1), nN(nChunk ? nChunk : 1);
   for (int i = 0; i < nChunk; i++) { make_chunk(b, (lint)i * chunkSize, chunkSize, &chunk[i], &nN[i]); nS[i] = chunkSize; }
   for (int i = 0; i < nChunk; i++) kmer_main(&chunk[i], nN[i], nS[i], k, device);          /* main.cu:222 */
   struct read remain; lint rnS = gnS - (lint)nChunk * chunkSize, rnN;
   make_chunk(b, (lint)nChunk * chunkSize, rnS, &remain, &rnN);
   kmer_main(&remain, rnN, rnS, k, device);                                                 /* main.cu:300 */
   print_freq(argv[2], chunk.data(), nS.data(), nChunk, k);                                 /* main.cu:303 */
   print_freq(argv[2], &remain, &rnS, 1, k);                                                /* main.cu:305 */
   return 0;
}
"""


@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_kmer_main_shim_executed_by_a_main_shaped_caller_reproduces_the_goldens(derived_fasta, tmp_path, name):
    """The boundary RUN, not only linked: INTEGRATION.md's kmer_main() (cfrk_amd/host/kmer_main_shim.cpp) is
    compiled with a caller shaped like the reference's main() and this repo's host parser into an executable,
    which is fed the golden-derived FASTA; kmer_main(&rd, nN, nS, 2, 0) is called as src/main.cu:222,300 call it
    and the PrintFreq-format output must be the reference's golden byte for byte (test/test.sh:13-19)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    (tmp_path / "tipos.h").write_text(_TIPOS_H)
    (tmp_path / "caller.cpp").write_text(_MAIN_SHAPED_CALLER)
    exe = tmp_path / "refmain"
    lib = os.path.join(root, "cfrk_amd")
    subprocess.check_call([gxx, "-O1", "-std=c++17", "-pthread", "-I" + str(tmp_path), "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(lib, "host"), str(tmp_path / "caller.cpp"),
                           os.path.join(lib, "host", "kmer_main_shim.cpp"), os.path.join(lib, "host", "cfrk_host.cpp"),
                           "-L" + lib, "-lcfrk_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = tmp_path / "out.cfrk"
    subprocess.check_call([str(exe), derived_fasta[name], str(out), "2", "12", "8192"])
    assert out.read_bytes() == open(os.path.join(GOLDEN, f"out-{name}.cfrk"), "rb").read()
    # several full chunks + remainder: only the remainder reaches the file (main.cu:303-305), as refsem restates
    subprocess.check_call([str(exe), derived_fasta[name], str(out), "3", "12", "300"])
    assert out.read_bytes() == refsem.reference_cfrk_bytes(open(derived_fasta[name], "rb").read(), 3, 300)


@pytest.mark.parametrize("k,canonical", [(31, True), (28, False), (21, True), (32, True)])
def test_one_word_leaves_with_more_keys_than_their_table_are_split_by_record(ctx, k, canonical):
    """The one-word path (16 <= k <= 32) shares an overfull leaf by RECORD too: with a capacity hint above
    2.7e8 the partition kernel writes five more minimizer-hash bits into the header's top byte and 2^s
    workgroups take a leaf, each the records its bits name -- every record expanded once, where key-subset
    passes expand every record in every pass.  All-distinct input (~4000 distinct k-mers per leaf), hint
    above / below the switch: both give the digest of the general HBM-table path; and the same switch
    forced on a small deep input (anchored truncated runs, record table in use, exact re-layout, chunked
    counting, export by leaf) gives the oracle's list."""
    import cfrk_amd
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    L = 150
    R = 250_000_000 // (L - k + 1)                                     # 2.5e8 k-mers, ~all distinct (the list of the smaller hint holds 2.68e8)
    data, _, _ = orc.synth_reads(0, R, L, 0, uniform=True)
    K = R * (L - k + 1)
    digests = []
    for hint in (300_000_000, 134_000_000):                            # record subsets on / off
        g = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
        g.add(data)
        d = g.digest()
        assert d[1] == K and d[0] > 0.99 * K
        digests.append(d)
        del g
    gh = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_FORCE_HASH, 300_000_000)
    gh.add(data)
    assert digests[0] == digests[1] == gh.digest()
    del gh
    for G, dbg in ((30_000, 0), (1_200, 0), (30_000, cfrk_amd.lib.CFRK_DEBUG_SMALL_PIPELINE), (5_000_000, cfrk_amd.lib.CFRK_DEBUG_FORCE_RT_OVERFLOW)):
        small, _, _ = orc.synth_reads(0, 40_000, 150, G)
        small = small.copy()
        small[::1013] = -1
        want = orc.global_count(small, k, orc.ORC_CANONICAL if canonical else 0, threads=4)
        g = cfrk_amd.GlobalCounter(ctx, k, flags, 8_000_000)
        g.set_debug_flags(cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS | dbg)
        g.add(small)
        lo, hi, cnt = g.export()
        g.set_debug_flags(0)
        assert len(lo) == len(want[0])
        assert (lo == want[0]).all() and (cnt.astype(np.uint64) == want[2]).all()
        del g


def test_two_word_leaves_with_more_keys_than_their_table_are_split_by_record(ctx):
    """k = 63, all-distinct input with ~4000 distinct k-mers per leaf (the LDS table has 4096 slots):
    with a capacity hint above 2.7e8 the records carry more minimizer-hash bits and several workgroups
    share a leaf, each taking the records its bits name (every record expanded once); with a smaller
    hint an overfull leaf is split by key.  Both give the digest of the general HBM-table path."""
    import cfrk_amd
    R, L, k = 3_000_000, 150, 63
    data, _, _ = orc.synth_reads(0, R, L, 0, uniform=True)             # 2.64e8 k-mers, ~all distinct
    K = R * (L - k + 1)
    digests = []
    for hint in (300_000_000, 134_000_000):                            # record subsets on / off
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, hint)
        g.add(data)
        d = g.digest()
        assert d[1] == K and d[0] > 0.999 * K
        digests.append(d)
        del g
    gh = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL | cfrk_amd.CFRK_FORCE_HASH, 300_000_000)
    gh.add(data)
    assert digests[0] == digests[1] == gh.digest()
    # the same switch on a small, deep input (anchored truncated runs, record table in use), with four, two
    # and one sub-values per workgroup (the library picks from the expected runs per leaf; the knob is for tests)
    small, _, _ = orc.synth_reads(0, 40_000, 150, 30_000)
    want = orc.global_count(small, k, orc.ORC_CANONICAL, threads=4)
    for hb in (0, 2, 1):                                               # library's choice, two, one sub-values
        g = cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_CANONICAL, 100_000)
        g.set_debug_flags(cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS)
        g.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP2_SUBVALUE_BITS, hb)
        try:
            g.add(small)
            lo, hi, cnt = g.export()
        finally:
            g.set_debug_flags(0)
            g.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP2_SUBVALUE_BITS, 0)
        assert (lo == want[0]).all() and (hi == want[1]).all() and (cnt.astype(np.uint64) == want[2]).all()


# ------------------------------------------------------------------ pipelined runs exchange (round 5)

def _pipelined_exchange(ctx, data, R, L, k, flags, world, ngroups, hint, seg_cap=1 << 18, dbg=0, owner_subvalue_bits=0):
    """`world` emulated ranks on one GPU through the PIPELINED runs exchange: CFRK_RUNS_DEFER add, export_runs_async,
    per group export_runs_wait + "all-to-all" (host copies) + merge_runs_group_device on every owner.
    -> {key: count} over all owners, or None when a rank's export refused (overflow / spill / small segment)"""
    import cfrk_amd
    per_rank = []                                     # [rank][group] -> (rows per owner, host copy of the group's segments)
    for r in range(world):
        r0, r1 = R * r // world, R * (r + 1) // world
        shard = np.ascontiguousarray(data[r0 * (L + 1):r1 * (L + 1)])
        g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, hint)
        g.set_debug_flags(dbg)
        try:
            g.add(shard)
            d = ctx.alloc(ngroups * world * seg_cap * 16)
            g.export_runs_async(d, seg_cap, world, ngroups)
            groups = []
            for gi in range(ngroups):
                try:
                    rows = g.export_runs_wait(gi)
                except cfrk_amd.CfrkError as e:
                    assert e.code in (-4, -9), e
                    ctx.sync()
                    ctx.free(d)
                    return None
                host = np.empty((world * seg_cap, 2), np.uint64)
                ctx.d2h(host, d + gi * world * seg_cap * 16)
                groups.append((rows, host))
            ctx.sync()
            ctx.free(d)
        finally:
            g.set_debug_flags(0)
        per_rank.append(groups)
    merged = {}
    for owner in range(world):
        og = cfrk_amd.GlobalCounter(ctx, k, flags, hint)
        og.set_debug_flags(dbg & (cfrk_amd.lib.CFRK_DEBUG_NO_ANCHORS | cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS))
        og.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP2_SUBVALUE_BITS, owner_subvalue_bits)   # (1: one sub-value per workgroup, four workgroups per leaf)
        bufs = []
        for gi in range(ngroups):
            segs, recv = [], []
            for r in range(world):
                rows, host = per_rank[r][gi]
                segs.append(host[owner * seg_cap:owner * seg_cap + rows[owner]])
                recv.append(rows[owner])
            buf = np.concatenate(segs)
            d = ctx.alloc(len(buf) * 16)
            ctx.h2d(d, buf)
            og.merge_runs_group_device(d, recv, gi, ngroups)      # enqueued: the buffer stays until the export below
            bufs.append(d)
        lo, hi, cnt = og.export()
        og.set_debug_flags(0)
        og.set_debug_param(cfrk_amd.lib.CFRK_PARAM_MSP2_SUBVALUE_BITS, 0)
        for d in bufs:
            ctx.free(d)
        keys = [int(x) for x in lo] if k <= 32 else [(int(h) << 64) | int(l) for l, h in zip(lo, hi)]
        for key, c in zip(keys, cnt):
            assert key not in merged                 # owners hold disjoint key sets
            merged[key] = int(c)
    return merged


@pytest.mark.parametrize("k,canonical,world,ngroups,G,dbg", [
    (31, True, 2, 2, 300_000, 0), (31, True, 8, 2, 300_000, 0), (31, True, 8, 1, 300_000, 0), (21, True, 4, 3, 300_000, 0),
    (28, False, 3, 2, 300_000, 0), (16, True, 2, 4, 5_000_000, 0), (32, True, 5, 2, 300_000, 0),
    (31, True, 4, 2, 300_000, "rt_overflow"), (31, True, 4, 2, 300_000, "no_anchors"), (25, True, 3, 2, 300_000, "chunked"),
    (31, False, 4, 16, 300_000, 0),
    # two-word keys (msp2.hip: 32-byte records travel as two rows)
    (63, True, 4, 2, 300_000, 0), (33, False, 3, 2, 300_000, 0), (47, True, 8, 3, 300_000, 0), (64, True, 2, 1, 300_000, 0),
    (63, True, 4, 2, 300_000, "rt_overflow"), (63, True, 4, 2, 300_000, "no_anchors"), (55, True, 3, 2, 300_000, "note_pos"),
    (40, True, 5, 16, 300_000, 0),
    # ... whose leaves are shared by sub-value (configs[4]-sized hints; forced here): the spare word travels, the owner
    # counts a leaf with several workgroups
    (63, True, 4, 2, 300_000, "subsets"), (40, False, 3, 3, 300_000, "subsets"), (64, True, 8, 1, 300_000, "subsets"),
    (47, True, 2, 2, 300_000, "subsets_rt_overflow"), (63, True, 4, 2, 300_000, "subsets_wg1"), (40, False, 3, 3, 300_000, "subsets_wg2")])
def test_pipelined_runs_exchange_emulated_ranks_equal_the_oracle(ctx, k, canonical, world, ngroups, G, dbg):
    """the pipelined form of the strong-scaling exchange (cfrk_global_export_runs_async / _wait,
    cfrk_global_merge_runs_group_device): the sender deduplicates and packs one group of leaves after the
    other straight into the send buffer (no gather pass, notes decided in LDS), the owner's leaf kernel reads the N lists
    of a leaf in place (no layout / scatter kernels, notes expanded where they are read) -- the union of the owners'
    results equals the oracle's count of ALL reads, key by key.  (Genome 300 000, 12 x coverage: the leaf streams of a
    deferred add keep their fixed stride -- it cannot lay them out again; that refusal is the next test's.)
    Variants: a world that does not divide 65 536, one to
    sixteen groups, leaves whose record table is forced to overflow (runs leave undeduplicated), no anchoring, a chunked
    add, forward-strand counting."""
    import cfrk_amd
    R, L = 24_000, 150
    data, _, _ = orc.synth_reads(0, R, L, G)
    data = data.copy()
    data[::1013] = -1
    data.reshape(R, L + 1)[:, L] = -1
    flags = cfrk_amd.CFRK_CANONICAL if canonical else 0
    bits = {0: 0, "rt_overflow": cfrk_amd.CFRK_DEBUG_FORCE_RT_OVERFLOW, "no_anchors": cfrk_amd.lib.CFRK_DEBUG_NO_ANCHORS,
            "chunked": cfrk_amd.CFRK_DEBUG_SMALL_PIPELINE, "note_pos": cfrk_amd.lib.CFRK_DEBUG_SMALL_WAVE_CAP,
            "subsets": cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS,
            "subsets_rt_overflow": cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS | cfrk_amd.CFRK_DEBUG_FORCE_RT_OVERFLOW,
            "subsets_wg1": cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS, "subsets_wg2": cfrk_amd.lib.CFRK_DEBUG_RECORD_SUBSETS}[dbg]
    # (subsets_wg1 / _wg2: the owner's workgroups take one / two sub-values each: four / two workgroups share a leaf)
    merged = _pipelined_exchange(ctx, data, R, L, k, flags, world, ngroups, 2 * G, dbg=bits,
                                 owner_subvalue_bits={"subsets_wg1": 1, "subsets_wg2": 2}.get(dbg, 0))
    assert merged is not None
    wlo, whi, wcnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0, threads=4)
    assert len(merged) == len(wlo)
    if k <= 32:
        assert all(merged[int(a)] == int(b) for a, b in zip(wlo, wcnt))
    else:
        assert all(merged[(int(h) << 64) | int(a)] == int(b) for a, h, b in zip(wlo, whi, wcnt))


def test_pipelined_runs_exchange_refuses_what_it_cannot_carry_and_the_classic_export_follows(ctx):
    """(1) a segment capacity that is too small: export_runs_wait returns CFRK_ERR_SMALL_BUF for the group, nothing is
    written beyond a segment, and the CLASSIC export still works on the same job (the leaf streams were not touched);
    (2) a shard whose leaf streams overflow their fixed stride (deep coverage of a 1200-base genome): the deferred add
    cannot lay them out again -- export_runs_wait returns CFRK_ERR_STATE, and so does the classic export: the rank adds
    again without CFRK_RUNS_DEFER; (3) call-sequence errors."""
    import cfrk_amd
    R, L, k = 24_000, 150, 31
    data, _, _ = orc.synth_reads(0, R, L, 300_000)
    data = data.copy()
    flags = cfrk_amd.CFRK_CANONICAL
    g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, 600_000)
    g.add(data)
    lpp = g.leaves_per_part(2)
    cap = lpp // 2 + 2 + 64                                   # the header and 64 rows: far too small
    d = ctx.alloc(2 * 2 * cap * 16 + 4096)
    guard = np.full(256, 0xABABABABABABABAB, np.uint64)
    ctx.h2d(d + 2 * 2 * cap * 16, guard)
    g.export_runs_async(d, cap, 2, 2)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.export_runs_wait(0)
    assert e.value.code == -9
    ctx.sync()
    back = np.empty(256, np.uint64)
    ctx.d2h(back, d + 2 * 2 * cap * 16)
    assert (back == guard).all()
    ctx.free(d)
    big = ctx.alloc((1 << 20) * 16)
    rows = g.export_runs_device(big, 1 << 20, 2)              # the classic export deduplicates first, then packs
    assert sum(rows) > 0
    with pytest.raises(cfrk_amd.CfrkError) as e:              # ... after which the streams are no longer what the pipelined form reads
        g.export_runs_async(big, 1 << 18, 2, 2)
    assert e.value.code == -4
    ctx.free(big)
    # (2)
    tiny, _, _ = orc.synth_reads(0, R, L, 1_200)
    g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY | cfrk_amd.CFRK_RUNS_DEFER, 60_000)
    g.add(tiny)
    d = ctx.alloc(2 * 2 * (1 << 18) * 16)
    g.export_runs_async(d, 1 << 18, 2, 2)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.export_runs_wait(0)
    assert e.value.code == -4
    ctx.sync()
    with pytest.raises(cfrk_amd.CfrkError) as e:
        g.export_runs_device(d, 1 << 19, 2)
    assert e.value.code == -4
    g = cfrk_amd.GlobalCounter(ctx, k, flags | cfrk_amd.CFRK_RUNS_ONLY, 60_000)     # without the flag the add settles
    g.add(tiny)
    assert sum(g.export_runs_device(d, 1 << 19, 2)) > 0
    # (3)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        cfrk_amd.GlobalCounter(ctx, k, cfrk_amd.CFRK_RUNS_DEFER, 1000)              # the flag goes with CFRK_RUNS_ONLY
    assert e.value.code == -1
    og = cfrk_amd.GlobalCounter(ctx, k, flags, 60_000)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        og.merge_runs_group_device(d, [10, 10], 1, 2)         # groups are merged in order
    assert e.value.code == -4
    # a segment whose header is garbage is not followed, and the job says so
    junk = np.full((4096, 2), 0x1234567812345678, np.uint64)
    ctx.h2d(d, junk)
    og.merge_runs_group_device(d, [2048, 2048], 0, 16)
    with pytest.raises(cfrk_amd.CfrkError) as e:
        og.finish()
    assert e.value.code == -6
    ctx.free(d)
