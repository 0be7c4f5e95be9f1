"""Oracle vs the reference's own goldens and hand-made cases (CPU only)."""
import os

import numpy as np
import pytest

from . import oracle_lib as orc
from . import refsem
from .conftest import GOLDEN


def _enc(s):
    return refsem._CODE[np.frombuffer(s.encode(), np.uint8)]


@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_golden_k2_byte_exact(derived_fasta, name):
    """test/test.sh:13-19: `cfrk seqN.fasta out.cfrk 2 12 8192` == test/out-seqN.cfrk."""
    raw = open(derived_fasta[name], "rb").read()
    got = refsem.reference_cfrk_bytes(raw, 2, 8192)
    want = open(os.path.join(GOLDEN, f"out-{name}.cfrk"), "rb").read()
    assert got == want


def test_golden_seq2_fixture_matches_generator(derived_fasta):
    """the committed seq2 pre-image is exactly what derive_fasta.py produces"""
    committed = open(os.path.join(GOLDEN, "seq2-derived.fasta"), "rb").read()
    assert committed == open(derived_fasta["seq2"], "rb").read()


def test_survey_worked_example():
    """SURVEY 8a worked example (k=2 and k=3 row sums) on ACGTACGTAC / AAAANAAAA / TTTTT / ACGT\\nACGT"""
    raw = b">a\nACGTACGTAC\n>b\nAAAANAAAA\n>c\nTTTTT\n>d\nACGT\nACGT\n"
    reads = refsem.read_fasta_compat(raw)
    assert [len(r) for r in reads] == [10, 9, 5, 9]
    data, start, length = refsem.flatten(reads)
    f = orc.per_read_dense(data, start, length, 2, orc.ORC_COMPAT)
    AC, CG, GT, TA, TT, AA = 1, 6, 11, 12, 15, 0
    assert f[0, AC] == 3 and f[0, CG] == 2 and f[0, GT] == 2 and f[0, TA] == 2 and f[0, TT] == 2
    assert f[0].sum() == 11
    assert f[1, AA] == 6 and f[1].sum() == 6
    assert f[2, TT] == 6 and f[2].sum() == 6      # 4 own + 2 spilled by read d's embedded newline
    assert f[3, AC] == 2 and f[3, CG] == 2 and f[3, GT] == 2 and f[3].sum() == 6
    f3 = orc.per_read_dense(data, start, length, 3, orc.ORC_COMPAT)
    assert list(f3.sum(axis=1)) == [12, 5, 7, 4]


def test_compat_1024_window_cap():
    """src/kmer_kernel.cu:85 with blockDim 1024: a 2000-base read contributes 1024 windows"""
    rng = np.random.default_rng(0)
    r = rng.integers(0, 4, 2000).astype(np.int8)
    data, start, length = refsem.flatten([r])
    f = orc.per_read_dense(data, start, length, 2, orc.ORC_COMPAT)
    assert f.sum() == 1024
    fn = orc.per_read_dense(data, start, length, 2, 0)
    assert fn.sum() == 1999


def test_compat_first_read_spill_dropped_and_tail_windows():
    # k=3: read 0 has 1 tail window (k-2) -> Freq[-1], dropped; read 1's tail lands on row 0 bin 63
    data, start, length = refsem.flatten([_enc("ACGTA"), _enc("CCCCC")])
    f = orc.per_read_dense(data, start, length, 3, orc.ORC_COMPAT)
    assert f[0].sum() == 3 + 1 and f[0, 63] == 1
    assert f[1].sum() == 3 and f[1, 0b010101] == 3


def test_native_equals_compat_when_no_invalid_k2():
    rng = np.random.default_rng(1)
    reads = [rng.integers(0, 4, n).astype(np.int8) for n in (150, 151, 2, 1, 300)]
    data, start, length = refsem.flatten(reads)
    a = orc.per_read_dense(data, start, length, 2, orc.ORC_COMPAT)
    b = orc.per_read_dense(data, start, length, 2, 0)
    assert (a == b).all()


def test_float_index_exact_up_to_k12_and_not_beyond():
    """src/kmer_kernel.cu:38 accumulates through float: exact for k <= 12 only"""
    rng = np.random.default_rng(2)
    r = rng.integers(0, 4, 4000).astype(np.int8)
    data, _, _ = refsem.flatten([r])
    for k in (2, 7, 12):
        assert (orc.compute_index(data, k, True) == orc.compute_index(data, k, False)).all()
    assert (orc.compute_index(data, 14, True) != orc.compute_index(data, 14, False)).any()


def test_global_equals_column_sum_of_native_dense():
    rng = np.random.default_rng(3)
    reads = [rng.integers(-1, 4, int(n)).astype(np.int8) for n in rng.integers(1, 200, 50)]
    data, start, length = refsem.flatten(reads)
    for k in (1, 2, 5):
        dense = orc.per_read_dense(data, start, length, k, 0).sum(axis=0)
        klo, khi, cnt = orc.global_count(data, k, 0)
        ref = np.zeros(4 ** k, np.int64)
        ref[klo.astype(np.int64)] = cnt.astype(np.int64)
        assert (ref == dense).all()
        assert (khi == 0).all()


def _py_count(data, k, canonical):
    """pure-Python loops (small cases only)"""
    out = {}
    n = len(data)
    for s in range(n - k + 1):
        w = data[s:s + k]
        if (w < 0).any():
            continue
        f = 0
        r = 0
        for i, c in enumerate(w):
            f = (f << 2) | int(c)
            r |= (3 - int(c)) << (2 * i)
        key = min(f, r) if canonical else f
        out[key] = out.get(key, 0) + 1
    return out


@pytest.mark.parametrize("k", [1, 3, 15, 16, 31, 32, 33, 47, 63, 64])
@pytest.mark.parametrize("canonical", [False, True])
def test_global_count_vs_python(k, canonical):
    rng = np.random.default_rng(k)
    reads = [rng.integers(0, 4, int(n)).astype(np.int8) for n in rng.integers(1, 120, 30)]
    reads.append(np.array([0] * 70 + [-1] + [3] * 70, np.int8))     # homopolymers + an N
    reads.append(np.array([3] * 100, np.int8))
    data, _, _ = refsem.flatten(reads)
    klo, khi, cnt = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0)
    want = _py_count(data, k, canonical)
    got = {(int(h) << 64) | int(l): int(c) for l, h, c in zip(klo, khi, cnt)}
    assert got == want
    keys = [(int(h), int(l)) for l, h in zip(klo, khi)]
    assert keys == sorted(keys)
    # the partition + radix-sort variant (large cases, cpu_baseline) gives the identical list
    for threads in (1, 3):
        l2, h2, c2 = orc.global_count(data, k, orc.ORC_CANONICAL if canonical else 0, threads=threads)
        assert (l2 == klo).all() and (h2 == khi).all() and (c2 == cnt).all()


def test_synth_reads_deterministic_and_strand():
    d, s, l = orc.synth_reads(0, 64, 50, 1000)
    d2, _, _ = orc.synth_reads(32, 32, 50, 1000)
    assert (d[32 * 51:] == d2).all()
    assert (d.reshape(64, 51)[:, 50] == -1).all() and (l == 50).all() and (s == np.arange(64) * 51).all()
    g = np.array([orc.splitmix64(1 + j) & 3 for j in range(1000)], np.int8)
    for r in range(64):
        pos = orc.splitmix64(2 ^ r) % (1000 - 50 + 1)
        sl = g[pos:pos + 50]
        if orc.splitmix64(3 ^ r) & 1:
            sl = (3 - sl)[::-1]
        assert (d.reshape(64, 51)[r, :50] == sl).all()


def test_digest_order_independent():
    d, _, _ = orc.synth_reads(0, 200, 80, 3000)
    klo, khi, cnt = orc.global_count(d, 21, orc.ORC_CANONICAL)
    a = orc.digest(klo, khi, cnt)
    p = np.random.default_rng(0).permutation(len(klo))
    assert orc.digest(klo[p], khi[p], cnt[p]) == a
    assert a[0] == len(klo) and a[1] == 200 * (80 - 21 + 1)


def test_format_cfrk():
    f = np.arange(32, dtype=np.int32)
    assert orc.format_cfrk(f, 2) == (" ".join(f"{i}:{i}" for i in range(16)) + " \n" +
                                     " ".join(f"{i}:{i + 16}" for i in range(16)) + " ").encode()


def test_empty_output_when_reads_multiple_of_chunk():
    """main.cu:270-305: gnS % chunkSize == 0 -> empty file; 5 reads chunk 2 -> only read 5"""
    raw = b"".join(b">r%d\nACGTACGT\n" % i for i in range(4))
    assert refsem.reference_cfrk_bytes(raw, 2, 2) == b""
    raw5 = raw + b">r4\nTTTTGGGG\n"
    out = refsem.reference_cfrk_bytes(raw5, 2, 2)
    assert out.count(b"\n") == 0 and b"15:3 " in out and b"10:3 " in out


def test_no_final_newline_drops_last_base():
    """fastaIO.h:53: len = strlen - 1 drops the last base when the file lacks a final newline"""
    reads = refsem.read_fasta_compat(b">a\nACGT")
    assert [len(r) for r in reads] == [3]


@pytest.mark.parametrize("k", [5, 15, 31, 32, 33, 63, 64])
def test_sorted_variant_equals_hash_variant_on_synthetic_reads(k):
    """orc_global_count_sorted (threads > 0) against orc_global_count on reads with deep coverage,
    invalid bases and both strands: same keys, same counts, same digest"""
    data, _, _ = orc.synth_reads(0, 4000, 150, 20000)
    data = data.copy()
    data[::991] = -1
    a = orc.global_count(data, k, orc.ORC_CANONICAL)
    b = orc.global_count(data, k, orc.ORC_CANONICAL, threads=5)
    assert all((x == y).all() for x, y in zip(a, b))
    assert orc.digest(*a, two_word=k > 32) == orc.digest(*b, two_word=k > 32)


@pytest.mark.parametrize("R,L,G,k,canonical,uniform", [
    (3000, 150, 20000, 31, True, False), (3000, 150, 20000, 31, False, False),
    (2000, 250, 50000, 63, True, False), (1500, 100, 9000, 15, True, False),
    (700, 60, 60, 40, True, True), (5000, 151, 4000, 33, False, False),
    (1, 150, 150, 31, True, False), (0, 150, 150, 31, True, False)])
def test_bounded_memory_digest_equals_the_plain_count(R, L, G, k, canonical, uniform):
    """orc_synth_digest (reads generated block by block, key space counted in slices: what counts the
    full-size configs on the GPU box's host) gives the digest of orc_global_count on the materialised
    reads, for any number of threads and slices"""
    fl = orc.ORC_CANONICAL if canonical else 0
    d, _, _ = orc.synth_reads(0, R, L, G, uniform=uniform)
    want = orc.digest(*orc.global_count(d, k, fl), two_word=k > 32)
    for threads, slices in [(1, 1), (3, 1), (4, 5), (7, 16)]:
        assert orc.synth_digest(R, L, G, k, fl, threads, slices, uniform=uniform) == want, (threads, slices)
    assert orc.synth_digest(R, L, G, k, fl, 2, 3, uniform=uniform, progress=lambda i, n: None) == want


@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_float_index_equals_exact_index_on_the_golden_preimages_up_to_k12(derived_fasta, name):
    """The reference accumulates the index through float (src/kmer_kernel.cu:38).  On the inputs
    that reproduce its goldens the float path and the exact-integer path give identical Index[]
    arrays and identical compat rows for every k <= 12 -- the range in which the reference is
    numerically defined -- so the HIP path (exact integers) is checked against the same numbers
    the reference would produce there."""
    raw = open(derived_fasta[name], "rb").read()
    reads = refsem.read_fasta_compat(raw)
    data, start, length = refsem.flatten(reads)
    for k in range(1, 13):
        assert (orc.compute_index(data, k, True) == orc.compute_index(data, k, False)).all(), k
    sub = slice(0, 64)
    d2, s2, l2 = refsem.flatten(reads[sub])
    for k in (2, 6):
        a = orc.per_read_dense(d2, s2, l2, k, orc.ORC_COMPAT | orc.ORC_FLOAT_INDEX)
        b = orc.per_read_dense(d2, s2, l2, k, orc.ORC_COMPAT)
        assert (a == b).all()
