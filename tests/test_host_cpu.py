"""Host-side C++ (ingest, chunk views, .cfrk text) against the Python restatement of the
reference's host semantics and the oracle's formatter.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from . import oracle_lib as orc
from . import refsem
from .conftest import GOLDEN, ROOT


class Batch(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_int8)), ("start", C.POINTER(C.c_int64)),
                ("length", C.POINTER(C.c_int32)), ("nN", C.c_int64), ("nS", C.c_int64)]


@pytest.fixture(scope="module")
def host():
    so = os.path.join(ROOT, "cfrk_amd", "libcfrk_host.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cfrk_amd", "host"), "../libcfrk_host.so"],
                          stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.cfrk_host_parse_fasta.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Batch)]
    L.cfrk_host_read_fasta.argtypes = [C.c_char_p, C.c_int, C.POINTER(Batch)]
    L.cfrk_host_free_batch.argtypes = [C.POINTER(Batch)]
    L.cfrk_host_format_dense.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_size_t]
    L.cfrk_host_format_dense.restype = C.c_size_t
    L.cfrk_host_format_dense_mt.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_size_t, C.c_int]
    L.cfrk_host_format_dense_mt.restype = C.c_size_t
    L.cfrk_host_format_sparse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
    L.cfrk_host_format_sparse.restype = C.c_size_t
    L.cfrk_host_chunk.argtypes = [C.POINTER(Batch), C.c_int64, C.c_int64, C.POINTER(C.POINTER(C.c_int8)),
                                  C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int64)]
    return L


def _parse(L, raw, flags):
    b = Batch()
    rc = L.cfrk_host_parse_fasta(raw, len(raw), flags, C.byref(b))
    if rc:
        return rc, None
    data = np.ctypeslib.as_array(b.data, (max(b.nN, 1),))[:b.nN].copy()
    start = np.ctypeslib.as_array(b.start, (max(b.nS, 1),))[:b.nS].copy()
    length = np.ctypeslib.as_array(b.length, (max(b.nS, 1),))[:b.nS].copy()
    L.cfrk_host_free_batch(C.byref(b))
    return 0, (data, start, length)


CASES = [
    b">a\nACGTACGTAC\n>b\nAAAANAAAA\n>c\nTTTTT\n>d\nACGT\nACGT\n",
    b">a\nacgtn\n>b\nRYKM\n\n\n\n",
    b">only\nACGT",                       # no final newline: compat drops the last base
    b">x\nAC\r\nGT\r\n",
    b">h1 some text > more\nAAAA\n>h2\nC\n",
]


@pytest.mark.parametrize("raw", CASES)
def test_compat_ingest_matches_reference_semantics(host, raw):
    rc, got = _parse(host, raw, 1)
    assert rc == 0
    want = refsem.flatten(refsem.read_fasta_compat(raw))
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()


def test_native_ingest_joins_lines_and_keeps_last_base(host):
    rc, (data, start, length) = _parse(host, b">a\nAC\r\nGT\n>b\nTTTT", 0)
    assert rc == 0
    assert list(length) == [4, 4] and list(start) == [0, 5]
    assert list(data) == [0, 1, 2, 3, -1, 3, 3, 3, 3, -1]


def test_large_batch_is_encoded_by_several_threads_identically(host):
    """batches of >= 8 MB of codes are encoded by a thread pool (records split by bytes): ragged
    multi-line records, lower case and N, checked against the one-record-at-a-time reference"""
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"ACGTacgtN", np.uint8)
    parts = []
    reads = []
    for i in range(1500):
        n = int(rng.integers(1, 20000)) if i % 7 else int(rng.integers(1, 400000))
        seq = alphabet[rng.integers(0, len(alphabet), n)]
        reads.append(seq)
        parts.append(b">r%d\n" % i)
        w = int(rng.integers(20, 200))
        for o in range(0, n, w):
            parts.append(seq[o:o + w].tobytes() + b"\n")
    raw = b"".join(parts)
    rc, (data, start, length) = _parse(host, raw, 0)
    assert rc == 0 and len(data) >= (8 << 20)
    lut = np.full(256, -1, np.int8)
    for ch, v in zip(b"ACGT", range(4)):
        lut[ch] = v
        lut[ch + 32] = v
    want = np.concatenate([np.concatenate([lut[r], np.array([-1], np.int8)]) for r in reads])
    assert list(length) == [len(r) for r in reads]
    assert (start == np.concatenate([[0], np.cumsum([len(r) + 1 for r in reads])[:-1]])).all()
    assert data.shape == want.shape and (data == want).all()


def test_ingest_errors(host):
    assert _parse(host, b"ACGT\n>a\nAC\n", 1)[0] == -2
    assert _parse(host, b">a\n>b\nAC\n", 1)[0] == -3          # UB in the reference: rejected
    b = Batch()
    assert host.cfrk_host_read_fasta(b"/nonexistent/file.fasta", 1, C.byref(b)) == -1
    rc, (data, start, length) = _parse(host, b"", 1)
    assert rc == 0 and len(data) == 0 and len(length) == 0


def test_read_fasta_from_a_file_equals_parsing_its_bytes(host, tmp_path):
    raw = b">a\nACGTACGTAC\n>b\nAAAANAAAA\nCC\n>c\nTTTTT\n" * 500
    path = tmp_path / "x.fasta"
    path.write_bytes(raw)
    for flags in (0, 1):
        b = Batch()
        assert host.cfrk_host_read_fasta(str(path).encode(), flags, C.byref(b)) == 0
        data = np.ctypeslib.as_array(b.data, (b.nN,)).copy()
        length = np.ctypeslib.as_array(b.length, (b.nS,)).copy()
        host.cfrk_host_free_batch(C.byref(b))
        rc, (wdata, _, wlength) = _parse(host, raw, flags)
        assert rc == 0 and (data == wdata).all() and (length == wlength).all()
    empty = tmp_path / "empty.fasta"
    empty.write_bytes(b"")
    b = Batch()
    assert host.cfrk_host_read_fasta(str(empty).encode(), 0, C.byref(b)) == 0 and b.nS == 0
    host.cfrk_host_free_batch(C.byref(b))


@pytest.mark.parametrize("name", ["seq1", "seq2"])
def test_golden_through_host_ingest_and_writer(host, derived_fasta, name):
    """reference goldens: C++ ingest -> oracle counts -> C++ writer == golden bytes"""
    raw = open(derived_fasta[name], "rb").read()
    rc, (data, start, length) = _parse(host, raw, 1)
    assert rc == 0
    freq = orc.per_read_dense(data, start, length, 2, orc.ORC_COMPAT)
    n = host.cfrk_host_format_dense(freq.ctypes.data, len(length), 2, None, 0)
    buf = C.create_string_buffer(n)
    assert host.cfrk_host_format_dense(freq.ctypes.data, len(length), 2, buf, n) == n
    assert buf.raw[:n] == open(os.path.join(GOLDEN, f"out-{name}.cfrk"), "rb").read()


def test_writer_matches_oracle_formatter(host):
    rng = np.random.default_rng(0)
    f = rng.integers(0, 100000, 5 * 64).astype(np.int32)
    n = host.cfrk_host_format_dense(f.ctypes.data, 5, 3, None, 0)
    buf = C.create_string_buffer(n)
    host.cfrk_host_format_dense(f.ctypes.data, 5, 3, buf, n)
    assert buf.raw[:n] == orc.format_cfrk(f, 3)
    keys = np.array([0, 7, 2 ** 63 + 5], np.uint64)
    cnt = np.array([1, 4000000000, 9], np.uint32)
    n = host.cfrk_host_format_sparse(keys.ctypes.data, cnt.ctypes.data, 3, None, 0)
    buf = C.create_string_buffer(n)
    host.cfrk_host_format_sparse(keys.ctypes.data, cnt.ctypes.data, 3, buf, n)
    assert buf.raw[:n] == b"0:1\n7:4000000000\n9223372036854775813:9\n"


def test_sparse_two_word_text_and_binary_form_round_trip(host):
    """k > 32 keys as "hi:lo:count" lines; the CFRKGLB1 binary form (12- / 20-byte records) written
    and parsed back, truncated or foreign images refused"""
    host.cfrk_host_format_sparse2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
    host.cfrk_host_format_sparse2.restype = C.c_size_t
    host.cfrk_host_write_binary.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
    host.cfrk_host_write_binary.restype = C.c_size_t
    host.cfrk_host_read_binary.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p]
    lo = np.array([5, 2 ** 64 - 1, 0], np.uint64)
    hi = np.array([0, 0, 2 ** 61 + 3], np.uint64)
    cnt = np.array([1, 4000000000, 9], np.uint32)
    n = host.cfrk_host_format_sparse2(lo.ctypes.data, hi.ctypes.data, cnt.ctypes.data, 3, None, 0)
    buf = C.create_string_buffer(n)
    host.cfrk_host_format_sparse2(lo.ctypes.data, hi.ctypes.data, cnt.ctypes.data, 3, buf, n)
    assert buf.raw[:n] == b"0:5:1\n0:18446744073709551615:4000000000\n2305843009213693955:0:9\n"
    # keys_hi NULL: the one-word form
    n1 = host.cfrk_host_format_sparse2(lo.ctypes.data, None, cnt.ctypes.data, 3, None, 0)
    b1 = C.create_string_buffer(n1)
    host.cfrk_host_format_sparse2(lo.ctypes.data, None, cnt.ctypes.data, 3, b1, n1)
    assert b1.raw[:n1] == b"5:1\n18446744073709551615:4000000000\n0:9\n"
    for k, flags, use_hi in ((63, 1, True), (31, 0, False), (32, 1, False), (33, 0, True)):
        nb = host.cfrk_host_write_binary(k, flags, lo.ctypes.data, hi.ctypes.data if use_hi else None, cnt.ctypes.data, 3, None, 0)
        assert nb == 32 + 3 * (20 if k > 32 else 12)
        img = C.create_string_buffer(nb)
        assert host.cfrk_host_write_binary(k, flags, lo.ctypes.data, hi.ctypes.data if use_hi else None, cnt.ctypes.data, 3, img, nb) == nb
        assert img.raw[:8] == b"CFRKGLB1"
        kk, ff, nn = C.c_int(), C.c_int(), C.c_uint64()
        rlo, rhi, rc_ = np.zeros(3, np.uint64), np.ones(3, np.uint64), np.zeros(3, np.uint32)
        assert host.cfrk_host_read_binary(img.raw, nb, C.byref(kk), C.byref(ff), C.byref(nn), rlo.ctypes.data, rhi.ctypes.data, rc_.ctypes.data) == 0
        assert (kk.value, nn.value) == (k, 3) and (ff.value & 1) == flags and bool(ff.value & 2) == (k > 32)
        assert (rlo == lo).all() and (rc_ == cnt).all() and (rhi == (hi if k > 32 else 0)).all()
        assert host.cfrk_host_read_binary(img.raw, nb - 1, None, None, None, None, None, None) == -1     # truncated
        bad = bytearray(img.raw[:nb]); bad[32 + (16 if k > 32 else 8)] ^= 1                       # a count byte: the sum check
        assert host.cfrk_host_read_binary(bytes(bad), nb, None, None, None, None, None, None) == -1
    assert host.cfrk_host_read_binary(b"x" * 64, 64, None, None, None, None, None, None) == -1


def test_chunk_views_are_chunk_relative(host):
    raw = b"".join(b">r%d\n%s\n" % (i, b"ACGT" * (i + 1)) for i in range(5))
    b = Batch()
    assert host.cfrk_host_parse_fasta(raw, len(raw), 1, C.byref(b)) == 0
    data = C.POINTER(C.c_int8)(); length = C.POINTER(C.c_int32)(); nN = C.c_int64()
    start = (C.c_int64 * 2)()
    assert host.cfrk_host_chunk(C.byref(b), 2, 2, C.byref(data), start, C.byref(length), C.byref(nN)) == 0
    assert list(start) == [0, 13] and nN.value == 13 + 17 and length[0] == 12 and length[1] == 16
    assert data[0] == 0 and data[12] == -1
    assert host.cfrk_host_chunk(C.byref(b), 4, 2, C.byref(data), start, C.byref(length), C.byref(nN)) == -1
    host.cfrk_host_free_batch(C.byref(b))


@pytest.mark.parametrize("threads", [2, 3, 8, 64])
def test_dense_text_is_the_same_with_any_number_of_formatter_threads(host, threads):
    rng = np.random.default_rng(threads)
    nS, k = 37, 3
    f = rng.integers(-5, 2000, nS * 4 ** k).astype(np.int32)
    n = host.cfrk_host_format_dense(f.ctypes.data, nS, k, None, 0)
    one = C.create_string_buffer(n)
    host.cfrk_host_format_dense(f.ctypes.data, nS, k, one, n)
    assert host.cfrk_host_format_dense_mt(f.ctypes.data, nS, k, None, 0, threads) == n
    many = C.create_string_buffer(n)
    assert host.cfrk_host_format_dense_mt(f.ctypes.data, nS, k, many, n, threads) == n
    assert one.raw == many.raw
    # fewer rows than threads, and no rows at all
    assert host.cfrk_host_format_dense_mt(f.ctypes.data, 2, k, None, 0, threads) == \
        host.cfrk_host_format_dense(f.ctypes.data, 2, k, None, 0)
    assert host.cfrk_host_format_dense_mt(f.ctypes.data, 0, k, None, 0, threads) == 0
