"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (cfrk_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle.so")

ORC_COMPAT = 0x1
ORC_CANONICAL = 0x2
ORC_FLOAT_INDEX = 0x4


def build():
    src = [os.path.join(ORACLE_DIR, f) for f in ("cfrk_oracle.c", "cfrk_oracle.h", "Makefile")]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        p8, p32, p64, pu64 = (C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                              C.POINTER(C.c_uint64))
        L.orc_per_read_dense.argtypes = [p8, p64, p32, C.c_int64, C.c_int64, C.c_int, C.c_int, p32]
        L.orc_per_read_dense.restype = C.c_int
        L.orc_compute_index.argtypes = [p8, C.c_int64, C.c_int, C.c_int, p64]
        L.orc_compute_index.restype = None
        L.orc_global_count.argtypes = [p8, C.c_int64, C.c_int, C.c_int, C.POINTER(pu64),
                                       C.POINTER(pu64), C.POINTER(pu64)]
        L.orc_global_count.restype = C.c_int64
        L.orc_global_count_sorted.argtypes = [p8, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(pu64),
                                              C.POINTER(pu64), C.POINTER(pu64)]
        L.orc_global_count_sorted.restype = C.c_int64
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_digest.argtypes = [pu64, pu64, pu64, C.c_int64, C.c_int, pu64]
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_synth_reads.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_uint64,
                                      C.c_uint64, C.c_uint64, C.c_int, p8, p64, p32]
        L.orc_synth_digest.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, pu64]
        L.orc_synth_digest.restype = C.c_int
        L.orc_synth_digest_slices.argtypes = [C.c_int64, C.c_int, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint64,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, pu64]
        L.orc_synth_digest_slices.restype = C.c_int
        L.orc_format_cfrk.argtypes = [p32, C.c_int64, C.c_int, C.c_char_p, C.c_size_t]
        L.orc_format_cfrk.restype = C.c_size_t
        L.orc_encode_base.argtypes = [C.c_int]
        L.orc_encode_base.restype = C.c_int8
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def per_read_dense(data, start, length, k, flags):
    data = np.ascontiguousarray(data, np.int8)
    start = np.ascontiguousarray(start, np.int64)
    length = np.ascontiguousarray(length, np.int32)
    nS = len(length)
    freq = np.zeros(max(nS, 0) * 4 ** k, np.int32)
    rc = lib().orc_per_read_dense(_p(data, C.c_int8), _p(start, C.c_int64), _p(length, C.c_int32),
                                  len(data), nS, k, flags, _p(freq, C.c_int32))
    if rc != 0:
        raise ValueError(f"orc_per_read_dense rc={rc}")
    return freq.reshape(nS, 4 ** k)


def compute_index(data, k, float_index=False):
    data = np.ascontiguousarray(data, np.int8)
    out = np.zeros(len(data), np.int64)
    lib().orc_compute_index(_p(data, C.c_int8), len(data), k, int(float_index), _p(out, C.c_int64))
    return out


def global_count(data, k, flags=0, threads=0):
    """-> (keys_lo, keys_hi, counts) sorted by (hi, lo); keys_hi is zeros for k <= 32.
    threads = 0: the plain hash-table restatement; threads > 0: the partition + radix-sort
    variant on that many threads (same result, for the large cases)."""
    data = np.ascontiguousarray(data, np.int8)
    lo, hi, cnt = (C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)())
    if threads > 0:
        n = lib().orc_global_count_sorted(_p(data, C.c_int8), len(data), k, flags, threads,
                                          C.byref(lo), C.byref(hi), C.byref(cnt))
    else:
        n = lib().orc_global_count(_p(data, C.c_int8), len(data), k, flags,
                                   C.byref(lo), C.byref(hi), C.byref(cnt))
    if n < 0:
        raise ValueError(f"orc_global_count rc={n}")
    klo = np.ctypeslib.as_array(lo, (max(n, 1),))[:n].copy()
    kc = np.ctypeslib.as_array(cnt, (max(n, 1),))[:n].copy()
    khi = np.ctypeslib.as_array(hi, (max(n, 1),))[:n].copy()
    lib().orc_free(hi)
    lib().orc_free(lo)
    lib().orc_free(cnt)
    return klo, khi, kc


def digest(keys_lo, keys_hi, counts, two_word=False):
    klo = np.ascontiguousarray(keys_lo, np.uint64)
    khi = np.ascontiguousarray(keys_hi if keys_hi is not None else np.zeros_like(klo), np.uint64)
    cnt = np.ascontiguousarray(counts, np.uint64)
    out = np.zeros(4, np.uint64)
    lib().orc_digest(_p(klo, C.c_uint64), _p(khi, C.c_uint64), _p(cnt, C.c_uint64), len(klo),
                     int(two_word), _p(out, C.c_uint64))
    return tuple(int(x) for x in out)


def synth_reads(r0, R, L, Glen, seedG=1, seedR=2, seedS=3, uniform=False):
    data = np.empty(R * (L + 1), np.int8)
    start = np.empty(R, np.int64)
    length = np.empty(R, np.int32)
    lib().orc_synth_reads(r0, R, L, Glen, seedG, seedR, seedS, int(uniform),
                          _p(data, C.c_int8), _p(start, C.c_int64), _p(length, C.c_int32))
    return data, start, length


def synth_digest(R, L, Glen, k, flags=0, threads=1, slices=1, seedG=1, seedR=2, seedS=3, uniform=False,
                 progress=None):
    """Digest of the global count of reads [0, R) of the generator, in bounded memory (the reads are
    never materialised; the key space is counted in `slices` rounds).  progress(slice_done, slices):
    called between slices (the digest terms of the slices are added / xored here)."""
    if progress is None:
        out = np.zeros(4, np.uint64)
        rc = lib().orc_synth_digest(R, L, Glen, seedG, seedR, seedS, int(uniform), k, flags, threads, slices,
                                    _p(out, C.c_uint64))
        if rc != 0:
            raise ValueError(f"orc_synth_digest rc={rc}")
        return tuple(int(x) for x in out)
    M = (1 << 64) - 1
    acc = [0, 0, 0, 0]
    for s in range(slices):
        out = np.zeros(4, np.uint64)
        rc = lib().orc_synth_digest_slices(R, L, Glen, seedG, seedR, seedS, int(uniform), k, flags, threads,
                                           s, s + 1, slices, _p(out, C.c_uint64))
        if rc != 0:
            raise ValueError(f"orc_synth_digest_slices rc={rc}")
        o = [int(x) for x in out]
        acc = [(acc[0] + o[0]) & M, (acc[1] + o[1]) & M, (acc[2] + o[2]) & M, acc[3] ^ o[3]]
        progress(s + 1, slices)
    return tuple(acc)


def format_cfrk(freq, k):
    freq = np.ascontiguousarray(freq, np.int32)
    nS = freq.size // 4 ** k
    n = lib().orc_format_cfrk(_p(freq, C.c_int32), nS, k, None, 0)
    buf = C.create_string_buffer(n + 1)
    lib().orc_format_cfrk(_p(freq, C.c_int32), nS, k, buf, n)
    return buf.raw[:n]


def splitmix64(x):
    return int(lib().orc_splitmix64(x & 0xFFFFFFFFFFFFFFFF))
