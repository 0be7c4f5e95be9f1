"""ctypes binding of libcfrk_hip.so and the host-side mirror of the reference interface.

Reference interface mirrored here (paths under /root/reference/):
  struct read { char *data; int *length; lint *start; int *Freq; }      src/tipos.h:23-30
  void kmer_main(struct read *rd, lint nN, lint nS, int k, ushort device) src/kmer_main.cu:20
"""
import ctypes as C
import os
import re

import numpy as np

CFRK_COMPAT = 0x1
CFRK_CANONICAL = 0x2
CFRK_FORCE_HASH = 0x4
CFRK_RUNS_ONLY = 0x8
CFRK_FLOAT_INDEX = 0x10
CFRK_RUNS_DEFER = 0x20
CFRK_DEBUG_FORCE_RT_OVERFLOW = 0x1   # cfrk_debug_set_flags
CFRK_DEBUG_SMALL_WAVE_CAP = 0x2
CFRK_DEBUG_NO_ANCHORS = 0x4
CFRK_DEBUG_RECORD_SUBSETS = 0x8
CFRK_DEBUG_NO_PIPELINE = 0x10
CFRK_DEBUG_SMALL_PIPELINE = 0x20
CFRK_DEBUG_NO_RADIX16 = 0x40
CFRK_DEBUG_NO_SMALL_LEAVES = 0x80
CFRK_ERR_COUNT_OVERFLOW = -10        # finish / digest / export: some count was held at CFRK_COUNT_MAX
CFRK_ERR_RUNS_REFUSED = -11          # a CFRK_RUNS_ONLY add that needs more than one pass
CFRK_COUNT_MAX = 0xFFFFFFFE
CFRK_PARAM_MSP_CHUNKS, CFRK_PARAM_L2_SLACK_COMPLETE, CFRK_PARAM_L2_SLACK_TRUNCATED, CFRK_PARAM_MSP2_SUBVALUE_BITS = 0, 1, 2, 3   # cfrk_debug_set_param

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcfrk_hip.so")
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "cfrk_abi.h")

_lib = None


class CfrkError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"{what}: {detail}" if detail else what)


def library_path():
    return _SO


def abi_symbols():
    """every function include/cfrk_abi.h declares"""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfrk_[a-z0-9_]+)\s*\(", text)))


def load_library():
    """dlopen libcfrk_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise CfrkError(-100, "libcfrk_hip.so is missing",
                        "build it with `make -C cfrk_amd/csrc` or __graft_entry__.build(); "
                        "there is no CPU fallback")
    L = C.CDLL(_SO)
    vp, i64, i32, u64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint64
    sig = {
        "cfrk_abi_version": ([], C.c_int),
        "cfrk_strerror": ([C.c_int], C.c_char_p),
        "cfrk_last_error": ([vp], C.c_char_p),
        "cfrk_device_count": ([C.POINTER(C.c_int)], C.c_int),
        "cfrk_ctx_create": ([C.c_int, vp, C.POINTER(vp)], C.c_int),
        "cfrk_ctx_destroy": ([vp], None),
        "cfrk_ctx_sync": ([vp], C.c_int),
        "cfrk_device_alloc": ([vp, C.c_size_t, C.POINTER(vp)], C.c_int),
        "cfrk_device_free": ([vp, vp], C.c_int),
        "cfrk_memcpy_h2d": ([vp, vp, vp, C.c_size_t], C.c_int),
        "cfrk_memcpy_d2h": ([vp, vp, vp, C.c_size_t], C.c_int),
        "cfrk_memcpy_peer": ([vp, vp, vp, vp, C.c_size_t], C.c_int),
        "cfrk_per_read_dense": ([vp, vp, vp, vp, i64, i64, i32, i32, vp], C.c_int),
        "cfrk_per_read_dense_device": ([vp, vp, vp, vp, i64, i64, i32, i32, vp], C.c_int),
        "cfrk_global_begin": ([vp, i32, i32, u64], C.c_int),
        "cfrk_global_add": ([vp, vp, vp, vp, i64, i64], C.c_int),
        "cfrk_global_add_device": ([vp, vp, i64], C.c_int),
        "cfrk_global_merge_device": ([vp, vp, vp, vp, i64], C.c_int),
        "cfrk_global_finish": ([vp, C.POINTER(u64)], C.c_int),
        "cfrk_global_export": ([vp, vp, vp, vp, u64, C.POINTER(u64)], C.c_int),
        "cfrk_global_export_device": ([vp, vp, vp, vp, u64, C.c_int, C.POINTER(u64)], C.c_int),
        "cfrk_global_digest": ([vp, C.POINTER(u64)], C.c_int),
        "cfrk_global_last_add_ms": ([vp, C.POINTER(C.c_float)], C.c_int),
        "cfrk_global_leaves_per_part": ([C.c_int], C.c_int),
        "cfrk_global_export_leaves_device": ([vp, vp, vp, vp, u64, C.c_int, C.POINTER(u64), vp], C.c_int),
        "cfrk_global_merge_leaves_device": ([vp, vp, vp, vp, C.POINTER(u64), vp, C.c_int], C.c_int),
        "cfrk_global_export_runs_device": ([vp, vp, u64, C.c_int, C.POINTER(u64)], C.c_int),
        "cfrk_global_merge_runs_device": ([vp, vp, C.POINTER(u64), C.c_int], C.c_int),
        "cfrk_global_export_runs_async": ([vp, vp, u64, C.c_int, C.c_int], C.c_int),
        "cfrk_global_export_runs_wait": ([vp, C.c_int, C.POINTER(u64)], C.c_int),
        "cfrk_global_merge_runs_group_device": ([vp, vp, C.POINTER(u64), C.c_int, C.c_int, C.c_int], C.c_int),
        "cfrk_global_runs_group_ms": ([vp, C.c_int, C.POINTER(C.c_float)], C.c_int),
        "cfrk_debug_msp_info": ([vp, C.POINTER(u64)], C.c_int),
        "cfrk_debug_set_mem_budget": ([vp, u64], C.c_int),
        "cfrk_debug_device_bytes": ([vp, C.POINTER(u64)], C.c_int),
        "cfrk_debug_set_flags": ([vp, C.c_uint32], C.c_int),
        "cfrk_debug_set_param": ([vp, C.c_int, C.c_double], C.c_int),
        "cfrk_debug_last_add_passes": ([vp, C.POINTER(C.c_int)], C.c_int),
        "cfrk_synth_reads_device": ([vp, i64, i64, i32, i64, u64, u64, u64, i32, vp, vp, vp], C.c_int),
    }
    for name, (args, res) in sig.items():
        f = getattr(L, name, None)
        if f is None:       # an OLDER build swapped in by tools/ab.sh: the entry point raises when it is called
            continue        # (tests/test_abi_cpu.py checks that the product exports every symbol the header declares)
        f.argtypes = args
        f.restype = res
    _lib = L
    return L


def device_count():
    """usable devices (hipGetDeviceCount through the C ABI); raises when the library is missing"""
    n = C.c_int(0)
    L = load_library()
    rc = L.cfrk_device_count(C.byref(n))
    if rc != 0:
        raise CfrkError(rc, "cfrk_device_count", L.cfrk_strerror(rc).decode())
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP stream + persistent device pool on one GPU (cfrk_ctx)."""

    def __init__(self, device=0, stream=None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.cfrk_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise CfrkError(rc, "cfrk_ctx_create", self._L.cfrk_strerror(rc).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.cfrk_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, rc, what):
        if rc != 0:
            raise CfrkError(rc, f"{what}: {self._L.cfrk_strerror(rc).decode()}",
                            self._L.cfrk_last_error(self._h).decode())

    def sync(self):
        self.check(self._L.cfrk_ctx_sync(self._h), "cfrk_ctx_sync")

    def device_bytes(self):
        """device memory the context holds (pool + global table), without the caller's buffers"""
        n = C.c_uint64()
        self.check(self._L.cfrk_debug_device_bytes(self._h, C.byref(n)), "cfrk_debug_device_bytes")
        return n.value

    # -- raw device buffers --------------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self._L.cfrk_device_alloc(self._h, nbytes, C.byref(p)), "cfrk_device_alloc")
        return p.value

    def free(self, dptr):
        self.check(self._L.cfrk_device_free(self._h, C.c_void_p(dptr)), "cfrk_device_free")

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self._L.cfrk_memcpy_h2d(self._h, C.c_void_p(dptr), _ptr(arr), arr.nbytes), "cfrk_memcpy_h2d")

    def d2h(self, arr, dptr):
        self.check(self._L.cfrk_memcpy_d2h(self._h, _ptr(arr), C.c_void_p(dptr), arr.nbytes), "cfrk_memcpy_d2h")

    # -- per-read dense (kmer_main) --------------------------------------------------------
    def per_read_dense(self, data, start, length, k, flags=CFRK_COMPAT):
        data = np.ascontiguousarray(data, np.int8)
        start = np.ascontiguousarray(start, np.int64)
        length = np.ascontiguousarray(length, np.int32)
        nS = len(length)
        if len(start) != nS:
            raise ValueError("start and length differ in size")
        freq = np.empty(nS * 4 ** k if 1 <= k <= 15 else 0, np.int32)
        self.check(self._L.cfrk_per_read_dense(self._h, _ptr(data), _ptr(start), _ptr(length),
                                               len(data), nS, k, flags, _ptr(freq)),
                   "cfrk_per_read_dense")
        return freq.reshape(nS, -1) if nS else freq.reshape(0, 4 ** k)

    def synth_reads_device(self, r0, R, L, Glen, d_data, d_start=None, d_length=None,
                           seedG=1, seedR=2, seedS=3, uniform=False):
        self.check(self._L.cfrk_synth_reads_device(self._h, r0, R, L, Glen, seedG, seedR, seedS,
                                                   int(uniform), C.c_void_p(d_data),
                                                   C.c_void_p(d_start) if d_start else None,
                                                   C.c_void_p(d_length) if d_length else None),
                   "cfrk_synth_reads_device")


class GlobalCounter:
    """Global (all-reads) k-mer counts: begin / add* / finish / export."""

    def __init__(self, ctx, k, flags=0, capacity_hint=0):
        self.ctx, self.k, self.flags = ctx, k, flags
        self._L = ctx._L
        ctx.check(self._L.cfrk_global_begin(ctx._h, k, flags, capacity_hint), "cfrk_global_begin")

    def add(self, data, start=None, length=None):
        data = np.ascontiguousarray(data, np.int8)
        if start is not None:
            start = np.ascontiguousarray(start, np.int64)
            length = np.ascontiguousarray(length, np.int32)
        nS = 0 if length is None else len(length)
        self.ctx.check(self._L.cfrk_global_add(self.ctx._h, _ptr(data), _ptr(start), _ptr(length),
                                               len(data), nS), "cfrk_global_add")

    def add_device(self, d_data, nN):
        self.ctx.check(self._L.cfrk_global_add_device(self.ctx._h, C.c_void_p(d_data), nN),
                       "cfrk_global_add_device")

    def merge_device(self, d_lo, d_hi, d_cnt, n):
        self.ctx.check(self._L.cfrk_global_merge_device(self.ctx._h, C.c_void_p(d_lo),
                                                        C.c_void_p(d_hi) if d_hi else None,
                                                        C.c_void_p(d_cnt), n),
                       "cfrk_global_merge_device")

    def finish(self, allow_saturated=False):
        n = C.c_uint64()
        rc = self._L.cfrk_global_finish(self.ctx._h, C.byref(n))
        if not (allow_saturated and rc == CFRK_ERR_COUNT_OVERFLOW):
            self.ctx.check(rc, "cfrk_global_finish")
        return n.value

    def digest(self, allow_saturated=False):
        """allow_saturated: a result with counts held at CFRK_COUNT_MAX (CFRK_ERR_COUNT_OVERFLOW) is returned
        instead of raised -- the digest is that of the saturated result"""
        out = (C.c_uint64 * 4)()
        rc = self._L.cfrk_global_digest(self.ctx._h, out)
        if not (allow_saturated and rc == CFRK_ERR_COUNT_OVERFLOW):
            self.ctx.check(rc, "cfrk_global_digest")
        return tuple(int(x) for x in out)

    def leaves_per_part(self, parts):
        return int(self._L.cfrk_global_leaves_per_part(parts))

    def export_leaves_device(self, d_keys, d_cnt, cap, parts, d_leaf_counts, d_keys_hi=0):
        """per-leaf owner export (d_keys_hi: high key words, k > 32); raises CfrkError(code -4)
        when the result is not in leaf form"""
        pc = (C.c_uint64 * parts)()
        self.ctx.check(self._L.cfrk_global_export_leaves_device(self.ctx._h, C.c_void_p(d_keys),
                                                                C.c_void_p(d_keys_hi) if d_keys_hi else None,
                                                                C.c_void_p(d_cnt), cap, parts, pc,
                                                                C.c_void_p(d_leaf_counts)),
                       "cfrk_global_export_leaves_device")
        return [int(x) for x in pc]

    def merge_leaves_device(self, d_keys, d_cnt, recv_counts, d_leaf_counts, d_keys_hi=0):
        parts = len(recv_counts)
        rc = (C.c_uint64 * parts)(*[int(x) for x in recv_counts])
        self.ctx.check(self._L.cfrk_global_merge_leaves_device(self.ctx._h, C.c_void_p(d_keys),
                                                               C.c_void_p(d_keys_hi) if d_keys_hi else None,
                                                               C.c_void_p(d_cnt), rc,
                                                               C.c_void_p(d_leaf_counts), parts),
                       "cfrk_global_merge_leaves_device")

    def export_runs_device(self, d_packed, cap_rows, parts):
        """CFRK_RUNS_ONLY job: per-leaf deduplicated runs, one packed segment per owner -> rows per
        segment; raises CfrkError(code -4) when the shard's runs are not all in the leaf streams"""
        pr = (C.c_uint64 * parts)()
        self.ctx.check(self._L.cfrk_global_export_runs_device(self.ctx._h, C.c_void_p(d_packed), cap_rows, parts, pr),
                       "cfrk_global_export_runs_device")
        return [int(x) for x in pr]

    def merge_runs_device(self, d_packed, recv_rows):
        parts = len(recv_rows)
        rr = (C.c_uint64 * parts)(*[int(x) for x in recv_rows])
        self.ctx.check(self._L.cfrk_global_merge_runs_device(self.ctx._h, C.c_void_p(d_packed), rr, parts),
                       "cfrk_global_merge_runs_device")

    # -- pipelined runs exchange (CFRK_RUNS_ONLY | CFRK_RUNS_DEFER jobs; one-word keys) ----------------------
    def export_runs_async(self, d_packed, seg_cap_rows, parts, ngroups):
        """enqueue deduplication + packing of every group of leaves into the send buffer; returns at once"""
        self.ctx.check(self._L.cfrk_global_export_runs_async(self.ctx._h, C.c_void_p(d_packed), seg_cap_rows, parts, ngroups),
                       "cfrk_global_export_runs_async")
        self._runs_parts = parts

    def export_runs_wait(self, group):
        """wait for group `group` only -> rows per owner segment; CfrkError -9 (segment too small) / -4 (overflow, spill)"""
        pr = (C.c_uint64 * self._runs_parts)()
        self.ctx.check(self._L.cfrk_global_export_runs_wait(self.ctx._h, group, pr), "cfrk_global_export_runs_wait")
        return [int(x) for x in pr]

    def merge_runs_group_device(self, d_recv, recv_rows, group, ngroups):
        """owner: count group `group` from the received segments, read in place; enqueued, no host synchronisation"""
        parts = len(recv_rows)
        rr = (C.c_uint64 * parts)(*[int(x) for x in recv_rows])
        self.ctx.check(self._L.cfrk_global_merge_runs_group_device(self.ctx._h, C.c_void_p(d_recv), rr, parts, group, ngroups),
                       "cfrk_global_merge_runs_group_device")

    def runs_group_ms(self, group):
        ms = C.c_float()
        self.ctx.check(self._L.cfrk_global_runs_group_ms(self.ctx._h, group, C.byref(ms)), "cfrk_global_runs_group_ms")
        return ms.value

    def msp_info(self):
        out = (C.c_uint64 * 9)()
        self.ctx.check(self._L.cfrk_debug_msp_info(self.ctx._h, out), "cfrk_debug_msp_info")
        names = ("l1_records", "l1_max_bin", "l1_cap", "l2_records", "l2_max_leaf", "l2_cap",
                 "spilled_records", "spilled_kmers", "list_entries")
        return dict(zip(names, (int(x) for x in out)))

    def set_mem_budget(self, nbytes):
        self.ctx.check(self._L.cfrk_debug_set_mem_budget(self.ctx._h, int(nbytes)), "cfrk_debug_set_mem_budget")

    def set_debug_flags(self, flags):
        self.ctx.check(self._L.cfrk_debug_set_flags(self.ctx._h, int(flags)), "cfrk_debug_set_flags")

    def set_debug_param(self, which, value):
        self.ctx.check(self._L.cfrk_debug_set_param(self.ctx._h, int(which), float(value)), "cfrk_debug_set_param")

    def last_add_passes(self):
        n = C.c_int()
        self.ctx.check(self._L.cfrk_debug_last_add_passes(self.ctx._h, C.byref(n)), "cfrk_debug_last_add_passes")
        return n.value

    def last_add_ms(self):
        ms = C.c_float()
        self.ctx.check(self._L.cfrk_global_last_add_ms(self.ctx._h, C.byref(ms)), "cfrk_global_last_add_ms")
        return ms.value

    def export(self, allow_saturated=False):
        """-> (keys_lo, keys_hi, counts) sorted by (hi, lo)"""
        n = self.finish(allow_saturated)
        lo = np.empty(n, np.uint64)
        hi = np.empty(n, np.uint64)
        cnt = np.empty(n, np.uint32)
        got = C.c_uint64()
        rc = self._L.cfrk_global_export(self.ctx._h, _ptr(lo), _ptr(hi), _ptr(cnt), n, C.byref(got))
        if not (allow_saturated and rc == CFRK_ERR_COUNT_OVERFLOW):
            self.ctx.check(rc, "cfrk_global_export")
        assert got.value == n
        return lo, hi, cnt

    def export_device(self, d_lo, d_hi, d_cnt, cap, parts=1):
        pc = (C.c_uint64 * parts)()
        self.ctx.check(self._L.cfrk_global_export_device(self.ctx._h, C.c_void_p(d_lo),
                                                         C.c_void_p(d_hi) if d_hi else None,
                                                         C.c_void_p(d_cnt), cap, parts, pc),
                       "cfrk_global_export_device")
        return [int(x) for x in pc]


class Read:
    """Mirror of `struct read` (src/tipos.h:23-30): data / length / start in, Freq out."""

    def __init__(self, data, length, start):
        self.data = np.ascontiguousarray(data, np.int8)
        self.length = np.ascontiguousarray(length, np.int32)
        self.start = np.ascontiguousarray(start, np.int64)
        self.Freq = None


_default_ctx = {}


def kmer_main(rd, nN, nS, k, device=0, flags=CFRK_COMPAT):
    """Drop-in mirror of kmer_main (src/kmer_main.cu:20): fills rd.Freq (nS x 4^k int32).

    Same argument meaning as the reference; errors raise CfrkError instead of printing and
    continuing (src/kmer_main.cu:59-63) or exit(1) (src/kmer_main.cu:51-56)."""
    if nN != len(rd.data) or nS != len(rd.length):
        raise ValueError("nN / nS do not match the buffers")
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    rd.Freq = ctx.per_read_dense(rd.data, rd.start, rd.length, k, flags).reshape(-1)
    return rd
