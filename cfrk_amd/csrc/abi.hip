// abi.hip -- the C-ABI entry points of libcfrk_hip.so (include/cfrk_abi.h).
// Replaces kmer_main() (/root/reference/src/kmer_main.cu:20-128): context + persistent device
// pool instead of per-call cudaMalloc/cudaFree, error codes instead of printf/exit.
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>
#include <new>
#include <numeric>
#include <vector>

#include "msp.h"
#include "table.h"

int cfrk_fail(cfrk_ctx *ctx, int code, const char *fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return code;
}

int cfrk_pool_get(cfrk_ctx *ctx, int slot, size_t bytes, void **out) {
  cfrk_buf &b = ctx->pool[slot];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    if (b.p) { HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = (bytes + 255) & ~(size_t)255;
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
  }
  *out = b.p;
  return CFRK_OK;
}

extern "C" {

int cfrk_abi_version(void) { return CFRK_ABI_VERSION; }

const char *cfrk_strerror(int code) {
  switch (code) {
    case CFRK_OK: return "ok";
    case CFRK_ERR_ARG: return "invalid argument";
    case CFRK_ERR_NOMEM: return "out of device or pinned memory";
    case CFRK_ERR_HIP: return "HIP runtime error";
    case CFRK_ERR_STATE: return "call sequence violated";
    case CFRK_ERR_LAYOUT: return "data/start/length do not describe the struct-read layout";
    case CFRK_ERR_TABLE_FULL: return "global table overflowed (raise capacity_hint)";
    case CFRK_ERR_ALIGN: return "device pointer not 16-byte aligned";
    case CFRK_ERR_NO_DEVICE: return "no usable gfx950 device";
    case CFRK_ERR_SMALL_BUF: return "output buffer too small";
    case CFRK_ERR_COUNT_OVERFLOW: return "a key's count reached 2^32 - 2 and was saturated";
    case CFRK_ERR_RUNS_REFUSED: return "the CFRK_RUNS_ONLY add does not fit device memory in one pass";
    default: return "unknown error";
  }
}

const char *cfrk_last_error(const cfrk_ctx *ctx) { return ctx ? ctx->err : ""; }

int cfrk_device_count(int *count) {
  if (!count) return CFRK_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return CFRK_ERR_NO_DEVICE; }
  *count = n;
  return CFRK_OK;
}

int cfrk_ctx_create(int device, void *hip_stream, cfrk_ctx **out) {
  if (!out) return CFRK_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CFRK_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return CFRK_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return CFRK_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CFRK_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CFRK_ERR_NO_DEVICE;  // gfx950 code objects only
  cfrk_ctx *ctx = new (std::nothrow) cfrk_ctx();
  if (!ctx) return CFRK_ERR_NOMEM;
  memset(ctx, 0, sizeof *ctx);
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return CFRK_ERR_HIP; }
    ctx->own_stream = true;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipMalloc((void **)&ctx->g_stats, ST_NWORDS * sizeof(uint64_t)) != hipSuccess ||
      hipHostMalloc((void **)&ctx->h_stats, ST_NWORDS * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
    cfrk_ctx_destroy(ctx);
    return CFRK_ERR_HIP;
  }
  *out = ctx;
  return CFRK_OK;
}

static void global_release(cfrk_ctx *ctx) {
  if (ctx->g_keys_lo) (void)hipFree(ctx->g_keys_lo);
  if (ctx->g_keys_hi) (void)hipFree(ctx->g_keys_hi);
  if (ctx->g_counts) (void)hipFree(ctx->g_counts);
  ctx->g_keys_lo = ctx->g_keys_hi = nullptr;
  ctx->g_counts = nullptr;
  ctx->g_cap = 0;
  ctx->g_active = false;
  ctx->g_table_cleared = false;
}

void cfrk_ctx_destroy(cfrk_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  cfrk_msp_destroy(ctx);
  global_release(ctx);
  for (int i = 0; i < BUF_NSLOTS; ++i)
    if (ctx->pool[i].p) (void)hipFree(ctx->pool[i].p);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->g_stats) (void)hipFree(ctx->g_stats);
  if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
  if (ctx->h_runs) (void)hipHostFree(ctx->h_runs);
  for (int g = 0; g < CFRK_RUNS_MAX_GROUPS; ++g) if (ctx->runs_ev[g]) (void)hipEventDestroy(ctx->runs_ev[g]);
  if (ctx->stage_ev[0]) (void)hipEventDestroy(ctx->stage_ev[0]);
  if (ctx->stage_ev[1]) (void)hipEventDestroy(ctx->stage_ev[1]);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int cfrk_ctx_sync(cfrk_ctx *ctx) {
  if (!ctx) return CFRK_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

int cfrk_device_alloc(cfrk_ctx *ctx, size_t bytes, void **dptr) {
  if (!ctx || !dptr) return CFRK_ERR_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 16));
  return CFRK_OK;
}

int cfrk_device_free(cfrk_ctx *ctx, void *dptr) {
  if (!ctx) return CFRK_ERR_ARG;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (dptr) HIP_TRY(ctx, hipFree(dptr));
  return CFRK_OK;
}

int cfrk_memcpy_h2d(cfrk_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return CFRK_ERR_ARG;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

int cfrk_memcpy_d2h(cfrk_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return CFRK_ERR_ARG;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

int cfrk_memcpy_peer(cfrk_ctx *dst_ctx, void *dst, cfrk_ctx *src_ctx, const void *src, size_t bytes) {
  if (!dst_ctx || !src_ctx || (bytes && (!dst || !src))) return CFRK_ERR_ARG;
  if (bytes == 0) return CFRK_OK;
  // every failure is reported on dst_ctx: several owners (threads, each with its own dst_ctx) may copy from ONE
  // source context at the same time, and a context's error text has a single writer -- its own thread
  HIP_TRY(dst_ctx, hipSetDevice(src_ctx->device));
  HIP_TRY(dst_ctx, hipStreamSynchronize(src_ctx->stream));      // (draining a stream from several threads is safe)
  HIP_TRY(dst_ctx, hipSetDevice(dst_ctx->device));
  if (dst_ctx->device != src_ctx->device) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst_ctx->device, src_ctx->device) == hipSuccess && can) {
      const hipError_t e = hipDeviceEnablePeerAccess(src_ctx->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // the copy still works, staged by the runtime
      else (void)hipGetLastError();
    }
    HIP_TRY(dst_ctx, hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device, bytes, dst_ctx->stream));
  } else {
    HIP_TRY(dst_ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dst_ctx->stream));
  }
  return CFRK_OK;
}

/* ------------------------------------------------------------------ per-read dense */

static int dense_check(cfrk_ctx *ctx, int64_t nN, int64_t nS, int k) {
  if (!ctx) return CFRK_ERR_ARG;
  if (k < 1 || k > 15) return cfrk_fail(ctx, CFRK_ERR_ARG, "k=%d outside 1..15 (POW(k) is 1U<<2k in an int)", k);
  if (nN < 0 || nS < 0) return cfrk_fail(ctx, CFRK_ERR_ARG, "negative size");
  return CFRK_OK;
}

int cfrk_per_read_dense_device(cfrk_ctx *ctx, const int8_t *d_data, const int64_t *d_start,
                               const int32_t *d_length, int64_t nN, int64_t nS, int k, int flags,
                               int32_t *d_freq) {
  int rc = dense_check(ctx, nN, nS, k);
  if (rc) return rc;
  if (nS == 0) return CFRK_OK;
  if (!d_data || !d_start || !d_length || !d_freq) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return cfrk_launch_dense(ctx, d_data, d_start, d_length, nN, nS, k, flags, d_freq);
}

int cfrk_per_read_dense(cfrk_ctx *ctx, const int8_t *data, const int64_t *start,
                        const int32_t *length, int64_t nN, int64_t nS, int k, int flags,
                        int32_t *freq_out) {
  int rc = dense_check(ctx, nN, nS, k);
  if (rc) return rc;
  if (nS == 0) return CFRK_OK;
  if (!data || !start || !length || !freq_out) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t fourk = (size_t)1 << (2 * k);
  const size_t freq_bytes = (size_t)nS * fourk * sizeof(int32_t);
  // the reference's up-front estimate (src/kmer_main.cu:44-56), against free memory
  size_t free_b = 0, total_b = 0;
  HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  size_t need = (size_t)nN + (size_t)nS * 16 + freq_bytes;
  size_t have = free_b + ctx->pool[BUF_DATA].cap + ctx->pool[BUF_START].cap +
                ctx->pool[BUF_LENGTH].cap + ctx->pool[BUF_FREQ].cap + ctx->pool[BUF_SPILL].cap;
  if (need > have)
    return cfrk_fail(ctx, CFRK_ERR_NOMEM, "required %zu B, available %zu B", need, have);
  void *d_data, *d_start, *d_length, *d_freq;
  if ((rc = cfrk_pool_get(ctx, BUF_DATA, (size_t)nN + 64, &d_data))) return rc;
  if ((rc = cfrk_pool_get(ctx, BUF_START, (size_t)nS * 8, &d_start))) return rc;
  if ((rc = cfrk_pool_get(ctx, BUF_LENGTH, (size_t)nS * 4, &d_length))) return rc;
  if ((rc = cfrk_pool_get(ctx, BUF_FREQ, freq_bytes, &d_freq))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d_data, data, (size_t)nN, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_start, start, (size_t)nS * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_length, length, (size_t)nS * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = cfrk_launch_dense(ctx, (const int8_t *)d_data, (const int64_t *)d_start,
                         (const int32_t *)d_length, nN, nS, k, flags, (int32_t *)d_freq);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(freq_out, d_freq, freq_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

/* ------------------------------------------------------------------ global counting */

int cfrk_global_begin(cfrk_ctx *ctx, int k, int flags, uint64_t capacity_hint) {
  if (!ctx) return CFRK_ERR_ARG;
  if (k < 1 || k > 64) return cfrk_fail(ctx, CFRK_ERR_ARG, "k=%d outside 1..64", k);
  if ((flags & CFRK_RUNS_ONLY) && (k < 16 || k > 64 || (flags & CFRK_FORCE_HASH)))
    return cfrk_fail(ctx, CFRK_ERR_ARG, "CFRK_RUNS_ONLY needs a partitioned path (16 <= k <= 64)");
  if ((flags & CFRK_RUNS_DEFER) && !(flags & CFRK_RUNS_ONLY)) return cfrk_fail(ctx, CFRK_ERR_ARG, "CFRK_RUNS_DEFER goes with CFRK_RUNS_ONLY");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // Was the HBM table left untouched by the previous job (the partitioned path only writes it
  // on spill)?  Then it is still all-empty and the 12 B/slot clear below can be skipped.
  // (h_stats = pinned snapshot of the device flags, copied at the end of the last add.)
  const bool table_clean_prev = ctx->g_table_cleared && !cfrk_msp_table_written(ctx) &&
                                ctx->h_stats_valid && ctx->h_stats[ST_SPILLED] == 0;
  bool table_clean = table_clean_prev;
  ctx->h_stats_valid = false;
  cfrk_msp_reset(ctx);
  if (capacity_hint == 0) capacity_hint = 1ull << 24;
  // distinct keys cannot exceed 4^k
  if (k < 31) capacity_hint = std::min<uint64_t>(capacity_hint, 1ull << (2 * k));
  int lg = 10;
  while (lg < 40 && (1ull << lg) < capacity_hint * 2) ++lg;   // load factor <= 0.5 at the hint
  const uint64_t cap = 1ull << lg;
  const bool two = k > 32;
  if (cap != ctx->g_cap || two != ctx->g_two || (two && !ctx->g_keys_hi)) {
    global_release(ctx);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->g_keys_lo, cap * 8));
    if (two) HIP_TRY(ctx, hipMalloc((void **)&ctx->g_keys_hi, cap * 8));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->g_counts, cap * 4));
    ctx->g_cap = cap;
    table_clean = false;
  }
  if (two != ctx->g_two) table_clean = false;
  ctx->g_log2cap = lg;
  ctx->g_k = k;
  ctx->g_flags = flags;
  ctx->g_two = two;
  // one-word: empty = all-ones key.  two-word: the COUNT word is the slot state (0 = empty).
  if (!table_clean) {
    if (!two) HIP_TRY(ctx, hipMemsetAsync(ctx->g_keys_lo, 0xFF, cap * 8, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_counts, 0, cap * 4, ctx->stream));
  }
  ctx->g_table_cleared = true;
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats, 0, ST_NWORDS * 8, ctx->stream));
  ctx->g_active = true;
  ctx->ev_valid = false;
  return CFRK_OK;
}

int cfrk_global_add_device(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN) {
  if (!ctx) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "cfrk_global_add before cfrk_global_begin");
  if (nN < 0) return cfrk_fail(ctx, CFRK_ERR_ARG, "negative size");
  if (nN == 0) return CFRK_OK;
  if (!d_data) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  if (((uintptr_t)d_data & 15) != 0) return cfrk_fail(ctx, CFRK_ERR_ALIGN, "d_data %p", (const void *)d_data);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc;
  rc = CFRK_ERR_NOMEM;
  ctx->last_passes = 0;
  if (cfrk_radix_prefers(ctx, nN)) {
    rc = cfrk_radix_count(ctx, d_data, nN);
    if (rc == CFRK_ERR_NOMEM && cfrk_msp_usable(ctx)) { (void)hipGetLastError(); rc = cfrk_msp_count(ctx, d_data, nN); }   // (k = 16)
  } else if (cfrk_msp_usable(ctx)) rc = cfrk_msp_count(ctx, d_data, nN);
  else if (cfrk_msp2_usable(ctx)) rc = cfrk_msp2_count(ctx, d_data, nN);
  if (rc == CFRK_ERR_NOMEM && (cfrk_msp_usable(ctx) || cfrk_radix_usable(ctx) || cfrk_msp2_usable(ctx)) &&
      !(ctx->msp && ctx->msp->pending)) {     // (CFRK_INTERNAL_FLOOD: no second attempt, straight to the general path)
    // The pool only grows: buffers sized by earlier jobs of this context (another k, a larger batch) may be
    // what stands in the way.  Nothing of the partitioned paths is live between adds unless a result list is
    // pending: give their buffers back and plan the batch once more before giving the fast path up.
    static const int trim[] = {BUF_SCRATCH, BUF_MSP_L1, BUF_MSP_L2, BUF_MSP_OUTK, BUF_MSP_OUTC, BUF_MSP_AUX, BUF_MSP_OUTH,
                               BUF_MSP_ACCK, BUF_MSP_ACCH, BUF_MSP_ACCC, BUF_MSP_LAYOUT, BUF_MSP_OVF, BUF_MSP_OVF1,
                               BUF_MSP_LAYOUT1, BUF_EXPORT_LO, BUF_EXPORT_HI, BUF_EXPORT_CNT};
    size_t freed = 0;
    (void)hipGetLastError();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int slot : trim) {
      cfrk_buf &b = ctx->pool[slot];
      if (b.p) { freed += b.cap; (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    }
    if (freed) {
      if (cfrk_radix_prefers(ctx, nN)) rc = cfrk_radix_count(ctx, d_data, nN);
      if (rc == CFRK_ERR_NOMEM && cfrk_msp_usable(ctx)) { (void)hipGetLastError(); rc = cfrk_msp_count(ctx, d_data, nN); }
      else if (rc == CFRK_ERR_NOMEM && cfrk_msp2_usable(ctx)) rc = cfrk_msp2_count(ctx, d_data, nN);
    }
  }
  if ((rc == CFRK_ERR_NOMEM || rc == CFRK_INTERNAL_FLOOD) && (ctx->g_flags & CFRK_RUNS_ONLY))
    return cfrk_fail(ctx, CFRK_ERR_RUNS_REFUSED, rc == CFRK_ERR_NOMEM ? "the shard's record buffers do not fit device memory"
                                                                        : "2^32 records in one leaf stream: count without CFRK_RUNS_ONLY");
  if (rc == CFRK_INTERNAL_FLOOD) {
    // a single-key flood wrapped a 32-bit stream cursor before anything of this add was counted: the general path
    // (one saturating HBM atomic per occurrence) counts it instead
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CWRAP, 0, sizeof(uint64_t), ctx->stream));
    rc = CFRK_ERR_NOMEM;
  }
  if (rc == CFRK_ERR_NOMEM) {
    // no partitioned path for this k, or its record buffers (about 7 bytes per input byte) do
    // not fit next to the caller's data: count with the general HBM-table path instead
    (void)hipGetLastError();
    rc = cfrk_msp_flush_to_table(ctx);
    if (rc) return rc;
    cfrk_msp_note_table_write(ctx);
    rc = cfrk_hash_count(ctx, d_data, nN);
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->ev_valid = true;
  // snapshot of the flags for the next begin(); lands before its stream synchronisation
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_stats, ctx->g_stats, ST_NWORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  ctx->h_stats_valid = true;
  return CFRK_OK;
}

// background check of the struct-read layout (joined by failed() or by the destructor: the function
// below leaves early on HIP errors)
struct LayoutCheck {
  std::vector<std::thread> th;
  std::vector<int64_t> bad;
  std::vector<int> why;
  const int8_t *data = nullptr; const int64_t *start = nullptr; const int32_t *length = nullptr;
  int64_t nN = 0, nS = 0;
  void begin(const int8_t *d, const int64_t *s, const int32_t *l, int64_t nn, int64_t ns) {
    data = d; start = s; length = l; nN = nn; nS = ns;
    const int64_t piece = 1 << 20;
    unsigned nt = (unsigned)std::min<int64_t>((nS + piece - 1) / piece, 8);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if (nt == 0) nt = 1;
    bad.assign(nt, -1);
    why.assign(nt, 0);
    auto piece_fn = [this](unsigned t, unsigned nt) {
      const int64_t i0 = nS * t / nt, i1 = nS * (t + 1) / nt;
      for (int64_t i = i0; i < i1; ++i) {
        // A thread's first read is compared with a predecessor another thread validates: every
        // term is range-checked before it is used, so a corrupt table (negative starts, sums that
        // overflow) never turns into an out-of-bounds read of data[].
        int w = 0;
        if (start[i] < 0 || start[i] > nN || length[i] < 0) w = 1;
        else if (i && (start[i - 1] < 0 || start[i - 1] > nN || length[i - 1] < 0 ||
                       start[i] != start[i - 1] + (int64_t)length[i - 1] + 1)) w = 1;
        else if (!i && start[i] != 0) w = 1;
        else if ((int64_t)length[i] + 1 > nN - start[i]) w = 2;
        else { const int8_t term = data[start[i] + length[i]]; if (term >= 0 && term <= 3) w = 3; }
        if (w) { bad[t] = i; why[t] = w; return; }
      }
    };
    // std::thread may throw (resource exhaustion): nothing may cross the extern "C" boundary, so
    // the pieces no thread could be started for are checked right here
    unsigned started = 0;
    try {
      for (; started < nt; ++started) th.emplace_back(piece_fn, started, nt);
    } catch (...) {
      for (unsigned t = started; t < nt; ++t) piece_fn(t, nt);
    }
  }
  void join() { for (auto &x : th) if (x.joinable()) x.join(); }
  // true (and the context's error text set) when the layout is not the reference's
  bool failed(cfrk_ctx *ctx) {
    if (!start) return false;
    join();
    for (size_t t = 0; t < bad.size(); ++t) {
      if (bad[t] < 0) continue;
      const int64_t i = bad[t];
      // (the predecessor's fields may themselves be garbage: wrap-around arithmetic, text only)
      const int64_t want = i ? (int64_t)((uint64_t)start[i - 1] + (uint64_t)(int64_t)length[i - 1] + 1u) : 0;
      if (why[t] == 1) cfrk_fail(ctx, CFRK_ERR_LAYOUT, "read %lld: start %lld, expected %lld", (long long)i, (long long)start[i], (long long)want);
      else if (why[t] == 2) cfrk_fail(ctx, CFRK_ERR_LAYOUT, "read %lld runs past nN", (long long)i);
      else cfrk_fail(ctx, CFRK_ERR_LAYOUT, "read %lld has no terminator", (long long)i);
      return true;
    }
    const int64_t pos = nS ? start[nS - 1] + (int64_t)length[nS - 1] + 1 : 0;
    if (pos != nN) { cfrk_fail(ctx, CFRK_ERR_LAYOUT, "sum(length)+nS = %lld but nN = %lld", (long long)pos, (long long)nN); return true; }
    return false;
  }
  ~LayoutCheck() { join(); }
};

int cfrk_global_add(cfrk_ctx *ctx, const int8_t *data, const int64_t *start, const int32_t *length,
                    int64_t nN, int64_t nS) {
  if (!ctx) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "cfrk_global_add before cfrk_global_begin");
  if (nN < 0 || nS < 0) return cfrk_fail(ctx, CFRK_ERR_ARG, "negative size");
  if (nN == 0) return CFRK_OK;
  if (!data) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  // struct-read layout (src/fastaIO.h:74-102, src/main.cu:195-200): read i occupies
  // [start[i], start[i]+length[i]) and is followed by one terminator byte.  Every read is checked
  // against its predecessor, so ranges of reads are independent: up to eight threads check them
  // WHILE the batch is staged and copied below (10^7 reads: 16 ms of cache misses that used to come
  // first); nothing is counted before they have all passed.
  LayoutCheck lc;
  if (start && length) lc.begin(data, start, length, nN, nS);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  void *d_data;
  int rc;
  // the previous add may still be reading BUF_DATA
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = cfrk_pool_get(ctx, BUF_DATA, (size_t)nN + 64, &d_data))) return rc;
  // The runtime's own path for pageable memory moves 46 GB/s here (it pins the caller's pages chunk
  // by chunk); a bounce buffer of our own, filled by one thread, made 26-32 GB/s, filled by eight 45.
  HIP_TRY(ctx, hipMemcpyAsync(d_data, data, (size_t)nN, hipMemcpyHostToDevice, ctx->stream));
  if (lc.failed(ctx)) return CFRK_ERR_LAYOUT;
  return cfrk_global_add_device(ctx, (const int8_t *)d_data, nN);
}

int cfrk_global_merge_device(cfrk_ctx *ctx, const uint64_t *d_lo, const uint64_t *d_hi,
                             const uint32_t *d_cnt, int64_t n) {
  if (!ctx) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge before begin");
  if (n < 0) return cfrk_fail(ctx, CFRK_ERR_ARG, "negative size");
  if (n == 0) return CFRK_OK;
  if (!d_lo || !d_cnt || (ctx->g_two && !d_hi)) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = cfrk_msp_flush_to_table(ctx);
  if (rc) return rc;
  cfrk_msp_note_table_write(ctx);
  return cfrk_hash_merge(ctx, d_lo, d_hi, d_cnt, n);
}

int cfrk_global_finish(cfrk_ctx *ctx, uint64_t *n_distinct) {
  uint64_t d[4] = {0, 0, 0, 0};
  int rc = cfrk_global_digest(ctx, d);
  if (rc && rc != CFRK_ERR_COUNT_OVERFLOW) return rc;
  if (n_distinct) *n_distinct = d[0];     // (a saturated result is complete: every key is there, some counts are CFRK_COUNT_MAX)
  return rc;
}

int cfrk_global_digest(cfrk_ctx *ctx, uint64_t out[4]) {
  if (!ctx || !out) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "digest before begin");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ResultSrc src;
  bool use_list = false;
  int rc = cfrk_msp_resolve(ctx, &src, &use_list);
  if (rc) return rc;
  uint64_t st[ST_NWORDS];
  if ((rc = cfrk_result_scan(ctx, use_list ? &src : nullptr, st))) return rc;
  if (st[ST_OVERFLOW]) return cfrk_fail(ctx, CFRK_ERR_TABLE_FULL, "table of %llu slots overflowed", (unsigned long long)ctx->g_cap);
  out[0] = st[ST_DIG0]; out[1] = st[ST_DIG1]; out[2] = st[ST_DIG2]; out[3] = st[ST_DIG3];
  if (st[ST_SAT]) return cfrk_fail(ctx, CFRK_ERR_COUNT_OVERFLOW, "a key occurred 2^32 - 2 times or more: its count is held at 0xFFFFFFFE");
  return CFRK_OK;
}

int cfrk_global_export_device(cfrk_ctx *ctx, uint64_t *d_lo, uint64_t *d_hi, uint32_t *d_cnt,
                              uint64_t cap, int parts, uint64_t *part_counts) {
  if (!ctx || !part_counts || parts < 1 || parts > 1024) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "export before begin");
  if (cap && (!d_lo || !d_cnt || (ctx->g_two && !d_hi))) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ResultSrc src;
  bool use_list = false;
  int rc = cfrk_msp_resolve(ctx, &src, &use_list);
  if (rc) return rc;
  return cfrk_result_export(ctx, use_list ? &src : nullptr, d_lo, d_hi, d_cnt, cap, parts, part_counts);
}

int cfrk_global_export(cfrk_ctx *ctx, uint64_t *keys_lo, uint64_t *keys_hi, uint32_t *counts,
                       uint64_t cap, uint64_t *n_out) {
  if (!ctx || !n_out) return CFRK_ERR_ARG;
  uint64_t n = 0;
  int rc = cfrk_global_finish(ctx, &n);
  const bool saturated = rc == CFRK_ERR_COUNT_OVERFLOW;   // the result is exported all the same, the code returned at the end
  if (rc && !saturated) return rc;
  *n_out = n;
  if (n > cap) return cfrk_fail(ctx, CFRK_ERR_SMALL_BUF, "%llu entries, room for %llu", (unsigned long long)n, (unsigned long long)cap);
  if (n == 0) return CFRK_OK;
  if (!keys_lo || !counts) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  void *d_lo, *d_hi = nullptr, *d_cnt;
  if ((rc = cfrk_pool_get(ctx, BUF_EXPORT_LO, n * 8, &d_lo))) return rc;
  if (ctx->g_two && (rc = cfrk_pool_get(ctx, BUF_EXPORT_HI, n * 8, &d_hi))) return rc;
  if ((rc = cfrk_pool_get(ctx, BUF_EXPORT_CNT, n * 4, &d_cnt))) return rc;
  uint64_t pc = 0;
  rc = cfrk_global_export_device(ctx, (uint64_t *)d_lo, (uint64_t *)d_hi, (uint32_t *)d_cnt, n, 1, &pc);
  if (rc) return rc;
  // sorted on the device (export_sort.hip), then copied straight into the caller's buffers
  const uint64_t *s_lo, *s_hi;
  const uint32_t *s_cnt;
  if ((rc = cfrk_sort_export(ctx, (const uint64_t *)d_lo, ctx->g_two ? (const uint64_t *)d_hi : nullptr,
                             (const uint32_t *)d_cnt, n, &s_lo, &s_hi, &s_cnt))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(keys_lo, s_lo, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (keys_hi) {
    if (s_hi) HIP_TRY(ctx, hipMemcpyAsync(keys_hi, s_hi, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    else memset(keys_hi, 0, n * 8);
  }
  HIP_TRY(ctx, hipMemcpyAsync(counts, s_cnt, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (saturated) return cfrk_fail(ctx, CFRK_ERR_COUNT_OVERFLOW, "a key occurred 2^32 - 2 times or more: its count is held at 0xFFFFFFFE");
  return CFRK_OK;
}

int cfrk_global_last_add_ms(cfrk_ctx *ctx, float *ms) {
  if (!ctx || !ms) return CFRK_ERR_ARG;
  if (!ctx->ev_valid) return cfrk_fail(ctx, CFRK_ERR_STATE, "no add recorded");
  HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
  HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return CFRK_OK;
}

/* ------------------------------------------------------------------ synthetic reads */

int cfrk_synth_reads_device(cfrk_ctx *ctx, int64_t r0, int64_t R, int L, int64_t Glen,
                            uint64_t seedG, uint64_t seedR, uint64_t seedS, int uniform,
                            int8_t *d_data, int64_t *d_start, int32_t *d_length) {
  if (!ctx) return CFRK_ERR_ARG;
  if (r0 < 0 || R < 0 || L < 1 || (!uniform && Glen < L)) return cfrk_fail(ctx, CFRK_ERR_ARG, "bad generator parameters");
  if (R == 0) return CFRK_OK;
  if (!d_data) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return cfrk_launch_synth(ctx, r0, R, L, Glen, seedG, seedR, seedS, uniform, d_data, d_start, d_length);
}

}  // extern "C"
