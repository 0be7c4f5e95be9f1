// global_hash.hip -- global k-mer counting into one open-addressing table in HBM.
//
// Semantics: the guarded ComputeFreq of the reference (/root/reference/src/kmer_kernel.cu:52-70)
// summed over reads: a window counts iff its k codes are all valid (ComputeIndex's -1 rule,
// src/kmer_kernel.cu:36-46).  The reference keeps nS dense rows of 4^k ints
// (src/kmer_main.cu:47); this keeps one sparse table, so k up to 64 is representable.
//
// This is the general path and the spill target of the minimizer-partitioned fast path
// (msp.hip): every occurrence costs one HBM atomic, so it is bound by the chip's scattered
// atomic rate, not by HBM bandwidth.
//
// Table (SoA, capacity 2^n, linear probing):
//   k <= 32: keys_lo[] u64 (empty = all ones), counts[] u32.  Claim = 64-bit CAS on the key;
//            the all-ones key itself (k = 32, TTT...T, non-canonical) is counted in a side word.
//   k  > 32: keys_lo[], keys_hi[] u64, counts[] u32 doubles as the slot state
//            (0 empty, 0xFFFFFFFF locked, else count): CAS 0->locked, store both key words,
//            release, publish count.  No 128-bit atomics needed.
#include "table.h"

#include <algorithm>
#include <vector>

namespace {

// ---- k <= 32: packed front end ---------------------------------------------------------------
// A wave takes 2 KiB tiles of the flat code buffer; lane l owns the 32 window starts at
// tile + 32*l.  Loads are 2 x dwordx4 per lane, bases are packed 2 bits each in registers, the
// 31 look-ahead bases come from lane l+1 by one cross-lane shuffle (lane 63: one extra load).
template <bool CANON>
__global__ __launch_bounds__(256) void hash_count1_kernel(const int8_t *__restrict__ data,
                                                          int64_t nN, int k, TableView t) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ntiles = (nN + 2047) >> 11;
  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int64_t off = (tile << 11) + 32 * lane;
    uint32_t b0, b1, bad;
    dev_load_chunk32(data, off, nN, b0, b1, bad);
    uint32_t n0 = dev_lane_next(b0), n1 = dev_lane_next(b1), nbad = dev_lane_next(bad);
    if (lane == 63) dev_load_chunk32(data, off + 32, nN, n0, n1, nbad);
    const uint64_t hi = ((uint64_t)b0 << 32) | b1;
    const uint64_t lo = ((uint64_t)n0 << 32) | n1;
    const uint64_t M = ((uint64_t)bad << 32) | nbad;
    for (int i = 0; i < 32; ++i) {
      // window i: bases i .. i+k-1 of the 64-base string hi:lo
      const uint64_t x = i ? ((hi << (2 * i)) | (lo >> (64 - 2 * i))) : hi;
      const uint64_t inval = (M << i) >> (64 - k);
      if (inval == 0) {
        uint64_t key = x >> (64 - 2 * k);
        if (CANON) {
          const uint64_t rc = dev_revcomp64(key, k);
          key = rc < key ? rc : key;
        }
        table_add1(t, key, 1u);
      }
    }
  }
}

// ---- k > 32: byte-wise rolling, 128-bit keys ---------------------------------------------------
template <bool CANON>
__global__ __launch_bounds__(256) void hash_count2_kernel(const int8_t *__restrict__ data,
                                                          int64_t nN, int k, TableView t) {
  typedef unsigned __int128 u128;
  const int SEG = 64;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t nseg = (nN + SEG - 1) / SEG;
  const u128 mask = (k == 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1);
  for (int64_t s = tid; s < nseg; s += nthreads) {
    const int64_t s0 = s * SEG;
    const int64_t s1 = min(s0 + SEG, nN);
    const int64_t end = min(s1 + k - 1, nN);
    u128 fwd = 0, rc = 0;
    int run = 0;
    for (int64_t p = s0; p < end; ++p) {
      const int c = (int)data[p];
      if (c < 0 || c > 3) { run = 0; continue; }
      fwd = ((fwd << 2) | (u128)(unsigned)c) & mask;
      rc = (rc >> 2) | ((u128)(unsigned)(3 - c) << (2 * (k - 1)));
      if (++run >= k) {
        const u128 key = (CANON && rc < fwd) ? rc : fwd;
        table_add2(t, (uint64_t)key, (uint64_t)(key >> 64), 1u);
      }
    }
  }
}

__global__ __launch_bounds__(256) void hash_merge_kernel(const uint64_t *__restrict__ lo,
                                                         const uint64_t *__restrict__ hi,
                                                         const uint32_t *__restrict__ cnt,
                                                         int64_t n, int two, TableView t) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n; i += nthreads) {
    const uint32_t c = cnt[i];
    if (c == 0) continue;
    if (two) table_add2(t, lo[i], hi[i], c);
    else table_add1(t, lo[i], c);
  }
}

__device__ __forceinline__ bool src_read(const ResultSrc &r, uint64_t s, uint64_t &lo, uint64_t &hi,
                                         uint32_t &c) {
  c = r.cnt[s];
  hi = 0;
  if (r.kind == 1 || r.kind == 3) {     // two-word table (count word = slot state) / two-word list
    if (c == 0) return false;
    lo = r.lo[s]; hi = r.hi[s];
    return true;
  }
  lo = r.lo[s];
  if (r.kind == 2) return c != 0;
  return lo != CFRK_EMPTY_KEY && c != 0;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((unsigned long long)v, o);
  return v;
}
__device__ __forceinline__ uint64_t wave_xor64(uint64_t v) {
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor((unsigned long long)v, o);
  return v;
}

// distinct / sum / weighted sum / xor digest over the result
__global__ __launch_bounds__(256) void result_scan_kernel(ResultSrc r) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const bool two = r.kind == 1 || r.kind == 3;
  uint64_t d = 0, s = 0, w = 0, x = 0;
  for (uint64_t i = tid; i < r.n; i += nthreads) {
    uint64_t lo, hi; uint32_t c;
    if (!src_read(r, i, lo, hi, c)) continue;
    const uint64_t kh = two ? lo + dev_splitmix64(hi) : lo;
    if (c >= CFRK_COUNT_MAX) r.stats[ST_SAT] = 1;      // (a saturated count, here or on the rank this entry came from)
    d += 1; s += c;
    w += (uint64_t)c * dev_splitmix64(kh);
    x ^= dev_splitmix64(kh ^ (uint64_t)c);
  }
  if (tid == 0 && !two) {
    const uint64_t ones = sat_ones(r.stats[ST_ONES]);
    if (ones >= CFRK_COUNT_MAX) r.stats[ST_SAT] = 1;
    if (ones) {
      d += 1; s += ones;
      w += ones * dev_splitmix64(CFRK_EMPTY_KEY);
      x ^= dev_splitmix64(CFRK_EMPTY_KEY ^ ones);
    }
  }
  d = wave_sum64(d); s = wave_sum64(s); w = wave_sum64(w); x = wave_xor64(x);
  if ((threadIdx.x & 63) == 0) {
    if (d) atomicAdd((unsigned long long *)&r.stats[ST_DIG0], (unsigned long long)d);
    if (s) atomicAdd((unsigned long long *)&r.stats[ST_DIG1], (unsigned long long)s);
    if (w) atomicAdd((unsigned long long *)&r.stats[ST_DIG2], (unsigned long long)w);
    if (x) atomicXor((unsigned long long *)&r.stats[ST_DIG3], (unsigned long long)x);
  }
}

__device__ __forceinline__ uint32_t owner_of(uint64_t lo, uint64_t hi, bool two, int parts) {
  const uint64_t m = two ? dev_mix64(lo ^ dev_mix64(hi)) : dev_mix64(lo);
  return (uint32_t)((m & 0xFFFFFFFFull) % (uint32_t)parts);
}

// Owner partition, both passes wave-aggregated: one atomic per (wave, part) instead of one per
// entry -- 10^8 entries hammering `parts` cursor words would serialise on those few addresses.
constexpr int MAX_FAST_PARTS = 16;

// pass 1: entries per owner part
__global__ __launch_bounds__(256) void result_export_count_kernel(ResultSrc r, int parts,
                                                                  unsigned long long *part_n) {
  const int lane = threadIdx.x & 63;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const bool two = r.kind == 1 || r.kind == 3;
  for (uint64_t base = (uint64_t)tid - lane; base < r.n; base += nthreads) {
    const uint64_t i = base + lane;
    uint64_t lo = 0, hi = 0; uint32_t c = 0;
    const bool occ = (i < r.n) && src_read(r, i, lo, hi, c);
    const uint32_t own = (occ && parts > 1) ? owner_of(lo, hi, two, parts) : 0u;
    if (parts <= MAX_FAST_PARTS) {
      for (int p = 0; p < parts; ++p) {
        const unsigned long long m = __ballot(occ && own == (uint32_t)p);
        if (lane == 0 && m) atomicAdd(&part_n[p], (unsigned long long)__popcll(m));
      }
    } else if (occ) {
      atomicAdd(&part_n[own], 1ull);
    }
  }
  if (tid == 0 && !two && r.stats[ST_ONES])
    atomicAdd(&part_n[owner_of(CFRK_EMPTY_KEY, 0, false, parts)], 1ull);
}

// pass 2: scatter into the owner segments (part_cursor starts at the exclusive prefix)
__global__ __launch_bounds__(256) void result_export_scatter_kernel(ResultSrc r, int parts,
                                                                    unsigned long long *part_cursor,
                                                                    uint64_t *__restrict__ out_lo,
                                                                    uint64_t *__restrict__ out_hi,
                                                                    uint32_t *__restrict__ out_cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const bool two = r.kind == 1 || r.kind == 3;
  for (uint64_t base = (uint64_t)tid - lane; base < r.n; base += nthreads) {
    const uint64_t i = base + lane;
    uint64_t lo = 0, hi = 0; uint32_t c = 0;
    const bool occ = (i < r.n) && src_read(r, i, lo, hi, c);
    const uint32_t own = (occ && parts > 1) ? owner_of(lo, hi, two, parts) : 0u;
    unsigned long long dst = 0;
    if (parts <= MAX_FAST_PARTS) {
      for (int p = 0; p < parts; ++p) {
        const bool mine = occ && own == (uint32_t)p;
        const unsigned long long m = __ballot(mine);
        unsigned long long b = 0;
        if (lane == 0 && m) b = atomicAdd(&part_cursor[p], (unsigned long long)__popcll(m));
        b = __shfl((unsigned long long)b, 0);
        if (mine) dst = b + __popcll(m & ((1ull << lane) - 1));
      }
    } else if (occ) {
      dst = atomicAdd(&part_cursor[own], 1ull);
    }
    if (occ) {
      out_lo[dst] = lo;
      if (two) out_hi[dst] = hi;
      out_cnt[dst] = c;
    }
  }
  if (tid == 0 && !two) {
    const uint64_t ones = r.stats[ST_ONES];
    if (ones) {
      const unsigned long long dst = atomicAdd(&part_cursor[owner_of(CFRK_EMPTY_KEY, 0, false, parts)], 1ull);
      out_lo[dst] = CFRK_EMPTY_KEY;
      out_cnt[dst] = sat_ones(ones);
    }
  }
}

int grid_for(const cfrk_ctx *ctx, int64_t items_per_block_hint, int64_t items) {
  int64_t want = (items + items_per_block_hint - 1) / items_per_block_hint;
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 8));
}

}  // namespace

TableView cfrk_table_view(const cfrk_ctx *ctx) {
  TableView t;
  t.lo = ctx->g_keys_lo; t.hi = ctx->g_keys_hi; t.cnt = ctx->g_counts; t.stats = ctx->g_stats;
  t.mask = ctx->g_cap - 1;
  t.shift = 64 - ctx->g_log2cap;
  return t;
}

static ResultSrc table_src(const cfrk_ctx *ctx) {
  ResultSrc r;
  r.lo = ctx->g_keys_lo; r.hi = ctx->g_keys_hi; r.cnt = ctx->g_counts; r.n = ctx->g_cap;
  r.kind = ctx->g_two ? 1 : 0;
  r.stats = ctx->g_stats;
  return r;
}

int cfrk_hash_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN) {
  const bool canon = (ctx->g_flags & CFRK_CANONICAL) != 0;
  TableView t = cfrk_table_view(ctx);
  if (!ctx->g_two) {
    const int grid = grid_for(ctx, 4 * 2048, nN);
    if (canon) hipLaunchKernelGGL((hash_count1_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_data, nN, ctx->g_k, t);
    else hipLaunchKernelGGL((hash_count1_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_data, nN, ctx->g_k, t);
  } else {
    const int grid = grid_for(ctx, 256 * 64, nN);
    if (canon) hipLaunchKernelGGL((hash_count2_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_data, nN, ctx->g_k, t);
    else hipLaunchKernelGGL((hash_count2_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_data, nN, ctx->g_k, t);
  }
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}

int cfrk_hash_merge(cfrk_ctx *ctx, const uint64_t *lo, const uint64_t *hi, const uint32_t *cnt,
                    int64_t n) {
  TableView t = cfrk_table_view(ctx);
  const int grid = grid_for(ctx, 256, n);
  hipLaunchKernelGGL(hash_merge_kernel, dim3(grid), dim3(256), 0, ctx->stream, lo, hi, cnt, n,
                     ctx->g_two ? 1 : 0, t);
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}

int cfrk_result_scan(cfrk_ctx *ctx, const ResultSrc *src, uint64_t st[ST_NWORDS]) {
  ResultSrc r = src ? *src : table_src(ctx);
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_DIG0, 0, 4 * sizeof(uint64_t), ctx->stream));
  const int grid = grid_for(ctx, 256 * 16, (int64_t)r.n);
  hipLaunchKernelGGL(result_scan_kernel, dim3(grid), dim3(256), 0, ctx->stream, r);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, ST_NWORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

int cfrk_result_export(cfrk_ctx *ctx, const ResultSrc *src, uint64_t *d_lo, uint64_t *d_hi,
                       uint32_t *d_cnt, uint64_t cap, int parts, uint64_t *part_counts) {
  ResultSrc r = src ? *src : table_src(ctx);
  void *scratch;
  int rc = cfrk_pool_get(ctx, BUF_SCRATCH, (size_t)parts * 8, &scratch);
  if (rc) return rc;
  unsigned long long *d_part = (unsigned long long *)scratch;
  const int grid = grid_for(ctx, 256 * 16, (int64_t)r.n);
  HIP_TRY(ctx, hipMemsetAsync(d_part, 0, (size_t)parts * 8, ctx->stream));
  hipLaunchKernelGGL(result_export_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, r, parts, d_part);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<uint64_t> n(parts), cur(parts);
  HIP_TRY(ctx, hipMemcpyAsync(n.data(), d_part, (size_t)parts * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t total = 0;
  for (int p = 0; p < parts; ++p) { cur[p] = total; total += n[p]; part_counts[p] = n[p]; }
  if (total > cap) return cfrk_fail(ctx, CFRK_ERR_SMALL_BUF, "%llu entries, room for %llu", (unsigned long long)total, (unsigned long long)cap);
  if (total == 0) return CFRK_OK;
  HIP_TRY(ctx, hipMemcpyAsync(d_part, cur.data(), (size_t)parts * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(result_export_scatter_kernel, dim3(grid), dim3(256), 0, ctx->stream, r, parts, d_part, d_lo, d_hi, d_cnt);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}
