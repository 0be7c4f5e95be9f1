// common.h -- context, error plumbing and device helpers shared by the gfx950 kernels.
// Product code: must never include or link anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cfrk_abi.h"

#define CFRK_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
// internal (never leaves the library): a partitioned path found 2^32 records in one stream (ST_CWRAP) -- a single-key
// flood; cfrk_global_add_device counts the add through the HBM table instead
#define CFRK_INTERNAL_FLOOD (-100)
#define CFRK_MAX_PROBE (1u << 22)
#define CFRK_RUNS_MAX_GROUPS 16

enum {  // pool slots
  BUF_DATA = 0, BUF_START, BUF_LENGTH, BUF_FREQ, BUF_SPILL, BUF_EXPORT_LO, BUF_EXPORT_HI,
  BUF_EXPORT_CNT, BUF_SCRATCH, BUF_MSP_L1, BUF_MSP_L2, BUF_MSP_OUTK, BUF_MSP_OUTC, BUF_MSP_AUX, BUF_MSP_OUTH,
  BUF_MSP_ACCK, BUF_MSP_ACCH, BUF_MSP_ACCC,   // lists of the passes of a multi-pass add, merged per leaf at the end
  BUF_MSP_LAYOUT,                             // exact second-level layout (stream bases and sizes)
  BUF_MSP_OVF,                                // records that did not fit their leaf stream (a few)
  BUF_MSP_OVF1, BUF_MSP_LAYOUT1,              // the same for the level-1 regions
  BUF_RUNS_AUX,                               // pipelined runs exchange: segment cursors, used rows per group
  BUF_NSLOTS
};

enum {  // device stats words (uint64 each)
  ST_OVERFLOW = 0, ST_ONES, ST_CURSOR, ST_DIG0, ST_DIG1, ST_DIG2, ST_DIG3, ST_SPILLED, ST_AUX0,
  ST_AUX1, ST_L2OVF /* leaf streams were too small by a lot: the second level is redone with exact sizes */,
  ST_OVFN /* records parked in the overflow buffer (leaf streams too small by a little) */,
  ST_MULTISEG /* a leaf was counted in several key-subset passes: its list entries are not contiguous */,
  ST_L1OVF /* level-1 regions were too small by a lot: the first level is redone with exact sizes */,
  ST_OVFN1 /* records parked because their level-1 region was full (a few) */,
  ST_SAT /* a count reached CFRK_COUNT_MAX and was held there (finish / digest / export: CFRK_ERR_COUNT_OVERFLOW) */,
  ST_CWRAP /* a 32-bit region / stream cursor of a partitioned path wrapped (>= 2^32 records in one region: a
              single-key flood): the add is counted again through the HBM table */,
  ST_NWORDS = 18
};

struct cfrk_buf { void *p; size_t cap; };

struct cfrk_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  char err[512];
  cfrk_buf pool[BUF_NSLOTS];
  void *pinned; size_t pinned_cap;
  int num_cus;
  // global-count state
  bool g_active;
  int g_k, g_flags;
  bool g_two;            // two-word keys (k > 32)
  int g_log2cap;
  uint64_t g_cap;
  uint64_t *g_keys_lo, *g_keys_hi;
  uint32_t *g_counts;
  uint64_t *g_stats;     // device, ST_NWORDS
  bool g_table_cleared;  // the table was cleared by the last begin()
  uint64_t *h_stats;     // pinned host snapshot of g_stats taken at the end of the last add
  bool h_stats_valid;
  hipEvent_t ev0, ev1;
  bool ev_valid;
  hipEvent_t stage_ev[2];   // H2D staging (cfrk_global_add)
  // pipelined runs exchange (cfrk_global_export_runs_async / _wait): one event per group, the groups' used rows in pinned memory
  hipEvent_t runs_ev[CFRK_RUNS_MAX_GROUPS];
  uint64_t *h_runs;         // pinned, CFRK_RUNS_MAX_GROUPS x (64 + 1) words
  int runs_groups, runs_parts;
  uint64_t runs_seg_cap;
  // minimizer-partitioned fast path (msp.hip)
  struct cfrk_msp *msp;
  int last_passes;       // passes the most recent add took on a partitioned path (1 unless memory was short)
  uint32_t dbg_flags;    // cfrk_debug_set_flags
  size_t mem_budget;     // 0 = what the device has free; else a cap on the partitioned paths' buffers (diagnostics)
  double dbg_param[4];   // cfrk_debug_set_param (0 = the library's own choice)
};

int cfrk_fail(cfrk_ctx *ctx, int code, const char *fmt, ...);
int cfrk_pool_get(cfrk_ctx *ctx, int slot, size_t bytes, void **out);

#define HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return cfrk_fail((ctx), (e_ == hipErrorOutOfMemory) ? CFRK_ERR_NOMEM : CFRK_ERR_HIP,   \
                       "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
  } while (0)

// ---- kernels' host-side launchers (one per .hip file) ------------------------------------
int cfrk_launch_dense(cfrk_ctx *ctx, const int8_t *d_data, const int64_t *d_start,
                      const int32_t *d_length, int64_t nN, int64_t nS, int k, int flags,
                      int32_t *d_freq);
int cfrk_launch_synth(cfrk_ctx *ctx, int64_t r0, int64_t R, int L, int64_t Glen, uint64_t seedG,
                      uint64_t seedR, uint64_t seedS, int uniform, int8_t *d_data, int64_t *d_start,
                      int32_t *d_length);
int cfrk_hash_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);
int cfrk_hash_merge(cfrk_ctx *ctx, const uint64_t *lo, const uint64_t *hi, const uint32_t *cnt,
                    int64_t n);
int cfrk_sort_export(cfrk_ctx *ctx, const uint64_t *d_lo, const uint64_t *d_hi, const uint32_t *d_cnt, uint64_t n,
                     const uint64_t **s_lo, const uint64_t **s_hi, const uint32_t **s_cnt);
struct ResultSrc;   // table.h: the table itself (NULL) or a compact (key,count) list
int cfrk_result_scan(cfrk_ctx *ctx, const ResultSrc *src, uint64_t stats_host[ST_NWORDS]);
int cfrk_result_export(cfrk_ctx *ctx, const ResultSrc *src, uint64_t *d_lo, uint64_t *d_hi,
                       uint32_t *d_cnt, uint64_t cap, int parts, uint64_t *part_counts);

// ---- device helpers -----------------------------------------------------------------------
#ifdef __HIPCC__

// Diagnostics counter bumped by many lanes at once: the active lanes elect one that adds their
// number (10^9 atomics on ONE word take seconds: they serialise at the L2).  Counts events.
__device__ __forceinline__ void dev_count_event(uint64_t *word) {
  const unsigned long long m = __ballot(1);
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  if (lane == (unsigned)__ffsll((long long)m) - 1u)
    atomicAdd((unsigned long long *)word, (unsigned long long)__popcll(m));
}

// Neighbour-lane reads as DPP wave shifts (one VALU instruction; __shfl_down(x, 1) compiles to
// ds_bpermute + address arithmetic).  The lane without a neighbour reads 0.
__device__ __forceinline__ uint32_t dev_lane_next(uint32_t x) {       // value of lane + 1
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x130, 0xF, 0xF, true);   // wave_shl:1
}
__device__ __forceinline__ uint32_t dev_lane_prev(uint32_t x) {       // value of lane - 1
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xF, 0xF, true);   // wave_shr:1
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row_shr 1/2/4/8 inside the
// rows of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals on): no LDS traffic.
// __shfl_up() compiles to ds_bpermute_b32 + address arithmetic + a select -- five instructions and
// an LDS round trip per step, six dependent steps per scan.
__device__ __forceinline__ uint32_t dev_wave_scan_incl(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return x;
}

__device__ __forceinline__ uint64_t dev_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// slot / owner hash of a key
__device__ __forceinline__ uint64_t dev_mix64(uint64_t x) {
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return x;
}

// 4 code bytes (first base in the lowest byte) -> 8 bits, first base most significant
__device__ __forceinline__ uint32_t dev_pack4(uint32_t w) {
  return ((w & 0x03030303u) * 0x40100401u) >> 24;
}
// 4 code bytes -> 4 bits, bit 3 = first base; set where the byte is not in 0..3
__device__ __forceinline__ uint32_t dev_bad4(uint32_t w) {
  // bit 7 of a byte <- (bits 2..6 nonzero) | bit 7: the add cannot carry out of a byte (0x7C + 0x7F)
  const uint32_t f = (((w & 0x7C7C7C7Cu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
  // the flags at bits 7, 15, 23, 31 land on bits 35, 34, 33, 32 of the product (every partial
  // product hits its own bit: no carries); bits 4.. of the high word hold other partial products
  return __umulhi(f, 0x10080402u) & 15u;
}
__device__ __forceinline__ void dev_pack16(uint4 v, uint32_t &bases, uint32_t &bad) {
  bases = (dev_pack4(v.x) << 24) | (dev_pack4(v.y) << 16) | (dev_pack4(v.z) << 8) | dev_pack4(v.w);
  bad = (dev_bad4(v.x) << 12) | (dev_bad4(v.y) << 8) | (dev_bad4(v.z) << 4) | dev_bad4(v.w);
}

// One lane's 32-byte chunk at byte offset off (multiple of 32) of the flat code buffer:
// b0,b1 = bases 0..15 / 16..31 packed 2 bits each, first base most significant;
// bad bit (31-j) set when base j is invalid or lies at/after nN.
__device__ __forceinline__ void dev_load_chunk32(const int8_t *__restrict__ data, int64_t off,
                                                 int64_t nN, uint32_t &b0, uint32_t &b1,
                                                 uint32_t &bad) {
  if (off + 32 <= nN) {
    const uint4 *p = reinterpret_cast<const uint4 *>(data + off);
    uint4 v0 = p[0], v1 = p[1];
    uint32_t m0, m1;
    dev_pack16(v0, b0, m0);
    dev_pack16(v1, b1, m1);
    bad = (m0 << 16) | m1;
  } else {
    b0 = 0; b1 = 0; bad = 0;
    for (int j = 0; j < 32; ++j) {
      int c = (off + j < nN) ? (int)data[off + j] : -1;
      bool inv = (c < 0 || c > 3);
      uint32_t v = inv ? 0u : (uint32_t)c;
      if (j < 16) b0 |= v << (30 - 2 * j); else b1 |= v << (30 - 2 * (j - 16));
      if (inv) bad |= 1u << (31 - j);
    }
  }
}

// reverse complement of a k-mer held in the low 2k bits (first base most significant)
__device__ __forceinline__ uint64_t dev_revcomp64(uint64_t x, int k) {
  uint64_t r = __brevll(x);
  r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
  r = ~r;
  return r >> (64 - 2 * k);
}

#endif  // __HIPCC__
