// radix.hip -- global k-mer counting for k <= 15 on gfx950: keys fit 30 bits, so instead of
// hashing the key is radix-partitioned and counted by DIRECT ADDRESS in LDS.
//
//   keys are first scrambled by an invertible mix of the 2k-bit value (odd multiply, xorshift by
//   k) -- canonical k-mers are skewed low and real genomes are not uniform, scrambled keys are;
//   RX1  extract (canonical) k-mers (2 x dwordx4 per lane, 2-bit packing in registers), counting
//        sort a tile of 8192 keys by the top b1 bits in LDS, one HBM atomic per bin per tile,
//        coalesced copy-out of 4-byte keys;
//   RX2  split every region by the next b2 bits the same way;
//   RX3  one workgroup per leaf: an LDS array of 2^idx counters, ds_add_u32 by the low idx bits,
//        then the non-zero counters are un-scrambled and appended to the result list.
//   8 <= k <= 15: b1 = 8, idx = min(13, 2k - 11), b2 = 2k - 8 - idx (3..9): >= 2048 leaves.
//   k <= 7 (at most 16384 keys): no partition at all, every workgroup counts into a replicated
//   LDS table and adds it to a dense HBM array once.
//
// HBM never sees the bits a key's region already implies: level 1 stores the 2k-8 bits below the
// bin as a 16-bit plane plus (2k-8 > 16) an 8-bit plane -- 3 bytes per k-mer at k = 15 instead of
// 4 --, the leaves store the low idx <= 13 bits as 16-bit words (round 3: 24.1 -> 15 GB per launch
// on configs[1]); every occurrence is one LDS atomic.  Same semantics and
// the same result-list form as msp.hip (which covers 16 <= k <= 32); overflowing regions spill
// into the HBM table.
#include "msp.h"
#include "table.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int RX_NXG = 8;                 // workgroup-id groups (XCD affinity, speed only)
constexpr int RX_NREG = 32;               // sub-regions (cursors) per level-1 bin, RX_NREG / RX_NXG per XCD:
                                          // the returning atomics on one cursor serialise (see msp_dev.h: NXG)
constexpr int RX1_THREADS = 256, RX1_KEYS = RX1_THREADS * 32;
constexpr int RX2_THREADS = 512, RX2_PER = 16, RX2_KEYS = RX2_THREADS * RX2_PER;
constexpr int RX3_THREADS = 256;
constexpr int RX_IDX_MAX = 13;

struct RxView {
  // level 1: the 2k - b1 bits of a key below its bin, low 16 in k1lo, the rest (if any: hi8) in k1hi
  uint16_t *k1lo; uint8_t *k1hi; uint32_t hi8; uint32_t *cnt1; uint64_t cap1;      // 2^b1 x RX_NREG regions
  uint16_t *key2; uint32_t *cnt2; uint64_t cap2;      // 2^(b1+b2) leaves: the low idx bits of their keys
  // exact layout after a leaf stream overflowed the fixed stride (few distinct keys, each seen
  // very often): leaf l starts at key lbase[l] and holds exactly lcap[l] keys
  const uint64_t *lbase; const uint32_t *lcap; uint32_t exact;
  // the same for the level-1 regions (one amplicon: all keys in a few bins): region r starts at
  // key rbase[r] and holds exactly rcap[r] keys
  const uint64_t *rbase; const uint32_t *rcap; uint32_t exact1;
  uint64_t *out_keys; uint32_t *out_cnt; uint64_t out_cap;
  uint64_t *stats;
  int k, b1, b2, idx;
  uint32_t mul, inv, kmask;                           // scramble: x * mul mod 4^k, x ^= x >> k
};

// level-1 region / cursor of (bin, sub-region): sub-region major -- the cursors a workgroup reserves
// from lie side by side (memory-side atomics: one request per touched 64 bytes, see msp.hip l1_reg)
__device__ __forceinline__ uint32_t rx_reg(const RxView &v, uint32_t bin, uint32_t sub) { return (sub << v.b1) | bin; }
__device__ __forceinline__ uint32_t rx_mix(const RxView &v, uint32_t key) {
  uint32_t x = (key * v.mul) & v.kmask;
  return x ^ (x >> v.k);
}
__device__ __forceinline__ uint32_t rx_unmix(const RxView &v, uint32_t x) {
  x ^= x >> v.k;                                       // k = half the width: an involution
  return (x * v.inv) & v.kmask;
}

template <int NB>
__device__ __forceinline__ void rx_scan(const uint32_t *cnt, uint32_t *off, uint32_t *wtot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t x = 0, incl = 0;
  if (tid < NB) {
    x = cnt[tid];
    incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == 63) wtot[wave] = incl;
  }
  __syncthreads();
  if (tid < NB) {
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    off[tid] = base + incl - x;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------ RX1
template <bool CANON>
__global__ __launch_bounds__(RX1_THREADS) void rx1_kernel(const int8_t *__restrict__ data, int64_t nN,
                                                          RxView v, TableView t) {
  __shared__ uint32_t sorted[RX1_KEYS];
  __shared__ uint32_t hist[256], loff[256], gbase[256], fill[256];
  __shared__ uint32_t wtot[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int k = v.k;
  hist[tid] = 0; fill[tid] = 0;
  __syncthreads();

  const int64_t off = ((int64_t)blockIdx.x * RX1_THREADS + tid) * 32;
  uint32_t b0, b1w, bad;
  dev_load_chunk32(data, off, nN, b0, b1w, bad);
  uint32_t n0 = dev_lane_next(b0), n1 = dev_lane_next(b1w), nbad = dev_lane_next(bad);
  if (lane == 63) dev_load_chunk32(data, off + 32, nN, n0, n1, nbad);
  (void)n1;
  const int sh1 = 2 * k - v.b1;
  // The k-mer at position i is the top 2k bits of the 32-bit window of the base string that starts
  // there (one v_alignbit with a static shift), its reverse complement the low 2k bits of a window
  // of the reverse-complemented string that ENDS where the k-mer starts (msp_dev.h: msp_minimizers),
  // its validity the top k bits of a window of the invalid-base mask: 32-bit work throughout
  // (64-bit shifts and a bit reversal per position took twice the instructions).
  const uint32_t D[3] = {b0, b1w, n0};
  uint32_t R[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    uint32_t x = __brev(D[i]);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    R[3 - i] = ~x;
  }
  R[0] = 0;
  const int fsh = 32 - 2 * k;
  const uint32_t vlim = 1u << (32 - k);            // a window whose top k mask bits are clear is below this

  uint32_t keys[32];
  uint32_t V = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int o = 2 * i, q = o >> 5, r = o & 31;
    const uint32_t X = r ? __builtin_amdgcn_alignbit(D[q], D[q + 1], 32 - r) : D[q];
    uint32_t key = X >> fsh;
    if (CANON) {
      const int o2 = 96 - 2 * i, q2 = o2 >> 5, r2 = o2 & 31;
      const uint32_t Y = r2 ? __builtin_amdgcn_alignbit(R[q2], R[q2 + 1], 32 - r2) : R[q2];
      key = min(key, Y & v.kmask);
    }
    const uint32_t Wm = i ? __builtin_amdgcn_alignbit(bad, nbad, 32 - i) : bad;
    const bool ok = Wm < vlim;
    key = rx_mix(v, key);
    keys[i] = key;
    if (ok) { V |= 1u << i; atomicAdd(&hist[key >> sh1], 1u); }
  }
  __syncthreads();
  uint32_t my_base = 0;
  {
    const uint32_t c = hist[tid];
    if (c) my_base = atomicAdd(&v.cnt1[rx_reg(v, tid, blockIdx.x & (RX_NREG - 1))], c);
  }
  rx_scan<256>(hist, loff, wtot);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    if (V & (1u << i)) {
      const uint32_t b = keys[i] >> sh1;
      sorted[loff[b] + atomicAdd(&fill[b], 1u)] = keys[i];
    }
  }
  gbase[tid] = my_base;
  __syncthreads();
  const uint32_t total = loff[255] + hist[255];
  for (uint32_t p = tid; p < total; p += RX1_THREADS) {
    const uint32_t key = sorted[p];
    const uint32_t b = key >> sh1;
    const uint32_t dst = gbase[b] + (p - loff[b]);
    const uint32_t reg = rx_reg(v, b, blockIdx.x & (RX_NREG - 1));
    if (v.exact1) {
      if (dst < v.rcap[reg]) {
        v.k1lo[v.rbase[reg] + dst] = (uint16_t)key;
        if (v.hi8) v.k1hi[v.rbase[reg] + dst] = (uint8_t)((key & ((1u << sh1) - 1u)) >> 16);
      } else { t.stats[ST_SPILLED] = 1; table_add1(t, (uint64_t)rx_unmix(v, key), 1u); }   // cannot happen
    } else if (dst < v.cap1) {
      v.k1lo[(uint64_t)reg * v.cap1 + dst] = (uint16_t)key;
      if (v.hi8) v.k1hi[(uint64_t)reg * v.cap1 + dst] = (uint8_t)((key & ((1u << sh1) - 1u)) >> 16);
    } else {
      v.stats[ST_L1OVF] = 1;       // the cursor keeps counting: the host redoes RX1 with exact sizes
    }
  }
}

// ------------------------------------------------------------------------------------------ RX2
__global__ __launch_bounds__(RX2_THREADS) void rx2_kernel(int tiles_per_sub, RxView v, TableView t) {
  __shared__ uint32_t sorted[RX2_KEYS];
  __shared__ uint32_t hist[512], loff[512], gbase[512], fill[512];
  __shared__ uint32_t wtot[8];
  const int tid = threadIdx.x;
  const uint32_t nb1 = 1u << v.b1;
  const uint32_t xg = blockIdx.x & (RX_NXG - 1), seq = blockIdx.x / RX_NXG;
  const uint32_t per_bin = (uint32_t)RX_NREG * (uint32_t)tiles_per_sub;
  // bins are dealt to the 8 workgroup-id groups round robin (nb1 may be smaller than 8)
  const uint32_t bins_per_group = (nb1 + RX_NXG - 1) / RX_NXG;
  const uint32_t bl = seq / per_bin;
  if (bl >= bins_per_group) return;
  const uint32_t bin1 = xg + RX_NXG * bl;
  if (bin1 >= nb1) return;
  const uint32_t sub = (seq % per_bin) / (uint32_t)tiles_per_sub;
  const uint32_t tile = (seq % per_bin) % (uint32_t)tiles_per_sub;
  const uint32_t reg = rx_reg(v, bin1, sub);
  const uint64_t n = min((uint64_t)v.cnt1[reg], v.exact1 ? (uint64_t)v.rcap[reg] : v.cap1);
  const uint64_t r0 = (uint64_t)tile * RX2_KEYS;
  if (r0 >= n) return;
  const uint32_t nt = (uint32_t)min((uint64_t)RX2_KEYS, n - r0);
  hist[tid] = 0; fill[tid] = 0;
  __syncthreads();
  const uint64_t rb = (v.exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1) + r0;
  const uint16_t *slo = v.k1lo + rb;
  const uint8_t *shi = v.k1hi + rb;
  const uint32_t m2 = (1u << v.b2) - 1u;
  // 16-byte loads of the 16-bit plane (eight keys) and 8-byte loads of the 8-bit plane (a region
  // starts on a 16-element boundary unless the level was laid out again: cap1 is a multiple of 16)
  // instead of one load per key; which key a thread holds does not matter to a counting sort
  uint32_t kk[RX2_PER];                          // the 2k - b1 bits below the bin
  static_assert(RX2_PER % 8 == 0, "keys are loaded eight at a time");
  const bool vec = !v.exact1;
  auto idx_of = [&](int i) {                     // tile position of key i of this thread
    return vec ? (((uint32_t)(i >> 3) * RX2_THREADS + tid) << 3) + (uint32_t)(i & 7) : (uint32_t)i * RX2_THREADS + tid;
  };
  if (vec) {
#pragma unroll
    for (int q = 0; q < RX2_PER / 8; ++q) {
      const uint32_t idx8 = ((uint32_t)q * RX2_THREADS + tid) << 3;
      uint4 x = make_uint4(0u, 0u, 0u, 0u);
      uint2 h = make_uint2(0u, 0u);
      if (idx8 < nt) {
        x = *reinterpret_cast<const uint4 *>(slo + idx8);
        if (v.hi8) h = *reinterpret_cast<const uint2 *>(shi + idx8);
      }
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t lo16 = (w[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
        const uint32_t hb = ((c < 4 ? h.x : h.y) >> (8 * (c & 3))) & 0xFFu;
        kk[8 * q + c] = (hb << 16) | lo16;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < RX2_PER; ++i) {
      const uint32_t idx = (uint32_t)i * RX2_THREADS + tid;
      kk[i] = (idx < nt) ? ((uint32_t)slo[idx] | (v.hi8 ? (uint32_t)shi[idx] << 16 : 0u)) : 0u;
    }
  }
#pragma unroll
  for (int i = 0; i < RX2_PER; ++i)
    if (idx_of(i) < nt) atomicAdd(&hist[(kk[i] >> v.idx) & m2], 1u);
  __syncthreads();
  uint32_t my_base = 0;
  {
    const uint32_t c = hist[tid];
    if (c) my_base = atomicAdd(&v.cnt2[((uint64_t)bin1 << v.b2) + tid], c);
  }
  rx_scan<512>(hist, loff, wtot);
#pragma unroll
  for (int i = 0; i < RX2_PER; ++i) {
    if (idx_of(i) < nt) {
      const uint32_t b = (kk[i] >> v.idx) & m2;
      sorted[loff[b] + atomicAdd(&fill[b], 1u)] = kk[i];
    }
  }
  gbase[tid] = my_base;
  __syncthreads();
  for (uint32_t p = tid; p < nt; p += RX2_THREADS) {
    const uint32_t key = sorted[p];
    const uint32_t b = (key >> v.idx) & m2;
    const uint32_t dst = gbase[b] + (p - loff[b]);
    const uint64_t leaf = ((uint64_t)bin1 << v.b2) + b;
    if (v.exact) {
      if (dst < v.lcap[leaf]) v.key2[v.lbase[leaf] + dst] = (uint16_t)(key & ((1u << v.idx) - 1u));
      else { t.stats[ST_SPILLED] = 1; table_add1(t, (uint64_t)rx_unmix(v, (bin1 << (2 * v.k - v.b1)) | key), 1u); }   // cannot happen
    } else if (dst < v.cap2) {
      v.key2[leaf * v.cap2 + dst] = (uint16_t)(key & ((1u << v.idx) - 1u));
    } else {
      v.stats[ST_L2OVF] = 1;       // the cursor keeps counting: the host redoes RX2 with exact sizes
    }
  }
}

// exact leaf layout from the demand the first attempt counted (single workgroup)
__global__ __launch_bounds__(1024) void rx_layout_kernel(const uint32_t *__restrict__ cnt2, uint32_t nleaf,
                                                         uint64_t *__restrict__ lbase, uint32_t *__restrict__ lcap) {
  __shared__ unsigned long long part[1024];
  const uint32_t per = (nleaf + 1023u) / 1024u;
  const uint32_t tid = threadIdx.x;
  unsigned long long s = 0;
  for (uint32_t i = 0; i < per; ++i) { const uint32_t l = tid * per + i; if (l < nleaf) s += cnt2[l]; }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; ++i) { const unsigned long long x = part[i]; part[i] = run; run += x; }
  }
  __syncthreads();
  unsigned long long run = part[tid];
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t l = tid * per + i;
    if (l < nleaf) { const uint32_t c = cnt2[l]; lbase[l] = run; lcap[l] = c; run += c; }
  }
}

// ------------------------------------------------------------------------------------------ RX3
__global__ __launch_bounds__(RX3_THREADS) void rx3_kernel(RxView v) {
  __shared__ uint32_t cnt[1 << RX_IDX_MAX];
  __shared__ unsigned long long wg_base;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t leaf = blockIdx.x;
  const uint64_t n = min((uint64_t)v.cnt2[leaf], v.exact ? (uint64_t)v.lcap[leaf] : v.cap2);
  if (n == 0) return;
  const uint32_t nidx = 1u << v.idx, imask = nidx - 1u;
  // few counters per leaf (small k): every thread group gets its own replica, or all 256 threads
  // would serialise on a handful of LDS words
  const int rlog = min(RX_IDX_MAX - v.idx, 8);
  const uint32_t rmask = (1u << rlog) - 1u;
  for (uint32_t s = tid; s < (nidx << rlog); s += RX3_THREADS) cnt[s] = 0;
  __syncthreads();
  const uint16_t *src = v.exact ? v.key2 + v.lbase[leaf] : v.key2 + (uint64_t)leaf * v.cap2;
  {
    // 16-byte loads (eight 16-bit keys), four of them in flight per thread: the stream is read as
    // uint4 from the 16-byte boundary below its first key, elements outside [0, n) are skipped
    const uint32_t head = (uint32_t)((reinterpret_cast<uintptr_t>(src) >> 1) & 7u);
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src - head);
    const uint64_t n4 = (head + n + 7) >> 3;                 // uint4 elements that hold keys
    const uint64_t lo = head, hi = head + n;                 // valid element range in the aligned view
    constexpr int INFL = 4;
    for (uint64_t q0 = 0; q0 < n4; q0 += (uint64_t)INFL * RX3_THREADS) {
      uint4 x[INFL];
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const uint64_t q = q0 + (uint64_t)u * RX3_THREADS + tid;
        x[u] = make_uint4(0u, 0u, 0u, 0u);
        if (q < n4) x[u] = src4[q];
      }
#pragma unroll
      for (int u = 0; u < INFL; ++u) {
        const uint64_t e = 8 * (q0 + (uint64_t)u * RX3_THREADS + tid);
        const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const uint32_t key = (w[c >> 1] >> (16 * (c & 1))) & 0xFFFFu;
          if (e + c >= lo && e + c < hi) atomicAdd(&cnt[((key & imask) << rlog) | ((uint32_t)tid & rmask)], 1u);
        }
      }
    }
  }
  __syncthreads();
  if (rlog) {                       // fold the replicas into replica 0
    for (uint32_t s = tid; s < nidx; s += RX3_THREADS) {
      uint32_t c = 0;
      for (uint32_t r = 0; r <= rmask; ++r) c += cnt[(s << rlog) | r];
      cnt[s << rlog] = c;
    }
    __syncthreads();
  }
  // compaction: every thread counts the non-zero counters among its own (counter q * 256 + tid: the
  // reads are conflict-free), one block scan gives it a range of the workgroup's reservation, and it
  // writes its entries there -- no ballot and no LDS atomic per 256 counters (the two passes of 32
  // such steps were half of this kernel's instructions at k = 15)
  __shared__ uint32_t wsum[RX3_THREADS / 64];
  uint32_t mine = 0;
  for (uint32_t s = tid; s < nidx; s += RX3_THREADS) mine += (cnt[s << rlog] != 0u) ? 1u : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < RX3_THREADS / 64; ++w) { const uint32_t x = wsum[w]; base += (w < (tid >> 6)) ? x : 0u; total += x; }
  if (tid == 0) wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)total);
  __syncthreads();
  unsigned long long dst = wg_base + base + incl - mine;
  for (uint32_t s = tid; s < nidx; s += RX3_THREADS) {
    const uint32_t c = cnt[s << rlog];
    if (c) {
      const uint32_t mixed = (leaf << v.idx) | s;
      if (dst < v.out_cap) { v.out_keys[dst] = (uint64_t)rx_unmix(v, mixed); v.out_cnt[dst] = c; }
      else v.stats[ST_OVERFLOW] = 1;
      ++dst;
    }
  }
}

// ------------------------------------------------------------------------------- k <= 7: direct
// 4^k <= 16384 counters: every workgroup keeps the whole (replicated) table in LDS over a
// grid-stride of tiles and adds it to a dense HBM array once; a one-workgroup kernel turns the
// dense array into the result list.
constexpr int RXS_THREADS = 256;
template <bool CANON>
__global__ __launch_bounds__(RXS_THREADS) void rxs_count_kernel(const int8_t *__restrict__ data, int64_t nN,
                                                                int k, uint32_t *__restrict__ dense) {
  __shared__ uint32_t cnt[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t nkey = 1u << (2 * k);
  int rlog = 14 - 2 * k;           // k = 7: the table fills the 64 KiB, no replicas
  if (rlog > 6) rlog = 6;
  const uint32_t rmask = (1u << rlog) - 1u;
  for (uint32_t s = tid; s < (nkey << rlog); s += RXS_THREADS) cnt[s] = 0;
  __syncthreads();
  const int64_t ntiles = (nN + RXS_THREADS * 32 - 1) / (RXS_THREADS * 32);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t off = (tile * RXS_THREADS + tid) * 32;
    uint32_t b0, b1w, bad;
    dev_load_chunk32(data, off, nN, b0, b1w, bad);
    uint32_t n0 = dev_lane_next(b0), n1 = dev_lane_next(b1w), nbad = dev_lane_next(bad);
    if (lane == 63) dev_load_chunk32(data, off + 32, nN, n0, n1, nbad);
    const uint64_t hi = ((uint64_t)b0 << 32) | b1w;
    const uint64_t lo = ((uint64_t)n0 << 32) | n1;
    const uint64_t M = ((uint64_t)bad << 32) | nbad;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const uint64_t x = i ? ((hi << (2 * i)) | (lo >> (64 - 2 * i))) : hi;
      if (((M << i) >> (64 - k)) == 0) {
        uint32_t key = (uint32_t)(x >> (64 - 2 * k));
        if (CANON) key = min(key, (uint32_t)dev_revcomp64((uint64_t)key, k));
        atomicAdd(&cnt[(key << rlog) | ((uint32_t)lane & rmask)], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t s = tid; s < nkey; s += RXS_THREADS) {
    uint32_t c = 0;
    for (uint32_t r = 0; r <= rmask; ++r) c += cnt[(s << rlog) | r];
    if (c) atomicAdd(&dense[s], c);
  }
}

__global__ __launch_bounds__(1024) void rxs_emit_kernel(const uint32_t *__restrict__ dense, uint32_t nkey,
                                                        RxView v) {
  __shared__ uint32_t total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  for (uint32_t s = threadIdx.x; s < nkey; s += blockDim.x) {
    const uint32_t c = dense[s];
    if (c) {
      const uint32_t dst = atomicAdd(&total, 1u);
      if (dst < v.out_cap) { v.out_keys[dst] = s; v.out_cnt[dst] = c; }
      else v.stats[ST_OVERFLOW] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) v.stats[ST_CURSOR] = total;
}

uint32_t inv_odd32(uint32_t a) {        // a * x == 1 (mod 2^32), Newton
  uint32_t x = a;
  for (int i = 0; i < 5; ++i) x *= 2u - a * x;
  return x;
}

}  // namespace

bool cfrk_radix_usable(const cfrk_ctx *ctx) {
  return !ctx->g_two && ctx->g_k >= 1 && ctx->g_k <= 15 && !(ctx->g_flags & CFRK_FORCE_HASH);
}

int cfrk_radix_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  int rc;
  if (ms->pending && (rc = cfrk_msp_flush_to_table(ctx))) return rc;
  const int k = ctx->g_k;
  RxView v;
  memset(&v, 0, sizeof v);
  v.k = k;
  if (k <= 7) {
    void *p;
    const uint32_t nkey = 1u << (2 * k);
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, (size_t)nkey * 4, &p))) return rc;
    uint32_t *dense = (uint32_t *)p;
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
    v.out_keys = (uint64_t *)p;
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
    v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
    v.stats = ctx->g_stats;
    HIP_TRY(ctx, hipMemsetAsync(dense, 0, (size_t)nkey * 4, ctx->stream));
    const int64_t ntiles = (nN + RXS_THREADS * 32 - 1) / (RXS_THREADS * 32);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, (int64_t)ctx->num_cus * 8));
    if (ctx->g_flags & CFRK_CANONICAL) hipLaunchKernelGGL((rxs_count_kernel<true>), dim3(grid), dim3(RXS_THREADS), 0, ctx->stream, d_data, nN, k, dense);
    else hipLaunchKernelGGL((rxs_count_kernel<false>), dim3(grid), dim3(RXS_THREADS), 0, ctx->stream, d_data, nN, k, dense);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(rxs_emit_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)dense, nkey, v);
    HIP_TRY(ctx, hipGetLastError());
    ms->view.out_keys = v.out_keys; ms->view.out_cnt = v.out_cnt; ms->view.out_cap = v.out_cap;
    ms->view.stats = v.stats;
    ms->view.cnt1 = nullptr;
    ms->pending = true;
    ms->leaf_form = false;
    ms->list_n_valid = false;
    return CFRK_OK;
  }
  // at least 2048 leaves (one workgroup each in RX3), at most 2^13 counters per leaf
  v.b1 = 8;
  v.idx = std::max(0, std::min(RX_IDX_MAX, 2 * k - 11));
  v.b2 = 2 * k - v.b1 - v.idx;
  v.kmask = (uint32_t)((1ull << (2 * k)) - 1ull);
  v.mul = 0x9E3779B1u;
  v.inv = inv_odd32(v.mul);
  const uint64_t nb1 = 1ull << v.b1, nleaf = 1ull << (v.b1 + v.b2);
  const uint64_t cap1 = ((uint64_t)((double)nN / (double)(nb1 * RX_NREG) * 1.3) + 4096 + 15) & ~15ull;   // (RX2 loads 8 keys at a time)
  const uint64_t cap2 = (uint64_t)((double)nN / (double)nleaf * 1.5) + 1024;
  v.hi8 = (2 * k - v.b1 > 16) ? 1u : 0u;
  const size_t l1_elems = (size_t)nb1 * RX_NREG * cap1;
  const size_t need = l1_elems * 3 + (size_t)nleaf * cap2 * 2 + (size_t)ctx->g_cap * 12;
  const size_t have = ctx->pool[BUF_MSP_L1].cap + ctx->pool[BUF_MSP_L2].cap + ctx->pool[BUF_MSP_OUTK].cap + ctx->pool[BUF_MSP_OUTC].cap;
  if (need > have) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    if (need > have + free_b) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "radix path needs %zu B, %zu B available", need, have + free_b);
  }
  void *p;
  // (the 16-bit plane, then the 8-bit plane; 64 bytes to spare: RX2's last vector loads may read past a region's keys)
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, l1_elems * 3 + 64, &p))) return rc;
  v.k1lo = (uint16_t *)p; v.k1hi = (uint8_t *)(v.k1lo + l1_elems); v.cap1 = cap1;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)nleaf * cap2 * 2 + 64, &p))) return rc;
  v.key2 = (uint16_t *)p; v.cap2 = cap2;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, (size_t)(nb1 * RX_NREG + nleaf) * 4, &p))) return rc;
  v.cnt1 = (uint32_t *)p; v.cnt2 = v.cnt1 + nb1 * RX_NREG;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_keys = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats;
  TableView t = cfrk_table_view(ctx);

  HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, (size_t)(nb1 * RX_NREG + nleaf) * 4, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
  const int64_t tiles = (nN + (int64_t)RX1_KEYS - 1) / RX1_KEYS;
  if (tiles > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
  const bool canon = (ctx->g_flags & CFRK_CANONICAL) != 0;
  v.exact = 0; v.lbase = nullptr; v.lcap = nullptr;
  v.exact1 = 0; v.rbase = nullptr; v.rcap = nullptr;
  int64_t tiles_per_sub = (int64_t)((cap1 + RX2_KEYS - 1) / RX2_KEYS);
  const int64_t bins_per_group = (int64_t)((nb1 + RX_NXG - 1) / RX_NXG);
  const size_t nreg = (size_t)nb1 * RX_NREG;
  // Few distinct keys, each seen very often (deep coverage of a small genome, a single amplicon):
  // their level-1 regions or their leaves overflow the fixed stride.  The cursors counted the exact
  // demand: the level is laid out back to back (all keys together never exceed the buffer) and its
  // kernel runs again (see msp.hip).
  bool run_rx1 = true, settled = false;
  for (int attempt = 0; attempt < 4 && !settled; ++attempt) {
    if (run_rx1) {
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L1OVF, 0, sizeof(uint64_t), ctx->stream));
      if (canon) hipLaunchKernelGGL((rx1_kernel<true>), dim3((unsigned)tiles), dim3(RX1_THREADS), 0, ctx->stream, d_data, nN, v, t);
      else hipLaunchKernelGGL((rx1_kernel<false>), dim3((unsigned)tiles), dim3(RX1_THREADS), 0, ctx->stream, d_data, nN, v, t);
      HIP_TRY(ctx, hipGetLastError());
    }
    const int64_t g2 = bins_per_group * RX_NXG * RX_NREG * tiles_per_sub;
    if (g2 > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, sizeof(uint64_t), ctx->stream));
    hipLaunchKernelGGL(rx2_kernel, dim3((unsigned)g2), dim3(RX2_THREADS), 0, ctx->stream, (int)tiles_per_sub, v, t);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t st[ST_NWORDS];
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st[ST_L1OVF]) {
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT1, nreg * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
      uint64_t *rbase = (uint64_t *)p;
      uint32_t *rcap = (uint32_t *)(rbase + nreg);
      hipLaunchKernelGGL(rx_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt1, (uint32_t)nreg, rbase, rcap);
      HIP_TRY(ctx, hipGetLastError());
      std::vector<uint32_t> c1(nreg);
      HIP_TRY(ctx, hipMemcpyAsync(c1.data(), v.cnt1, nreg * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      uint64_t maxreg = 0;
      for (size_t r = 0; r < nreg; ++r) maxreg = std::max<uint64_t>(maxreg, c1[r]);
      tiles_per_sub = (int64_t)((maxreg + RX2_KEYS - 1) / RX2_KEYS);     // the largest region decides RX2's grid
      HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, (nreg + (size_t)nleaf) * sizeof(uint32_t), ctx->stream));   // cnt1 and cnt2
      v.exact1 = 1; v.rbase = rbase; v.rcap = rcap;
      run_rx1 = true;
      continue;
    }
    if (st[ST_L2OVF]) {
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT, (size_t)nleaf * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
      uint64_t *lbase = (uint64_t *)p;
      uint32_t *lcap = (uint32_t *)(lbase + nleaf);
      hipLaunchKernelGGL(rx_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt2, (uint32_t)nleaf, lbase, lcap);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemsetAsync(v.cnt2, 0, (size_t)nleaf * sizeof(uint32_t), ctx->stream));
      v.exact = 1; v.lbase = lbase; v.lcap = lcap;
      run_rx1 = false;
      continue;
    }
    settled = true;
  }
  if (!settled) return cfrk_fail(ctx, CFRK_ERR_STATE, "the key regions did not settle after an exact layout");
  hipLaunchKernelGGL(rx3_kernel, dim3((unsigned)nleaf), dim3(RX3_THREADS), 0, ctx->stream, v);
  HIP_TRY(ctx, hipGetLastError());
  // the result list lives where msp.hip keeps its own: digest / export / fold are shared
  ms->view.out_keys = v.out_keys; ms->view.out_cnt = v.out_cnt; ms->view.out_cap = v.out_cap;
  ms->view.stats = v.stats;
  ms->view.cnt1 = nullptr;
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}
