// radix.hip -- global k-mer counting for k <= 16 on gfx950: keys fit 32 bits, so instead of
// hashing the key is radix-partitioned and counted by DIRECT ADDRESS in LDS.
//
//   keys are first scrambled by one odd multiply mod 4^k (a bijection) -- canonical k-mers are skewed low and
//   real genomes are not uniform, the product's top bits are;
//   RX1  extract (canonical) k-mers (2 x dwordx4 per lane, 2-bit packing in registers), counting
//        sort a tile of 8192 keys by the top b1 = 8 bits in LDS (one returning LDS atomic per key), one HBM
//        atomic per bin per tile;
//   RX2  split every region by the next b2 bits the same way (leaf streams written in padded groups of eight);
//   RX3  persistent workgroups walk the leaves: an LDS array of counters addressed by the low idx bits, then the
//        non-zero counters are un-scrambled and appended to the result list through an LDS buffer (one cursor
//        atomic per ~10 leaves).
//   Leaf shape, chosen on the HOST from k and the batch size (cfrk_radix_count):
//     8 <= k <= 12: idx = 2k - 11 (5..13) counters of 32 bits, replicated when there are fewer than 2^13;
//     13 <= k <= 15: idx = 14 -- 2^14 keys per leaf as PACKED 16-bit counters in 2^13 words -- while a leaf is
//        expected to stay well under 2^16 elements (exact below that: no half can carry; a leaf that turns out
//        larger is counted in two passes over its stream with 32-bit counters), else idx = 13, 32-bit counters;
//     k = 16 (round 5): idx = 15, packed 16-bit counters in 2^14 words (64 KB of LDS, 512 threads), b2 = 9 --
//        for batches whose leaves stay under 2^16 elements (~4e9 bases); larger ones take msp.hip;
//     b2 = 2k - 8 - idx (3..9): >= 2048 leaves.
//   k <= 7 (at most 16384 keys): no partition at all, every workgroup counts into a replicated
//   LDS table and adds it to a dense HBM array once.
//
// HBM never sees the bits a key's region already implies: level 1 stores the 2k-8 bits below the
// bin as a 16-bit plane plus (2k-8 >= 16) an 8-bit plane -- 3 bytes per k-mer at k = 15 and 16 instead of
// 4 --, the leaves store the low idx <= 15 bits as 16-bit words; everything moves in groups of eight
// elements (16- and 8-byte vectors; segments are padded, see RX_PAD); every occurrence is one LDS atomic.
// Same semantics and the same result-list form as msp.hip (which covers 16 <= k <= 32); a level that
// overflows its regions is laid out again with exact sizes.
#include "msp.h"
#include "table.h"
#include "msp_dev.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int RX_NXG = 8;                 // workgroup-id groups (XCD affinity, speed only)
constexpr int RX_NREG = 32;               // sub-regions (cursors) per level-1 bin, RX_NREG / RX_NXG per XCD:
                                          // the returning atomics on one cursor serialise (see msp_dev.h: NXG)
constexpr int RX1_THREADS = 512, RX1_PER = 16, RX1_KEYS = RX1_THREADS * RX1_PER;
constexpr int RX2_THREADS = 512, RX2_PER = 16, RX2_KEYS = RX2_THREADS * RX2_PER;   // (32 per thread: 11 % pads instead of 22 %, but 114 VGPRs and two workgroups per CU: RX2 2.33 ms against 2.1)
constexpr int RX_IDX_MAX = 13;            // 2^13 32-bit counter words per leaf workgroup (k <= 15); k = 16: 2^14

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
  uint32_t mul, inv, kmask;                           // scramble: x * mul mod 4^k
  uint32_t dbg;                                       // cfrk_debug_set_flags (timing ablations)
};

// level-1 region / cursor of (bin, sub-region): sub-region major -- the cursors a workgroup reserves
// from lie side by side (memory-side atomics: one request per touched 64 bytes, see msp.hip l1_reg)
__device__ __forceinline__ uint32_t rx_reg(const RxView &v, uint32_t bin, uint32_t sub) { return (sub << v.b1) | bin; }
// The scramble is one odd multiply mod 4^k (a bijection): the level-1 bin is the product's TOP bits, which
// mix every bit of the key -- canonical k-mers are skewed low, the product's top bits are not.  (Until round 4
// an xorshift by k followed: two more instructions per position in RX1 for bits whose balance nothing needs --
// the low idx bits only address a leaf's counters, and an unbalanced level is laid out again anyway.)
__device__ __forceinline__ uint32_t rx_mix(const RxView &v, uint32_t key) { return (key * v.mul) & v.kmask; }
__device__ __forceinline__ uint32_t rx_unmix(const RxView &v, uint32_t x) { return (x * v.inv) & v.kmask; }

// WIDE STORES (round 4).  What the two partition kernels cost was their copy-out, and not the scatter: a
// wave-wide 2-byte (or 1-byte) store occupies the CU's address path for ~20 clocks whatever it carries, 128 bytes or
// 1024 (profiles/r04/c2_radix_ablations.txt: RX1 2.17 ms, 1.16 without its copy-out, 1.95 with the 16-bit plane written
// back to back instead of scattered).  RX2 therefore writes the leaf streams in groups of EIGHT keys: a sub-bin's segment
// of a tile is rounded up to a multiple of eight with pad elements (all ones: no key of a leaf looks like that), every
// leaf cursor advances in multiples of eight and every leaf stream starts on one, and a LANE copies its sub-bin's groups
// as whole 16-byte vectors; the leaf kernel sends the pads to a spare counter.  (RX1 was built the same way and measured
// slower, see there: level 1 is written key by key.)
constexpr uint32_t RX_PAD = 0xFFFFFFFFu;
constexpr int RX_GROUP = 8;
// RX1 writes key by key (see there): sorted position p of bin b lands at element dabs + p of the level-1 buffer
// while p < plim (the part of the bin's reservation that fits its region).  One ds_read_b128 per key.
struct RxDst { uint32_t dlo, dhi, plim, pad; };
static_assert(sizeof(RxDst) == 16, "one LDS read");

// ------------------------------------------------------------------------------------------ RX1
// 512 threads x 16 window starts (a lane holds 16 keys + 8 packed ranks: eight waves per SIMD).  ONE returning LDS
// atomic per key (its rank inside the bin; the rank waits in a register for the scan -- no second "fill" atomic),
// window starts that are no k-mer go to a dummy bin of their lane and to spare slots instead of around the atomics
// in branches, LDS-only barriers and DPP scans (msp_dev.h), 32-bit arithmetic throughout.
// This kernel is bound by its LDS work (~4 LDS operations per key, the atomics and the sort's scattered writes with
// 3-5-way bank conflicts): variants that pad every bin's segment to groups of eight and copy them out as 16- and 8-byte
// vectors (as RX2 does) were measured SLOWER, 2.48 / 2.58 ms against 2.15 -- the second plane's LDS write and the pad
// fill cost more than the wide stores save (profiles/r04/c2_radix_ablations.txt) -- so level 1 is written key by key
// and is not padded.
__device__ __forceinline__ void rx_load_chunk16(const int8_t *__restrict__ data, int64_t off, int64_t nN, uint32_t &b, uint32_t &bad) {
  if (off + 16 <= nN) {
    dev_pack16(*reinterpret_cast<const uint4 *>(data + off), b, bad);
  } else {                                       // the buffer's last bytes
    b = 0; bad = 0;
    for (int j = 0; j < 16; ++j) {
      const int c = (off + j < nN) ? (int)data[off + j] : -1;
      const bool inv = (c < 0 || c > 3);
      b |= (inv ? 0u : (uint32_t)c) << (30 - 2 * j);
      if (inv) bad |= 1u << (15 - j);
    }
  }
}

// K16: k = 16 -- a key fills its 32-bit register, so a window start that is no k-mer cannot be told by a "key" whose
// bin is 256 + lane; its bin is worked out from the validity mask again where it is needed (three instructions
// per key more; an instantiation of its own so that k <= 15 does not pay)
template <bool CANON, bool K16>
__global__ __launch_bounds__(RX1_THREADS) void rx1_kernel(const int8_t *__restrict__ data, int64_t nN,
                                                          RxView v, TableView t) {
  __shared__ uint32_t sorted[RX1_KEYS + 64];                   // + 64 spare slots for the dummy bins
  __shared__ uint32_t hist[320], loff[320];                    // (bins 256 .. 319: one dummy bin per lane)
  __shared__ RxDst dst[256];
  __shared__ uint32_t wtot[4];
  (void)t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int k = v.k;
  if (tid < 320) hist[tid] = 0;
  if (tid < 64) loff[256 + tid] = (uint32_t)RX1_KEYS;
  lds_barrier();

  const int64_t off = ((int64_t)blockIdx.x * RX1_THREADS + tid) * RX1_PER;
  uint32_t b0, bad0;
  rx_load_chunk16(data, off, nN, b0, bad0);
  uint32_t n0 = dev_lane_next(b0), nbad = dev_lane_next(bad0);
  if (lane == 63) rx_load_chunk16(data, off + 16, nN, n0, nbad);
  const int sh1 = 2 * k - v.b1;
  // The k-mer at position i is the top 2k bits of the 32-bit window of the base string that starts
  // there (one v_alignbit with a static shift), its reverse complement the low 2k bits of a window
  // of the reverse-complemented string that ENDS where the k-mer starts (msp_dev.h: msp_minimizers),
  // its validity the top k bits of the invalid-base mask shifted to the position: 32-bit work throughout.
  const uint32_t D[2] = {b0, n0};
  uint32_t R[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    uint32_t x = __brev(D[i]);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    R[2 - i] = ~x;
  }
  R[0] = 0;
  const uint32_t M = (bad0 << 16) | (nbad & 0xFFFFu);          // bit (31 - j): base j of the 32 is invalid
  const int fsh = 32 - 2 * k;
  const uint32_t vlim = 1u << (32 - k);            // a window whose top k mask bits are clear is below this
  const uint32_t mul = v.mul, kmask = v.kmask;

  uint32_t keys[RX1_PER], rk2[RX1_PER / 2];                   // (two ranks per register)
  const uint32_t dummy = 256u + (uint32_t)lane;               // this lane's dummy bin
  const uint32_t inval = K16 ? 0u : (dummy << sh1);           // ... and its "key" (sh1 <= 22: fits)
  auto bin_of = [&](int i) { return (!K16 || (M << i) < vlim) ? (keys[i] >> sh1) : dummy; };
#pragma unroll
  for (int i = 0; i < RX1_PER; ++i) {
    const uint32_t X = i ? __builtin_amdgcn_alignbit(D[0], D[1], 32 - 2 * i) : D[0];
    uint32_t key = K16 ? X : (X >> fsh);
    if (CANON) {
      const int o2 = 64 - 2 * i, q2 = o2 >> 5, r2 = o2 & 31;
      const uint32_t Y = r2 ? __builtin_amdgcn_alignbit(R[q2], R[q2 + 1], 32 - r2) : R[q2];
      key = min(key, K16 ? Y : (Y & kmask));
    }
    const uint32_t Wm = M << i;
    key = K16 ? key * mul : ((key * mul) & kmask);
    if (!K16) key = (Wm < vlim) ? key : inval;
    keys[i] = key;
    const uint32_t r_ = atomicAdd(&hist[bin_of(i)], 1u);
    rk2[i >> 1] = (i & 1) ? (rk2[i >> 1] | (r_ << 16)) : r_;
  }
  lds_barrier();
  // ---- thread b < 256 owns bin b: one reservation per non-empty bin, bin offsets, destinations ----
  const uint32_t subreg = blockIdx.x & (RX_NREG - 1);
  const uint32_t c = tid < 256 ? hist[tid] : 0u;
  uint32_t my_base = 0;                                  // (returning atomic: consumed after the sort)
  if (c) my_base = atomicAdd(&v.cnt1[rx_reg(v, tid, subreg)], c);
  block_scan<256, true>(hist, loff, wtot);
#pragma unroll
  for (int i = 0; i < RX1_PER; ++i) {
    const uint32_t r_ = (i & 1) ? (rk2[i >> 1] >> 16) : (rk2[i >> 1] & 0xFFFFu);
    sorted[min(loff[bin_of(i)] + r_, (uint32_t)RX1_KEYS + 63u)] = keys[i];
  }
  if (tid < 256) {
    const uint32_t reg = rx_reg(v, tid, subreg);
    const uint64_t cap = v.exact1 ? (uint64_t)v.rcap[reg] : v.cap1;
    const uint64_t at = v.exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1;
    const uint32_t lo = loff[tid];
    uint64_t d = at + my_base - lo;
    if (v.dbg & CFRK_ABL_RX1_LINEAR) d = (uint64_t)(blockIdx.x & 255u) * 16384u;   // (timing: the tile stays in its XCD's L2)
    const uint64_t room = cap > (uint64_t)my_base ? cap - my_base : 0;
    RxDst e;
    e.dlo = (uint32_t)d; e.dhi = (uint32_t)(d >> 32);
    e.plim = (uint32_t)min((uint64_t)lo + room, (uint64_t)0xFFFFFFFFu);
    e.pad = 0;
    dst[tid] = e;
    // the cursor keeps counting past the region: the host then redoes RX1 with exact sizes
    if ((uint64_t)c > room) v.stats[ST_L1OVF] = 1;
  }
  lds_barrier();
  if (v.dbg & CFRK_ABL_RX1_NO_OUT) return;
  // ---- copy out in bin order: consecutive lanes -> consecutive elements of one region ----
  const uint32_t total = loff[255] + hist[255];
  const uint32_t lowmask = (1u << sh1) - 1u;
  for (uint32_t p = tid; p < total; p += RX1_THREADS) {
    const uint32_t key = sorted[p];
    const RxDst e = dst[key >> sh1];
    if (p < e.plim) {
      const uint64_t at = (((uint64_t)e.dhi << 32) | e.dlo) + p;
      v.k1lo[at] = (uint16_t)(key & lowmask);          // (below 16 bits per key the bin's bits would read as a pad)
      if (v.hi8) v.k1hi[at] = (uint8_t)((key & lowmask) >> 16);
    }
  }
}

// ------------------------------------------------------------------------------------------ RX2
// One tile of RX2_KEYS elements of one level-1 region, split by the next b2 bits (<= 512 sub-bins) into the
// leaves' 16-bit streams.  Same economy as RX1: one returning LDS atomic per key, pads and the elements beyond
// the tile's end go to a dummy bin of their lane, thread b writes sub-bin b's groups as 16-byte vectors.
template <bool VEC>   // VEC: level 1 in its fixed-stride layout (regions start on 16-element boundaries: vector loads)
__global__ __launch_bounds__(RX2_THREADS) void rx2_kernel(int tiles_per_sub, RxView v, TableView t) {
  constexpr int SLOTS = RX2_KEYS + 512 * (RX_GROUP - 1) + 8;
  constexpr int TRASH = SLOTS;
  static_assert((SLOTS + 64) % 8 == 0, "whole vectors");
  __shared__ uint4 s4[(SLOTS + 64) / 8];
  __shared__ uint32_t hist[576], hp[512], loff[576];
  __shared__ unsigned long long dabs[512];
  __shared__ uint16_t vb[(SLOTS + 64) / 8];
  __shared__ uint32_t wtot[8];
  (void)t;
  uint16_t *const s16 = reinterpret_cast<uint16_t *>(s4);
  const int tid = threadIdx.x, lane = tid & 63;
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
  hist[tid] = 0;
  if (tid < 64) { hist[512 + tid] = 0; loff[512 + tid] = (uint32_t)TRASH; }
  for (int s = tid; s < (SLOTS + 64) / 8; s += RX2_THREADS) s4[s] = make_uint4(RX_PAD, RX_PAD, RX_PAD, RX_PAD);
  lds_barrier();
  const uint64_t rb = (v.exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1) + r0;
  const uint32_t idxb = (uint32_t)v.idx, m2 = (1u << v.b2) - 1u;
  const uint32_t klim = 1u << (2 * v.k - v.b1);              // a key of this level is below it, a pad is not
  uint32_t kk[RX2_PER], rk2[RX2_PER / 2];        // the 2k - b1 bits below the bin; ranks inside the sub-bins, two per register
  static_assert(RX2_PER % 8 == 0, "keys are loaded eight at a time");
  // 16-byte loads of the 16-bit plane (eight keys) and 8-byte loads of the 8-bit plane (a region starts on a
  // 16-element boundary unless the level was laid out again: cap1 is a multiple of 16); elements beyond the
  // region's keys become pads.  Which key a thread holds does not matter to a counting sort.
  if (VEC) {
    const uint4 *slo = reinterpret_cast<const uint4 *>(v.k1lo + rb);
    const uint2 *shi = reinterpret_cast<const uint2 *>(v.k1hi + rb);
    const bool full = nt == (uint32_t)RX2_KEYS;
#pragma unroll
    for (int q = 0; q < RX2_PER / 8; ++q) {
      const uint32_t g = (uint32_t)q * RX2_THREADS + tid;      // group of eight elements
      uint4 x = make_uint4(RX_PAD, RX_PAD, RX_PAD, RX_PAD);
      uint2 h = make_uint2(RX_PAD, RX_PAD);
      if (8u * g < nt) {
        x = slo[g];
        if (v.hi8) h = shi[g]; else h = make_uint2(0u, 0u);
      }
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t lo16 = (c & 1) ? (w[c >> 1] >> 16) : (w[c >> 1] & 0xFFFFu);
        const uint32_t hw = c < 4 ? h.x : h.y;
        const uint32_t hb = ((c & 3) == 3) ? (hw >> 24) : ((hw >> (8 * (c & 3))) & 0xFFu);
        kk[8 * q + c] = (hb << 16) | lo16;
      }
      if (!full) {                                             // (the region's last tile: wave-uniform)
#pragma unroll
        for (int c = 0; c < 8; ++c) if (8u * g + (uint32_t)c >= nt) kk[8 * q + c] = RX_PAD;
      }
    }
  } else {
    const uint16_t *slo = v.k1lo + rb;
    const uint8_t *shi = v.k1hi + rb;
#pragma unroll
    for (int i = 0; i < RX2_PER; ++i) {
      const uint32_t idx = (uint32_t)i * RX2_THREADS + tid;
      kk[i] = (idx < nt) ? ((uint32_t)slo[idx] | (v.hi8 ? (uint32_t)shi[idx] << 16 : 0u)) : RX_PAD;
    }
  }
  const uint32_t dummy = 512u + (uint32_t)lane;
#pragma unroll
  for (int i = 0; i < RX2_PER; ++i) {
    const uint32_t b = (kk[i] < klim) ? ((kk[i] >> idxb) & m2) : dummy;
    const uint32_t r_ = atomicAdd(&hist[b], 1u);
    rk2[i >> 1] = (i & 1) ? (rk2[i >> 1] | (r_ << 16)) : r_;
  }
  lds_barrier();
  const uint32_t c = hist[tid];                          // (non-zero only for tid <= m2)
  const uint32_t cp = (c + (RX_GROUP - 1)) & ~(uint32_t)(RX_GROUP - 1);
  hp[tid] = cp;
  uint32_t my_base = 0;
  const uint64_t leaf = ((uint64_t)bin1 << v.b2) + (uint32_t)tid;
  if (cp) my_base = atomicAdd(&v.cnt2[leaf], cp);
  block_scan<512, true>(hp, loff, wtot);
  const uint32_t imask = (1u << idxb) - 1u;
#pragma unroll
  for (int i = 0; i < RX2_PER; ++i) {
    const uint32_t b = (kk[i] < klim) ? ((kk[i] >> idxb) & m2) : dummy;
    const uint32_t r_ = (i & 1) ? (rk2[i >> 1] >> 16) : (rk2[i >> 1] & 0xFFFFu);
    s16[min(loff[b] + r_, (uint32_t)TRASH + 63u)] = (uint16_t)(kk[i] & imask);
  }
  // copy out: consecutive lanes take consecutive 16-byte groups of the padded tile (a wave-wide store then touches
  // ~27 streams instead of 64: what a store costs grows with the lines it touches); thread b first publishes where
  // sub-bin b's groups go and marks them
  {
    uint32_t nv = 0;
    uint64_t at = 0;
    if (cp) {
      const uint64_t cap = v.exact ? (uint64_t)v.lcap[leaf] : v.cap2;
      at = (v.exact ? v.lbase[leaf] : leaf * v.cap2) + my_base;
      const uint64_t room = cap > (uint64_t)my_base ? cap - my_base : 0;
      if ((uint64_t)cp > room) {
        v.stats[ST_L2OVF] = 1;                              // the cursor keeps counting: the host redoes RX2 with exact sizes
        if (my_base >= 0x80000000u) v.stats[ST_CWRAP] = 1;  // ... unless a single-key flood is about to wrap it: counted through the HBM table
      }
      nv = (uint32_t)min((uint64_t)cp, room) / RX_GROUP;
    }
    const uint32_t src = loff[tid] / RX_GROUP;
    dabs[tid] = at - (uint64_t)loff[tid];                // element of group g of the tile: dabs + 8 g
    for (uint32_t j = 0; j < cp / RX_GROUP; ++j) vb[src + j] = (j < nv) ? (uint16_t)tid : (uint16_t)0xFFFFu;
  }
  lds_barrier();
  if (v.dbg & CFRK_ABL_RX2_NO_OUT) return;
  const uint32_t totalv = (loff[511] + hp[511]) / RX_GROUP;
  for (uint32_t g = tid; g < totalv; g += RX2_THREADS) {
    const uint32_t b = vb[g];
    if (b != 0xFFFFu) reinterpret_cast<uint4 *>(v.key2 + dabs[b])[g] = s4[g];
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
// 2^13 LDS counters per leaf (2^idx of them, replicated 2^rlog times when a leaf has fewer), one ds_add per
// key; a leaf's stream is whole vectors of eight elements (pads go to a spare counter of their lane).
// PERSISTENT workgroups: a workgroup walks leaves b, b + grid, ...; the next leaf's size is asked for before
// the current leaf is counted and its first keys before the current leaf is scanned.
// ONE CURSOR ATOMIC PER ~10 LEAVES: 131 072 returning atomics on the single result cursor were 0.73 of this
// kernel's 1.7 ms (same-address atomics serialise at the memory side, ~5.5 ns each;
// profiles/r04/c2_radix_ablations.txt).  A workgroup therefore collects the (key, count) entries of its leaves in
// an LDS buffer and takes one range from the cursor when the buffer fills up (C2: ~76 entries per leaf); the
// flush also writes the list in whole coalesced lines instead of one 8 + 4 byte pair per thread.
constexpr int RX3_INFL = 6;            // 16-byte loads in flight per thread (12 288 elements per round of the workgroup)
constexpr int RX3_OB = 704;            // buffered result entries (5.5 KB: four workgroups still fit a CU's LDS)
// NCLOG = log2 of the 32-bit counter words: 13 with 256 threads (32 KB: four workgroups per CU) for k <= 15, 14 with
// 512 threads (64 KB: two per CU, the same sixteen waves) for k = 16, whose leaves hold 2^15 keys as packed counters
template <int NCLOG, int RX3_THREADS>
__global__ __launch_bounds__(RX3_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rx3_kernel(RxView v, uint32_t nleaf) {
  constexpr int NC = 1 << NCLOG;                 // 32-bit counter words
  __shared__ uint4 cnt4[(NC + 64) / 4];          // + one spare counter per lane for the pads
  __shared__ uint32_t ob_key[RX3_OB], ob_cnt[RX3_OB];
  __shared__ unsigned long long wg_base;
  __shared__ uint32_t wsum[RX3_THREADS / 64];
  uint32_t *const cnt = reinterpret_cast<uint32_t *>(cnt4);
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t nidx = 1u << v.idx;
  // idx = 14 (k >= 13): a leaf has 2^14 keys and the 2^13 words hold them as PACKED 16-bit counters -- exact as long as
  // the leaf holds fewer than 2^16 elements (no half can carry into its neighbour), which the leaf's size tells before
  // anything is counted; a larger leaf is counted in two passes over its stream, one per half of its key range, with
  // 32-bit counters.  Half as many leaves as with idx = 13: half the per-leaf work here, half the sub-bins (twice as
  // long segments, half the pads) in RX2.
  // idx = 13: one 32-bit counter per key.  idx < 13 (small k): 2^idx counters replicated 2^rlog times -- or all 256
  // threads would serialise on a handful of LDS words (nidx << rlog = 2^13 for every k this path serves: idx >= 5).
  const bool wide = v.idx > NCLOG;
  const int rlog = wide ? 0 : min(NCLOG - v.idx, 8);
  const uint32_t rmask = (1u << rlog) - 1u, rep = (uint32_t)tid & rmask;
  const uint32_t spare = (uint32_t)NC + (uint32_t)lane;
  constexpr int NQ = NC / 4 / RX3_THREADS;       // uint4 groups of counter words per thread (8)
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  uint32_t leaf = blockIdx.x;
  if (leaf >= nleaf) return;
#pragma unroll
  for (int j = 0; j < NQ; ++j) cnt4[j * RX3_THREADS + tid] = zero4;
  if (tid < 16) cnt4[NC / 4 + tid] = zero4;

  auto leaf_n = [&](uint32_t l) { return (uint32_t)min((uint64_t)v.cnt2[l], v.exact ? (uint64_t)v.lcap[l] : v.cap2); };
  auto leaf_p = [&](uint32_t l) {                 // (a leaf's stream starts on a multiple of eight elements)
    return reinterpret_cast<const uint4 *>(v.exact ? v.key2 + v.lbase[l] : v.key2 + (uint64_t)l * v.cap2);
  };
  uint4 x[RX3_INFL];
  auto load_round = [&](const uint4 *p, uint32_t n4, uint32_t q0) {
#pragma unroll
    for (int u = 0; u < RX3_INFL; ++u) {
      const uint32_t q = q0 + (uint32_t)u * RX3_THREADS + tid;
      x[u] = make_uint4(RX_PAD, RX_PAD, RX_PAD, RX_PAD);
      if (q < n4) x[u] = p[q];
    }
  };
  // mode 0: one 32-bit counter per key (idx = 13); 1: replicated (idx < 13); 2: packed 16-bit (idx = 14, small leaf);
  // 3: idx = 14, the keys of half `hsel` only, 32-bit counters
  auto count_round = [&](int mode, uint32_t hsel) {
#pragma unroll
    for (int u = 0; u < RX3_INFL; ++u) {
      const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t key = (c & 1) ? (w[c >> 1] >> 16) : (w[c >> 1] & 0xFFFFu);       // (a pad is 0xFFFF)
        if (mode == 0) atomicAdd(&cnt[min(key, spare)], 1u);
        else if (mode == 1) atomicAdd(&cnt[key < nidx ? ((key << rlog) | rep) : spare], 1u);
        else if (mode == 2) atomicAdd(&cnt[min(key >> 1, spare)], (key & 1u) ? 0x10000u : 1u);
        else atomicAdd(&cnt[(key >> NCLOG) == hsel ? (key & (uint32_t)(NC - 1)) : spare], 1u);
      }
    }
  };
  // flush the buffered entries: one range of the result list, written in order
  uint32_t fill = 0;                               // (the same in every thread)
  auto flush = [&]() {
    if (tid == 0) wg_base = (v.dbg & CFRK_ABL_RX3_NO_CURSOR) ? (unsigned long long)blockIdx.x * 16384ull
                                                              : atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)fill);
    __syncthreads();
    const unsigned long long b = wg_base;
    for (uint32_t i = tid; i < fill; i += RX3_THREADS) {
      if (b + i < v.out_cap) { v.out_keys[b + i] = (uint64_t)rx_unmix(v, ob_key[i]); v.out_cnt[b + i] = ob_cnt[i]; }
      else v.stats[ST_OVERFLOW] = 1;
    }
    lds_barrier();                                 // (the buffer and wg_base are free again)
    fill = 0;
  };

  uint32_t n = leaf_n(leaf);
  const uint4 *cur = leaf_p(leaf);
  load_round(cur, n / RX_GROUP, 0);
  lds_barrier();                                  // the counters are clear
  for (;;) {
    const uint32_t next = leaf + gridDim.x;
    const bool more = next < nleaf;
    uint32_t n_next = 0;
    if (more) n_next = leaf_n(next);              // (in flight while this leaf is counted)
    const uint32_t n4 = n / RX_GROUP;
    const int mode = !wide ? (rlog ? 1 : 0) : (n < 65536u ? 2 : 3);
    const uint32_t npass = mode == 3 ? 2u : 1u;
    const uint4 *nxp = leaf_p(more ? next : leaf);
    for (uint32_t pass = 0; pass < npass; ++pass) {
      if (pass) { load_round(cur, n4, 0); }       // (the second half of a large leaf: its stream once more)
      if (!(v.dbg & CFRK_ABL_RX3_NO_COUNT)) count_round(mode, pass);
      for (uint32_t q0 = (uint32_t)RX3_INFL * RX3_THREADS; q0 < n4; q0 += (uint32_t)RX3_INFL * RX3_THREADS) {
        load_round(cur, n4, q0);
        count_round(mode, pass);
      }
      lds_barrier();
      if (more && pass + 1u == npass) load_round(nxp, n_next / RX_GROUP, 0);   // (in flight while this leaf is scanned and its entries collected)
      if (!n) continue;                           // (wave-uniform: one leaf per workgroup)
      if (rlog) {                                 // fold the replicas into replica 0
        for (uint32_t s = tid; s < nidx; s += RX3_THREADS) {
          uint32_t c = 0;
          for (uint32_t r = 0; r <= rmask; ++r) c += cnt[(s << rlog) | r];
          cnt[s << rlog] = c;
        }
        lds_barrier();
      }
      // compaction: a thread counts the non-zero counters among its own, one block scan gives it a range of
      // the leaf's entries.  Modes 0, 2, 3: its counters are eight groups of four consecutive words (two counters
      // per word in mode 2); mode 1: counter s sits at word s << rlog.
      uint32_t mine = 0;
      if (mode == 1) {
        for (uint32_t s = tid; s < nidx; s += RX3_THREADS) mine += (cnt[s << rlog] != 0u) ? 1u : 0u;
      } else {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const uint4 c4 = cnt4[j * RX3_THREADS + tid];
          if (mode == 2) mine += ((c4.x & 0xFFFFu) != 0u) + ((c4.x >> 16) != 0u) + ((c4.y & 0xFFFFu) != 0u) + ((c4.y >> 16) != 0u) +
                                 ((c4.z & 0xFFFFu) != 0u) + ((c4.z >> 16) != 0u) + ((c4.w & 0xFFFFu) != 0u) + ((c4.w >> 16) != 0u);
          else mine += (c4.x != 0u) + (c4.y != 0u) + (c4.z != 0u) + (c4.w != 0u);
        }
      }
      const uint32_t incl = dev_wave_scan_incl(mine);
      if (lane == 63) wsum[tid >> 6] = incl;
      lds_barrier();
      uint32_t base = 0, total = 0;
#pragma unroll
      for (int w = 0; w < RX3_THREADS / 64; ++w) { const uint32_t xw = wsum[w]; base += (w < (tid >> 6)) ? xw : 0u; total += xw; }
      // where this leaf's entries go: behind the buffered ones, or -- more entries than the buffer
      // holds -- straight to a range of the list of their own
      const bool direct = total > (uint32_t)RX3_OB;
      if (fill && (direct || fill + total > (uint32_t)RX3_OB)) flush();
      unsigned long long dbase = 0;
      if (direct) {
        if (tid == 0) wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)total);
        __syncthreads();
        dbase = wg_base;
      }
      uint32_t d = (direct ? 0u : fill) + base + incl - mine;
      const uint32_t leaf_hi = (leaf << v.idx) | (mode == 3 ? pass << NCLOG : 0u);
      auto emit = [&](uint32_t s, uint32_t c) {
        if (direct) {
          if (dbase + d < v.out_cap) { v.out_keys[dbase + d] = (uint64_t)rx_unmix(v, leaf_hi | s); v.out_cnt[dbase + d] = c; }
          else v.stats[ST_OVERFLOW] = 1;
        } else { ob_key[d] = leaf_hi | s; ob_cnt[d] = c; }
        ++d;
      };
      if (mode == 1) {
        for (uint32_t s = tid; s < nidx; s += RX3_THREADS) {
          const uint32_t c = cnt[s << rlog];
          if (c) emit(s, c);
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < NQ; ++j) cnt4[j * RX3_THREADS + tid] = zero4;
      } else {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (mine) {
            const uint4 c4 = cnt4[j * RX3_THREADS + tid];
            const uint32_t w0 = 4u * (uint32_t)(j * RX3_THREADS + tid);
            const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (mode == 2) {
                if (cw[q] & 0xFFFFu) emit(2u * (w0 + (uint32_t)q), cw[q] & 0xFFFFu);
                if (cw[q] >> 16) emit(2u * (w0 + (uint32_t)q) + 1u, cw[q] >> 16);
              } else if (cw[q]) emit(w0 + (uint32_t)q, cw[q]);
            }
          }
          cnt4[j * RX3_THREADS + tid] = zero4;     // (this thread is the only reader of these four)
        }
      }
      if (!direct) fill += total;
      lds_barrier();                              // the counters are clear again (and wsum / wg_base are free)
    }
    if (!more) break;
    leaf = next; n = n_next; cur = nxp;
  }
  if (fill) flush();
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

// k = 16: msp.hip's window is four k-mers there (2.2 x the step of k = 31 on the same reads, profiles/r05/ksweep_*);
// the radix path takes the batch while its 2^17 leaves of 2^15 keys stay well under 2^16 elements each (packed 16-bit
// counters), i.e. up to ~4e9 bases per add.  Not for CFRK_RUNS_ONLY jobs (the runs exchange is msp.hip's).
constexpr int RX16_IDX = 15, RX16_B2 = 9;
static double rx_pad2(int b2) { return 1.0 + 3.5 * (double)(1u << b2) / (double)RX2_KEYS; }
bool cfrk_radix_prefers(const cfrk_ctx *ctx, int64_t nN) {
  if (cfrk_radix_usable(ctx)) return true;
  if (ctx->g_two || ctx->g_k != 16 || (ctx->g_flags & (CFRK_FORCE_HASH | CFRK_RUNS_ONLY)) || (ctx->dbg_flags & CFRK_DEBUG_NO_RADIX16)) return false;
  const double per_leaf = (double)nN / (double)(1u << (8 + RX16_B2)) * rx_pad2(RX16_B2);
  return per_leaf * 1.5 <= 49152.0;
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
  // at least 2048 leaves (one workgroup each in RX3), at most 2^13 counter words per leaf (k = 16: 2^14)
  v.b1 = 8;
  if (k == 16) v.idx = RX16_IDX;
  else if (2 * k - 11 > RX_IDX_MAX) {
    // k >= 13: 2^14 keys per leaf as packed 16-bit counters (half the leaves, half RX2's fan-out) pay while a leaf
    // stays under 2^16 elements; a leaf above that is counted in two passes over its stream, so a batch whose MEAN
    // leaf comes near it (k = 13 from ~2e8 bases, k = 14 from ~8e8, k = 15 from ~3e9) keeps 2^13 keys per leaf and
    // 32-bit counters instead (measured at configs[1]'s size: profiles/r05/ksweep_*)
    const int b2w = 2 * k - v.b1 - (RX_IDX_MAX + 1);
    const double per_leaf = (double)nN / (double)(1ull << (v.b1 + b2w)) * rx_pad2(b2w);
    v.idx = (per_leaf * 1.5 <= 49152.0) ? RX_IDX_MAX + 1 : RX_IDX_MAX;
  } else v.idx = std::max(0, 2 * k - 11);
  v.b2 = 2 * k - v.b1 - v.idx;
  v.kmask = (uint32_t)((1ull << (2 * k)) - 1ull);
  v.mul = 0x9E3779B1u;
  v.dbg = ctx->dbg_flags;
  v.inv = inv_odd32(v.mul);
  const uint64_t nb1 = 1ull << v.b1, nleaf = 1ull << (v.b1 + v.b2);
  // (a tile of RX2 rounds each of its <= 512 segments up to eight elements, ~3.5 pads each; level-1 regions start on
  //  multiples of 16 elements, leaf streams on multiples of 8)
  const double pad2 = rx_pad2(v.b2);
  const uint64_t cap1 = ((uint64_t)((double)nN / (double)(nb1 * RX_NREG) * 1.3) + 4096 + 15) & ~15ull;
  const uint64_t cap2 = ((uint64_t)((double)nN / (double)nleaf * 1.5 * pad2) + 1024 + 7) & ~7ull;
  // the 8-bit plane also exists when the level's keys fill the 16-bit plane exactly (k = 12): a pad must not look like a key
  v.hi8 = (2 * k - v.b1 >= 16) ? 1u : 0u;
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
      const dim3 g1((unsigned)tiles), b1d(RX1_THREADS);
      if (k == 16) {
        if (canon) hipLaunchKernelGGL((rx1_kernel<true, true>), g1, b1d, 0, ctx->stream, d_data, nN, v, t);
        else hipLaunchKernelGGL((rx1_kernel<false, true>), g1, b1d, 0, ctx->stream, d_data, nN, v, t);
      } else if (canon) hipLaunchKernelGGL((rx1_kernel<true, false>), g1, b1d, 0, ctx->stream, d_data, nN, v, t);
      else hipLaunchKernelGGL((rx1_kernel<false, false>), g1, b1d, 0, ctx->stream, d_data, nN, v, t);
      HIP_TRY(ctx, hipGetLastError());
    }
    const int64_t g2 = bins_per_group * RX_NXG * RX_NREG * tiles_per_sub;
    if (g2 > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, sizeof(uint64_t), ctx->stream));
    if (v.exact1) hipLaunchKernelGGL((rx2_kernel<false>), dim3((unsigned)g2), dim3(RX2_THREADS), 0, ctx->stream, (int)tiles_per_sub, v, t);
    else hipLaunchKernelGGL((rx2_kernel<true>), dim3((unsigned)g2), dim3(RX2_THREADS), 0, ctx->stream, (int)tiles_per_sub, v, t);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t st[ST_NWORDS];
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st[ST_CWRAP]) return CFRK_INTERNAL_FLOOD;      // (nothing of this add has been counted yet)
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
  // persistent: four workgroups (32 KB of counters each) per CU walk the leaves
  if (k == 16) hipLaunchKernelGGL((rx3_kernel<14, 512>), dim3((unsigned)std::min<uint64_t>(nleaf, (uint64_t)std::max(1, ctx->num_cus) * 2)), dim3(512), 0, ctx->stream, v, (uint32_t)nleaf);
  else hipLaunchKernelGGL((rx3_kernel<RX_IDX_MAX, 256>), dim3((unsigned)std::min<uint64_t>(nleaf, (uint64_t)std::max(1, ctx->num_cus) * 4)), dim3(256), 0, ctx->stream, v, (uint32_t)nleaf);
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
