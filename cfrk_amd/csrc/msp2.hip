// msp2.hip -- minimizer-partitioned global k-mer counting for 33 <= k <= 64 (two-word keys).
//
// Same pipeline as msp.hip (which covers 16 <= k <= 32) with three differences:
//   * the minimizer of a k-mer is taken over the CENTRAL 18 of its k-m+1 canonical m-mers
//     (m = 13 for even k, 14 for odd k, so that the margin c on both sides is equal: the choice is
//     then strand-symmetric).  A run (same minimizer occurrence) still holds at most 18 k-mers,
//     so the minimizer front end of msp.hip (msp_dev.h: msp_minimizers<18>) is reused as is --
//     a lane owns the 32 WINDOW positions y of its chunk, i.e. the k-mers starting at y - c;
//   * records are 32 bytes: 96 bases (6 dwords; a run spans at most 18 + 63 bases) + header;
//   * the leaf kernel keeps 16-byte keys in LDS (claimed through the count word: 0 empty,
//     LOCK while the claiming lane writes the key); complete runs go through a record table
//     first, as in msp.hip; results go to a two-word list.
// Anything that does not fit is counted in the two-word HBM table (table_add2).
//
// Semantics: the guarded ComputeFreq of /root/reference/src/kmer_kernel.cu:52-70 summed over
// reads, like global_hash.hip.
#include "msp.h"
#include "table.h"
#include "msp_dev.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {
#include "msp_runs.h"

// Window (m-mers per k-mer the minimizer is taken over = longest run): all but a margin of c = 1 on
// both sides of the k-mer's k - m + 1 m-mers (m = 13 / 14 for even / odd k, so the count is even),
// i.e. 18 at k = 33, 20 at k = 34 ... and 30 from k = 44 on: a run of 30 k-mers is k + 29 <= 93 bases
// and fits the 96-base record, position tags are 5 bits, and a lane's change mask covers 32 + W - 1 <= 64
// positions with one to spare for the right-end flag.  Longer runs = fewer records (2 / (W + 1) per
// position).
constexpr int W2_LONG = 30;
__host__ __device__ constexpr int msp2_window(int k) {
  const int nm = k - ((k & 1) ? 14 : 13) + 1;
  return nm - 2 > W2_LONG ? W2_LONG : nm - 2;
}
constexpr int Q1_THREADS = 512, Q1_WAVES = Q1_THREADS / 64;
constexpr int Q1_OWN = 60;                 // owner lanes 1..60; lane 0 and lanes 61..63 are halo lanes
constexpr int Q1_RCAP_BASE = 1536;         // 32-byte records staged in LDS per workgroup (48 KB: three workgroups per CU)
constexpr int Q2_THREADS = 1024, Q2_PER = 2, Q2_TILE = Q2_THREADS * Q2_PER;
constexpr int Q2_GROUP = 4;                      // consecutive tiles per workgroup (next tile prefetched)
constexpr int Q3_THREADS = 1024;
constexpr int T2_LOG = 12, T2 = 1 << T2_LOG;   // LDS slots per leaf (16-byte keys)
constexpr uint32_t T2_LOCK = 0xFFFFFFFFu;

struct Rec2 { uint4 a, b; };               // a = bases dwords 0..3, b = {dwords 4, 5, 0, header}

struct View2 {
  Rec2 *rec1; uint32_t *cnt1; uint64_t cap1;            // B1 x NXG regions
  Rec2 *rec2; uint32_t *cnt2; uint64_t cap2c, cap2t;    // per leaf: complete stream, 3 truncated classes
  uint64_t *out_lo, *out_hi; uint32_t *out_cnt; uint64_t out_cap;
  uint64_t *leaf_off; uint32_t *leaf_n;                  // where each leaf's entries sit in the result list (shared leaves: one
                                                         // segment per sub-value, entry (leaf << sub_bits) | sub-value)
  // exact layout after leaf streams overflowed the fixed stride (see msp.hip): stream (leaf, class)
  // starts at record lbase[NCLS * leaf + class] and holds exactly lcap[...] records
  const uint64_t *lbase; const uint32_t *lcap; uint32_t exact;
  Rec2 *ovf; uint32_t ovf_cap;                           // parking for a few overflowing records
  // the same for the level-1 regions (msp.hip): region reg starts at record rbase[reg], holds rcap[reg]
  const uint64_t *rbase; const uint32_t *rcap; uint32_t exact1;
  Rec2 *ovf1; uint32_t ovf1_cap;
  uint32_t count_only;                                   // second-level kernel: count the streams' records, write nothing
  uint32_t dbg;                                          // cfrk_debug_set_flags
  uint32_t sel_mask, sel_val, sel_bits;                  // leaf subset of this pass (msp.h: MspView)
  uint32_t sub_bits;                                     // records carry so many more minimizer-hash bits in b.z (0: none)
  uint32_t hbits;                                        // a workgroup of a shared leaf counts 2^hbits sub-values, one after the other
  uint64_t *stats;
};

// Truncated runs are split by length class as well, so that in the leaf kernel the lanes of a
// wave expand records of similar length.
constexpr int NCLS = 4;                                   // 0..2 truncated (n<=4, <=10, >10), 3 complete
constexpr int NSUB = NCLS * B2;
__device__ __forceinline__ uint32_t cls_of(uint32_t w) {
  const uint32_t n = (w & 63u) + 1u;
  return ((w & 192u) == 192u) ? 3u : (n <= 4u ? 0u : (n <= 10u ? 1u : 2u));
}
__device__ __forceinline__ uint32_t sub_of(uint32_t w) { return (((w >> 8) & (B2 - 1)) << 2) | cls_of(w); }

typedef unsigned __int128 u128;

// k-mers of a record, one by one: fwd / rc rolling over the 192-bit base string
struct Roll2 {
  u128 fwd, rc, kmask;
  uint64_t t0, t1, t2;     // bases not yet consumed, first one in the top bits of t0
  int rcsh;
  // j0: first k-mer to produce (the record's base string is shifted left by j0 bases first)
  __device__ __forceinline__ void init(const Rec2 &r, int k, int j0 = 0) {
    uint64_t s0 = ((uint64_t)r.a.x << 32) | r.a.y, s1 = ((uint64_t)r.a.z << 32) | r.a.w;
    uint64_t s2 = ((uint64_t)r.b.x << 32) | r.b.y;
    if (j0) {                                        // 1 <= j0 <= 31
      const int sh = 2 * j0;
      s0 = (s0 << sh) | (s1 >> (64 - sh));
      s1 = (s1 << sh) | (s2 >> (64 - sh));
      s2 <<= sh;
    }
    kmask = (k == 64) ? ~(u128)0 : ((((u128)1) << (2 * k)) - 1);
    rcsh = 2 * k - 2;
    const u128 top = ((u128)s0 << 64) | s1;            // bases 0..63
    fwd = top >> (128 - 2 * k);                         // k <= 64
    {
      // reverse complement of the 2k-bit value: reverse all 128 bits, swap the bits of every
      // pair back, complement, drop the 128-2k low bits
      uint64_t rh = __brevll((uint64_t)fwd), rl = __brevll((uint64_t)(fwd >> 64));
      rh = ((rh >> 1) & 0x5555555555555555ull) | ((rh & 0x5555555555555555ull) << 1);
      rl = ((rl >> 1) & 0x5555555555555555ull) | ((rl & 0x5555555555555555ull) << 1);
      rc = ((((u128)~rh) << 64) | (u128)~rl) >> (128 - 2 * k);
    }
    // remaining bases start at base k (33..64)
    const int sh = 2 * k - 64;                          // 2..64 bits into s1
    if (sh == 64) { t0 = s2; t1 = 0; }
    else { t0 = (s1 << sh) | (s2 >> (64 - sh)); t1 = s2 << sh; }
    t2 = 0;
  }
  __device__ __forceinline__ void next() {
    const unsigned nb = (unsigned)(t0 >> 62);
    t0 = (t0 << 2) | (t1 >> 62);
    t1 <<= 2;
    fwd = ((fwd << 2) | (u128)nb) & kmask;
    rc = (rc >> 2) | ((u128)(3u - nb) << rcsh);
  }
};

// (record and table view by value: references pin the caller's copies to scratch memory)
__device__ __noinline__ void spill_record2(uint4 ra, uint4 rb, int k, bool canon, TableView t, uint32_t weight = 1u) {
  const Rec2 rec = {ra, rb};
  t.stats[ST_SPILLED] = 1;
  dev_count_event(&t.stats[ST_AUX0]);
  const int nk = (int)(rec.b.w & 63u) + 1;
  Roll2 r;
  r.init(rec, k);
  for (int j = 0; j < nk; ++j) {
    const u128 key = (canon && r.rc < r.fwd) ? r.rc : r.fwd;
    table_add2(t, (uint64_t)key, (uint64_t)(key >> 64), weight);
    r.next();
  }
}

// ---------------------------------------------------------------------------------------- Q1
// level-1 region / cursor of (bin, sub-region): sub-region major, so that a workgroup's 256
// reservations (memory-side atomics, one request per touched 64 bytes) are 16 requests (msp.hip: l1_reg)
__host__ __device__ __forceinline__ uint32_t q1_reg(uint32_t bin, uint32_t xg) { return xg * (uint32_t)B1 + bin; }
__device__ __forceinline__ uint64_t q1_cap(const View2 &v, uint32_t reg) { return v.exact1 ? (uint64_t)v.rcap[reg] : v.cap1; }
__device__ __forceinline__ uint64_t q1_at(const View2 &v, uint32_t reg) { return v.exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1; }
// a record that found its level-1 region full: a few are parked (and counted through the HBM table
// afterwards), beyond that ST_L1OVF is raised -- the cursors keep counting, and the host lays the
// level out again with the exact sizes and reruns the kernel (msp.hip: l1_put)
// (cold, and a leaf function: a whole view, or a call from here, would go through scratch memory)
__device__ __noinline__ void q1_park(uint64_t *stats, Rec2 *ovf1, uint32_t ovf1_cap, uint4 ra, uint4 rb) {
  if (*(volatile uint64_t *)&stats[ST_L1OVF] != 0) return;
  const unsigned long long o = atomicAdd((unsigned long long *)&stats[ST_OVFN1], 1ull);
  if (o < ovf1_cap) { ovf1[o].a = ra; ovf1[o].b = rb; }
  else stats[ST_L1OVF] = 1;
}
__device__ __forceinline__ void q1_overflow(const View2 &v, uint4 ra, uint4 rb, int k, bool canon, const TableView &t) {
  // (with the exact layout this cannot happen; if it does the flag makes the host try again and give up)
  q1_park(v.stats, v.ovf1, v.exact1 ? 0u : v.ovf1_cap, ra, rb);
}
// Emission is wave-balanced as in msp.hip's msp_p1b_kernel: a lane lists its run starts, the wave
// stages its base string, run terminators, validity and the leaf id of every position in LDS, and
// lane i builds the wave's i-th record; records wait in registers for the bin offsets and go to
// LDS in bin order (the staging bytes are reused).
constexpr int Q1_TR = 3;                           // balanced trips held in registers (192 runs per wave; ~131 expected at k = 63, ~170 at k = 33)
constexpr int Q1_STAGE_BASE = 4096 + 3 * 512 + Q1_TR * 128;   // staging bytes per wave

struct Stage2 {
  const uint16_t *leaf; const uint32_t *str; const uint64_t *E, *Wv;
  const uint8_t *sub;            // SUB_BITS more bits of the minimizer hash per position, or nullptr
};
// record of the run that starts at window position d & 31 of lane d >> 5
__device__ __forceinline__ Rec2 q1_build(const Stage2 &st, uint32_t d, int k, int c, int wmax) {
  const uint32_t L = d >> 5, a = d & 31u;
  const uint64_t Es = st.E[L], Ws = st.Wv[L];
  const uint32_t leaf = st.leaf[d];
  // 192 bits of the wave's base string from base d - c: dwords idx0 .. idx0+6, funnel-shifted
  const uint32_t P = 2u * (d - (uint32_t)c) - 2u, idx0 = P >> 5, sh = 30u - (P & 31u);
  uint32_t D[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) D[i] = st.str[idx0 + i];
  const uint64_t rest = Es << (a + 1);
  const int n = min(__clzll(rest) + 1, wmax);
  // which ends of the run are closed (a minimizer change between valid k-mers): bit 6 left, bit 7
  // right; both = a complete run (see msp.hip)
  const uint32_t complete = (((uint32_t)(Ws >> (63 - a)) & 1u) << 6) | (((uint32_t)(Ws >> (62 - a - n)) & 1u) << 7);
  const int nb = n + k - 1;                                     // bases that belong to the run
  uint32_t T[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    uint32_t x = __builtin_amdgcn_alignbit(D[i], D[i + 1], sh);
    const int keep = 2 * nb - 32 * i;                           // bits of this dword inside the run
    x = (keep >= 32) ? x : ((keep <= 0) ? 0u : (x & (~0u << (32 - keep))));
    T[i] = x;
  }
  Rec2 rec;
  rec.a = make_uint4(T[0], T[1], T[2], T[3]);
  rec.b = make_uint4(T[4], T[5], st.sub ? (uint32_t)st.sub[d] : 0u, (leaf << 8) | complete | (uint32_t)(n - 1));
  return rec;
}

// SUB: a job with far more distinct k-mers than the leaf tables hold (capacity hint > ~2.7e8) also
// stages SUB_BITS bits of every position's minimizer hash that the leaf id does not use (bits 24..27
// and 7 of the packed minimum: hash bits, equal for every occurrence of a k-mer) and writes them to
// the record's spare word: the leaf kernel then splits an overfull leaf by RECORD, not by key, so
// that every record is expanded once.  2 KB more staging per wave: two workgroups per CU.
constexpr int SUB_BITS = 5;
template <int W2, bool SUB>
__global__ __launch_bounds__(Q1_THREADS, (SUB ? 4 : 6)) void msp2_p1_kernel(const int8_t *__restrict__ data, int64_t nN,
                                                             int k, int m, int c, int canon, int64_t tile0,
                                                             View2 v, TableView t) {
  constexpr int NH = 32 + W2 - 1;
  constexpr int Q1_STAGE = Q1_STAGE_BASE + (SUB ? 2048 : 0);
  constexpr int Q1_RCAP = SUB ? 2048 : Q1_RCAP_BASE;
  static_assert(Q1_WAVES * Q1_STAGE <= Q1_RCAP * 32, "staging fits the record arena");
  __shared__ Rec2 arena[Q1_RCAP];                  // per-wave staging, later the bin-sorted records
  __shared__ uint32_t hist[B1], loff[B1], gbase[B1];
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t nrec_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint8_t *const stage = reinterpret_cast<uint8_t *>(arena) + wave * Q1_STAGE;
  uint16_t *const s_leaf = reinterpret_cast<uint16_t *>(stage);               // [64 lanes][32 positions]
  uint32_t *const s_str = reinterpret_cast<uint32_t *>(stage + 4096);         // 128 dwords of bases
  uint64_t *const s_E = reinterpret_cast<uint64_t *>(stage + 4096 + 512);     // run terminators
  uint64_t *const s_W = reinterpret_cast<uint64_t *>(stage + 4096 + 1024);    // validity of position p-1
  uint16_t *const s_dsc = reinterpret_cast<uint16_t *>(stage + 4096 + 1536);  // (lane << 5) | position
  uint8_t *const s_sub = stage + Q1_STAGE_BASE;                                // [64 lanes][32 positions] (SUB only)

  if (tid < B1) hist[tid] = 0;
  lds_barrier();

  // ---- A: own chunk and its neighbours (by shuffle) ----
  const int64_t wave_g = (tile0 + blockIdx.x) * Q1_WAVES + wave;
  const int64_t chunk = wave_g * Q1_OWN + lane - 1;
  const int64_t off = chunk * 32;
  uint32_t b0 = 0, b1 = 0, bd1 = 0xFFFFFFFFu;
  if (chunk >= 0) dev_load_chunk32(data, off, nN, b0, b1, bd1);
  const uint32_t bd0 = dev_lane_prev(bd1), bd2 = dev_lane_next(bd1);
  const uint32_t bd3 = dev_lane_next(bd2), bd4 = dev_lane_next(bd3);
  const uint64_t hi = ((uint64_t)b0 << 32) | b1;
  const uint64_t mid = ((uint64_t)dev_lane_next(b0) << 32) | dev_lane_next(b1);

  // validity of the k-mer that belongs to window position y: bases y-c .. y-c+k-1.  I = invalid
  // mask with origin at base -32; smear over the k following bases, then read at offset 32-c.
  uint64_t I0 = ((uint64_t)bd0 << 32) | bd1, I1 = ((uint64_t)bd2 << 32) | bd3, I2 = (uint64_t)bd4 << 32;
  {
    int w = 1;
#pragma unroll
    for (int st = 0; st < 6; ++st) {
      if (2 * w <= k) {
        I0 |= (I0 << w) | (I1 >> (64 - w));
        I1 |= (I1 << w) | (I2 >> (64 - w));
        I2 |= I2 << w;
        w *= 2;
      }
    }
    if (k > w) {
      const int s = k - w;
      I0 |= (I0 << s) | (I1 >> (64 - s));
      I1 |= (I1 << s) | (I2 >> (64 - s));
    }
  }
  const int o = 32 - c;                                        // 15..31
  const uint64_t Vx = ~((I0 << o) | (I1 >> (64 - o)));
  const uint32_t V = (uint32_t)(Vx >> 32);
  const uint32_t prevV = dev_lane_prev(V) & 1u;

  uint32_t H[NH];
  const uint64_t Cx = msp_minimizers<W2>(hi, mid, chunk, m, H);
  const uint64_t E = Cx | ~Vx | (1ull << (63 - NH));
  const uint32_t Vprev = (V >> 1) | (prevV << 31);
  uint32_t S = V & ((uint32_t)(Cx >> 32) | ~Vprev);
  const bool owner = lane >= 1 && lane <= Q1_OWN && off < nN + 32;   // a k-mer may START in the previous chunk
  if (!owner) S = 0;

  // ---- B1: stage what records are made of; list the run starts ----
  {
    const LeafPack LP = leaf_pack(H);
    uint4 *lp = reinterpret_cast<uint4 *>(s_leaf + lane * 32);
    lp[0] = make_uint4(LP.w[0], LP.w[1], LP.w[2], LP.w[3]);
    lp[1] = make_uint4(LP.w[4], LP.w[5], LP.w[6], LP.w[7]);
    lp[2] = make_uint4(LP.w[8], LP.w[9], LP.w[10], LP.w[11]);
    lp[3] = make_uint4(LP.w[12], LP.w[13], LP.w[14], LP.w[15]);
    reinterpret_cast<uint2 *>(s_str)[lane] = make_uint2(b0, b1);
    s_E[lane] = E;
    s_W[lane] = (Vx >> 1) | ((uint64_t)prevV << 63);
    if (SUB) {
      uint32_t sw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          const uint32_t h = H[4 * i + j];
          x = (x << 8) | (((h >> 24) & 15u) << 1) | ((h >> 7) & 1u);
        }
        sw[i] = x;
      }
      uint4 *sp = reinterpret_cast<uint4 *>(s_sub + lane * 32);
      sp[0] = make_uint4(sw[0], sw[1], sw[2], sw[3]);
      sp[1] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
    }
  }
  const Stage2 st = {s_leaf, s_str, s_E, s_W, SUB ? s_sub : nullptr};
  uint32_t cnt_w;
  uint32_t widx, S2 = 0;                           // S2: run starts beyond the balanced phase's capacity
  const uint32_t wcap = (v.dbg & CFRK_DEBUG_SMALL_WAVE_CAP) ? 64u : (uint32_t)(Q1_TR * 64);
  {
    const uint32_t mine = (uint32_t)__popc(S);
    const uint32_t incl = dev_wave_scan_incl(mine);
    cnt_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    widx = incl - mine;
    const uint32_t tag = (uint32_t)lane << 5;
    while (S) {
      const int a = __clz(S);
      const uint32_t bit = 0x80000000u >> a;
      S &= ~bit;
      if (widx < wcap) s_dsc[widx] = (uint16_t)(tag | (uint32_t)a);
      else S2 |= bit;
      ++widx;
    }
  }
  lds_wave_sync();

  // more runs in this wave than the balanced phase holds (pathological input): append directly
  while (S2) {
    const int a = __clz(S2);
    S2 &= ~(0x80000000u >> a);
    const Rec2 rec = q1_build(st, ((uint32_t)lane << 5) | (uint32_t)a, k, c, W2);
    if (((rec.b.w >> 8) & v.sel_mask) != v.sel_val) continue;    // not a leaf of this pass
    const uint32_t reg = q1_reg(rec.b.w >> 16, blockIdx.x & (NXG - 1));
    const uint32_t dst = atomicAdd(&v.cnt1[reg], 1u);
    if (dst < q1_cap(v, reg)) v.rec1[q1_at(v, reg) + dst] = rec;
    else q1_overflow(v, rec.a, rec.b, k, canon != 0, t);
  }

  // ---- B2: lane i builds the wave's i-th record ----
  Rec2 rc[Q1_TR];
  uint32_t rk[Q1_TR];                              // rank inside the record's bin; ~0: no record
  cnt_w = min(cnt_w, wcap);
#pragma unroll
  for (int tr = 0; tr < Q1_TR; ++tr) {
    rk[tr] = 0xFFFFFFFFu;
    rc[tr].a = make_uint4(0, 0, 0, 0);
    rc[tr].b = make_uint4(0, 0, 0, 0);
    const uint32_t i = (uint32_t)(tr * 64 + lane);
    if (i < cnt_w) {
      rc[tr] = q1_build(st, s_dsc[i], k, c, W2);
      if (((rc[tr].b.w >> 8) & v.sel_mask) == v.sel_val)     // (all leaves, unless the batch takes several passes)
        rk[tr] = atomicAdd(&hist[rc[tr].b.w >> 16], 1u);
    }
    __builtin_amdgcn_sched_barrier(0);             // one trip at a time: interleaved trips spill registers
  }
  lds_barrier();

  // ---- C: one global reservation per non-empty bin; bin offsets ----
  uint32_t my_base = 0;
  if (tid < B1) {
    const uint32_t cnum = hist[tid];
    if (cnum) my_base = atomicAdd(&v.cnt1[q1_reg(tid, blockIdx.x & (NXG - 1))], cnum);
  }
  block_scan<B1, true>(hist, loff, wtot);                // ends with a barrier: the staging area is dead
  if (tid == B1 - 1) nrec_s = loff[tid] + hist[tid];
#pragma unroll
  for (int tr = 0; tr < Q1_TR; ++tr) {
    if (rk[tr] != 0xFFFFFFFFu) {
      const uint32_t pos = loff[rc[tr].b.w >> 16] + rk[tr];
      if (pos < (uint32_t)Q1_RCAP) arena[pos] = rc[tr];
    }
  }
  if (tid < B1) gbase[tid] = my_base;
  lds_barrier();

  // ---- D: copy out in bin order ----
  auto put = [&](uint32_t b, uint32_t dst, const Rec2 &rec) {
    const uint32_t reg = q1_reg(b, blockIdx.x & (NXG - 1));
    if (dst < q1_cap(v, reg)) v.rec1[q1_at(v, reg) + dst] = rec;
    else q1_overflow(v, rec.a, rec.b, k, canon != 0, t);
  };
  if (nrec_s > (uint32_t)Q1_RCAP) {                // records beyond the LDS arena go to their reserved places
#pragma unroll
    for (int tr = 0; tr < Q1_TR; ++tr) {
      if (rk[tr] != 0xFFFFFFFFu) {
        const uint32_t b = rc[tr].b.w >> 16;
        if (loff[b] + rk[tr] >= (uint32_t)Q1_RCAP) put(b, gbase[b] + rk[tr], rc[tr]);
      }
    }
  }
  // consecutive lanes write consecutive 16-byte HALVES of consecutive records: every store
  // instruction fills whole 32-byte sectors (a record written as two strided halves leaves every
  // sector half-written by the first store: the L2 then fetches the line to merge, and writes it twice)
  const uint32_t nrec = min(nrec_s, (uint32_t)Q1_RCAP);
  const uint4 *arena4 = reinterpret_cast<const uint4 *>(arena);
  uint4 *out4 = reinterpret_cast<uint4 *>(v.rec1);
  for (uint32_t q = tid; q < 2 * nrec; q += Q1_THREADS) {
    const uint32_t p = q >> 1;
    const uint32_t b = arena[p].b.w >> 16;
    const uint32_t dst = gbase[b] + (p - loff[b]);
    const uint32_t reg = q1_reg(b, blockIdx.x & (NXG - 1));
    if (dst < q1_cap(v, reg)) out4[(q1_at(v, reg) + dst) * 2 + (q & 1u)] = arena4[q];
    else if (!(q & 1u)) q1_overflow(v, arena[p].a, arena[p].b, k, canon != 0, t);
  }
}

// ---------------------------------------------------------------------------------------- Q2
// msp.hip's P2 on 32-byte records (same header word, same sub-bins, same XCD-affine order)
__global__ __launch_bounds__(Q2_THREADS) void msp2_p2_kernel(int groups_per_bin, int k, int canon, View2 v,
                                                             TableView t) {
  __shared__ Rec2 sorted[Q2_TILE];
  __shared__ uint32_t hist[NSUB], loff[NSUB], gbase[NSUB];
  __shared__ uint32_t wtot[Q2_THREADS / 64];
  __shared__ uint32_t rpre[NXG + 1];             // exclusive prefix of the bin's sub-region sizes
  __shared__ unsigned long long rfirst[NXG];     // first record of every sub-region (looked up once: msp.hip's P2)
  static_assert(NSUB == Q2_THREADS, "one sub-bin per thread");
  static_assert(NXG <= 64, "one wave scans the sub-region sizes");
  const int tid = threadIdx.x;
  const uint32_t xg = blockIdx.x & (NXCD - 1), seq = blockIdx.x / NXCD;
  const uint32_t b1 = xg + NXCD * (seq / (uint32_t)groups_per_bin);
  const uint32_t grp = seq % (uint32_t)groups_per_bin;
  // the bin's sub-regions are read as one stream (as in msp.hip's P2)
  if (tid < 64) {
    const uint32_t c = (tid < NXG) ? (uint32_t)min((uint64_t)v.cnt1[q1_reg(b1, tid)], q1_cap(v, q1_reg(b1, tid))) : 0u;
    const uint32_t incl = dev_wave_scan_incl(c);
    if (tid < NXG) { rpre[tid] = incl - c; rfirst[tid] = q1_at(v, q1_reg(b1, tid)); }
    if (tid == 63) rpre[NXG] = incl;
  }
  lds_barrier();
  const uint64_t n = rpre[NXG];
  // Q2_GROUP consecutive tiles per workgroup, the next tile's records requested before the
  // current tile is sorted and written (as in msp.hip's P2)
  const uint64_t g0r = (uint64_t)grp * Q2_GROUP * Q2_TILE;
  if (g0r >= n) return;
  auto fetch = [&](uint64_t idx) {
    uint32_t lo = 0, hi = NXG;                 // invariant: rpre[lo] <= idx < rpre[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (rpre[mid] <= idx) lo = mid; else hi = mid;
    }
    return v.rec1[rfirst[lo] + (idx - rpre[lo])];
  };
  const Rec2 zrec = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  Rec2 nx[Q2_PER];
#pragma unroll
  for (int i = 0; i < Q2_PER; ++i) {
    const uint64_t idx = g0r + (uint64_t)i * Q2_THREADS + tid;
    nx[i] = zrec;
    if (idx < n) nx[i] = fetch(idx);
  }
  for (int tt = 0; tt < Q2_GROUP; ++tt) {
    const uint64_t r0 = g0r + (uint64_t)tt * Q2_TILE;
    if (r0 >= n) break;
    const uint32_t nt = (uint32_t)min((uint64_t)Q2_TILE, n - r0);
    Rec2 r[Q2_PER];
#pragma unroll
    for (int i = 0; i < Q2_PER; ++i) r[i] = nx[i];
    if (tt + 1 < Q2_GROUP) {
#pragma unroll
      for (int i = 0; i < Q2_PER; ++i) {
        const uint64_t idx = r0 + Q2_TILE + (uint64_t)i * Q2_THREADS + tid;
        nx[i] = zrec;
        if (idx < n) nx[i] = fetch(idx);
      }
    }
    hist[tid] = 0;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < Q2_PER; ++i) {
      const uint32_t idx = (uint32_t)i * Q2_THREADS + tid;
      if (idx < nt) atomicAdd(&hist[sub_of(r[i].b.w)], 1u);
    }
    lds_barrier();
    uint32_t g0 = 0;
    {
      const int lane = tid & 63, wave = tid >> 6;
      const uint32_t x0 = hist[tid];
      if (x0) g0 = atomicAdd(&v.cnt2[b1 * NSUB + tid], x0);
      // (sizing pass of a batch that would not fit otherwise: the streams are then laid out with
      //  exactly the room they need and the kernel runs again)
      if (v.count_only) { lds_barrier(); continue; }
      const uint32_t incl = dev_wave_scan_incl(x0);
      if (lane == 63) wtot[wave] = incl;
      lds_barrier();
      uint32_t bs = 0;
      for (int w = 0; w < wave; ++w) bs += wtot[w];
      loff[tid] = bs + incl - x0;
      hist[tid] = 0;
      lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < Q2_PER; ++i) {
      const uint32_t idx = (uint32_t)i * Q2_THREADS + tid;
      if (idx < nt) {
        const uint32_t sb = sub_of(r[i].b.w);
        sorted[loff[sb] + atomicAdd(&hist[sb], 1u)] = r[i];
      }
    }
    gbase[tid] = g0;
    lds_barrier();
    for (uint32_t p = tid; p < nt; p += Q2_THREADS) {
      const Rec2 rec = sorted[p];
      const uint32_t sb = sub_of(rec.b.w);
      const uint32_t dst = gbase[sb] + (p - loff[sb]);
      const uint64_t leaf = ((uint64_t)b1 * NSUB + sb) >> 2;
      const uint32_t cls = sb & 3u;
      uint64_t cap = (cls == 3u) ? v.cap2c : v.cap2t;
      uint64_t at = (leaf >> v.sel_bits) * (v.cap2c + 3 * v.cap2t) + ((cls == 3u) ? 0 : v.cap2c + cls * v.cap2t);
      if (v.exact) { cap = v.lcap[b1 * NSUB + sb]; at = v.lbase[b1 * NSUB + sb]; }
      if (dst < cap) v.rec2[at + dst] = rec;
      else if (dst >= 0x80000000u) v.stats[ST_CWRAP] = 1;                 // a flood on its way to wrapping the 32-bit cursor (msp.hip)
      else if (v.exact) spill_record2(rec.a, rec.b, k, canon != 0, t);    // cannot happen: cap is the exact count
      else if (*(volatile uint64_t *)&v.stats[ST_L2OVF] == 0) {
        // too small by a little: park the record; by a lot: the host redoes Q2 with exact sizes
        const unsigned long long o = atomicAdd((unsigned long long *)&v.stats[ST_OVFN], 1ull);
        if (o < v.ovf_cap) v.ovf[o] = rec;
        else v.stats[ST_L2OVF] = 1;
      }
    }
    lds_barrier();                                       // the LDS buffers are reused by the next tile
  }
}

// ---------------------------------------------------------------------------------------- Q3
// (T2L: log2 of the table's slots -- T2_LOG, or T2_LOG_SMALL in the small-leaf instantiation of the leaf kernel)
template <int T2L = T2_LOG>
__device__ __forceinline__ uint32_t t2_slot(uint64_t lo, uint64_t hi) {
  const uint32_t x = (uint32_t)lo ^ (uint32_t)(lo >> 32) ^ ((uint32_t)hi * 0x85EBCA77u) ^ (uint32_t)(hi >> 32);
  return (x * 0x9E3779B1u) >> (32 - T2L);
}

// One probe step for one key per lane (see msp.hip's kt_try for why it is written this way):
// cnts[h] is the slot state -- 0 empty, T2_LOCK while the claiming lane writes the key, else the
// count.  The lane's state is its slot h with T2_DONE or-ed in once the key is counted.
constexpr uint32_t T2_DONE = 0x80000000u;
constexpr int T2_TRIPS = 96;
constexpr int T2_TRIPS_SPLIT = 24;
// SAT: the count saturates at CFRK_COUNT_MAX (the merge of lists whose counts nothing bounds; the count word is the
// slot state, so a wrapped value must never show: compare-and-swap).  The leaf kernel's own counts cannot overflow
// (msp.hip: HUGE_LEAF).
template <bool SAT = false, int T2L = T2_LOG>
__device__ __forceinline__ void t2_step(ulonglong2 *keys, uint32_t *cnts, uint64_t lo, uint64_t hi,
                                        uint32_t add, uint32_t &h) {
  constexpr uint32_t T2 = 1u << T2L;
  const bool p = (int32_t)h >= 0;
  const uint32_t hh = h & (T2 - 1);
  const uint32_t cst = cnts[hh];
  // the key may only be read after the state: a published count vouches for the key words
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const ulonglong2 kk = keys[hh];
  const bool empty = cst == 0u, locked = cst == T2_LOCK;
  const bool match = !empty && !locked && kk.x == lo && kk.y == hi;
  uint32_t won = 0u;
  if (p && empty) {
    if (atomicCAS(&cnts[hh], 0u, T2_LOCK) == 0u) {
      keys[hh] = make_ulonglong2(lo, hi);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      atomicExch(&cnts[hh], add);
      won = 1u;
    }
  }
  if (SAT) {
    if (p && match) {
      uint32_t cur = cst;
      for (;;) {
        const uint32_t want = (cur > CFRK_COUNT_MAX - add) ? CFRK_COUNT_MAX : cur + add;
        const uint32_t old = atomicCAS(&cnts[hh], cur, want);
        if (old == cur) break;
        cur = old;
      }
    }
  } else if (p && match) atomicAdd(&cnts[hh], add);
  // an empty slot lost to another lane, or a locked one, is read again
  const bool stay = match || empty || locked;
  const uint32_t nh = stay ? hh : ((hh + 1) & (T2 - 1));
  h = (p && !match && won == 0u) ? nh : (h | T2_DONE);
}

// key subsets as in msp.hip: selector bits from the slot hash's unused bits
// (rmask / rval: a shared leaf's workgroup counts the records of one of its sub-values at a time --
//  the extra minimizer-hash bits in b.z; 0 / 0 elsewhere)
struct KeySubset2 { uint32_t mask, val, rmask, rval; };
__device__ __forceinline__ bool in_subset2(uint64_t lo, uint64_t hi, KeySubset2 ss) {
  const uint32_t x = (uint32_t)lo ^ (uint32_t)(lo >> 32) ^ ((uint32_t)hi * 0x85EBCA77u) ^ (uint32_t)(hi >> 32);
  return (((x * 0x9E3779B1u) >> 6) & ss.mask) == ss.val;
}

// expand one record per lane (valid lanes), every k-mer counted `add` times; every lane of the
// wave must call.  Only keys of subset `ss`; a key without room goes to the HBM table when
// ovf == nullptr, else raises the LDS flag *ovf (the caller then redoes the subset in halves).
template <bool CANON, int T2L = T2_LOG>
__device__ __forceinline__ void count_record2(ulonglong2 *keys, uint32_t *cnts, const Rec2 &rec, uint32_t add,
                                              bool valid, int k, const TableView &t,
                                              KeySubset2 ss = KeySubset2{0u, 0u, 0u, 0u}, uint32_t *ovf = nullptr,
                                              int part = 0, int parts = 1,
                                              const uint8_t *tb = nullptr, uint32_t tb_n = 0u) {
  // tb[0 .. tb_n): lengths (in k-mers) of the truncated runs that are prefixes of this record: k-mer J
  // of the record is counted once more for every one of them that is longer than J (msp.hip)
  if (ovf && *(volatile uint32_t *)ovf) return;
  // a record may be shared by `parts` lanes, each expanding a contiguous share of its k-mers
  const int nall = (valid && (rec.b.z & ss.rmask) == ss.rval) ? (int)(rec.b.w & 63u) + 1 : 0;
  // (parts <= 8, nall <= 64: the quotient by multiplication with ceil(2^16 / parts) is exact)
  const int per = (parts == 1) ? nall : (int)(((uint32_t)(nall + parts - 1) * ((65536u + (uint32_t)parts - 1u) / (uint32_t)parts)) >> 16);
  const int j0 = part * per;
  const int nk = min(nall, j0 + per);
  // The lists are walked ONCE when every lane of the wave has at most 8 k-mers to expand and at most
  // 127 noted runs (the usual case: 4 lanes share a run of <= 30 k-mers): a lane counts its list's
  // entries by d = clamp(length - j0, 0, 8) in nine 7-bit fields of one 64-bit register; k-mer j0 + q is
  // counted once more for every entry with d > q, i.e. the fields' suffix sums, kept as eight bytes.
  // (One pass over the list per K-MER was a third of the expansion's instructions at k = 63.)
  unsigned long long exq = 0ull;
  const bool tb_fast = tb && !__ballot(valid && (nk - j0 > 8 || tb_n > 127u));
  if (tb_fast) exq = noted_counts8(tb, tb_n, j0);
  Roll2 roll;
  roll.init(rec, k, j0);
  for (int j = j0; __ballot(j < nk); ++j) {
    uint32_t addj = add;
    if (tb_fast) {
      addj += (uint32_t)(exq >> (8 * ((j - j0) & 7))) & 255u;
    } else if (tb) {
      for (uint32_t e = 0; __ballot(e < tb_n); ++e)         // (as many steps as the wave's longest list)
        addj += (e < tb_n && (uint32_t)tb[e] > (uint32_t)j) ? 1u : 0u;
    }
    const u128 key = (CANON && roll.rc < roll.fwd) ? roll.rc : roll.fwd;
    const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
    uint32_t h = t2_slot<T2L>(lo, hi) | ((j < nk && in_subset2(lo, hi, ss)) ? 0u : T2_DONE);
    // (a pass that may still be split gives up early: probing a nearly full table is the slow way
    //  to find out that it is full)
    // (the pass over the whole leaf or sub-value probes on: see msp.hip)
    const int trips = ovf ? (ss.mask == 0u ? T2_TRIPS : T2_TRIPS_SPLIT) : T2_TRIPS;
    for (int it = 0; it < trips && __ballot((int32_t)h >= 0); ++it) t2_step<false, T2L>(keys, cnts, lo, hi, addj, h);
    if ((int32_t)h >= 0) {
      if (ovf) {
        *ovf = 1u;
      } else {
        t.stats[ST_SPILLED] = 1;
        dev_count_event(&t.stats[ST_AUX1]);
        table_add2(t, lo, hi, addj);
      }
    }
    roll.next();
  }
}

// Record table as in msp.hip: complete runs of a leaf are mostly byte-identical across reads.
// Entry = Rec2 whose header word is the state: 0 empty, R2_LOCK while being written, else
// count << 6 | (n-1).
constexpr int R2_LOG = 10, R2 = 1 << R2_LOG;
static_assert(R2 == Q3_THREADS, "phase 2 lists the record table with one slot per thread");
constexpr uint32_t R2_LOCK = 0xFFFFFFFFu;
constexpr uint32_t R2_EMPTY = 62u;                 // n-1 = 62 does not occur
constexpr uint32_t R2_DONE = 0x80000000u;
constexpr int R2_TRIPS = 96;
constexpr int TL2_PER = 4, TL2_CAP = TL2_PER * 1024;   // truncated runs that may be anchored per leaf
constexpr int FL2_CAP = 1024;                           // ... of which so many may lack a twin
constexpr int T2_LOG_MID = 11;                          // (q3_body: the mid-size instantiation)

__device__ __forceinline__ uint32_t r2_slot(const Rec2 &r) {
  uint32_t h = (r.a.x * 0x9E3779B1u) ^ (r.a.y * 0x85EBCA77u) ^ (r.a.z * 0xC2B2AE3Du) ^ (r.a.w * 0x27D4EB2Fu) ^
               (r.b.x * 0x165667B1u) ^ (r.b.y * 0xD3A2646Cu) ^ ((r.b.w & 63u) * 0xFD7046C5u);
  h = (h ^ (h >> 15)) * 0x2C1B3C6Du;
  return h >> (32 - R2_LOG);
}

// The record table is keyed by the record's FIRST k-mer (its top 2k bits, 66 .. 128 of them): a
// truncated run that is a prefix of a complete run finds its twin by probing from the same slot
// (msp.hip explains the scheme).
__device__ __forceinline__ uint32_t r2_slot_k(const Rec2 &r, int k, int log = R2_LOG) {
  const int nb = 2 * k - 64;                       // bits of the first k-mer beyond a.x, a.y: 2 .. 64
  const uint32_t z = (nb >= 32) ? r.a.z : (r.a.z & ~(0xFFFFFFFFu >> nb));
  const uint32_t w = (nb <= 32) ? 0u : ((nb >= 64) ? r.a.w : (r.a.w & ~(0xFFFFFFFFu >> (nb - 32))));
  uint32_t h = (r.a.x * 0x9E3779B1u) ^ (r.a.y * 0x85EBCA77u) ^ (z * 0xC2B2AE3Du) ^ (w * 0x27D4EB2Fu);
  h = (h ^ (h >> 15)) * 0x2C1B3C6Du;
  return h >> (32 - log);
}
// do the first `len` bases (33 <= len <= 96) of two records agree?
__device__ __forceinline__ bool rec2_prefix_equal(const Rec2 &e, const Rec2 &r, int len) {
  auto mk = [&](int i) {                           // mask of word i: bits 32 i .. 32 i + 31 of the string
    const int b = 2 * len - 32 * i;
    return (b >= 32) ? 0xFFFFFFFFu : ((b <= 0) ? 0u : ~(0xFFFFFFFFu >> b));
  };
  return ((e.a.x ^ r.a.x) | (e.a.y ^ r.a.y) | ((e.a.z ^ r.a.z) & mk(2)) | ((e.a.w ^ r.a.w) & mk(3)) |
          ((e.b.x ^ r.b.x) & mk(4)) | ((e.b.y ^ r.b.y) & mk(5))) == 0u;
}
// reverse complement of a record's run (len bases, 33 <= len <= 96); header word unchanged
__device__ __forceinline__ Rec2 revcomp_record2(const Rec2 &rec, int len) {
  auto rcw = [](uint32_t x) {                      // reverse the 16 bases of a word and complement them
    x = __brev(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return ~x;
  };
  // the 96-base field reversed: the reverse complement preceded by 96 - len complemented pad bases
  uint64_t s0 = ((uint64_t)rcw(rec.b.y) << 32) | rcw(rec.b.x);
  uint64_t s1 = ((uint64_t)rcw(rec.a.w) << 32) | rcw(rec.a.z);
  uint64_t s2 = ((uint64_t)rcw(rec.a.y) << 32) | rcw(rec.a.x);
  int sh = 2 * (96 - len);                         // 0 .. 126: shift the pad out at the top
  if (sh >= 64) { s0 = s1; s1 = s2; s2 = 0; sh -= 64; }
  if (sh) {
    s0 = (s0 << sh) | (s1 >> (64 - sh));
    s1 = (s1 << sh) | (s2 >> (64 - sh));
    s2 <<= sh;
  }
  Rec2 out = rec;
  out.a = make_uint4((uint32_t)(s0 >> 32), (uint32_t)s0, (uint32_t)(s1 >> 32), (uint32_t)s1);
  out.b.x = (uint32_t)(s2 >> 32); out.b.y = (uint32_t)s2;
  return out;
}

// insert-or-count one record per lane; state = slot h with R2_DONE or-ed in once placed.  On return lanes still without
// R2_DONE found no place.  (mask: slots - 1; inc: the record's multiplicity << 6 -- 1 << 6 unless the stream holds
// deduplicated runs, see "multi-GPU by runs")
// The record table is SPLIT: bases 0..63 (ra), bases 64..95 (rb), the spare word (rz; nullptr where it is always
// zero) and the state words (rst) in arrays of their own.  With 32-byte entries the state word of every slot sits in LDS
// bank 7 mod 8 -- a wave's 64 atomics land on 4 of the 32 banks -- and a wave's 16-byte reads of random entries use half
// the banks (77 % of the sender kernel's LDS-active cycles were bank conflicts, profiles/r05/pipelined_exchange_kernels_k63_*).
__device__ __forceinline__ void r2s_insert_loop(uint4 *ra, uint2 *rb, uint32_t *rz, uint32_t *rst, const Rec2 &rec, uint32_t &h,
                                                uint32_t mask, uint32_t inc = 1u << 6) {
  const uint32_t nm1 = rec.b.w & 63u;
  for (int it = 0; it < R2_TRIPS && __ballot((int32_t)h >= 0); ++it) {
    const bool p = (int32_t)h >= 0;
    const uint32_t hh = h & mask;
    const uint32_t st = rst[hh];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the bases only after the state
    const uint2 eb = rb[hh];
    const uint4 ea = ra[hh];
    // EMPTY and LOCK carry low header bits no record has, so they never compare equal
    const bool match = (((st ^ rec.b.w) & 63u) | (eb.x ^ rec.b.x) | (eb.y ^ rec.b.y) | (ea.x ^ rec.a.x) |
                        (ea.y ^ rec.a.y) | (ea.z ^ rec.a.z) | (ea.w ^ rec.a.w)) == 0u;
    const bool empty = st == R2_EMPTY;
    uint32_t won = 0u;
    if (p && empty) {
      if (atomicCAS(&rst[hh], R2_EMPTY, R2_LOCK) == R2_EMPTY) {
        ra[hh] = rec.a;
        rb[hh] = make_uint2(rec.b.x, rec.b.y);
        if (rz) rz[hh] = rec.b.z;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicExch(&rst[hh], inc | nm1);
        won = 1u;
      }
    }
    if (p && match) atomicAdd(&rst[hh], inc);
    const bool stay = match || empty || st == R2_LOCK;
    const uint32_t nh = stay ? hh : ((hh + 1) & mask);
    h = (p && !match && won == 0u) ? nh : (h | R2_DONE);
  }
}

// SHARED: 2^sub_bits workgroups per leaf (an instantiation of its own: the ordinary kernel is short
// of scalar registers as it is)
// mode & Q3_WEIGHTED: the complete streams hold DISTINCT runs with multiplicities (header = count << 6 |
// n-1, what msp2_dedupe_export_kernel leaves behind): an owner counting the runs its ranks sent
constexpr uint32_t Q3_WEIGHTED = 1u;
constexpr uint32_t Q3_SPLIT2 = 2u;     // every leaf starts with two key subsets (the mid-size instantiation on leaves of > ~1400 keys)
constexpr uint64_t Q3_HUGE_LEAF = 1ull << 25, Q3_HUGE_LEAF_SENDER = 1ull << 19;   // (msp.hip: HUGE_LEAF)
// LISTS (round 5, the owner of the PIPELINED runs exchange, msp.hip: p3_body): the leaf's runs are the N lists its ranks
// sent, read in place from the receive buffer -- two rows per record, a note expanded where it is read.
// T2L: log2 of the k-mer table's slots.  The leaf kernel at ONE 124 KB workgroup per CU spends 62 % of its wave cycles waiting
// (barriers, dependent LDS and L2 round trips; profiles/r04/k63_pmc_sq_counters.txt) with nothing to run beside it.  Where leaves
// are counted whole (not shared by sub-value) two smaller instantiations run TWO workgroups per CU at 64 VGPRs:
//   T2_LOG_SMALL = 10 (62 KB of LDS) for jobs that announce <= 512 table slots per leaf (small genomes: a leaf is ~17 us of
//     fixed work with hardly any data) -- 10 M reads of a 10^7-base genome: leaf kernel 4.3 -> 3.1 ms, k = 63 6.9 -> 5.4 ms per step;
//   T2_LOG_MID = 11 (79 KB: a quarter of the list of twin-less read ends, no spare-word array) for everything else -- a
//     C3-sized leaf (~3000 keys) is counted as two key subsets FROM THE START (Q3_SPLIT2: every record expanded twice,
//     half of its k-mers inserted each time), which costs less than the second workgroup gains: k = 63 on the C3 reads
//     leaf kernel 10.57 -> 9.41 ms, step 32.5 -> 31.1 ms.
// A leaf that outgrows its table is split further by key subset like any other.  (The 4096-slot instantiation stays for
// leaves shared by sub-value, for the owners' list reader, and under CFRK_DEBUG_NO_SMALL_LEAVES.)
template <bool CANON, bool SHARED, bool LISTS, int T2L = T2_LOG>
__device__ __forceinline__ void q3_body(int k, uint32_t mode, const View2 &v, const TableView &t, const P3ListsT<LISTS> &lx) {
  constexpr int T2 = 1 << T2L;                    // (shadows the file-level constant)
  __shared__ ulonglong2 keys[T2];
  __shared__ uint32_t cnts[T2];
  // the record table, SPLIT (r2s_insert_loop): slot s = {ra[s], {rb[s].x, rb[s].y, rz[s], rst[s]}}
  __shared__ uint4 ra[R2];
  __shared__ uint2 rb[R2];
  __shared__ uint32_t rz[SHARED ? R2 : 1], rst[R2];   // (the spare word only matters where leaves are shared)
  auto rt_get = [&](uint32_t s_) { const uint2 e = rb[s_]; return Rec2{ra[s_], make_uint4(e.x, e.y, SHARED ? rz[SHARED ? s_ : 0] : 0u, rst[s_])}; };
  __shared__ uint16_t occ_list[R2];
  // truncated runs anchored to their complete twin (msp.hip): per record-table slot the start of
  // the group of lengths noted with it, the lengths, and the runs without a twin
  __shared__ uint32_t th[R2 + 1];
  __shared__ uint8_t tbytes[TL2_CAP];
  constexpr int FLC = (T2L == T2_LOG_MID) ? 256 : FL2_CAP;   // (the mid-size instantiation has 512 bytes for it)
  __shared__ uint16_t flist[FLC];
  __shared__ uint32_t nfb, nfl;
  __shared__ uint32_t wsum2[Q3_THREADS / 64];
  constexpr int NSV = SHARED ? 4 : 1;             // sub-values a workgroup of a shared leaf counts, one after the other
  __shared__ uint32_t nhist[NSV * 32 + 1];       // distinct runs by (sub-value, length): counts, then list positions
  __shared__ uint32_t nocc;
  __shared__ uint32_t wg_total;
  __shared__ unsigned long long wg_base;
  __shared__ uint32_t rt_fail, kovf;             // record table / k-mer table ran out of room (msp.hip)
  __shared__ uint32_t stk[SHARED ? 192 : 40];     // subsets still to count: sub-value << 24 | bits << 16 | value
  __shared__ int sp;
  __shared__ uint32_t leaf_total, nseg;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool weighted = LISTS || (mode & Q3_WEIGHTED) != 0u;
  // sub_bits > 0: 2^sub_bits workgroups share a leaf, each taking the records whose extra
  // minimizer-hash bits (b.z, written by msp2_p1_kernel<true>) name it -- every occurrence of a
  // k-mer has the same minimizer, so the workgroups' key sets are disjoint.  The workgroups of a
  // leaf run next to each other on one XCD (workgroup b goes to XCD b % 8), whose L2 then serves
  // all but the first read of the leaf's streams.
  // A workgroup takes up to four sub-values (the low hbits of the extra bits) and counts them one
  // after the other from ONE pass over the leaf's streams: the record table holds the distinct runs
  // of all four (~4 x 120), the k-mer table one sub-value's keys at a time.  2^(sub_bits - hbits)
  // workgroups per leaf: every one of them reads the whole leaf, and that was most of the kernel's
  // time with eight of them.
  const uint32_t sub_bits = SHARED ? v.sub_bits : 0u;
  const uint32_t hbits = SHARED ? v.hbits : 0u, hmask = (1u << hbits) - 1u;
  const uint32_t gbits = sub_bits - hbits;
  const uint32_t rmask = (1u << gbits) - 1u;
  const uint32_t vq = blockIdx.x >> 3;
  const uint32_t rsel = vq & rmask;
  // (the grid holds the leaves of this pass only: were the others launched and left at once, the
  //  pass's leaves -- equal low bits -- would all sit on 8 / 2^sel_bits of the 8 XCDs)
  const uint32_t leaf = (SHARED ? ((((vq >> gbits) << 3) | (blockIdx.x & 7u))) : blockIdx.x) << v.sel_bits | v.sel_val;
  auto mine = [&](const Rec2 &r) { return !SHARED || ((r.b.z >> hbits) & rmask) == rsel; };
  uint64_t ns[NCLS];
  uint64_t total = 0;
  // LISTS: rank r's lists of this leaf -- l_base[r] = its first row, complete runs [l_cpre[r], l_cpre[r+1]) of the leaf's
  // complete "stream" (class 3), truncated runs + notes [l_tpre[r], l_tpre[r+1]) of its truncated one (class 0; l_nu[r]
  // records, then the notes)
  __shared__ uint32_t l_cpre[LISTS ? 65 : 1], l_tpre[LISTS ? 65 : 1], l_nu[LISTS ? 64 : 1];
  __shared__ const uint4 *l_base[LISTS ? 64 : 1];
  if constexpr (LISTS) {
    if (leaf >= lx.lcount) return;                     // (shared leaves: the grid is padded to whole groups of eight)
    if (tid < 64) {
      const int r = tid;
      uint32_t nd = 0, nu = 0, na = 0;
      const uint4 *base = nullptr;
      if (r < lx.parts) {
        const uint4 *hdr = lx.packed + lx.rr.rstart[r];
        const uint64_t rows = lx.rr.rows[r], hrows = 1ull + lx.lcount;
        bool ok = rows >= hrows;
        if (ok) { const uint4 h0 = hdr[0]; ok = h0.y == lx.lcount && h0.z == lx.ll0 && h0.w == RUNS2_MAGIC; }
        if (ok) {
          const uint4 e = hdr[1u + leaf];               // (LISTS: `leaf` is the local leaf of the group)
          const uint64_t tot = 2ull * ((uint64_t)e.y + e.z) + (e.w + (uint32_t)NOTES_PER_ROW - 1u) / (uint32_t)NOTES_PER_ROW;
          // (a segment that does not add up, or notes without a run they could point at, is not followed)
          if ((uint64_t)e.x + tot <= rows - hrows && (e.w == 0u || e.y != 0u)) { nd = e.y; nu = e.z; na = e.w; base = hdr + hrows + e.x; }
          else if (tot) ok = false;
        }
        if (!ok) v.stats[ST_OVERFLOW] = 1;                 // (reported by finish / digest: the result would be incomplete)
      }
      const uint32_t ci = dev_wave_scan_incl(nd), ti = dev_wave_scan_incl(nu + na);
      l_cpre[r] = ci - nd; l_tpre[r] = ti - (nu + na); l_nu[r] = nu; l_base[r] = base;
      if (r == 63) { l_cpre[64] = ci; l_tpre[64] = ti; }
    }
    __syncthreads();
    ns[0] = l_tpre[64]; ns[1] = 0; ns[2] = 0; ns[3] = l_cpre[64];
    total = ns[0] + ns[3];
  } else {
#pragma unroll
    for (int cl = 0; cl < NCLS; ++cl) {
      ns[cl] = min((uint64_t)v.cnt2[NCLS * leaf + cl],
                   v.exact ? (uint64_t)v.lcap[NCLS * leaf + cl] : (cl == 3 ? v.cap2c : v.cap2t));
      total += ns[cl];
    }
    if (total - 1ull >= Q3_HUGE_LEAF - 1ull) return;    // nothing to count -- or a HUGE leaf: msp2_huge_leaves_kernel counts it
  }
  auto l_find = [&](const uint32_t *pre, uint32_t i) {        // rank r with pre[r] <= i < pre[r + 1] (LISTS)
    uint32_t lo = 0, hi = 64;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= i) lo = mid; else hi = mid; }
    return lo;
  };
  for (int s = tid; s < T2; s += Q3_THREADS) cnts[s] = 0;
  rst[tid] = R2_EMPTY;
  if (SHARED) rz[SHARED ? tid : 0] = 0u;
  if (tid == 0) { wg_total = 0; nocc = 0; rt_fail = 0; kovf = 0; sp = 0; leaf_total = 0; nseg = 0; nfb = 0; nfl = 0; }
  if (tid < NSV * 32 + 1) nhist[tid] = 0;
  for (int s = tid; s < R2 + 1; s += Q3_THREADS) th[s] = 0;
  __syncthreads();

  const Rec2 *leaf_rec = LISTS ? nullptr : (v.exact ? v.rec2 + v.lbase[NCLS * leaf + 3] : v.rec2 + (uint64_t)(leaf >> v.sel_bits) * (v.cap2c + 3 * v.cap2t));
  const Rec2 zrec = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  // complete run i of the leaf / record i of its truncated class cl, wherever they lie
  auto ld_c = [&](uint64_t i) -> Rec2 {
    if constexpr (LISTS) {
      const uint32_t r = l_find(l_cpre, (uint32_t)i), j = (uint32_t)i - l_cpre[r];
      const uint4 *b = l_base[r];
      return Rec2{b[2 * j], b[2 * j + 1]};
    } else return leaf_rec[i];
  };
  auto ld_t = [&](int cl, uint64_t i) -> Rec2 {
    if constexpr (LISTS) {
      const uint32_t r = l_find(l_tpre, (uint32_t)i), j = (uint32_t)i - l_tpre[r];
      const uint4 *b = l_base[r];
      const uint32_t nd_r = l_cpre[r + 1] - l_cpre[r], nu_r = l_nu[r];
      const uint4 *st = b + 2ull * nd_r;
      if (j < nu_r) return Rec2{st[2 * j], st[2 * j + 1]};
      // a note: the first n k-mers of its twin, closed on the left only (msp2_runs_scatter_kernel does this in a pass of its own)
      const uint32_t note = reinterpret_cast<const uint16_t *>(st + 2ull * nu_r)[j - nu_r];
      const uint32_t ti = min(note >> 5, nd_r - 1u);                      // a position outside the list is not followed
      const uint4 ta = b[2 * ti], tb = b[2 * ti + 1];
      const uint32_t nm1 = min(note & 31u, tb.w & 31u);
      const int len = (int)nm1 + k;                                       // bases of the run: 33 .. 93
      auto mk = [&](int w) {                                              // mask of word w: bits 32 w .. 32 w + 31 of the string
        const int bb = 2 * len - 32 * w;
        return (bb >= 32) ? 0xFFFFFFFFu : ((bb <= 0) ? 0u : ~(0xFFFFFFFFu >> bb));
      };
      return Rec2{make_uint4(ta.x, ta.y, ta.z & mk(2), ta.w & mk(3)), make_uint4(tb.x & mk(4), tb.y & mk(5), tb.z, 64u | nm1)};
    } else {
      const Rec2 *src = v.exact ? v.rec2 + v.lbase[NCLS * leaf + cl] : leaf_rec + v.cap2c + (uint64_t)cl * v.cap2t;
      return src[i];
    }
  };
  if constexpr (LISTS) {
    if (total == 0) return;
    if (total >= Q3_HUGE_LEAF) {                       // (see Q3_HUGE_LEAF: counted in the HBM table, whose adds saturate)
      for (uint64_t i = tid; i < total; i += Q3_THREADS) {
        const Rec2 rec = (i < ns[3]) ? ld_c(i) : ld_t(0, i - ns[3]);
        spill_record2(rec.a, rec.b, k, CANON, t, (i < ns[3]) ? (rec.b.w >> 6) : 1u);
      }
      return;
    }
  }
  // ---- phase 1: complete runs, one record-table update per record; when the table runs out of
  //      room (more distinct runs than it holds: low coverage of a large genome) the dedupe is
  //      dropped and the whole leaf is counted from its streams
  //      Most records find their twin in the home slot (two LDS reads, an XOR/OR reduction, one
  //      LDS add, every lane busy); the rest are compacted across the wave into a 64-lane leftover
  //      set and only a full set runs the probe loop (as in msp.hip).  The next record is asked
  //      for before the current one is inserted: one workgroup per CU, nothing else hides the load.
  {
    Rec2 Lr = zrec;                    // leftover records, lanes [0, c)
    uint32_t Lh = 0;
    int c = 0;                         // wave-uniform
    auto drain = [&](int cnt) {
      uint32_t h = Lh | ((lane < cnt) ? 0u : R2_DONE);
      r2s_insert_loop(ra, rb, SHARED ? rz : nullptr, rst, Lr, h, R2 - 1, weighted ? (Lr.b.w & ~63u) : (1u << 6));
      if ((int32_t)h >= 0) rt_fail = 1u;
    };
    auto home = [&](const Rec2 &rec, bool valid) {
      const uint32_t h = r2_slot_k(rec, k);
      const uint32_t st = rst[h];
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the bases only after the state
      const uint2 eb = rb[h];
      const uint4 ea = ra[h];
      const bool match = valid && ((((st ^ rec.b.w) & 63u) | (eb.x ^ rec.b.x) | (eb.y ^ rec.b.y) | (ea.x ^ rec.a.x) |
                                    (ea.y ^ rec.a.y) | (ea.z ^ rec.a.z) | (ea.w ^ rec.a.w)) == 0u);
      // (the state words are an array of their own: a wave's 64 adds spread over all banks)
      if (match) atomicAdd(&rst[h], weighted ? (rec.b.w & ~63u) : (1u << 6));
      const bool left = valid && !match;
      const unsigned long long mask = __ballot(left);
      if (mask == 0ull) return;
      const int n = __popcll(mask);
      if (c + n > 64) { drain(c); c = 0; }
      // (the spare word travels only when leaves are shared: it holds the sub-value then)
      if (SHARED) {
        uint32_t set[9] = {Lr.a.x, Lr.a.y, Lr.a.z, Lr.a.w, Lr.b.x, Lr.b.y, Lr.b.z, Lr.b.w, Lh};
        const uint32_t mine_[9] = {rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, rec.b.y, rec.b.z, rec.b.w, h};
        wave_append<9>(set, mine_, left, mask, c, n);
        Lr.a = make_uint4(set[0], set[1], set[2], set[3]); Lr.b = make_uint4(set[4], set[5], set[6], set[7]); Lh = set[8];
      } else {
        uint32_t set[8] = {Lr.a.x, Lr.a.y, Lr.a.z, Lr.a.w, Lr.b.x, Lr.b.y, Lr.b.w, Lh};
        const uint32_t mine_[8] = {rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, rec.b.y, rec.b.w, h};
        wave_append<8>(set, mine_, left, mask, c, n);
        Lr.a = make_uint4(set[0], set[1], set[2], set[3]); Lr.b = make_uint4(set[4], set[5], 0u, set[6]); Lh = set[7];
      }
      c += n;
    };
    if (!SHARED) {
      // four loads in flight per lane: ONE workgroup runs on a CU, so what its 1024 threads have in flight is all
      // the CU has in flight -- with one prefetched record per thread (32 KB per CU) the scan ran at what a 2 us
      // round trip allows, ~4 TB/s for the chip (round 4)
      const uint64_t nr = (ns[3] + 63) & ~63ull;
      for (uint64_t r = tid; r < nr && !(v.dbg & CFRK_ABL_P3_NO_RTAB); r += 4 * Q3_THREADS) {
        Rec2 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          q[u] = zrec;
          if (r + (uint64_t)u * Q3_THREADS < ns[3]) q[u] = ld_c(r + (uint64_t)u * Q3_THREADS);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (r + (uint64_t)u * Q3_THREADS >= nr) break;      // (wave-uniform)
          home(q[u], r + (uint64_t)u * Q3_THREADS < ns[3]);
        }
      }
    } else {
      // a shared leaf: most records read here are another workgroup's.  Four loads are in flight per
      // lane (the stream comes from the XCD's L2, and latency, not work, bounds this loop)
      // The one lane in 2^s that holds a record of this workgroup is gathered into full sets of 64
      // before the table look-up: home() costs the wave ~100 instructions however few lanes take part.
      Rec2 Cr = zrec;
      int cc = 0;                        // wave-uniform
      auto cfeed = [&](const Rec2 &rec, bool keep) {
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) return;
        const int n = __popcll(mask);
        if (cc + n > 64) { home(Cr, lane < cc); cc = 0; }
        uint32_t set[8] = {Cr.a.x, Cr.a.y, Cr.a.z, Cr.a.w, Cr.b.x, Cr.b.y, Cr.b.z, Cr.b.w};
        const uint32_t mine_[8] = {rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, rec.b.y, rec.b.z, rec.b.w};
        wave_append<8>(set, mine_, keep, mask, cc, n);
        Cr.a = make_uint4(set[0], set[1], set[2], set[3]); Cr.b = make_uint4(set[4], set[5], set[6], set[7]);
        cc += n;
      };
      for (uint64_t r = tid; r < ((ns[3] + 63) & ~63ull) && !(v.dbg & CFRK_ABL_P3_NO_RTAB); r += 4 * Q3_THREADS) {
        Rec2 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          q[u] = zrec;
          if (r + (uint64_t)u * Q3_THREADS < ns[3]) q[u] = ld_c(r + (uint64_t)u * Q3_THREADS);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (r + (uint64_t)u * Q3_THREADS >= ((ns[3] + 63) & ~63ull)) break;      // (wave-uniform)
          cfeed(q[u], r + (uint64_t)u * Q3_THREADS < ns[3] && mine(q[u]));
        }
      }
      if (cc) home(Cr, lane < cc);
    }
    if (c) drain(c);
  }
  __syncthreads();
  const bool big = rt_fail != 0u;
  // ---- phase 2: every distinct complete record once (weight = multiplicity), listed longest
  //      first (a wave expands 64 records in lock-step for as many steps as its longest one),
  //      then the truncated runs by length class, long ones first
  {
    const uint32_t st = rst[tid];
    const bool occ = !big && st != R2_EMPTY;
    // list position: by sub-value (shared leaves), then longest first -- entry hidx of the histogram
    const uint32_t hidx = (SHARED ? (rz[SHARED ? tid : 0] & hmask) * 32u : 0u) + (31u - (st & 31u));
    uint32_t rank = 0;
    if (occ) rank = atomicAdd(&nhist[hidx], 1u);
    __syncthreads();
    if (tid < 64) {                                // exclusive prefix over the NSV * 32 entries (one wave, two rounds)
      uint32_t carry = 0;
#pragma unroll
      for (int base = 0; base < NSV * 32; base += 64) {
        const uint32_t own = (base + tid < NSV * 32) ? nhist[base + tid] : 0u;
        uint32_t incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t y = __shfl_up(incl, d);
          if (tid >= d) incl += y;
        }
        if (base + tid < NSV * 32) nhist[base + tid] = carry + incl - own;
        carry += __shfl(incl, 63);
      }
      if (tid == 0) { nhist[NSV * 32] = carry; nocc = carry; }
    }
    __syncthreads();
    if (occ) occ_list[nhist[hidx] + rank] = (uint16_t)tid;
    if (tid == 0) {
      // one item per sub-value (shared leaves: 2^hbits of them; pushed so that 0 comes first) ...
      int d0 = 0;
      for (int sv = (int)hmask; sv >= 0; --sv) {
        if (big) {
          // ... no dedupe: the leaf holds thousands of distinct runs.  Start with as many key subsets as
          // its size suggests (one per ~6000 records, 4 .. 32) instead of finding out by overflowing
          uint32_t b0 = 2u;
          while (b0 < 5u && (total >> (b0 + sub_bits)) > 6000ull) ++b0;
          for (uint32_t q = 0; q < (1u << b0); ++q) stk[d0++] = ((uint32_t)sv << 24) | (b0 << 16) | q;
        } else if (mode & Q3_SPLIT2) {
          // (the leaves of this job hold more keys than the table takes in one go: two key subsets from the start
          //  instead of finding out by overflowing)
          stk[d0++] = ((uint32_t)sv << 24) | (1u << 16) | 1u;
          stk[d0++] = ((uint32_t)sv << 24) | (1u << 16) | 0u;
        } else stk[d0++] = (uint32_t)sv << 24;
      }
      sp = d0;
    }
  }
  // ---- truncated runs: the first TL2_CAP of the three class streams (class 0 first) look for the
  //      complete run they are a prefix of (other strand: suffix) and leave their length with it
  const uint32_t c0 = (uint32_t)min(ns[0], (uint64_t)TL2_CAP);
  const uint32_t c1 = c0 + (uint32_t)min(ns[1], (uint64_t)(TL2_CAP - c0));
  const uint32_t tl = c1 + (uint32_t)min(ns[2], (uint64_t)(TL2_CAP - c1));
  auto trunc_at = [&](uint32_t g) -> Rec2 {                  // record g of the concatenated class streams
    const int cl = (g < c0) ? 0 : (g < c1) ? 1 : 2;
    const uint32_t i = g - ((cl == 0) ? 0u : (cl == 1) ? c0 : c1);
    return ld_t(cl, i);
  };
  const bool anchors_on = !big && !(v.dbg & (CFRK_DEBUG_NO_ANCHORS | CFRK_ABL_P3_NO_TRUNC));
  bool use_anchors = false;
  {
    constexpr uint32_t TW_NONE = 0xFFFFFFFFu, TW_TWIN = 0x80000000u;
    uint32_t tw[TL2_PER], trank[TL2_PER];
    // (all TL2_PER records of a thread are asked for before the first is looked up: one workgroup per CU, so four
    //  dependent round trips here were four times ~2 us with nothing else to run -- round 4)
    Rec2 trec[TL2_PER];
#pragma unroll
    for (int i = 0; i < TL2_PER; ++i) {
      const uint32_t g = (uint32_t)(i * Q3_THREADS + tid);
      trec[i] = zrec;
      if (anchors_on && g < tl) trec[i] = trunc_at(g);
    }
#pragma unroll
    for (int i = 0; i < TL2_PER; ++i) {
      const uint32_t g = (uint32_t)(i * Q3_THREADS + tid);
      bool valid = anchors_on && g < tl;
      tw[i] = TW_NONE; trank[i] = 0u;
      if (!anchors_on || !__ballot(valid)) continue;
      Rec2 rec = trec[i];
      valid = valid && mine(rec);
      const uint32_t nm1 = rec.b.w & 31u;
      if (valid) tw[i] = nm1;
      const bool lc = (rec.b.w & 64u) != 0u, rc_ = (rec.b.w & 128u) != 0u;
      const bool suf = CANON && valid && !lc && rc_;
      if (suf) rec = revcomp_record2(rec, (int)nm1 + k);
      const bool anchored = suf || (valid && lc && !rc_);
      uint32_t h = anchored ? r2_slot_k(rec, k) : R2_DONE;
      uint32_t found = TW_NONE;
      for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
        const bool p = (int32_t)h >= 0;
        const uint32_t hh = h & (uint32_t)(R2 - 1);
        const Rec2 e2 = rt_get(hh);
        const bool empty = e2.b.w == R2_EMPTY;
        // the twin holds at least as many k-mers and starts with the same nm1 + k bases
        const bool hit = p && !empty && (e2.b.w & 31u) >= nm1 && rec2_prefix_equal(e2, rec, (int)nm1 + k);
        found = hit ? hh : found;
        h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(R2 - 1)) : (h | R2_DONE);
      }
      if (found != TW_NONE) tw[i] = TW_TWIN | (found << 8) | (nm1 + 1u);
      const unsigned long long fb = __ballot(valid && found == TW_NONE);
      if (lane == 0 && fb) atomicAdd(&nfb, (uint32_t)__popcll(fb));
    }
    __syncthreads();
    use_anchors = anchors_on && nfb <= (uint32_t)FLC;
    if (use_anchors) {
#pragma unroll
      for (int i = 0; i < TL2_PER; ++i) {
        if (tw[i] == TW_NONE) continue;
        if (tw[i] & TW_TWIN) trank[i] = atomicAdd(&th[(tw[i] >> 8) & (uint32_t)(R2 - 1)], 1u);
        else flist[atomicAdd(&nfl, 1u)] = (uint16_t)(i * Q3_THREADS + tid);
      }
    }
    __syncthreads();
    if (use_anchors) {
      // group starts: exclusive prefix of the groups' sizes over the record table's slots
      const uint32_t own = th[tid];
      uint32_t incl = own;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
      }
      if (lane == 63) wsum2[tid >> 6] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (int w = 0; w < (tid >> 6); ++w) base += wsum2[w];
      const uint32_t goff = base + incl - own;
      __syncthreads();                               // (everybody has read its group's size)
      th[tid] = goff;
      if (tid == Q3_THREADS - 1) th[R2] = goff + own;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < TL2_PER; ++i)
        if (tw[i] != TW_NONE && (tw[i] & TW_TWIN)) tbytes[th[(tw[i] >> 8) & (uint32_t)(R2 - 1)] + trank[i]] = (uint8_t)(tw[i] & 63u);
    }
  }
  __syncthreads();
  // truncated runs of every class that the anchoring covered (they are skipped by the stream loops below)
  uint32_t cov[3] = {0u, 0u, 0u};
  if (use_anchors) { cov[0] = c0; cov[1] = c1 - c0; cov[2] = tl - c1; }
  constexpr uint32_t SUBSET_BITS_MAX = 8;
  bool first_pass = true;
  while (true) {
    const int depth = sp;
    if (depth == 0) break;
    const uint32_t item = stk[depth - 1];
    __syncthreads();
    if (tid == 0) { sp = depth - 1; kovf = 0; wg_total = 0; }
    const uint32_t bits = (item >> 16) & 0xFFu, sv = item >> 24;
    const KeySubset2 ss{(1u << bits) - 1u, item & 0xFFFFu, hmask, sv};
    uint32_t *ovf = (bits < SUBSET_BITS_MAX) ? &kovf : nullptr;
    if (!first_pass)
      for (int s = tid; s < T2; s += Q3_THREADS) cnts[s] = 0;
    first_pass = false;
    __syncthreads();
    if (!big) {
      // few distinct runs (~320) for 1024 lanes, and a wave works for as many steps as its longest
      // run has k-mers: up to four lanes share a record, each expanding a quarter of its k-mers
      // (a shared leaf: the part of the list that holds this pass's sub-value)
      const uint32_t occ0 = SHARED ? nhist[sv * 32u] : 0u;
      const uint32_t nocc_ = SHARED ? nhist[sv * 32u + 32u] - occ0 : nocc;
      // (up to eight: a shared leaf's sub-value has ~120 runs, and half-empty waves wait for LDS twice as long)
      int parts = 1;
      while (parts < 8 && nocc_ * (uint32_t)(parts + 1) <= (uint32_t)Q3_THREADS) ++parts;
      const uint32_t precip = (65536u + (uint32_t)parts - 1u) / (uint32_t)parts;     // i / parts for i < 8192
      const uint32_t nitems = (v.dbg & CFRK_ABL_P3_NO_CEXP) ? 0u : nocc_ * (uint32_t)parts;
      for (uint32_t i = tid; i < ((nitems + 63u) & ~63u); i += Q3_THREADS) {
        const bool valid = i < nitems;
        const uint32_t ri = (i * precip) >> 16;
        Rec2 rec = zrec;
        uint32_t slot = 0;
        if (valid) { slot = occ_list[occ0 + ri]; rec = rt_get(slot); }
        if (use_anchors) {
          // ... plus one for every truncated run of this locus that reaches the k-mer
          const uint32_t g0 = th[slot];
          count_record2<CANON, T2L>(keys, cnts, rec, rec.b.w >> 6, valid, k, t, ss, ovf, (int)(i - ri * (uint32_t)parts), parts,
                               tbytes + g0, valid ? th[slot + 1] - g0 : 0u);
        } else {
          count_record2<CANON, T2L>(keys, cnts, rec, rec.b.w >> 6, valid, k, t, ss, ovf, (int)(i - ri * (uint32_t)parts), parts);
        }
      }
      // truncated runs without a twin
      for (uint32_t i = tid; i < ((nfl + 63u) & ~63u) && use_anchors; i += Q3_THREADS) {
        const bool valid = i < nfl;
        Rec2 rec = zrec;
        if (valid) rec = trunc_at(flist[i]);
        count_record2<CANON, T2L>(keys, cnts, rec, 1u, valid, k, t, ss, ovf);
      }
    } else if (!SHARED) {
      for (uint64_t r = tid; r < ((ns[3] + 63) & ~63ull); r += Q3_THREADS) {
        const bool valid = r < ns[3];
        Rec2 rec = zrec;
        if (valid) rec = ld_c(r);
        count_record2<CANON, T2L>(keys, cnts, rec, weighted ? (rec.b.w >> 6) : 1u, valid, k, t, ss, ovf);
      }
    }
    if constexpr (!SHARED) {
      for (int cl = LISTS ? 0 : 2; cl >= 0 && !(v.dbg & CFRK_ABL_P3_NO_TRUNC); --cl) {
        Rec2 nxt = zrec;
        const uint64_t r_first = (uint64_t)cov[cl] + tid;        // (the anchored ones are done)
        if (r_first < ns[cl]) nxt = ld_t(cl, r_first);
        for (uint64_t r = r_first; r < (uint64_t)cov[cl] + ((ns[cl] - cov[cl] + 63) & ~63ull); r += Q3_THREADS) {
          const bool valid = r < ns[cl];
          const Rec2 rec = nxt;
          nxt = zrec;
          if (r + Q3_THREADS < ns[cl]) nxt = ld_t(cl, r + Q3_THREADS);
          count_record2<CANON, T2L>(keys, cnts, rec, 1u, valid, k, t, ss, ovf);
        }
      }
    } else {
      // a shared leaf: one lane in 2^sub_bits holds a record of this workgroup.  They are gathered
      // across the wave into full sets of 64 before they are expanded (lock-step: a wave expands
      // for as long as its longest record, however few of its lanes hold one)
      Rec2 Sr = zrec;
      int sc = 0;                        // wave-uniform
      auto sflush = [&]() {
        count_record2<CANON, T2L>(keys, cnts, Sr, 1u, lane < sc, k, t, KeySubset2{ss.mask, ss.val, 0u, 0u}, ovf);   // (sub-value: filtered when fed)
        sc = 0;
      };
      auto sfeed = [&](const Rec2 &rec, bool keep) {
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) return;
        const int n = __popcll(mask);
        if (sc + n > 64) sflush();
        uint32_t set[7] = {Sr.a.x, Sr.a.y, Sr.a.z, Sr.a.w, Sr.b.x, Sr.b.y, Sr.b.w};
        const uint32_t mine_[7] = {rec.a.x, rec.a.y, rec.a.z, rec.a.w, rec.b.x, rec.b.y, rec.b.w};
        wave_append<7>(set, mine_, keep, mask, sc, n);
        Sr.a = make_uint4(set[0], set[1], set[2], set[3]); Sr.b = make_uint4(set[4], set[5], 0u, set[6]);
        sc += n;
      };
      for (int cl = big ? 3 : 2; cl >= 0 && !(cl < 3 && (v.dbg & CFRK_ABL_P3_NO_TRUNC)); --cl) {
        if (LISTS && (cl == 1 || cl == 2)) continue;       // (an owner's truncated runs are one class)
        if (cl == 3 && weighted) {
          // (distinct runs with multiplicities and no room to merge them: rare enough to expand them where they lie)
          for (uint64_t r = tid; r < ((ns[3] + 63) & ~63ull); r += Q3_THREADS) {
            const bool valid = r < ns[3];
            Rec2 rec = zrec;
            if (valid) rec = ld_c(r);
            count_record2<CANON, T2L>(keys, cnts, rec, rec.b.w >> 6, valid && mine(rec), k, t, ss, ovf);
          }
          continue;
        }
        auto ld_cl = [&](uint64_t i) -> Rec2 { return (cl == 3) ? ld_c(i) : ld_t(cl, i); };
        const uint64_t done = (cl == 3) ? 0ull : (uint64_t)cov[cl];   // (the anchored ones are done)
        Rec2 nxt = zrec;
        const uint64_t r_first = done + tid;
        if (r_first < ns[cl]) nxt = ld_cl(r_first);
        for (uint64_t r = r_first; r < done + ((ns[cl] - done + 63) & ~63ull); r += Q3_THREADS) {
          const Rec2 rec = nxt;
          const bool keep = r < ns[cl] && mine(rec) && (rec.b.z & hmask) == sv;
          nxt = zrec;
          if (r + Q3_THREADS < ns[cl]) nxt = ld_cl(r + Q3_THREADS);
          sfeed(rec, keep);
        }
      }
      if (sc) sflush();
    }
    __syncthreads();
    if (kovf) {
      if (tid == 0) {
        const int d0 = sp;
        stk[d0] = (sv << 24) | ((bits + 1u) << 16) | ss.val;
        stk[d0 + 1] = (sv << 24) | ((bits + 1u) << 16) | ss.val | (1u << bits);
        sp = d0 + 2;
      }
      __syncthreads();
      continue;
    }

    // compaction to the two-word result list: one cursor atomic per workgroup and pass
    constexpr int NIT = T2 / Q3_THREADS;
    uint32_t wbase[NIT];
    auto occupied = [&](int s) { return cnts[s] != 0u; };
    wg_rank_slots<NIT, Q3_THREADS>(wbase, &wg_total, occupied);
    __syncthreads();
    if (tid == 0) {
      wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)wg_total);
      if (SHARED && !LISTS) {
        // one segment per sub-value; one that took several key-subset passes has no single segment
        const uint32_t sg = (leaf << sub_bits) | (rsel << hbits) | sv;
        if (bits == 0u) { v.leaf_off[sg] = wg_base; v.leaf_n[sg] = wg_total; }
        else if (wg_total) v.stats[ST_MULTISEG] = 1;
      } else if (!LISTS) {                           // (an owner's result is not kept in leaf form)
        if (nseg == 0) v.leaf_off[leaf] = wg_base;
        else if (wg_total) v.stats[ST_MULTISEG] = 1;
        if (wg_total) nseg = nseg + 1;
        leaf_total += wg_total;
        v.leaf_n[leaf] = leaf_total;
      }
    }
    __syncthreads();
    wg_emit_slots<NIT, Q3_THREADS>(wbase, wg_base, occupied, [&](int s, unsigned long long dst) {
      if (dst < v.out_cap) { v.out_lo[dst] = keys[s].x; v.out_hi[dst] = keys[s].y; v.out_cnt[dst] = cnts[s]; }
      else v.stats[ST_OVERFLOW] = 1;
    });
    __syncthreads();
  }
}

template <bool CANON, bool SHARED>
__global__ __launch_bounds__(Q3_THREADS) void msp2_p3_kernel(int k, uint32_t mode, View2 v, TableView t) {
  q3_body<CANON, SHARED, false>(k, mode, v, t, P3ListsT<false>{});
}
// ... for leaves of few distinct k-mers (q3_body: T2L): a 1024-slot k-mer table, two workgroups per CU
constexpr int T2_LOG_SMALL = 10;
// ... and of up to a few thousand: a 2048-slot table, two workgroups per CU (q3_body: T2L)
template <bool CANON>
__global__ __launch_bounds__(Q3_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void msp2_p3_mid_kernel(int k, uint32_t mode, View2 v, TableView t) {
  q3_body<CANON, false, false, T2_LOG_MID>(k, mode, v, t, P3ListsT<false>{});
}
template <bool CANON>
__global__ __launch_bounds__(Q3_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void msp2_p3_small_kernel(int k, uint32_t mode, View2 v, TableView t) {
  q3_body<CANON, false, false, T2_LOG_SMALL>(k, mode, v, t, P3ListsT<false>{});
}
// the owner of the pipelined runs exchange: workgroup b counts local leaf lx.ll0 + b from the lists its ranks sent
template <bool CANON, bool SHARED>
__global__ __launch_bounds__(Q3_THREADS) void msp2_p3_lists_kernel(int k, View2 v, TableView t, P3ListsT<true> lx) {
  q3_body<CANON, SHARED, true>(k, Q3_WEIGHTED, v, t, lx);
}

// HUGE leaves (msp.hip: HUGE_LEAF -- 2^25 records or more, a single-key flood): not counted in LDS, where a run's
// 26-bit multiplicity or a 32-bit count could overflow, but k-mer by k-mer in the HBM table, whose adds saturate.
// Launched behind the leaf kernel; 256 workgroups look at the leaves' sizes and normally find nothing.
__global__ __launch_bounds__(256) void msp2_huge_leaves_kernel(int k, int canon, int weighted, uint32_t nl, View2 v, TableView t) {
  __shared__ uint32_t nhuge, huge[256];
  if (threadIdx.x == 0) nhuge = 0;
  __syncthreads();
  auto size_of = [&](uint32_t leaf, int cl) {
    return min((uint64_t)v.cnt2[NCLS * leaf + cl], v.exact ? (uint64_t)v.lcap[NCLS * leaf + cl] : (cl == 3 ? v.cap2c : v.cap2t));
  };
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;          // every thread looks at one leaf's sizes
  if (i0 < nl) {
    const uint32_t leaf = (i0 << v.sel_bits) | v.sel_val;
    if (size_of(leaf, 0) + size_of(leaf, 1) + size_of(leaf, 2) + size_of(leaf, 3) >= Q3_HUGE_LEAF) huge[atomicAdd(&nhuge, 1u)] = i0;
  }
  __syncthreads();
  for (uint32_t q = 0; q < nhuge; ++q) {
    const uint32_t i = huge[q], leaf = (i << v.sel_bits) | v.sel_val;
    const Rec2 *l0 = v.rec2 + (uint64_t)i * (v.cap2c + 3 * v.cap2t);
    for (int cl = 0; cl < NCLS; ++cl) {
      const Rec2 *src = v.exact ? v.rec2 + v.lbase[NCLS * leaf + cl] : (cl == 3 ? l0 : l0 + v.cap2c + (uint64_t)cl * v.cap2t);
      const uint64_t n = size_of(leaf, cl);
      for (uint64_t j = threadIdx.x; j < n; j += 256) {
        const Rec2 r = src[j];
        spill_record2(r.a, r.b, k, canon != 0, t, (weighted && cl == 3) ? (r.b.w >> 6) : 1u);
      }
    }
  }
}

// exact layout of the second level from the demand the first attempt counted (see msp.hip)
// bytes that are not a base (read terminators, N): one 64-bit sum (a batch that does not fit is looked
// at before it is planned again: every such byte ends up to k k-mers)
__global__ __launch_bounds__(256) void msp2_count_invalid_kernel(const int8_t *__restrict__ data, int64_t nN,
                                                                 unsigned long long *__restrict__ out) {
  unsigned long long mine = 0;
  const int64_t n16 = nN >> 4;
  const uint4 *d4 = reinterpret_cast<const uint4 *>(data);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const uint4 x = d4[i];
    // a byte is a base iff its top six bits are clear
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t b = w[c] & 0xFCFCFCFCu;
      b |= b >> 4; b |= b >> 2; b |= b >> 1;               // bit 0 of every byte: any of its top six bits set
      mine += (unsigned long long)__popc(b & 0x01010101u);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n16 << 4; i < nN; ++i) mine += ((uint8_t)data[i] > 3u) ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// (either level: base = exclusive prefix sum of the n cursors, cap = the cursors themselves)
// sum of n cursors (one workgroup): how many records the first chunk of a batch made
__global__ __launch_bounds__(1024) void msp2_sum_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *out) {
  __shared__ unsigned long long tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) mine += cnt[i];
  atomicAdd(&tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = tot; out[1] = out[2] = out[3] = out[4] = 0; }
}

// ... and how they split into the four classes of streams: the records of sub-region 0 of every level-1
// bin (1/64 of the chunk, every bin in it) -- out[1 + class] += records of that class (msp.hip)
__global__ __launch_bounds__(256) void msp2_class_sample_kernel(View2 v, unsigned long long *out) {
  __shared__ uint32_t c[NCLS];
  if (threadIdx.x < NCLS) c[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t reg = q1_reg(blockIdx.x, 0u);
  const uint32_t n = (uint32_t)min((uint64_t)v.cnt1[reg], v.cap1);
  const uint32_t *w = reinterpret_cast<const uint32_t *>(v.rec1 + (uint64_t)reg * v.cap1) + 7;
  for (uint32_t i = threadIdx.x; i < n; i += 256u) atomicAdd(&c[cls_of(w[8 * (size_t)i])], 1u);
  __syncthreads();
  if (threadIdx.x < NCLS && c[threadIdx.x]) atomicAdd(&out[1 + threadIdx.x], (unsigned long long)c[threadIdx.x]);
}

__global__ __launch_bounds__(1024) void msp2_layout_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ base,
                                                           uint32_t *__restrict__ cap) {
  __shared__ unsigned long long part[1024];
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t tid = threadIdx.x;
  unsigned long long s = 0;
  for (uint32_t i = 0; i < per; ++i) { const uint32_t l = tid * per + i; if (l < n) s += cnt[l]; }
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
    if (l < n) { const uint32_t c = cnt[l]; base[l] = run; cap[l] = c; run += c; }
  }
}

// the few records that did not fit their leaf stream: counted k-mer by k-mer in the HBM table
__global__ __launch_bounds__(256) void msp2_spill_list_kernel(const Rec2 *__restrict__ recs, uint32_t n, int k,
                                                              int canon, TableView t) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) spill_record2(recs[i].a, recs[i].b, k, canon != 0, t);
}

// One workgroup per leaf: add the `parts` lists of that leaf (the passes of a multi-pass add) in
// an LDS table, exactly like the leaf kernel does -- no HBM atomics.
__global__ __launch_bounds__(Q3_THREADS) void msp2_merge_kernel(const uint64_t *__restrict__ in_lo,
                                                                const uint64_t *__restrict__ in_hi,
                                                                const uint32_t *__restrict__ in_cnt,
                                                                const uint64_t *__restrict__ seg_off,
                                                                const uint32_t *__restrict__ seg_n, int parts,
                                                                int leaves_per_part, View2 v, TableView t) {
  __shared__ ulonglong2 keys[T2];
  __shared__ uint32_t cnts[T2];
  __shared__ uint32_t wg_total;
  __shared__ unsigned long long wg_base;
  const int tid = threadIdx.x;
  const uint32_t ll = blockIdx.x;
  uint32_t total = 0;
  for (int p = 0; p < parts; ++p) total += seg_n[(size_t)p * leaves_per_part + ll];
  if (total == 0) return;
  for (int s = tid; s < T2; s += Q3_THREADS) cnts[s] = 0;
  if (tid == 0) wg_total = 0;
  __syncthreads();
  for (int p = 0; p < parts; ++p) {
    const uint32_t n = seg_n[(size_t)p * leaves_per_part + ll];
    const uint64_t off = seg_off[(size_t)p * leaves_per_part + ll];
    for (uint32_t i = tid; i < ((n + 63u) & ~63u); i += Q3_THREADS) {
      const bool valid = i < n;
      const uint64_t lo = valid ? in_lo[off + i] : 0ull, hi = valid ? in_hi[off + i] : 0ull;
      const uint32_t c = valid ? min(in_cnt[off + i], CFRK_COUNT_MAX) : 0u;
      uint32_t h = t2_slot(lo, hi) | ((valid && c != 0u) ? 0u : T2_DONE);
      for (int it = 0; it < T2_TRIPS && __ballot((int32_t)h >= 0); ++it) t2_step<true>(keys, cnts, lo, hi, c, h);
      if ((int32_t)h >= 0) {
        t.stats[ST_SPILLED] = 1;
        dev_count_event(&t.stats[ST_AUX1]);
        table_add2(t, lo, hi, c);
      }
    }
  }
  __syncthreads();
  constexpr int NIT = T2 / Q3_THREADS;
  uint32_t wbase[NIT];
  auto occupied = [&](int s) { return cnts[s] != 0u; };
  wg_rank_slots<NIT, Q3_THREADS>(wbase, &wg_total, occupied);
  __syncthreads();
  if (tid == 0) wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)wg_total);
  __syncthreads();
  wg_emit_slots<NIT, Q3_THREADS>(wbase, wg_base, occupied, [&](int s, unsigned long long dst) {
    if (dst < v.out_cap) { v.out_lo[dst] = keys[s].x; v.out_hi[dst] = keys[s].y; v.out_cnt[dst] = cnts[s]; }
    else v.stats[ST_OVERFLOW] = 1;
  });
}

// ---------------------------------------------------------------------------- multi-GPU by runs
// The exchange of msp.hip ("multi-GPU by runs") for two-word keys: a rank that only partitions
// (CFRK_RUNS_ONLY) deduplicates every leaf's complete runs in place, turns read ends that are a
// prefix of one of them into 16-bit notes and ships, per leaf, [distinct runs with multiplicities]
// [truncated runs][notes]; a 32-byte record travels as two 16-byte rows.  The owner lines the ranks'
// lists up as the streams of its leaves and runs the leaf kernel in Q3_WEIGHTED mode.
constexpr int DX2_INFL = 2;
// A note is 16 bits: position of the twin in the leaf's list << 5 | n-1 -- 11 bits of position
constexpr uint32_t NOTE_POS_MAX = 2047u;
static_assert((NOTE_POS_MAX << 5 | 29u) < 0xFFFFu, "a note never equals the padding value");

// stream cl (0..2 truncated, 3 complete) of a leaf and how many records it holds (one pass: sel_bits = 0)
__device__ __forceinline__ Rec2 *x2_stream(const View2 &v, uint32_t leaf, int cl) {
  if (v.exact) return v.rec2 + v.lbase[NCLS * leaf + cl];
  return v.rec2 + (uint64_t)leaf * (v.cap2c + 3 * v.cap2t) + (cl == 3 ? 0ull : v.cap2c + (uint64_t)cl * v.cap2t);
}
__device__ __forceinline__ uint32_t x2_count(const View2 &v, uint32_t leaf, int cl) {
  return (uint32_t)min((uint64_t)v.cnt2[NCLS * leaf + cl], v.exact ? (uint64_t)v.lcap[NCLS * leaf + cl] : (cl == 3 ? v.cap2c : v.cap2t));
}

// One workgroup per leaf.  The leaf's complete stream -> its DISTINCT runs, header = multiplicity << 6 |
// n-1 (the extra minimizer-hash bits in b.z stay), at the head of the stream; leaf_n[leaf] = how many.
// A leaf with more distinct runs than the table holds leaves as it is (multiplicity 1 each).
// Truncated runs that are a prefix of a distinct run of this rank (a suffix, read on the other strand)
// become notes: marked in place (b.w = RUN_NOTED, a.x = position of the run in the list << 5 | n-1);
// leaf_off[leaf] = how many.
// RX_LOG / DX2_THREADS: 2048 slots (64 KB of LDS) and 256 threads, two workgroups per CU, for leaves of a few
// hundred distinct runs; 4096 slots (128 KB) and 1024 threads, one per CU, for the leaves of a job with far
// more distinct k-mers than its leaf tables hold (configs[4]: ~2000 distinct runs per leaf and rank) --
// with the small table those leaves left undeduplicated: 53 GB per rank instead of 5.
template <int RX_LOG, int DX2_THREADS>
__global__ __launch_bounds__(DX2_THREADS) void msp2_dedupe_export_kernel(int k, int canon, View2 v) {
  constexpr int RX = 1 << RX_LOG;
  // the record table, SPLIT (r2s_insert_loop): slot s = {ra[s], {rb[s].x, rb[s].y, rz[s], rst[s]}}
  __shared__ uint4 ra[RX];
  __shared__ uint2 rb[RX];
  __shared__ uint32_t rz[RX], rst[RX];
  auto rt_get = [&](uint32_t s_) { const uint2 e = rb[s_]; return Rec2{ra[s_], make_uint4(e.x, e.y, rz[s_], rst[s_])}; };
  __shared__ uint16_t sidx[RX];                    // record-table slot -> position in the leaf's list
  __shared__ uint32_t wsum[DX2_THREADS / 64];
  __shared__ uint32_t rt_fail, noted;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t leaf = blockIdx.x;
  const uint32_t n1 = x2_count(v, leaf, 3);
  if (n1 == 0) return;
  Rec2 *const stream = x2_stream(v, leaf, 3);
  const Rec2 zrec = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  for (int s = tid; s < RX; s += DX2_THREADS) rst[s] = R2_EMPTY;
  // (a leaf of 2^19 complete runs or more leaves undeduplicated: msp.hip, HUGE_LEAF)
  const bool too_many = (uint64_t)n1 >= Q3_HUGE_LEAF_SENDER;
  if (tid == 0) { rt_fail = ((v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW) || too_many) ? 1u : 0u; noted = 0u; }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < n1 && !too_many; r0 += (uint32_t)DX2_INFL * DX2_THREADS) {
    Rec2 recs[DX2_INFL];
#pragma unroll
    for (int u = 0; u < DX2_INFL; ++u) {
      const uint32_t r = r0 + (uint32_t)u * DX2_THREADS + tid;
      recs[u] = zrec;
      if (r < n1) recs[u] = stream[r];
    }
#pragma unroll
    for (int u = 0; u < DX2_INFL; ++u) {
      const uint32_t r = r0 + (uint32_t)u * DX2_THREADS + tid;
      uint32_t h = r2_slot_k(recs[u], k, RX_LOG) | ((r < n1) ? 0u : R2_DONE);
      r2s_insert_loop(ra, rb, rz, rst, recs[u], h, RX - 1);
      if ((int32_t)h >= 0) rt_fail = 1u;
    }
  }
  __syncthreads();
  uint32_t nd;
  if (rt_fail) {
    // (every record was read before the barrier; the rewrite touches the header word only)
    for (uint32_t i = tid; i < n1; i += DX2_THREADS) stream[i].b.w = (1u << 6) | (stream[i].b.w & 63u);
    nd = n1;
  } else {
    // occupied slots -> head of the stream (eight slots per thread)
    constexpr int PER = RX / DX2_THREADS;
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mine += (rst[PER * tid + i] != R2_EMPTY) ? 1u : 0u;
    const uint32_t incl = dev_wave_scan_incl(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < DX2_THREADS / 64; ++w) { const uint32_t x = wsum[w]; base += (w < wave) ? x : 0u; total += x; }
    uint32_t at = base + incl - mine;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const Rec2 e = rt_get(PER * tid + i);
      sidx[PER * tid + i] = (uint16_t)at;
      if (e.b.w != R2_EMPTY) stream[at++] = e;
    }
    nd = total;
    __syncthreads();
    // truncated runs -> notes (the lookup of the leaf kernel's anchoring, msp2_p3_kernel)
    for (int cl = 0; cl < 3 && !(v.dbg & CFRK_DEBUG_NO_ANCHORS); ++cl) {
      const uint32_t nt = x2_count(v, leaf, cl);
      Rec2 *const trunc = x2_stream(v, leaf, cl);
      for (uint32_t g0 = 0; g0 < nt; g0 += DX2_THREADS) {
        const uint32_t g = g0 + tid;
        const bool valid = g < nt;
        Rec2 rec = zrec;
        if (valid) rec = trunc[g];
        const uint32_t nm1 = rec.b.w & 31u;
        const bool lc = (rec.b.w & 64u) != 0u, rc_ = (rec.b.w & 128u) != 0u;
        const bool suf = canon && valid && !lc && rc_;
        if (suf) rec = revcomp_record2(rec, (int)nm1 + k);
        const bool anchored = suf || (valid && lc && !rc_);
        uint32_t h = anchored ? r2_slot_k(rec, k, RX_LOG) : R2_DONE;
        uint32_t found = 0xFFFFFFFFu;
        for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
          const bool p = (int32_t)h >= 0;
          const uint32_t hh = h & (uint32_t)(RX - 1);
          const Rec2 e2 = rt_get(hh);
          const bool empty = e2.b.w == R2_EMPTY;
          const bool hit = p && !empty && (e2.b.w & 31u) >= nm1 && rec2_prefix_equal(e2, rec, (int)nm1 + k);
          found = hit ? hh : found;
          h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(RX - 1)) : (h | R2_DONE);
        }
        // (a twin beyond position 2047 of a long list cannot be named by a note: the run travels as a record;
        //  CFRK_DEBUG_SMALL_WAVE_CAP lowers the limit to 15 so that tests reach this at small sizes)
        const uint32_t pos_max = (v.dbg & CFRK_DEBUG_SMALL_WAVE_CAP) ? 15u : NOTE_POS_MAX;
        const bool hit = found != 0xFFFFFFFFu && (uint32_t)sidx[found] <= pos_max;
        if (hit) { trunc[g].a.x = ((uint32_t)sidx[found] << 5) | nm1; trunc[g].b.w = RUN_NOTED; }
        const unsigned long long hb = __ballot(hit);
        if (lane == 0 && hb) atomicAdd(&noted, (uint32_t)__popcll(hb));
      }
    }
    __syncthreads();
  }
  if (tid == 0) { v.leaf_n[leaf] = nd; v.leaf_off[leaf] = noted; }
}

// ------------------------------------------------------------------- multi-GPU by runs, PIPELINED (round 5)
// msp.hip's msp_dedupe_send_kernel for 32-byte records: one workgroup per leaf of a group deduplicates the leaf's
// complete runs in LDS, looks its read ends up, CLAIMS the leaf's rows in its owner's segment of the send buffer (one
// atomic) and writes [distinct runs, two rows each][truncated runs without a twin][notes] there -- the leaf streams are
// not rewritten, nothing is planned or gathered afterwards, the host is not asked.  (Layout: msp_runs.h.)
// DS2_LOG: 1024 slots (32 KB of LDS, four workgroups per CU: a shard's leaf is ~1000 records and half a dozen dependent
// round trips -- loads, a claim, a second look at the read ends -- and only other workgroups hide them) for leaves of a few
// hundred distinct runs, 2048 slots (two per CU) beyond.  (A home-slot fast path with the leftovers compacted across the
// wave, as in the leaf kernel, was measured SLOWER here, 2.47 against 2.10 ms: three records per thread do not pay for it.)
// SUB: the job's leaves are shared by sub-value (configs[4]-sized hints): the records' spare word b.z (extra minimizer-hash
// bits) is kept and travels; such leaves hold ~2000 distinct runs per rank: 4096 slots, 1024 threads, one workgroup per CU
// (msp2_dedupe_export_kernel's shapes).
// DS2_TCAP: read ends of a leaf that may become notes (the others travel as records: 32 bytes instead of 2) -- 2048 where
// a rank's leaf holds a few hundred, 6144 for the shared leaves of a configs[4]-sized job (~3800)
constexpr int DS2_INFL = 4;
template <int DS2_LOG, int DS2_THREADS, bool SUB, int DS2_TCAP = (SUB ? 6144 : 2048)>
__global__ __launch_bounds__(DS2_THREADS) void msp2_dedupe_send_kernel(int k, int canon, View2 v, RunsSend sg) {
  constexpr int RX = 1 << DS2_LOG;
  // SPLIT record table (r2s_insert_loop): bases 0..63, bases 64..95 and the state words are three arrays
  __shared__ uint4 ra[RX];
  __shared__ uint2 rb[RX];
  __shared__ uint32_t rst[RX];
  __shared__ uint32_t rz_[SUB ? RX : 1];
  uint32_t *const rz = SUB ? rz_ : nullptr;
  __shared__ uint16_t sidx[RX];                    // record-table slot -> position in the leaf's list
  __shared__ uint16_t tres[DS2_TCAP];              // truncated run g: its note, or 0xFFFF = travels as a record
  __shared__ uint32_t wsum[DS2_THREADS / 64];
  __shared__ uint32_t rt_fail, noted, cu, cn, row0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t leaf = sg.leaf0 + blockIdx.x;
  const uint32_t own = leaf % (uint32_t)sg.parts, ll = leaf / (uint32_t)sg.parts;
  uint4 *const seg = sg.packed + (uint64_t)own * sg.seg_cap;
  const uint32_t hrows = 1u + sg.lcount;
  uint4 *const entry = seg + 1u + (ll - sg.ll0);
  const uint32_t n1 = x2_count(v, leaf, 3);
  const uint32_t t0 = x2_count(v, leaf, 0);
  const uint64_t t1 = (uint64_t)t0 + x2_count(v, leaf, 1), nt64 = t1 + x2_count(v, leaf, 2);
  if ((uint64_t)n1 + nt64 == 0) { if (tid == 0) *entry = make_uint4(0u, 0u, 0u, 0u); return; }
  const Rec2 *const c3 = x2_stream(v, leaf, 3);
  const Rec2 *const s0 = x2_stream(v, leaf, 0), *const s1 = x2_stream(v, leaf, 1), *const s2 = x2_stream(v, leaf, 2);
  auto trunc_at = [&](uint64_t g) { return (g < t0) ? s0 + g : (g < t1) ? s1 + (g - t0) : s2 + (g - t1); };
  const Rec2 zrec = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  // (a leaf of 2^19 complete runs or more leaves undeduplicated: msp.hip, HUGE_LEAF; a row count beyond 32 bits cannot be claimed)
  const bool too_many = (uint64_t)n1 >= Q3_HUGE_LEAF_SENDER;
  const bool unclaimable = 2ull * ((uint64_t)n1 + nt64) >= 0xFFFFFFF0ull;
  // the first round of the complete stream is asked for before anything else is done
  Rec2 recs[DS2_INFL];
#pragma unroll
  for (int u = 0; u < DS2_INFL; ++u) {
    const uint32_t r = (uint32_t)u * DS2_THREADS + tid;
    recs[u] = zrec;
    if (r < n1 && !too_many) recs[u] = c3[r];
  }
  for (int s_ = tid; s_ < RX; s_ += DS2_THREADS) rst[s_] = R2_EMPTY;
  if (tid == 0) { rt_fail = ((v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW) || too_many) ? 1u : 0u; noted = 0u; cu = 0u; cn = 0u; }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < n1 && !too_many; r0 += (uint32_t)DS2_INFL * DS2_THREADS) {
    if (r0) {
#pragma unroll
      for (int u = 0; u < DS2_INFL; ++u) {
        const uint32_t r = r0 + (uint32_t)u * DS2_THREADS + tid;
        recs[u] = zrec;
        if (r < n1) recs[u] = c3[r];
      }
    }
#pragma unroll
    for (int u = 0; u < DS2_INFL; ++u) {
      const uint32_t r = r0 + (uint32_t)u * DS2_THREADS + tid;
      if (r0 + (uint32_t)u * DS2_THREADS >= n1) break;            // (wave-uniform)
      uint32_t h = r2_slot_k(recs[u], k, DS2_LOG) | ((r < n1) ? 0u : R2_DONE);
      r2s_insert_loop(ra, rb, rz, rst, recs[u], h, RX - 1);
      if ((int32_t)h >= 0) rt_fail = 1u;
    }
  }
  __syncthreads();
  const bool plain = rt_fail != 0u;                 // no deduplication: every complete run leaves with multiplicity 1, no notes
  const uint32_t nt = (uint32_t)min(nt64, (uint64_t)0xFFFFFFFFull);
  uint32_t nd, at0 = 0;
  constexpr int PER = RX / DS2_THREADS;
  if (plain) {
    nd = n1;
  } else {
    // occupied slots -> positions in the leaf's list (eight slots per thread)
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mine += (rst[PER * tid + i] != R2_EMPTY) ? 1u : 0u;
    const uint32_t incl = dev_wave_scan_incl(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < DS2_THREADS / 64; ++w) { const uint32_t x = wsum[w]; base += (w < wave) ? x : 0u; total += x; }
    at0 = base + incl - mine;
    uint32_t at = at0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      sidx[PER * tid + i] = (uint16_t)at;
      if (rst[PER * tid + i] != R2_EMPTY) ++at;
    }
    nd = total;
    __syncthreads();
    // truncated runs: which of the first DS2_TCAP are a prefix of a distinct complete run of this rank (a suffix, read on
    // the other strand)?  The verdict waits in LDS (the lookup of the leaf kernel's anchoring)
    const uint32_t tlook = (v.dbg & CFRK_DEBUG_NO_ANCHORS) ? 0u : min(nt, (uint32_t)DS2_TCAP);
    const uint32_t pos_max = (v.dbg & CFRK_DEBUG_SMALL_WAVE_CAP) ? 15u : NOTE_POS_MAX;
    for (uint32_t g0 = 0; g0 < tlook; g0 += DS2_THREADS) {
      const uint32_t g = g0 + tid;
      const bool valid = g < tlook;
      Rec2 rec = zrec;
      if (valid) rec = *trunc_at(g);
      const uint32_t nm1 = rec.b.w & 31u;
      const bool lc = (rec.b.w & 64u) != 0u, rc_ = (rec.b.w & 128u) != 0u;
      const bool suf = canon && valid && !lc && rc_;
      if (suf) rec = revcomp_record2(rec, (int)nm1 + k);
      const bool anchored = suf || (valid && lc && !rc_);
      uint32_t h = anchored ? r2_slot_k(rec, k, DS2_LOG) : R2_DONE;
      uint32_t found = 0xFFFFFFFFu;
      for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
        const bool p = (int32_t)h >= 0;
        const uint32_t hh = h & (uint32_t)(RX - 1);
        const uint2 eb2 = rb[hh];
        const Rec2 e2 = {ra[hh], make_uint4(eb2.x, eb2.y, 0u, rst[hh])};      // (the spare word takes no part in the comparison)
        const bool empty = e2.b.w == R2_EMPTY;
        const bool hit = p && !empty && (e2.b.w & 31u) >= nm1 && rec2_prefix_equal(e2, rec, (int)nm1 + k);
        found = hit ? hh : found;
        h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(RX - 1)) : (h | R2_DONE);
      }
      // (a twin beyond position 2047 of a long list cannot be named by a note: the run travels as a record)
      const bool hit = found != 0xFFFFFFFFu && (uint32_t)sidx[found & (uint32_t)(RX - 1)] <= pos_max;
      if (valid) tres[g] = hit ? (uint16_t)(((uint32_t)sidx[found] << 5) | nm1) : (uint16_t)0xFFFFu;
      const unsigned long long hb = __ballot(hit);
      if (lane == 0 && hb) atomicAdd(&noted, (uint32_t)__popcll(hb));
    }
    __syncthreads();
  }
  const uint32_t na = plain ? 0u : noted, nu = nt - na;
  const uint64_t rows = 2ull * ((uint64_t)nd + nu) + (na + NOTES_PER_ROW - 1) / NOTES_PER_ROW;
  if (tid == 0) {
    // one claim per leaf; a segment that runs out of room (or a flood that cannot be claimed at all) shows in its
    // cursor -- used rows > seg_cap -- and the host takes the classic exchange instead
    const uint32_t claim = unclaimable ? 0xFFFFFFFFu : (uint32_t)rows;
    const uint32_t pos = atomicAdd(&sg.cursor[own], claim);
    const bool fits = !unclaimable && (uint64_t)pos + rows <= sg.seg_cap - hrows && pos + claim >= pos;
    if (!fits) atomicMax(&sg.cursor[own], 0xFFFFFFF0u);              // (stays "too many" whatever is added later)
    *entry = fits ? make_uint4(pos, nd, nu, na) : make_uint4(0u, 0u, 0u, 0u);
    row0 = fits ? pos : 0xFFFFFFFFu;
  }
  __syncthreads();
  if (row0 == 0xFFFFFFFFu) return;
  uint4 *const dst = seg + hrows + row0;
  if (plain) {
    // (the halves as two vectors: a Rec2 copied as a whole goes through scratch memory here)
    for (uint32_t i = tid; i < n1; i += DS2_THREADS) {
      const uint4 a = c3[i].a;
      uint4 b = c3[i].b;
      b.w = (1u << 6) | (b.w & 63u);
      dst[2 * (uint64_t)i] = a; dst[2 * (uint64_t)i + 1] = b;
    }
    uint4 *const dt = dst + 2 * (uint64_t)nd;
    for (uint32_t i = tid; i < nt; i += DS2_THREADS) {
      const Rec2 *q = trunc_at(i);
      const uint4 a = q->a, b = q->b;
      dt[2 * (uint64_t)i] = a; dt[2 * (uint64_t)i + 1] = b;
    }
    return;
  }
  {
    uint32_t at = at0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t sl = PER * tid + i, st = rst[sl];
      if (st != R2_EMPTY) { const uint2 eb2 = rb[sl]; dst[2 * at] = ra[sl]; dst[2 * at + 1] = make_uint4(eb2.x, eb2.y, SUB ? rz_[sl] : 0u, st); ++at; }
    }
  }
  uint4 *const dt = dst + 2 * (uint64_t)nd;
  uint16_t *const notes = reinterpret_cast<uint16_t *>(dt + 2 * (uint64_t)nu);
  for (uint32_t i = tid; i < ((nt + 63u) & ~63u); i += DS2_THREADS) {
    const bool valid = i < nt;
    const uint32_t note = (valid && i < (uint32_t)DS2_TCAP && !(v.dbg & CFRK_DEBUG_NO_ANCHORS)) ? (uint32_t)tres[i] : 0xFFFFu;
    const bool isn = valid && note != 0xFFFFu;
    const unsigned long long mn = __ballot(isn), mu = __ballot(valid && !isn);
    uint32_t bn = 0, bu = 0;
    if (lane == 0) {
      if (mn) bn = atomicAdd(&cn, (uint32_t)__popcll(mn));
      if (mu) bu = atomicAdd(&cu, (uint32_t)__popcll(mu));
    }
    bn = __shfl(bn, 0); bu = __shfl(bu, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (isn) { const uint32_t at = bn + (uint32_t)__popcll(mn & below); if (at < na) notes[at] = (uint16_t)note; }
    else if (valid) {
      const uint32_t at = bu + (uint32_t)__popcll(mu & below);
      if (at < nu) { const Rec2 *q = trunc_at(i); const uint4 a = q->a, b = q->b; dt[2 * (uint64_t)at] = a; dt[2 * (uint64_t)at + 1] = b; }
    }
  }
  const uint32_t pad = (NOTES_PER_ROW - na % NOTES_PER_ROW) % NOTES_PER_ROW;
  if ((uint32_t)tid < pad) notes[na + tid] = 0xFFFFu;
}

// sender: what every leaf contributes -- n1 distinct complete runs, nt truncated runs as records, na as
// notes, rows in all (two per record) -- one thread per leaf
__global__ __launch_bounds__(256) void msp2_runs_sizes_kernel(View2 v, uint4 *__restrict__ sz, unsigned long long *__restrict__ plan_sync) {
  const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
  if (leaf < 72u) plan_sync[leaf] = 0ull;                      // (the plan kernel's look-back words)
  if (leaf >= (uint32_t)NLEAF) return;
  uint32_t n1 = 0, na = 0;
  uint32_t nt = x2_count(v, leaf, 0) + x2_count(v, leaf, 1) + x2_count(v, leaf, 2);
  if (v.cnt2[NCLS * leaf + 3]) {                               // (a leaf without complete runs never wrote its counts)
    n1 = v.leaf_n[leaf];
    na = min((uint32_t)v.leaf_off[leaf], nt);
  }
  nt -= na;
  sz[leaf] = make_uint4(n1, nt, na, 2u * (n1 + nt) + (na + NOTES_PER_ROW - 1) / NOTES_PER_ROW);
}

// sender: leaf -> [nd distinct complete runs][nu truncated runs][na notes, 8 per row] at row dst_off[leaf]
// of the send buffer (the three truncated streams hold records and noted records mixed)
__global__ __launch_bounds__(256) void msp2_runs_gather_kernel(View2 v, const uint64_t *__restrict__ dst_off, uint4 *__restrict__ out,
                                                               const uint64_t *__restrict__ plan_rows, const uint64_t *__restrict__ seg_start, int parts, uint64_t cap_rows) {
  __shared__ uint32_t cu, cn;
  if (plan_rows[parts] > cap_rows) return;         // the buffer is too small: nothing was planned
  const uint32_t leaf = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has1 = v.cnt2[NCLS * leaf + 3] != 0u;
  const uint32_t nd = has1 ? v.leaf_n[leaf] : 0u;
  const uint32_t t0 = x2_count(v, leaf, 0), t1 = t0 + x2_count(v, leaf, 1), nt = t1 + x2_count(v, leaf, 2);
  const uint32_t na = has1 ? min((uint32_t)v.leaf_off[leaf], nt) : 0u;
  const uint32_t nu = nt - na;
  if (threadIdx.x == 0) runs_write_header(out, seg_start, parts, blockIdx.x, nd, nu, na);
  const Rec2 *c3 = x2_stream(v, leaf, 3);
  const Rec2 *s0 = x2_stream(v, leaf, 0), *s1 = x2_stream(v, leaf, 1), *s2 = x2_stream(v, leaf, 2);
  auto trunc_at = [&](uint32_t g) { return (g < t0) ? s0 + g : (g < t1) ? s1 + (g - t0) : s2 + (g - t1); };
  uint4 *dst = out + dst_off[leaf];
  for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) { const Rec2 r = c3[i]; dst[2 * i] = r.a; dst[2 * i + 1] = r.b; }
  uint4 *dt = dst + 2 * (uint64_t)nd;
  if (na == 0u) {
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) { const Rec2 r = *trunc_at(i); dt[2 * i] = r.a; dt[2 * i + 1] = r.b; }
    return;
  }
  if (threadIdx.x == 0) { cu = 0u; cn = 0u; }
  __syncthreads();
  uint16_t *notes = reinterpret_cast<uint16_t *>(dt + 2 * (uint64_t)nu);
  for (uint32_t i = threadIdx.x; i < ((nt + 63u) & ~63u); i += blockDim.x) {
    const bool valid = i < nt;
    Rec2 rec = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    if (valid) rec = *trunc_at(i);
    const bool isn = valid && rec.b.w == RUN_NOTED;
    const unsigned long long mn = __ballot(isn), mu = __ballot(valid && !isn);
    uint32_t bn = 0, bu = 0;
    if (lane == 0) {
      if (mn) bn = atomicAdd(&cn, (uint32_t)__popcll(mn));
      if (mu) bu = atomicAdd(&cu, (uint32_t)__popcll(mu));
    }
    bn = __shfl(bn, 0); bu = __shfl(bu, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    // (the counts of the plan bound both: a stream that changed under us cannot write outside the leaf's rows)
    if (isn) { const uint32_t at = bn + (uint32_t)__popcll(mn & below); if (at < na) notes[at] = (uint16_t)rec.a.x; }
    else if (valid) { const uint32_t at = bu + (uint32_t)__popcll(mu & below); if (at < nu) { dt[2 * at] = rec.a; dt[2 * at + 1] = rec.b; } }
  }
  const uint32_t pad = (NOTES_PER_ROW - na % NOTES_PER_ROW) % NOTES_PER_ROW;
  if (threadIdx.x < pad) notes[na + threadIdx.x] = 0xFFFFu;
}

// owner: segment (source rank, local leaf) of the received buffer -> its place in the leaf's complete
// stream and its (one) stream of truncated runs.  A note becomes the run it stands for: the first n
// k-mers of its twin, closed on the left only.
__global__ __launch_bounds__(256) void msp2_runs_scatter_kernel(const uint4 *__restrict__ in, RunsRecv rr, int lpp, int k,
                                                                const uint64_t *__restrict__ src_off,
                                                                const uint64_t *__restrict__ dst1, const uint64_t *__restrict__ dst0,
                                                                Rec2 *__restrict__ rec2) {
  const uint32_t seg = blockIdx.x;
  const uint32_t r = seg / (uint32_t)lpp, ll = seg - r * (uint32_t)lpp;
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(in + rr.rstart[r]);
  const uint32_t nd = hdr[3 * ll], nt = hdr[3 * ll + 1], na = hdr[3 * ll + 2];
  const uint4 *src = in + src_off[seg];
  for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) { Rec2 q; q.a = src[2 * i]; q.b = src[2 * i + 1]; rec2[dst1[seg] + i] = q; }
  const uint4 *st = src + 2 * (uint64_t)nd;
  for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) { Rec2 q; q.a = st[2 * i]; q.b = st[2 * i + 1]; rec2[dst0[seg] + i] = q; }
  const uint16_t *notes = reinterpret_cast<const uint16_t *>(st + 2 * (uint64_t)nt);
  for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) {          // (nd > 0: the layout kernel checked)
    const uint32_t note = notes[i];
    const uint32_t ti = min(note >> 5, nd - 1u);                      // a position outside the list is not followed
    const uint4 ta = src[2 * ti], tb = src[2 * ti + 1];
    const uint32_t nm1 = min(note & 31u, tb.w & 31u);
    const int len = (int)nm1 + k;                                     // bases of the run: 33 .. 93
    auto mk = [&](int w) {                                            // mask of word w: bits 32 w .. 32 w + 31 of the string
      const int b = 2 * len - 32 * w;
      return (b >= 32) ? 0xFFFFFFFFu : ((b <= 0) ? 0u : ~(0xFFFFFFFFu >> b));
    };
    Rec2 q;
    q.a = make_uint4(ta.x, ta.y, ta.z & mk(2), ta.w & mk(3));
    q.b = make_uint4(tb.x & mk(4), tb.y & mk(5), tb.z, 64u | nm1);
    rec2[dst0[seg] + nt + i] = q;
  }
}

}  // namespace

bool cfrk_msp2_usable(const cfrk_ctx *ctx) {
  return ctx->g_two && ctx->g_k >= 33 && ctx->g_k <= 64 && !(ctx->g_flags & CFRK_FORCE_HASH);
}

// expected records per input byte: one per minimizer change (2 / (W + 1)) plus read ends, scaled by the
// measured share of positions that start a k-mer when the batch had to be looked at (dens_scale)
static double msp2_density(const cfrk_ctx *ctx) {
  const int W2 = msp2_window(ctx->g_k);
  const double sc = (ctx->msp && ctx->msp->dens_scale > 0.0) ? ctx->msp->dens_scale : 1.0;
  return (2.0 / (W2 + 1)) * sc + 1.0 / 64.0;
}

// ... with the leaf streams sized from a counting pass (at most one record per expected record)
static size_t msp2_need_lean(const cfrk_ctx *ctx, int64_t span) {
  const double expect = (double)span * msp2_density(ctx);
  const uint64_t cap1 = (uint64_t)(expect / (B1 * NXG) * 1.35) + 2048;
  return (size_t)B1 * NXG * cap1 * 32 + (size_t)(expect * 32) + (size_t)ctx->g_cap * 20;
}

// ... counted in chunks (one level-1 buffer of at most ~6 GB, leaf streams sized by class from the first
// chunk's records: ~1.4 x the records the batch really makes, msp2_count_tiles; the density estimate behind
// `expect` is itself an upper bound, and an attempt that runs out of memory is planned again the old way)
static size_t msp2_need_chunked(const cfrk_ctx *ctx, int64_t span) {
  const double expect = (double)span * msp2_density(ctx);
  const uint64_t cap1 = (uint64_t)(expect / (B1 * NXG) * 1.35) + 2048;
  const double l1 = std::min((double)B1 * NXG * cap1 * 32.0, 6e9 * 1.35 + (double)B1 * NXG * 2048 * 32.0);
  return (size_t)l1 + (size_t)(expect * 1.6 * 32) + (size_t)NLEAF * 1280 * 32 + (size_t)ctx->g_cap * 20;
}

static size_t msp2_need(const cfrk_ctx *ctx, int64_t span) {
  const double expect = (double)span * msp2_density(ctx);
  const uint64_t cap1 = (uint64_t)(expect / (B1 * NXG) * 1.35) + 2048;
  const uint64_t cap2c = (uint64_t)(expect / NLEAF * 2.1) + 96;
  const uint64_t cap2t = (uint64_t)(expect / NLEAF * 0.4) + 96;
  return (size_t)B1 * NXG * cap1 * 32 + (size_t)NLEAF * (cap2c + 3 * cap2t) * 32 + (size_t)ctx->g_cap * 20;
}

// Sub-values per workgroup of a shared leaf, as a power of two: four (one pass over the leaf's streams for
// four sub-values: the leaf is read by 2^(sub_bits - 2) workgroups only) unless the distinct runs of four
// sub-values would crowd the 1024-entry record table -- short windows make more runs per k-mer: at k = 33
// (window 18) a leaf of configs[4]'s shape holds ~3400 distinct runs, 860 per four of its sixteen
// sub-values, the table overflowed in most workgroups and the leaf was counted without deduplication in
// key-subset passes (1.15 s for the shard that takes 0.17 s at k = 63).
static uint32_t msp2_hbits(const cfrk_ctx *ctx, uint32_t sub_bits, uint64_t per_leaf_cap) {
  double runs_per_leaf = (double)per_leaf_cap / 2.0 * 4.0 / (double)(msp2_window(ctx->g_k) + 1);   // (table load 0.5; both strands)
  // even k takes its minimizers from 13-mers: 3.4e7 canonical values, each at ~30 loci of a 10^9-base genome,
  // all of them in one (leaf, sub-value) -- the loads of the sub-values scatter twice as much as with 14-mers
  // (measured on configs[4]'s shape, four / two / one sub-values per workgroup: k = 63 169 / 172 / 210 ms,
  //  k = 48 366 / 215 / 248, k = 40 581 / 220 / 257, k = 33 1133 / 283 / 335)
  if (!(ctx->g_k & 1)) runs_per_leaf *= 1.3;
  uint32_t hbits = std::min(sub_bits, 2u);
  while (hbits > 0 && runs_per_leaf * (double)(1u << hbits) / (double)(1u << sub_bits) > 640.0) --hbits;
  if (ctx->dbg_param[CFRK_PARAM_MSP2_SUBVALUE_BITS] > 0) hbits = std::min<uint32_t>(sub_bits, (uint32_t)ctx->dbg_param[CFRK_PARAM_MSP2_SUBVALUE_BITS] - 1u);   // (tests)
  return hbits;
}

// one pass of Q1 -> Q2 -> Q3 over the Q1 tiles [tile0, tile0 + ntiles)
// slack >= 1 widens the per-leaf streams beyond what msp2_need() accounts for (memory permitting)
// sel_bits / sel_val / first: the leaf subset of this pass (msp.hip: msp_count_tiles)
// lean: the leaf streams get no room up front; the second-level kernel first only COUNTS them, they are
// laid out back to back with exactly that room, and the kernel runs again (a batch that would need
// more passes over the input otherwise: one more read of the level-1 records is cheaper than a pass)
// deduplicate every leaf's complete stream where it lies (the one-shot runs export reads the result)
static int msp2_dedupe_in_place(cfrk_ctx *ctx, const View2 &v) {
  const int k = ctx->g_k, canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  const int W2 = msp2_window(k);
  const double runs_per_leaf = (double)(ctx->g_cap / NLEAF) / 2.0 * 4.0 / (double)(W2 + 1);
  if (runs_per_leaf > 700.0 || (ctx->dbg_flags & CFRK_DEBUG_RECORD_SUBSETS))
    hipLaunchKernelGGL((msp2_dedupe_export_kernel<12, 1024>), dim3(NLEAF), dim3(1024), 0, ctx->stream, k, canon, v);
  else
    hipLaunchKernelGGL((msp2_dedupe_export_kernel<11, 256>), dim3(NLEAF), dim3(256), 0, ctx->stream, k, canon, v);
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}

static int msp2_count_tiles(cfrk_ctx *ctx, cfrk_msp *ms, const int8_t *d_data, int64_t nN, int64_t tile0,
                            int64_t ntiles, double slack, int sel_bits = 0, uint32_t sel_val = 0, bool first = true,
                            bool lean = false, bool chunked = false) {
  int rc;
  const int k = ctx->g_k;
  const int W2 = msp2_window(k);
  const int m = (k & 1) ? 14 : 13;                       // k - m + 1 - W2 must be even
  const int c = (k - m + 1 - W2) / 2;                    // 1 for k < 44, 1..11 (k = 44..64) with the window of 30
  const int canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  const int64_t span = std::min(nN + 32, ntiles * (int64_t)Q1_WAVES * Q1_OWN * 32);

  const double dens = msp2_density(ctx);
  const double expect_all = (double)span * dens;                 // records of the whole batch
  const double expect = expect_all / (double)(1u << sel_bits);   // ... of this pass
  const uint64_t cap1 = (uint64_t)(expect / (B1 * NXG) * 1.35) + 2048;
  // (a leaf of the pass holds ALL its records: the pass has fewer leaves, not lighter ones)
  const uint64_t cap2c = lean ? 0 : (uint64_t)(expect_all / NLEAF * 2.1 * slack) + 96;
  const uint64_t cap2t = lean ? 0 : (uint64_t)(expect_all / NLEAF * 0.4 * slack) + 96;
  const int64_t tiles_per_sub = (int64_t)(((uint64_t)NXG * cap1 + (uint64_t)Q2_TILE * Q2_GROUP - 1) / ((uint64_t)Q2_TILE * Q2_GROUP));   // tile groups per bin
  if (tiles_per_sub * B1 > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
  // chunked (msp.hip: msp_count_tiles): Q1(c), Q2(c) chunk after chunk on ONE level-1 buffer that holds a
  // chunk's records, leaf streams sized from what the first chunk really made; any overflow beyond the
  // parking buffers returns CFRK_ERR_SMALL_BUF and the caller plans the batch the old way
  const bool small_pipe = (ctx->dbg_flags & CFRK_DEBUG_SMALL_PIPELINE) != 0;
  int nchunks = 1;
  if (chunked) nchunks = (int)std::min<int64_t>(small_pipe ? 5 : (int64_t)((double)B1 * NXG * cap1 * 32.0 / 6e9) + 1, ntiles / (small_pipe ? 3 : 8192));
  if (nchunks < 2) { nchunks = 1; chunked = false; }
  const int64_t chunk_tiles = (ntiles + nchunks - 1) / nchunks;
  const uint64_t cap1c = chunked ? (uint64_t)(expect * ((double)chunk_tiles / (double)ntiles) / (B1 * NXG) * 1.35) + 2048 : cap1;
  void *p;
  View2 v;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, (size_t)B1 * NXG * cap1c * sizeof(Rec2), &p))) return rc;
  v.rec1 = (Rec2 *)p; v.cap1 = cap1c;
  v.rec2 = nullptr;
  if (!lean && !chunked) {
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(NLEAF >> sel_bits) * (cap2c + 3 * cap2t) * sizeof(Rec2), &p))) return rc;
    v.rec2 = (Rec2 *)p;
  }
  v.cap2c = cap2c; v.cap2t = cap2t; v.count_only = lean ? 1u : 0u;
  v.sel_mask = (1u << sel_bits) - 1u; v.sel_val = sel_val; v.sel_bits = (uint32_t)sel_bits;
  // far more distinct k-mers expected than the leaf tables hold (65536 x ~2500): records carry extra
  // minimizer-hash bits and 2^sub_bits workgroups share a leaf (~2000 distinct k-mers each)
  uint32_t sub_bits = 0;
  while (sub_bits < (uint32_t)SUB_BITS && ((ctx->g_cap / NLEAF) >> sub_bits) > 2048u) ++sub_bits;
  if (ctx->g_cap / NLEAF <= 4096u) sub_bits = 0;
  if ((ctx->dbg_flags & CFRK_DEBUG_RECORD_SUBSETS) && sub_bits < 2u) sub_bits = 2u;
  const bool sub = sub_bits != 0u;
  v.sub_bits = sub_bits;
  v.hbits = msp2_hbits(ctx, sub_bits, ctx->g_cap / NLEAF);
  // (leaf index: one entry per leaf, or per (leaf, sub-value) when leaves are shared)
  const size_t nseg = (size_t)NLEAF << sub_bits;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, nseg * 8 + ((size_t)B1 * NXG + (size_t)NCLS * NLEAF + nseg) * sizeof(uint32_t), &p))) return rc;
  v.leaf_off = (uint64_t *)p;
  v.cnt1 = (uint32_t *)(v.leaf_off + nseg); v.cnt2 = v.cnt1 + B1 * NXG; v.leaf_n = v.cnt2 + NCLS * NLEAF;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_lo = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTH, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_hi = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats; v.dbg = ctx->dbg_flags;
  TableView t = cfrk_table_view(ctx);

  // (cnt1, cnt2 and -- first pass only -- the leaf index and the list cursor)
  HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, ((size_t)B1 * NXG + (size_t)NCLS * NLEAF + (first ? nseg : 0)) * sizeof(uint32_t), ctx->stream));
  if (first) HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
  // Both levels are laid out for an input that spreads evenly over the minimizer space; one that does
  // not overflows its regions.  The cursors keep counting past the capacity, so after Q2 the host
  // knows the exact demand of both levels (one D2H + stream sync per add): a few overflowing records
  // were parked and are counted through the HBM table, more than that and the level is laid out again
  // back to back with exactly the room each region needs, and its kernel runs again (msp.hip).
  constexpr uint32_t OVF_CAP = 1u << 19;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OVF, (size_t)OVF_CAP * sizeof(Rec2), &p))) return rc;
  v.ovf = (Rec2 *)p; v.ovf_cap = (uint32_t)std::min<double>((double)OVF_CAP, expect / 256.0);
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OVF1, (size_t)OVF_CAP * sizeof(Rec2), &p))) return rc;
  v.ovf1 = (Rec2 *)p; v.ovf1_cap = v.ovf_cap;
  // a job that only partitions (CFRK_RUNS_ONLY) has no table of its own for parked records: any
  // overflow goes straight to the exact layout
  const bool runs_only = (ctx->g_flags & CFRK_RUNS_ONLY) != 0;
  if (runs_only) v.ovf_cap = v.ovf1_cap = 0;
  // CFRK_RUNS_DEFER (msp.hip): no deduplication here -- the pipelined export deduplicates group by group -- and, when the
  // streams have their fixed stride (not a chunked or a count-first add, whose layouts need a read-back), the add ends behind
  // Q2 unsynchronised
  const bool nodedupe = runs_only && (ctx->g_flags & CFRK_RUNS_DEFER);
  const bool defer = nodedupe && !chunked && !lean;
  v.exact = 0; v.lbase = nullptr; v.lcap = nullptr;
  v.exact1 = 0; v.rbase = nullptr; v.rcap = nullptr;
  const size_t nreg = (size_t)B1 * NXG;
  int64_t q2_groups = chunked ? (int64_t)(((uint64_t)NXG * cap1c + (uint64_t)Q2_TILE * Q2_GROUP - 1) / ((uint64_t)Q2_TILE * Q2_GROUP))
                              : tiles_per_sub;             // tile groups per bin Q2 is launched with
  auto launch_q1 = [&](int64_t t0, int64_t t1) -> int {
    const dim3 g1((unsigned)(t1 - t0)), b1(Q1_THREADS);
#define CFRK_Q1_CASE(WW) \
    case WW: \
      if (sub) hipLaunchKernelGGL((msp2_p1_kernel<WW, true>), g1, b1, 0, ctx->stream, d_data, nN, k, m, c, canon, t0, v, t); \
      else hipLaunchKernelGGL((msp2_p1_kernel<WW, false>), g1, b1, 0, ctx->stream, d_data, nN, k, m, c, canon, t0, v, t); \
      break;
    switch (W2) {
      CFRK_Q1_CASE(18) CFRK_Q1_CASE(20) CFRK_Q1_CASE(22) CFRK_Q1_CASE(24) CFRK_Q1_CASE(26) CFRK_Q1_CASE(28) CFRK_Q1_CASE(30)
      default: return cfrk_fail(ctx, CFRK_ERR_ARG, "no partition kernel for a window of %d", W2);
    }
#undef CFRK_Q1_CASE
    HIP_TRY(ctx, hipGetLastError());
    return CFRK_OK;
  };
  bool run_q1 = true, settled = false;
  uint64_t parked1 = 0, parked2 = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (chunked) {
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L1OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L1OVF, ST_OVFN1
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L2OVF, ST_OVFN
      for (int cc = 0; cc < nchunks; ++cc) {
        const int64_t t0 = tile0 + (int64_t)cc * chunk_tiles, t1 = std::min(tile0 + ntiles, t0 + chunk_tiles);
        if (t0 >= t1) break;
        if (cc > 0) HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, nreg * sizeof(uint32_t), ctx->stream));
        if ((rc = launch_q1(t0, t1))) return rc;
        if (cc == 0) {
          void *sp;
          if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, 64 * sizeof(uint64_t), &sp))) return rc;
          hipLaunchKernelGGL(msp2_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt1, (uint32_t)nreg, (uint64_t *)sp);
          HIP_TRY(ctx, hipGetLastError());
          hipLaunchKernelGGL(msp2_class_sample_kernel, dim3(B1), dim3(256), 0, ctx->stream, v, (unsigned long long *)sp);
          HIP_TRY(ctx, hipGetLastError());
          uint64_t made[1 + NCLS] = {0, 0, 0, 0, 0};        // records of the chunk; of the sample, by class
          HIP_TRY(ctx, hipMemcpyAsync(made, sp, sizeof made, hipMemcpyDeviceToHost, ctx->stream));
          HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
          // records of this pass's leaves in the whole batch (+3 %), per leaf, split by the sample: complete
          // runs up to 1.45 x their share of the mean leaf (1.3 x parks records of a few leaves, msp.hip), the largest class of truncated runs up to 1.6 x its share
          const double per_leaf = (double)made[0] * ((double)ntiles / (double)(t1 - t0)) * 1.03 / (double)(NLEAF >> sel_bits);
          const uint64_t sampled = made[1] + made[2] + made[3] + made[4];
          double fc = 0.75, ft = 0.3;
          if (sampled >= 4096) {
            fc = (double)made[4] / (double)sampled + 0.02;
            ft = (double)std::max(made[1], std::max(made[2], made[3])) / (double)sampled + 0.01;
          }
          // (the slack grows with how few distinct runs a leaf holds, msp.hip)
          const double lambda = std::max(4.0, (double)ctx->g_cap / 3.0 / (double)NLEAF * 4.0 / (double)(W2 + 1));
          const double fc_slack = std::min(4.0, std::max(1.45, 1.2 + 6.5 / std::sqrt(lambda)));
          const double ft_slack = std::min(4.0, std::max(1.6, 1.3 + 6.5 / std::sqrt(lambda)));
          v.cap2c = (uint64_t)(per_leaf * std::min(1.0, fc) * fc_slack) + 512; v.cap2t = (uint64_t)(per_leaf * std::min(1.0, ft) * ft_slack) + 256;
          if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(NLEAF >> sel_bits) * (v.cap2c + 3 * v.cap2t) * sizeof(Rec2), &p))) return rc;
          v.rec2 = (Rec2 *)p;
        }
        hipLaunchKernelGGL(msp2_p2_kernel, dim3((unsigned)(q2_groups * B1)), dim3(Q2_THREADS), 0, ctx->stream, (int)q2_groups, k, canon, v, t);
        HIP_TRY(ctx, hipGetLastError());
      }
      uint64_t stc[ST_NWORDS];
      HIP_TRY(ctx, hipMemcpyAsync(stc, ctx->g_stats, sizeof stc, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (stc[ST_CWRAP]) return CFRK_INTERNAL_FLOOD;                     // (msp.hip: nothing of this pass has been counted yet)
      if (stc[ST_L1OVF] || stc[ST_L2OVF]) return CFRK_ERR_SMALL_BUF;   // (no error text: the caller starts over, the old way)
      parked1 = stc[ST_OVFN1]; parked2 = stc[ST_OVFN];
      settled = true;
      break;
    }
    if (run_q1) {
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L1OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L1OVF, ST_OVFN1
      if ((rc = launch_q1(tile0, tile0 + ntiles))) return rc;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L2OVF, ST_OVFN
    hipLaunchKernelGGL(msp2_p2_kernel, dim3((unsigned)(q2_groups * B1)), dim3(Q2_THREADS), 0, ctx->stream,
                       (int)q2_groups, k, canon, v, t);
    HIP_TRY(ctx, hipGetLastError());
    if (defer) { settled = true; break; }             // (whether a region overflowed is looked at by the export)
    uint64_t st[ST_NWORDS];
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st[ST_CWRAP]) return CFRK_INTERNAL_FLOOD;
    if (st[ST_L1OVF]) {
      // exact level-1 layout; Q2 ran on an incomplete level 1 and is redone as well
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT1, nreg * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
      uint64_t *rbase = (uint64_t *)p;
      uint32_t *rcap = (uint32_t *)(rbase + nreg);
      hipLaunchKernelGGL(msp2_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt1, (uint32_t)nreg, rbase, rcap);
      HIP_TRY(ctx, hipGetLastError());
      {
        // the heaviest bin decides how many tile groups per bin Q2 needs from now on
        std::vector<uint32_t> c1(nreg);
        HIP_TRY(ctx, hipMemcpyAsync(c1.data(), v.cnt1, nreg * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        uint64_t maxbin = 0, all = 0;
        for (int b = 0; b < B1; ++b) {
          uint64_t sum = 0;
          for (int r = 0; r < NXG; ++r) sum += c1[q1_reg((uint32_t)b, (uint32_t)r)];
          maxbin = std::max(maxbin, sum);
          all += sum;
        }
        if (all > (uint64_t)B1 * NXG * cap1) {           // (more records than the density estimate allowed for)
          if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, (size_t)all * sizeof(Rec2), &p))) return rc;
          v.rec1 = (Rec2 *)p;
        }
        q2_groups = (int64_t)((maxbin + (uint64_t)Q2_TILE * Q2_GROUP - 1) / ((uint64_t)Q2_TILE * Q2_GROUP)) + 1;
        if (q2_groups * B1 > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
      }
      HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, (nreg + (size_t)NCLS * NLEAF) * sizeof(uint32_t), ctx->stream));   // cnt1 and cnt2
      v.exact1 = 1; v.rbase = rbase; v.rcap = rcap;
      run_q1 = true;
      continue;
    }
    parked1 = st[ST_OVFN1];
    if (st[ST_L2OVF] || v.count_only) {
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT, (size_t)NCLS * NLEAF * (sizeof(uint64_t) + 2 * sizeof(uint32_t)), &p))) return rc;
      uint64_t *lbase = (uint64_t *)p;
      uint32_t *lcap = (uint32_t *)(lbase + NCLS * NLEAF);
      hipLaunchKernelGGL(msp2_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt2, (uint32_t)(NCLS * NLEAF), lbase, lcap);
      HIP_TRY(ctx, hipGetLastError());
      if (v.count_only) {
        // the streams' buffer: exactly what was counted (last stream's start + size)
        uint64_t lb = 0; uint32_t lc = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&lb, lbase + NCLS * NLEAF - 1, sizeof lb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&lc, lcap + NCLS * NLEAF - 1, sizeof lc, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(lb + lc + 1) * sizeof(Rec2), &p))) return rc;
        v.rec2 = (Rec2 *)p;
        v.count_only = 0;
      }
      HIP_TRY(ctx, hipMemsetAsync(v.cnt2, 0, (size_t)NCLS * NLEAF * sizeof(uint32_t), ctx->stream));
      v.exact = 1; v.lbase = lbase; v.lcap = lcap;
      run_q1 = false;
      continue;
    }
    parked2 = st[ST_OVFN];
    settled = true;
    break;
  }
  if (!settled) return cfrk_fail(ctx, CFRK_ERR_STATE, "the record regions did not settle after an exact layout");
  if (parked1 && v.ovf1_cap) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(parked1, v.ovf1_cap);
    hipLaunchKernelGGL(msp2_spill_list_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, (const Rec2 *)v.ovf1, n, k, canon, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (parked2 && v.ovf_cap) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(parked2, v.ovf_cap);
    hipLaunchKernelGGL(msp2_spill_list_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, (const Rec2 *)v.ovf, n, k, canon, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (runs_only) {
    // deduplicate the leaves where they lie (unless deferred); the streams stay for the export
    static_assert(sizeof(View2) <= sizeof(ms->view2), "cfrk_msp::view2 holds a View2");
    memcpy(ms->view2, &v, sizeof v);
    if (!nodedupe && (rc = msp2_dedupe_in_place(ctx, v))) return rc;
    ms->pending = false;
    ms->runs_ready = true;
    ms->runs_deduped = !nodedupe; ms->runs_unchecked = defer;
    ms->leaf_form = false;
    ms->list_n_valid = false;
    return CFRK_OK;
  }
  {
    // (a shared leaf: one workgroup per four sub-values)
    const dim3 g3(((unsigned)NLEAF >> sel_bits) << (v.sub_bits - v.hbits)), b3(Q3_THREADS);
    // (the table holds 2 .. 4 x the announced distinct k-mers: 512 slots per leaf = ~300 expected keys, 0.3 of the small table)
    const bool small_leaves = !sub && ctx->g_cap / NLEAF <= 512u && !(ctx->dbg_flags & CFRK_DEBUG_NO_SMALL_LEAVES);
    if (sub) {
      if (canon) hipLaunchKernelGGL((msp2_p3_kernel<true, true>), g3, b3, 0, ctx->stream, k, 0u, v, t);
      else hipLaunchKernelGGL((msp2_p3_kernel<false, true>), g3, b3, 0, ctx->stream, k, 0u, v, t);
    } else if (!small_leaves && !(ctx->dbg_flags & CFRK_DEBUG_NO_SMALL_LEAVES)) {
      // (2048 slots per leaf announced = ~1200 expected keys = 0.6 of the table: beyond that, two subsets from the start)
      const uint32_t mode3 = ctx->g_cap / NLEAF > 2048u ? Q3_SPLIT2 : 0u;
      if (canon) hipLaunchKernelGGL((msp2_p3_mid_kernel<true>), g3, b3, 0, ctx->stream, k, mode3, v, t);
      else hipLaunchKernelGGL((msp2_p3_mid_kernel<false>), g3, b3, 0, ctx->stream, k, mode3, v, t);
    } else if (small_leaves) {
      if (canon) hipLaunchKernelGGL((msp2_p3_small_kernel<true>), g3, b3, 0, ctx->stream, k, 0u, v, t);
      else hipLaunchKernelGGL((msp2_p3_small_kernel<false>), g3, b3, 0, ctx->stream, k, 0u, v, t);
    } else {
      if (canon) hipLaunchKernelGGL((msp2_p3_kernel<true, false>), g3, b3, 0, ctx->stream, k, 0u, v, t);
      else hipLaunchKernelGGL((msp2_p3_kernel<false, false>), g3, b3, 0, ctx->stream, k, 0u, v, t);
    }
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(msp2_huge_leaves_kernel, dim3((((unsigned)NLEAF >> sel_bits) + 255u) / 256u), dim3(256), 0, ctx->stream, k, canon, 0, (uint32_t)NLEAF >> sel_bits, v, t);
  }
  HIP_TRY(ctx, hipGetLastError());
  ms->view.out_keys = v.out_lo; ms->view.out_hi = v.out_hi; ms->view.out_cnt = v.out_cnt;
  ms->view.out_cap = v.out_cap; ms->view.stats = v.stats; ms->view.cnt1 = nullptr;
  ms->view.leaf_off = v.leaf_off; ms->view.leaf_n = v.leaf_n; ms->view.seg_bits = v.sub_bits;
  ms->pending = true;
  ms->leaf_form = true;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// the result list buffers of ms->view must be set (out_keys / out_hi / out_cnt / out_cap)
int cfrk_msp2_merge_lists(cfrk_ctx *ctx, const uint64_t *d_lo, const uint64_t *d_hi, const uint32_t *d_cnt,
                          const uint64_t *d_seg_off, const uint32_t *d_seg_n, int parts, int leaves_per_part) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  View2 v;
  v.rec1 = nullptr; v.cnt1 = nullptr; v.cap1 = 0; v.rec2 = nullptr; v.cnt2 = nullptr; v.cap2c = v.cap2t = 0;
  v.out_lo = ms->view.out_keys; v.out_hi = ms->view.out_hi; v.out_cnt = ms->view.out_cnt; v.out_cap = ms->view.out_cap;
  v.leaf_off = nullptr; v.leaf_n = nullptr; v.stats = ctx->g_stats;
  v.exact = 0; v.lbase = nullptr; v.lcap = nullptr; v.ovf = nullptr; v.ovf_cap = 0; v.sel_mask = v.sel_val = v.sel_bits = 0; v.sub_bits = 0; v.hbits = 0; v.dbg = 0;
  v.exact1 = 0; v.rbase = nullptr; v.rcap = nullptr; v.ovf1 = nullptr; v.ovf1_cap = 0; v.count_only = 0;
  hipLaunchKernelGGL(msp2_merge_kernel, dim3(leaves_per_part), dim3(Q3_THREADS), 0, ctx->stream, d_lo, d_hi, d_cnt,
                     d_seg_off, d_seg_n, parts, leaves_per_part, v, cfrk_table_view(ctx));
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange by runs
// (cfrk_global_export_runs_device / cfrk_global_merge_runs_device of msp.hip for two-word keys;
// the callers have checked the arguments and the job's state)
int cfrk_msp2_export_runs(cfrk_ctx *ctx, void *d_packed, uint64_t cap_rows, int parts, uint64_t *part_rows) {
  cfrk_msp *ms = ctx->msp;
  View2 v;
  memcpy(&v, ms->view2, sizeof v);
  if (ms->runs_unchecked) {
    // a CFRK_RUNS_DEFER add: did its regions hold?  (an add without the flag lays an overflowing level out again)
    uint64_t st0[ST_NWORDS];
    int rc0 = cfrk_msp_sync_stats(ctx, st0);
    if (rc0) return rc0;
    if (st0[ST_L1OVF] || st0[ST_L2OVF] || st0[ST_OVFN] || st0[ST_OVFN1] || st0[ST_CWRAP])
      return cfrk_fail(ctx, CFRK_ERR_STATE, "the CFRK_RUNS_DEFER add overflowed a record region: add again without the flag");
    ms->runs_unchecked = false;
  }
  if (!ms->runs_deduped) {
    int rc0 = msp2_dedupe_in_place(ctx, v);
    if (rc0) return rc0;
    ms->runs_deduped = true;
  }
  const int lpp = (NLEAF + parts - 1) / parts;           // leaves per part (owner p: leaves p, p+parts, ...)
  const int hrows = runs_header_rows(lpp);
  int rc;
  void *p;
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, (NLEAF + 65 + ST_NWORDS + 1 + 64 + 72) * sizeof(uint64_t) + (size_t)NLEAF * sizeof(uint4), &p))) return rc;
  uint64_t *d_off = (uint64_t *)p, *d_rows = d_off + NLEAF;
  uint64_t *d_seg = d_rows + 65 + ST_NWORDS + 1;
  unsigned long long *d_sync = (unsigned long long *)(d_seg + 64);
  uint4 *d_sz = (uint4 *)(d_sync + 72);      // (16-byte aligned: the pool is, and NLEAF + 65 + ST_NWORDS + 1 is even)
  static_assert((NLEAF + 65 + ST_NWORDS + 1 + 64 + 72) % 2 == 0, "d_sz is 16-byte aligned");
  hipLaunchKernelGGL(msp2_runs_sizes_kernel, dim3(NLEAF / 256), dim3(256), 0, ctx->stream, v, d_sz, d_sync);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(msp_runs_plan_kernel, dim3(runs_plan_grid(parts, lpp)), dim3(1024), 0, ctx->stream, (const uint4 *)d_sz, parts, lpp, hrows, d_off,
                     d_rows + parts, d_seg, d_sync);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(msp2_runs_gather_kernel, dim3(NLEAF), dim3(256), 0, ctx->stream, v, (const uint64_t *)d_off, (uint4 *)d_packed,
                     (const uint64_t *)d_rows, (const uint64_t *)d_seg, parts, cap_rows);
  HIP_TRY(ctx, hipGetLastError());
  // [0, 65): all rows at [parts]; then the job's flags; then the segment starts -- ONE copy
  uint64_t h[65 + ST_NWORDS + 1 + 64];
  HIP_TRY(ctx, hipMemcpyAsync(d_rows + 65, ctx->g_stats, ST_NWORDS * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(h, d_rows, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const uint64_t *st = h + 65, *seg = h + 65 + ST_NWORDS + 1;
  if (st[ST_SPILLED] || st[ST_ONES]) return cfrk_fail(ctx, CFRK_ERR_STATE, "part of the batch was counted in the HBM table");
  if (h[parts] > cap_rows) return cfrk_fail(ctx, CFRK_ERR_SMALL_BUF, "%llu rows, room for %llu", (unsigned long long)h[parts], (unsigned long long)cap_rows);
  for (int q = 0; q < parts; ++q) part_rows[q] = (q + 1 < parts ? seg[q + 1] : h[parts]) - seg[q];
  return CFRK_OK;
}

int cfrk_msp2_merge_runs(cfrk_ctx *ctx, const void *d_packed, const uint64_t *recv_rows, int parts) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  if (ms->pending || ms->table_dirty) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_runs needs an empty job (call cfrk_global_begin first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  const int k = ctx->g_k;
  const int canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  const int lpp = (NLEAF + parts - 1) / parts;
  const int hrows = runs_header_rows(lpp);
  const size_t nseg = (size_t)parts * lpp;
  int rc;
  void *p;
  RunsRecv rr;
  memset(&rr, 0, sizeof rr);
  uint64_t rows_all = 0;
  for (int r = 0; r < parts; ++r) {
    if (recv_rows[r] < (uint64_t)hrows) return cfrk_fail(ctx, CFRK_ERR_ARG, "rank %d sent %llu rows, fewer than its header", r, (unsigned long long)recv_rows[r]);
    rr.rstart[r] = rows_all; rr.rows[r] = recv_rows[r];
    rows_all += recv_rows[r];
  }
  View2 v;
  memset(&v, 0, sizeof v);
  // shared leaves (msp2_count_tiles): the owner holds 1 / parts of the leaves, each as heavy as it is in
  // the whole job -- the capacity hint of an owner is its share of the job's distinct k-mers
  const uint64_t per_leaf = ctx->g_cap / NLEAF * (uint64_t)parts;
  uint32_t sub_bits = 0;
  while (sub_bits < (uint32_t)SUB_BITS && (per_leaf >> sub_bits) > 2048u) ++sub_bits;
  if (per_leaf <= 4096u) sub_bits = 0;
  if ((ctx->dbg_flags & CFRK_DEBUG_RECORD_SUBSETS) && sub_bits < 2u) sub_bits = 2u;
  v.sub_bits = sub_bits;
  v.hbits = msp2_hbits(ctx, sub_bits, per_leaf);
  const size_t nsub = (size_t)NLEAF << sub_bits;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, nsub * 8 + ((size_t)B1 * NXG + (size_t)NCLS * NLEAF + nsub) * sizeof(uint32_t), &p))) return rc;
  v.leaf_off = (uint64_t *)p;
  v.cnt1 = (uint32_t *)(v.leaf_off + nsub); v.cnt2 = v.cnt1 + B1 * NXG; v.leaf_n = v.cnt2 + NCLS * NLEAF;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT, (size_t)NCLS * NLEAF * (sizeof(uint64_t) + 2 * sizeof(uint32_t)), &p))) return rc;
  uint64_t *d_lbase = (uint64_t *)p;
  uint32_t *d_lcap = (uint32_t *)(d_lbase + NCLS * NLEAF);
  v.exact = 1; v.lbase = d_lbase; v.lcap = d_lcap;
  // (streams nobody fills -- truncated classes 1 and 2, leaves of other owners -- hold nothing)
  HIP_TRY(ctx, hipMemsetAsync(v.cnt2, 0, ((size_t)NCLS * NLEAF + nsub) * sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_lbase, 0, (size_t)NCLS * NLEAF * (sizeof(uint64_t) + sizeof(uint32_t)), ctx->stream));
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_lo = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTH, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_hi = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats; v.dbg = ctx->dbg_flags;
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, (nseg * 3 + 2) * sizeof(uint64_t) + nseg * sizeof(uint32_t), &p))) return rc;
  uint64_t *d_src = (uint64_t *)p, *d_d1 = d_src + nseg, *d_d0 = d_d1 + nseg, *d_out = d_d0 + nseg;
  uint32_t *d_segrows = (uint32_t *)(d_out + 2);
  HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 2 * sizeof(uint64_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
  TableView t = cfrk_table_view(ctx);
  // segment (source rank, local leaf): the ranks' headers say how large; all offsets on the device.
  // Stream 3 of a leaf takes the ranks' distinct runs, stream 0 all their truncated runs.
  hipLaunchKernelGGL((msp_runs_layout1_kernel<NCLS, 3, 0, 2>), dim3((unsigned)(lpp + 255) / 256), dim3(256), 0, ctx->stream, (const uint4 *)d_packed, rr, parts, lpp,
                     d_segrows, d_d1, d_d0, d_lcap, d_out);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL((msp_runs_layout_kernel<NCLS, 3, 0>), dim3((unsigned)parts + 1u), dim3(1024), 0, ctx->stream, rr, parts, lpp, hrows, (const uint32_t *)d_segrows,
                     d_src, d_d1, d_d0, d_lbase, (const uint32_t *)d_lcap, v.cnt2, d_out);
  HIP_TRY(ctx, hipGetLastError());
  // the headers are checked before anything is copied by them (msp.hip)
  uint64_t h[2];
  HIP_TRY(ctx, hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (h[1]) return cfrk_fail(ctx, CFRK_ERR_ARG, "a rank's header does not add up to the rows it sent");
  if (h[0] > rows_all * NOTES_PER_ROW) return cfrk_fail(ctx, CFRK_ERR_ARG, "the headers announce more records than the rows can hold");
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(h[0] ? h[0] : 1) * sizeof(Rec2), &p))) return rc;
  v.rec2 = (Rec2 *)p;
  hipLaunchKernelGGL(msp2_runs_scatter_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, (const uint4 *)d_packed, rr, lpp, k,
                     (const uint64_t *)d_src, (const uint64_t *)d_d1, (const uint64_t *)d_d0, v.rec2);
  HIP_TRY(ctx, hipGetLastError());
  {
    // the owner's leaves are local indices 0 .. lpp-1 (a shared leaf: one workgroup per four sub-values,
    // eight leaves side by side on the XCDs -- leaves beyond lpp hold nothing and leave at once)
    const unsigned gbits = sub_bits - v.hbits;
    const dim3 g3(sub_bits ? (((unsigned)lpp + 7u) & ~7u) << gbits : (unsigned)lpp), b3(Q3_THREADS);
    if (sub_bits) {
      if (canon) hipLaunchKernelGGL((msp2_p3_kernel<true, true>), g3, b3, 0, ctx->stream, k, Q3_WEIGHTED, v, t);
      else hipLaunchKernelGGL((msp2_p3_kernel<false, true>), g3, b3, 0, ctx->stream, k, Q3_WEIGHTED, v, t);
    } else {
      if (canon) hipLaunchKernelGGL((msp2_p3_kernel<true, false>), g3, b3, 0, ctx->stream, k, Q3_WEIGHTED, v, t);
      else hipLaunchKernelGGL((msp2_p3_kernel<false, false>), g3, b3, 0, ctx->stream, k, Q3_WEIGHTED, v, t);
    }
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(msp2_huge_leaves_kernel, dim3(((unsigned)lpp + 255u) / 256u), dim3(256), 0, ctx->stream, k, canon, 1, (uint32_t)lpp, v, t);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->ev_valid = true;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->h_stats_valid = false;                          // the leaf kernel may have spilled into the table
  ms->view.out_keys = v.out_lo; ms->view.out_hi = v.out_hi; ms->view.out_cnt = v.out_cnt;
  ms->view.out_cap = v.out_cap; ms->view.stats = v.stats; ms->view.cnt1 = nullptr;
  ms->view.leaf_off = v.leaf_off; ms->view.leaf_n = v.leaf_n; ms->view.seg_bits = v.sub_bits;
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange by runs, pipelined (msp.hip)
int cfrk_msp2_export_runs_async(cfrk_ctx *ctx, void *d_packed, uint64_t seg_cap_rows, int parts, int ngroups) {
  cfrk_msp *ms = ctx->msp;
  View2 v;
  memcpy(&v, ms->view2, sizeof v);
  const int k = ctx->g_k, canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  // (the table's size from the expected distinct runs per leaf, as in msp2_dedupe_in_place)
  const double runs_per_leaf = (double)(ctx->g_cap / NLEAF) / 2.0 * 4.0 / (double)(msp2_window(k) + 1);
  const bool small_tab = runs_per_leaf <= 400.0;
  const bool sub = v.sub_bits != 0u;
  return runs_export_async_host(ctx, d_packed, seg_cap_rows, parts, ngroups, [&](const RunsSend &sg) {
    if (sub) hipLaunchKernelGGL((msp2_dedupe_send_kernel<12, 1024, true>), dim3(sg.nleaf), dim3(1024), 0, ctx->stream, k, canon, v, sg);
    else if (small_tab) hipLaunchKernelGGL((msp2_dedupe_send_kernel<10, 256, false>), dim3(sg.nleaf), dim3(256), 0, ctx->stream, k, canon, v, sg);
    else hipLaunchKernelGGL((msp2_dedupe_send_kernel<11, 256, false>), dim3(sg.nleaf), dim3(256), 0, ctx->stream, k, canon, v, sg);
  });
}

int cfrk_msp2_merge_runs_group(cfrk_ctx *ctx, const void *d_recv, const uint64_t *recv_rows, int parts, int group, int ngroups) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  if (group != ms->lists_group) return cfrk_fail(ctx, CFRK_ERR_STATE, "groups are merged in order: expected group %d", ms->lists_group);
  if (group == 0 && (ms->pending || ms->table_dirty)) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_runs_group needs an empty job (call cfrk_global_begin first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int lpp = (NLEAF + parts - 1) / parts;
  if (ngroups > lpp) return cfrk_fail(ctx, CFRK_ERR_ARG, "more groups than leaves per owner");
  int rc;
  void *p;
  View2 v;
  if (group == 0) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    memset(&v, 0, sizeof v);
    // shared leaves (cfrk_msp2_merge_runs): the owner holds 1 / parts of the leaves, each as heavy as it is in the whole job
    const uint64_t per_leaf = ctx->g_cap / NLEAF * (uint64_t)parts;
    uint32_t sub_bits = 0;
    while (sub_bits < (uint32_t)SUB_BITS && (per_leaf >> sub_bits) > 2048u) ++sub_bits;
    if (per_leaf <= 4096u) sub_bits = 0;
    if ((ctx->dbg_flags & CFRK_DEBUG_RECORD_SUBSETS) && sub_bits < 2u) sub_bits = 2u;
    v.sub_bits = sub_bits;
    v.hbits = msp2_hbits(ctx, sub_bits, per_leaf);
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
    v.out_lo = (uint64_t *)p;
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTH, (size_t)ctx->g_cap * 8, &p))) return rc;
    v.out_hi = (uint64_t *)p;
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
    v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
    v.stats = ctx->g_stats; v.dbg = ctx->dbg_flags;
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
    memcpy(ms->view2, &v, sizeof v);
  } else {
    memcpy(&v, ms->view2, sizeof v);
  }
  P3ListsT<true> lx;
  memset(&lx, 0, sizeof lx);
  lx.packed = (const uint4 *)d_recv;
  uint64_t at = 0;
  for (int r = 0; r < parts; ++r) { lx.rr.rstart[r] = at; lx.rr.rows[r] = recv_rows[r]; at += recv_rows[r]; }
  lx.parts = parts;
  lx.ll0 = runs_ll0(lpp, group, ngroups);
  lx.lcount = runs_ll0(lpp, group + 1, ngroups) - lx.ll0;
  TableView t = cfrk_table_view(ctx);
  if (lx.lcount) {
    const bool canon = (ctx->g_flags & CFRK_CANONICAL) != 0;
    if (v.sub_bits) {
      // (one workgroup per four sub-values, eight leaves side by side on the XCDs: local leaves beyond the group's leave at once)
      const dim3 g3(((lx.lcount + 7u) & ~7u) << (v.sub_bits - v.hbits));
      if (canon) hipLaunchKernelGGL((msp2_p3_lists_kernel<true, true>), g3, dim3(Q3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
      else hipLaunchKernelGGL((msp2_p3_lists_kernel<false, true>), g3, dim3(Q3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
    } else {
      if (canon) hipLaunchKernelGGL((msp2_p3_lists_kernel<true, false>), dim3(lx.lcount), dim3(Q3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
      else hipLaunchKernelGGL((msp2_p3_lists_kernel<false, false>), dim3(lx.lcount), dim3(Q3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->ev_valid = true;
  ctx->h_stats_valid = false;                          // the leaf kernel may have spilled into the table
  ms->view.out_keys = v.out_lo; ms->view.out_hi = v.out_hi; ms->view.out_cnt = v.out_cnt;
  ms->view.out_cap = v.out_cap; ms->view.stats = v.stats; ms->view.cnt1 = nullptr;
  ms->view.leaf_off = nullptr; ms->view.leaf_n = nullptr; ms->view.seg_bits = 0;
  ms->lists_group = group + 1;
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}

int cfrk_msp2_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  int rc;
  const int64_t nchunks = (nN + 31) / 32 + 1;              // + the chunk that only holds k-mer tails
  const int64_t nwaves = (nchunks + Q1_OWN - 1) / Q1_OWN;
  const int64_t ntiles = (nwaves + Q1_WAVES - 1) / Q1_WAVES;
  if (ntiles > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
  const size_t have = ctx->pool[BUF_MSP_L1].cap + ctx->pool[BUF_MSP_L2].cap + ctx->pool[BUF_MSP_OUTK].cap +
                      ctx->pool[BUF_MSP_OUTC].cap + ctx->pool[BUF_MSP_OUTH].cap;
  const bool runs_only = (ctx->g_flags & CFRK_RUNS_ONLY) != 0;
  if (runs_only && ms->runs_ready) return cfrk_fail(ctx, CFRK_ERR_STATE, "a CFRK_RUNS_ONLY job takes one add");
  int groups = 1;
  if ((rc = cfrk_msp_plan_groups(ctx, nN + 32, ntiles, (int64_t)Q1_WAVES * Q1_OWN * 32, msp2_need,
                                 (size_t)ctx->g_cap * 20, have, &groups))) return rc;
  // a batch that does not fit in one pass with the generous fixed-stride streams: would it with streams
  // sized from a counting pass?  (fewer passes over the input for one more read of the level-1 records)
  bool lean = false;
  ms->dens_scale = 1.0;
  // a large batch is counted in chunks when that fits in one pass (msp2_count_tiles); a batch that
  // overflows the sizes measured from its first chunk comes back and is planned as before
  {
    int cg = 0;
    const bool big = (double)(nN + 32) * msp2_density(ctx) * 1.35 * 32.0 > 6e9 || (ctx->dbg_flags & CFRK_DEBUG_SMALL_PIPELINE);
    // (CFRK_RUNS_DEFER: a batch that fits one pass with fixed-stride streams takes that form -- it ends unsynchronised;
    //  a chunked add sizes its streams from a read-back)
    const bool want_defer = runs_only && (ctx->g_flags & CFRK_RUNS_DEFER) && groups == 1;
    if (big && !want_defer && !(ctx->dbg_flags & CFRK_DEBUG_NO_PIPELINE) && !ctx->mem_budget &&
        !(rc = cfrk_msp_plan_groups(ctx, nN + 32, ntiles, (int64_t)Q1_WAVES * Q1_OWN * 32, msp2_need_chunked, (size_t)ctx->g_cap * 20, have, &cg)) &&
        cg == 1) {
      if (ms->pending && (rc = cfrk_msp_flush_to_table(ctx))) return rc;
      ctx->last_passes = 1;
      rc = msp2_count_tiles(ctx, ms, d_data, nN, 0, ntiles, 1.0, 0, 0, true, false, true);
      // (nothing has been counted when the leaf streams do not fit or overflow: the batch starts over)
      if (rc != CFRK_ERR_SMALL_BUF && rc != CFRK_ERR_NOMEM) return rc;
      if (rc == CFRK_ERR_NOMEM) (void)hipGetLastError();
    }
  }
  if (groups != 1) {
    // first, how many positions start a k-mer at all: reads of a few k make far fewer records than
    // one per 2 / (W + 1) bytes (250-base reads at k = 63: 0.75 of it) -- a 5 TB/s look at the batch
    void *sp;
    if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, 64, &sp))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(sp, 0, 8, ctx->stream));
    hipLaunchKernelGGL(msp2_count_invalid_kernel, dim3((unsigned)std::min<int64_t>(ctx->num_cus * 16, (nN >> 12) + 1)), dim3(256), 0,
                       ctx->stream, d_data, nN, (unsigned long long *)sp);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t ninv = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ninv, sp, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double starts = std::max(0.0, (double)nN - (double)ninv * (double)ctx->g_k);
    ms->dens_scale = std::min(1.0, std::max(0.05, starts / (double)std::max<int64_t>(nN, 1)));
    if ((rc = cfrk_msp_plan_groups(ctx, nN + 32, ntiles, (int64_t)Q1_WAVES * Q1_OWN * 32, msp2_need,
                                   (size_t)ctx->g_cap * 20, have, &groups))) return rc;
  }
  if (groups != 1) {
    int lg = 1;
    if ((rc = cfrk_msp_plan_groups(ctx, nN + 32, ntiles, (int64_t)Q1_WAVES * Q1_OWN * 32, msp2_need_lean,
                                   (size_t)ctx->g_cap * 20, have, &lg))) return rc;
    if (lg != 0 && (groups == 0 || lg < groups)) { groups = lg; lean = true; }
  }
  if (groups == 0) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "partitioned path does not fit device memory");
  const int passes = groups;
  ctx->last_passes = passes;
  if (runs_only && passes != 1) return cfrk_fail(ctx, CFRK_ERR_RUNS_REFUSED, "a CFRK_RUNS_ONLY job must fit device memory in one pass");
  if (ms->pending && (rc = cfrk_msp_flush_to_table(ctx))) return rc;
  if (passes == 1 && lean) return msp2_count_tiles(ctx, ms, d_data, nN, 0, ntiles, 1.0, 0, 0, true, true);
  if (passes == 1) {
    // lumpy leaves (small genomes): up to twice the stream room when memory is plentiful (msp.hip)
    double slack = 1.0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const double l2 = (double)(nN + 32) * msp2_density(ctx) * 3.3 * 32;
      size_t budget = have + free_b;
      if (ctx->mem_budget) budget = std::min(budget, ctx->mem_budget);
      const double room = 0.5 * (double)budget - (double)msp2_need(ctx, nN + 32);
      if (room > 0 && l2 > 0) slack = std::min(2.0, 1.0 + room / l2);
    }
    return msp2_count_tiles(ctx, ms, d_data, nN, 0, ntiles, slack);
  }

  // several passes over the WHOLE input, each emitting and counting 1/passes of the leaves (msp.hip)
  int sel_bits = 0;
  while ((1 << sel_bits) < passes) ++sel_bits;
  for (int pass = 0; pass < passes; ++pass) {
    if ((rc = msp2_count_tiles(ctx, ms, d_data, nN, 0, ntiles, 1.0, sel_bits, (uint32_t)pass, pass == 0, lean))) {
      // a refusal after the first pass must not reach the caller's fallback (it would count
      // the finished passes twice)
      if (pass > 0 && (rc == CFRK_ERR_NOMEM || rc == CFRK_INTERNAL_FLOOD)) return cfrk_fail(ctx, CFRK_ERR_STATE, "out of device memory in pass %d of a multi-pass add", pass);
      return rc;
    }
  }
  return CFRK_OK;
}
