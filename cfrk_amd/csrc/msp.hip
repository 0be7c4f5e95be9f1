// msp.hip -- minimizer-partitioned global k-mer counting for gfx950 (16 <= k <= 32).
//
// Why: one HBM atomic per k-mer occurrence (global_hash.hip) runs at the chip's scattered-atomic
// rate (~2e10/s), two orders of magnitude under what HBM bandwidth allows.  Here HBM only sees
// streams, and every occurrence is counted with LDS atomics:
//
//   P1  msp_p1b_kernel read the flat code buffer once (2 x dwordx4 per lane, 2-bit packing in
//                      registers), compute for every k-mer the minimum hash over its W = k-m+1
//                      canonical m-mers (its minimizer), cut the k-mers into runs that share one
//                      minimizer occurrence ("super-k-mers"; content-defined, they may cross into
//                      the next lane's chunk), and append each run as one 16-B record
//                      {48 bases, leaf id, closed-end flags, n} to level-1 bin = leaf >> 8.  A
//                      workgroup stages its records in LDS, reserves space with ONE global
//                      atomic per non-empty bin per 15.6 KB of input, then copies out in bin
//                      order (coalesced).  Emission is wave-balanced: lane i builds the wave's
//                      i-th record.
//   P2  msp_p2_kernel  stream every level-1 region, split it 512 ways: leaf low byte x
//                      {truncated run, complete run} (LDS counting sort of 4096-record tiles,
//                      one global atomic per stream per tile, coalesced copy-out).
//   P3  msp_p3_kernel  one workgroup per leaf (65536 leaves), everything in LDS: complete runs
//                      are counted per RECORD in a record table (at depth most are identical),
//                      then every distinct record is expanded once with its multiplicity, and
//                      truncated runs with weight 1, into a bucketized k-mer table
//                      (ds_cmpst_b64 to claim, ds_add_u32 to count); the occupied slots are
//                      compacted to the output list with one cursor atomic per workgroup.
//                      Leaves with more distinct runs than the record table holds (and every
//                      leaf for k < 28) deduplicate in a table over the whole LDS pool first
//                      and list the distinct runs at the head of their own stream.
//   multi-GPU          leaves are disjoint in key space on every rank, so lists are exchanged
//                      and added per leaf, again in LDS (msp_gather_kernel, msp_merge_kernel).
//
// All occurrences of a k-mer share its minimizer, hence its leaf, so leaves are disjoint in key
// space and the output list is the final result.  Anything that does not fit (bin capacity, LDS
// table) is counted with table_add1() into the global HBM table instead ("64-bit HBM atomics
// only on spill") and ST_SPILLED is raised; the host then folds the list into the table.
//
// Semantics are those of global_hash.hip (the guarded ComputeFreq of
// /root/reference/src/kmer_kernel.cu:52-70 summed over reads).
#include "msp.h"
#include "table.h"
#include "msp_dev.h"

#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

namespace {
#include "msp_runs.h"

constexpr int P1_THREADS = 512;

constexpr int P2_THREADS = 512, P2_PER = 8;   // (4096-record tiles: two 75 KB workgroups per CU; 8192 and 2560 measured slower, profiles/r05/p2_tile_size_*)
constexpr int P2_GROUP = 4;                      // consecutive tiles per workgroup (next tile prefetched)

constexpr int P3_THREADS = 1024;
constexpr int TS_LOG = 12, TS = 1 << TS_LOG;   // LDS table slots per leaf

// count every k-mer of a record straight into the global HBM table
// (the table view by value: a reference makes every thread of the calling kernel store the view
// to scratch memory at kernel entry -- 48 bytes per thread that reach HBM)
__device__ __noinline__ void spill_record(uint4 rec, int k, bool canon, TableView t, uint32_t weight = 1u) {
  t.stats[ST_SPILLED] = 1;
  dev_count_event(&t.stats[ST_AUX0]);
  const int nk = (int)(rec.w & 63u) + 1;
  const uint64_t hi = ((uint64_t)rec.x << 32) | rec.y;
  const uint64_t lo = (uint64_t)rec.z << 32;
  for (int j = 0; j < nk; ++j) {
    const uint64_t x = j ? ((hi << (2 * j)) | (lo >> (64 - 2 * j))) : hi;
    uint64_t key = x >> (64 - 2 * k);
    if (canon) {
      const uint64_t rc = dev_revcomp64(key, k);
      key = rc < key ? rc : key;
    }
    table_add1(t, key, weight);
  }
}

// Wave layout: 64 lanes load 64 consecutive 32-byte chunks; lane 0 and lanes 62, 63 are halo
// lanes (they compute, never emit), lanes 1..61 own their chunk's 32 window starts.  A run is
// emitted by the lane that owns its FIRST k-mer and may extend up to W-1 positions into the next
// lane's chunk, so run boundaries depend on the sequence only -- not on where a read happens to
// sit in the buffer -- and every read covering a locus emits the same record.
constexpr int P1_OWN = 61;                       // owner lanes per wave
constexpr int P1_WAVES = P1_THREADS / 64;

// Level-1 region (and cursor) of bin `bin`, sub-region `xg`: sub-region major.  Global atomics
// execute at the memory side, one 64-byte request per touched 64 bytes: with the cursors of the
// 256 bins of one sub-region side by side, a workgroup's 256 reservations are 16 requests instead
// of 256.
__host__ __device__ __forceinline__ uint32_t l1_reg(uint32_t bin, uint32_t xg) { return xg * (uint32_t)B1 + bin; }
// level-1 bin of a record: the leaf id's high byte (header bits 16..23; the top byte may hold sub-value bits)
template <bool SUB>
__device__ __forceinline__ uint32_t bin_of(uint32_t w) { return SUB ? ((w >> 16) & (uint32_t)(B1 - 1)) : (w >> 16); }

// write record `rec` as entry `dst` of level-1 region `reg`; a full region parks a few records and
// beyond that raises ST_L1OVF (the cursors keep counting: the host redoes P1 with exact sizes)
// (EX = false: compiled for the fixed-stride layout only -- the fused pipeline kernel, which never
// runs on an exact layout, sheds the code and the scalar registers of that case)
template <bool EX>
__device__ __forceinline__ void l1_put(const MspView &v, uint32_t reg, uint32_t dst, uint4 rec, int k, bool canon,
                                       const TableView &t) {
  const bool exact1 = EX && v.exact1;
  const uint64_t cap = exact1 ? (uint64_t)v.rcap[reg] : v.cap1;
  const uint64_t at = exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1;
  if (dst < cap) {
    v.rec1[at + dst] = rec;
  } else if (exact1) {
    spill_record(rec, k, canon, t);                      // cannot happen: cap is the exact count
  } else if (*(volatile uint64_t *)&v.stats[ST_L1OVF] == 0) {
    const unsigned long long o = atomicAdd((unsigned long long *)&v.stats[ST_OVFN1], 1ull);
    if (o < v.ovf1_cap) v.ovf1[o] = rec;
    else v.stats[ST_L1OVF] = 1;
  }
}

// ---------------------------------------------------------------------------------------- P1
// Emission is wave-balanced.  Building a lane's records in a loop over ITS runs (first version)
// costs the wave max-over-lanes trips (5..6 at W = 18 for 2.8 runs per lane on average, 18 at
// W = 4) of a ~75-instruction body.  Here a lane only lists its run starts (position descriptors,
// a few instructions per trip); everything a record is made of is staged per wave in LDS -- the
// wave's 2-bit base string, run terminators, validity, the leaf id of every position -- and then
// lane i builds the wave's i-th record: ceil(runs / 64) trips with all lanes busy.  Records stay
// in registers until the workgroup's bin histogram is scanned and go to LDS directly in bin order
// (rank from the histogram atomic), so the staging area and the sorted records share one
// allocation.  Runs per wave grow as 2/(W+1): the trips held in registers are a template
// parameter (4 for W >= 16 ... 12 for W = 4), and so are registers and workgroups per CU.
// P1B_TR = balanced trips held in registers (64 runs per wave each)
// LDS of one P1 workgroup: per-wave staging (later the bin-sorted records), histogram, offsets
// SUB: a job with far more distinct k-mers than the leaf tables hold (capacity hint > ~2.7e8) also stages
// five bits of every position's minimizer hash that the leaf id does not use and writes them to the top
// byte of the record's header word: the leaf kernel then splits an overfull leaf by RECORD, not by key,
// so that every record is expanded once (msp2.hip does the same).  2 KB more staging per wave: two
// workgroups per CU instead of three.
constexpr int SUB_BITS = 5;
template <int P1B_TR, bool SUB = false>
struct P1Lds {
  static constexpr int STAGE = 4096 + 3 * 512 + P1B_TR * 128 + (SUB ? 2048 : 0);     // bytes of staging per wave
  static constexpr int ARENA = P1_WAVES * STAGE;
  static constexpr int BYTES = ARENA + 2 * B1 * 4 + 32;
};

// Workgroup-level body of the partition kernel: tile `tile` of the input, appended to sub-region
// `subreg` of every level-1 bin.
// (One tile per workgroup, not a persistent loop: a looping workgroup waits at the top of every
// tile for its input loads -- and, vmcnt being one in-order counter, for the previous tile's
// stores before them; measured 16-28 % slower, profiles/r03/persistent_p1_is_slower.txt.)
template <int W, int P1B_TR, bool EX, bool SUB>
__device__ __forceinline__ void p1_tile(uint4 *pool, const int8_t *__restrict__ data, int64_t nN, int k, int m, int canon,
                                        int64_t tile, uint32_t subreg, const MspView &v, const TableView &t) {
  constexpr int NH = 32 + W - 1;
  constexpr int P1B_STAGE = P1Lds<P1B_TR, SUB>::STAGE;
  // sorted records share the staging bytes; the last 2.5 KB hold the copy-out table (dabs, plim)
  constexpr int P1B_TAIL = B1 * 10;
  constexpr int P1B_RCAP = (P1_WAVES * P1B_STAGE - P1B_TAIL) / 16;
  uint4 *const arena = pool;                     // per-wave staging, later the bin-sorted records (P1B_RCAP of them)
  uint32_t *const hist = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(pool) + P1Lds<P1B_TR, SUB>::ARENA);
  uint32_t *const loff = hist + B1;
  uint32_t *const wtot = loff + B1;              // 4 words
  uint32_t *const nrec_p = wtot + 4;
  uint32_t *const gbase = hist;                  // the global bases take the histogram's place once it is scanned
  // copy-out fast path, worked out once per bin and tile by the thread that reserved the segment:
  // arena position p of bin b goes to record dabs[b] + p of the level-1 buffer while p < plim[b]
  // (at the end of the arena: the staging area is dead by then and three workgroups keep fitting a CU's LDS)
  unsigned long long *const dabs = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(arena) + P1B_RCAP * 16);
  uint16_t *const plim = reinterpret_cast<uint16_t *>(dabs + B1);
  static_assert((P1_WAVES * P1B_STAGE - P1B_TAIL) % 16 == 0, "the table starts on a record boundary");

  const int nkmax = min(48 - k + 1, 32);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint8_t *const stage = reinterpret_cast<uint8_t *>(arena) + wave * P1B_STAGE;
  uint16_t *const s_leaf = reinterpret_cast<uint16_t *>(stage);               // [64 lanes][32 positions]
  uint32_t *const s_str = reinterpret_cast<uint32_t *>(stage + 4096);         // 128 dwords of bases
  uint64_t *const s_E = reinterpret_cast<uint64_t *>(stage + 4096 + 512);     // run terminators
  uint64_t *const s_W = reinterpret_cast<uint64_t *>(stage + 4096 + 1024);    // validity of position p-1
  uint16_t *const s_dsc = reinterpret_cast<uint16_t *>(stage + 4096 + 1536);  // (lane << 5) | position
  uint8_t *const s_sub = stage + 4096 + 1536 + P1B_TR * 128;                  // [64 lanes][32 positions] (SUB only)
  if (tid < B1) hist[tid] = 0;
  lds_barrier();

  // ---- A: this lane's chunk, packed 2 bits per base; neighbours' chunks by shuffle ----
  const int64_t wave_g = tile * P1_WAVES + wave;
  const int64_t chunk = wave_g * P1_OWN + lane - 1;
  const int64_t off = chunk * 32;
  uint32_t b0 = 0, b1 = 0, bad = 0xFFFFFFFFu;
  if (chunk >= 0) dev_load_chunk32(data, off, nN, b0, b1, bad);
  const uint32_t n0 = dev_lane_next(b0), n1 = dev_lane_next(b1);
  const uint32_t nbad = dev_lane_next(bad), nnbad = dev_lane_next(dev_lane_next(bad));
  const uint64_t hi = ((uint64_t)b0 << 32) | b1;
  const uint64_t mid = ((uint64_t)n0 << 32) | n1;
  uint64_t Yh = ((uint64_t)bad << 32) | nbad, Yl = (uint64_t)nnbad << 32;
  {
    int w = 1;
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      if (2 * w <= k) {
        Yh |= (Yh << w) | (Yl >> (64 - w));
        Yl |= Yl << w;
        w *= 2;
      }
    }
    if (k > w) {
      Yh |= (Yh << (k - w)) | (Yl >> (64 - (k - w)));
      Yl |= Yl << (k - w);
    }
  }
  const uint64_t Vx = ~Yh;
  const uint32_t V = (uint32_t)(Vx >> 32);
  const uint32_t prevV = dev_lane_prev(V) & 1u;

  uint32_t H[NH];
  const uint64_t Cx = msp_minimizers<W>(hi, mid, chunk, m, H);
  const uint64_t E = Cx | ~Vx | (1ull << (63 - NH));
  const uint32_t Vprev = (V >> 1) | (prevV << 31);
  uint32_t S = V & ((uint32_t)(Cx >> 32) | ~Vprev);
  const bool owner = lane >= 1 && lane <= P1_OWN && off < nN;
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
      // bits 24..27 and 7 of the packed minimum: hash bits, equal for every occurrence of a k-mer, that
      // neither the leaf id (bits 8..23) nor the position tag (bits 0..4) uses; bit 7 -- the uniform one -- lowest
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
  if (v.dbg & CFRK_ABL_P1_NO_EMIT) {             // timing ablation: keep the front end alive, emit nothing
    if (S == 0x12345678u && (uint32_t)E == 0x9ABCDEFu) v.stats[ST_AUX0] = 1;
    return;
  }
  uint32_t cnt_w;
  uint32_t S2 = 0;                               // run starts beyond the balanced phase's capacity
  // (CFRK_DEBUG_SMALL_WAVE_CAP: one trip's worth, so that tests reach the direct-append path)
  const uint32_t wcap = (v.dbg & CFRK_DEBUG_SMALL_WAVE_CAP) ? 64u : (uint32_t)(P1B_TR * 64);
  {
    const uint32_t mine = (uint32_t)__popc(S);
    const uint32_t incl = dev_wave_scan_incl(mine);
    cnt_w = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t widx = incl - mine;
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
  lds_wave_sync();                               // the wave's staging is written: lanes read each other's

  // record of the run that starts at position d & 31 of lane d >> 5 (d is also the index of its leaf id)
  auto build = [&](uint32_t d) {
    const uint32_t L = d >> 5, a = d & 31u;
    const uint64_t Es = s_E[L], Ws = s_W[L];
    const uint32_t leaf = s_leaf[d];
    // 96 bits of the wave's base string from bit 2d: dwords idx0..idx0+3, funnel-shifted
    const uint32_t P = 2u * d - 2u, idx0 = P >> 5, sh = 30u - (P & 31u);
    const uint32_t D0 = s_str[idx0], D1 = s_str[idx0 + 1], D2 = s_str[idx0 + 2], D3 = s_str[idx0 + 3];
    const uint64_t rest = Es << (a + 1);
    const int n = min(__clzll(rest) + 1, nkmax);
    // An end of the run is "closed" when it is a minimizer change between valid k-mers (header
    // bit 6: left end, bit 7: right end) and open when the read or an invalid base cut it.  Both
    // closed = a complete run: every read covering this locus emits the same record.  One end
    // open = a prefix (suffix) of the complete run of its locus, with which it shares its first
    // (last) k-mer.
    const uint32_t flags = (((uint32_t)(Ws >> (63 - a)) & 1u) << 6) | (((uint32_t)(Ws >> (62 - a - n)) & 1u) << 7);
    uint4 rec;
    rec.x = __builtin_amdgcn_alignbit(D0, D1, sh);
    // bases after the run's last k-mer are cleared: equal runs -> byte-identical records
    const int z = 2 * (48 - (n + k - 1));        // < 64 for k >= 17
    uint64_t r12 = ((uint64_t)__builtin_amdgcn_alignbit(D1, D2, sh) << 32) | __builtin_amdgcn_alignbit(D2, D3, sh);
    r12 = (W < 8 && z >= 64) ? 0ull : ((r12 >> z) << z);
    rec.y = (uint32_t)(r12 >> 32);
    rec.z = (uint32_t)r12;
    rec.w = (leaf << 8) | flags | (uint32_t)(n - 1);
    if (SUB) rec.w |= (uint32_t)s_sub[d] << 24;
    return rec;
  };

  // more runs in this wave than the balanced phase holds (pathological input): append directly
  while (S2) {
    const int a = __clz(S2);
    S2 &= ~(0x80000000u >> a);
    const uint4 rec = build(((uint32_t)lane << 5) | (uint32_t)a);
    if (((rec.w >> 8) & v.sel_mask) != v.sel_val) continue;    // not a leaf of this pass
    const uint32_t reg = l1_reg(bin_of<SUB>(rec.w), subreg);
    const uint32_t dst = atomicAdd(&v.cnt1[reg], 1u);
    l1_put<EX>(v, reg, dst, rec, k, canon != 0, t);
  }

  // ---- B2: lane i builds the wave's i-th record ----
  uint4 rc[P1B_TR];
  uint32_t rk[P1B_TR];                           // rank inside the record's bin; ~0: no record
  cnt_w = min(cnt_w, wcap);
#pragma unroll
  for (int tr = 0; tr < P1B_TR; ++tr) {
    rk[tr] = 0xFFFFFFFFu;
    rc[tr] = make_uint4(0, 0, 0, 0);
    const uint32_t i = (uint32_t)(tr * 64 + lane);
    if (i < cnt_w) {
      rc[tr] = build(s_dsc[i]);
      if (((rc[tr].w >> 8) & v.sel_mask) == v.sel_val)       // (all leaves, unless the batch takes several passes)
        rk[tr] = atomicAdd(&hist[bin_of<SUB>(rc[tr].w)], 1u);
    }
  }
  lds_barrier();

  // ---- C: one global reservation per non-empty bin; bin offsets ----
  uint32_t my_base = 0;
  if (tid < B1) {
    const uint32_t c = hist[tid];
    if (c) my_base = atomicAdd(&v.cnt1[l1_reg(tid, subreg)], c);
  }
  block_scan<B1, true>(hist, loff, wtot);              // ends with a barrier: the staging area is dead
  if (tid == B1 - 1) *nrec_p = loff[tid] + hist[tid];   // (before this thread overwrites hist[tid] below)
#pragma unroll
  for (int tr = 0; tr < P1B_TR; ++tr) {
    if (rk[tr] != 0xFFFFFFFFu) {
      const uint32_t pos = loff[bin_of<SUB>(rc[tr].w)] + rk[tr];
      if (pos < (uint32_t)P1B_RCAP) arena[pos] = rc[tr];
    }
  }
  if (tid < B1) {
    gbase[tid] = my_base;
    const uint32_t reg = l1_reg(tid, subreg);
    const bool exact1 = EX && v.exact1;
    const uint64_t cap = exact1 ? (uint64_t)v.rcap[reg] : v.cap1;
    const uint64_t at = exact1 ? v.rbase[reg] : (uint64_t)reg * v.cap1;
    const uint32_t lo = loff[tid];
    dabs[tid] = at + my_base - lo;
    const uint64_t room = cap > (uint64_t)my_base ? cap - my_base : 0;       // records of this segment that fit the region
    plim[tid] = (uint16_t)min((uint64_t)lo + room, (uint64_t)0xFFFFu);
    static_assert(P1B_RCAP <= 0xFFFF, "arena positions fit 16 bits");
  }
  lds_barrier();

  // ---- D: copy out in bin order ----
  const uint32_t nrec_s = *nrec_p;
  const uint32_t nrec = min(nrec_s, (uint32_t)P1B_RCAP);
  for (uint32_t p = tid; p < nrec; p += P1_THREADS) {
    const uint4 rec = arena[p];
    const uint32_t b = bin_of<SUB>(rec.w);
    if (p < (uint32_t)plim[b]) v.rec1[dabs[b] + p] = rec;
    else l1_put<EX>(v, l1_reg(b, subreg), gbase[b] + (p - loff[b]), rec, k, canon != 0, t);   // region full: park / flag
  }
  if (nrec_s > (uint32_t)P1B_RCAP) {             // records beyond the LDS arena go to their reserved places
#pragma unroll
    for (int tr = 0; tr < P1B_TR; ++tr) {
      if (rk[tr] != 0xFFFFFFFFu) {
        const uint32_t b = bin_of<SUB>(rc[tr].w);
        if (loff[b] + rk[tr] >= (uint32_t)P1B_RCAP)
          l1_put<EX>(v, l1_reg(b, subreg), gbase[b] + rk[tr], rc[tr], k, canon != 0, t);
      }
    }
  }
}

template <int W, int P1B_TR, bool SUB>
__global__ __launch_bounds__(P1_THREADS, ((P1B_TR <= 6 && !SUB) ? 6 : 4)) void msp_p1b_kernel(const int8_t *__restrict__ data,
                                                             int64_t nN, int k, int m, int canon,
                                                             int64_t tile0, MspView v, TableView t) {
  __shared__ uint4 pool[P1Lds<P1B_TR, SUB>::BYTES / 16];
  const int64_t tile = tile0 + blockIdx.x;
  p1_tile<W, P1B_TR, true, SUB>(pool, data, nN, k, m, canon, tile, (uint32_t)tile & (v.nxg - 1), v, t);
}

// ---------------------------------------------------------------------------------------- P2
// sub-bin of a record inside its level-1 bin: leaf low byte x {truncated run, complete run}.
// Complete runs get a stream of their own so that P3 can count them per RECORD with every lane
// busy, instead of k-mer by k-mer.
// PERSISTENT grid of nwg workgroups (a multiple of 8).  Workgroup w belongs to XCD group w % 8 and
// that group walks ITS bins (b1 = group, group + 8, ...) one after the other: the group's work items
// -- P2_GROUP consecutive tiles of one bin -- are numbered bin after bin from the records the bins
// really hold (no empty items), and workgroup j of the group takes items j, j + nwg/8, ...  At any
// moment the workgroups of one XCD append to the leaf streams of one or two level-1 bins: a leaf
// stream is only ever written from one XCD and its frontier sectors merge in L2 (P2: 11.6 -> 10.0 ms).
// Two streams per leaf: truncated runs (class 0) and complete runs (class 1).  The leaf kernel
// sorts records by length itself, so finer classes would only shorten P2's write segments.
constexpr int NCLS = 2;
constexpr int NSUB = NCLS * B2;                           // 512 sub-bins of a level-1 bin
__device__ __forceinline__ uint32_t sub_of(uint32_t w) { return (((w >> 8) & (B2 - 1)) << 1) | ((((w >> 6) & 3u) == 3u) ? 1u : 0u); }

// LDS of one P2 workgroup with tiles of P2_THREADS * PER records (byte offsets into its pool)
template <int PER>
struct P2Lds {
  static constexpr int TILE = P2_THREADS * PER;
  static constexpr int DABS = TILE * 16;                   // u64[NSUB]
  static constexpr int HIST = DABS + NSUB * 8;             // u32[NSUB]
  static constexpr int LOFF = HIST + NSUB * 4;             // u32[NSUB]
  static constexpr int PLIM = LOFF + NSUB * 4;             // u16[NSUB]
  static constexpr int WTOT = PLIM + NSUB * 2;             // u32[P2_THREADS / 64]
  static constexpr int RPRE = WTOT + (P2_THREADS / 64) * 4;   // u32[NXG + 1]
  static constexpr int GPRE = RPRE + (NXG + 1) * 4;        // u32[B1 / NXCD + 1] (persistent form only)
  static constexpr int BYTES = (GPRE + (B1 / NXCD + 1) * 4 + 15) / 16 * 16;
};

// One work item of the second-level kernel: the tiles [grp * NGRP, (grp + 1) * NGRP) of level-1 bin
// b1 (all of it that exists; nothing if the bin holds fewer records).
// PER = records per thread and tile: 8 stand-alone (two workgroups of 75 KB per CU), 5 in the fused
// kernel (the LDS of a P1 workgroup, 80 VGPRs); NGRP = consecutive tiles per item.
template <int PER, bool EX, bool PF, int NGRP>
__device__ __forceinline__ void p2_item(uint4 *pool, uint32_t b1, uint32_t grp, int k, int canon, const MspView &v, const TableView &t) {
  constexpr int TILE = P2_THREADS * PER;
  static_assert(TILE <= 0xFFFF, "tile positions fit 16 bits");
  uint4 *const sorted = pool;
  uint8_t *const pb = reinterpret_cast<uint8_t *>(pool);
  // copy-out fast path, once per stream and tile: sorted position p of sub-bin sb goes to record
  // dabs[sb] + p of the leaf buffer while p < plim[sb]
  unsigned long long *const dabs = reinterpret_cast<unsigned long long *>(pb + P2Lds<PER>::DABS);
  uint32_t *const hist = reinterpret_cast<uint32_t *>(pb + P2Lds<PER>::HIST);   // hist doubles as the fill cursor
  uint32_t *const loff = reinterpret_cast<uint32_t *>(pb + P2Lds<PER>::LOFF);
  uint16_t *const plim = reinterpret_cast<uint16_t *>(pb + P2Lds<PER>::PLIM);
  uint32_t *const wtot = reinterpret_cast<uint32_t *>(pb + P2Lds<PER>::WTOT);
  uint32_t *const rpre = reinterpret_cast<uint32_t *>(pb + P2Lds<PER>::RPRE);   // exclusive prefix of the bin's sub-region sizes
  static_assert(NSUB <= P2_THREADS && NSUB % 64 == 0, "the scan below gives one thread per sub-bin");
  static_assert(NXG <= 64, "one wave scans the sub-region sizes");
  const int tid = threadIdx.x;
  const bool exact1 = EX && v.exact1, exact2 = EX && v.exact;
  // The bin's sub-regions are read as ONE stream (region after region): tiles are full except the
  // bin's last one, however many cursors P1 spreads its appends over.
  if (tid < 64) {
    const uint32_t c = ((uint32_t)tid < v.nxg) ? (uint32_t)min((uint64_t)v.cnt1[l1_reg(b1, tid)],
                                                               exact1 ? (uint64_t)v.rcap[l1_reg(b1, tid)] : v.cap1) : 0u;
    const uint32_t incl = dev_wave_scan_incl(c);
    if ((uint32_t)tid < v.nxg) rpre[tid] = incl - c;
    if (tid == 63) rpre[v.nxg] = incl;
  }
  lds_barrier();
  const uint64_t n = rpre[v.nxg];
  // a workgroup takes NGRP consecutive tiles and asks for the next tile's records before it
  // sorts and writes the current one: the load latency hides under the LDS work
  const uint64_t g0r = (uint64_t)grp * NGRP * TILE;
  if (g0r >= n) return;
  // record `idx` of the bin's stream lies in the sub-region `sr` with rpre[sr] <= idx < rpre[sr + 1]: one
  // binary search for a thread's first record, then the cursor only moves forward (a thread's
  // indices grow from fetch to fetch)
  uint32_t sr = 0;
  {
    uint32_t hi = v.nxg;                         // invariant: rpre[sr] <= idx < rpre[hi]
    const uint64_t idx = g0r + tid;
    while (hi - sr > 1) {
      const uint32_t mid = (sr + hi) >> 1;
      if (rpre[mid] <= idx) sr = mid; else hi = mid;
    }
  }
  auto fetch = [&](uint64_t idx) {
    while (sr + 1 < v.nxg && rpre[sr + 1] <= idx) ++sr;
    if (exact1) return v.rec1[v.rbase[l1_reg(b1, sr)] + (idx - rpre[sr])];
    return v.rec1[(uint64_t)l1_reg(b1, sr) * v.cap1 + (idx - rpre[sr])];
  };
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  // PF: the next tile's records are asked for before the current tile is sorted and written (PER
  // more uint4 registers).  The fused kernel's copy of this role goes without: it has 80 VGPRs,
  // and while it waits for its loads the partitioning workgroups of its CU keep the vector pipe busy.
  uint4 nx[PF ? PER : 1];
  if (PF) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint64_t idx = g0r + (uint64_t)i * P2_THREADS + tid;
      nx[PF ? i : 0] = zero4;
      if (idx < n) nx[PF ? i : 0] = fetch(idx);
    }
  }
  for (int tt = 0; tt < NGRP; ++tt) {
    const uint64_t r0 = g0r + (uint64_t)tt * TILE;
    if (r0 >= n) break;
    const uint32_t nt = (uint32_t)min((uint64_t)TILE, n - r0);
    uint4 r[PER];
    if (PF) {
#pragma unroll
      for (int i = 0; i < PER; ++i) r[i] = nx[PF ? i : 0];
      if (tt + 1 < NGRP) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const uint64_t idx = r0 + TILE + (uint64_t)i * P2_THREADS + tid;
          nx[PF ? i : 0] = zero4;
          if (idx < n) nx[PF ? i : 0] = fetch(idx);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const uint64_t idx = r0 + (uint64_t)i * P2_THREADS + tid;
        r[i] = zero4;
        if (idx < n) r[i] = fetch(idx);
      }
    }
    if (tid < NSUB) hist[tid] = 0;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t idx = (uint32_t)i * P2_THREADS + tid;
      if (idx < nt) atomicAdd(&hist[sub_of(r[i].w)], 1u);
    }
    lds_barrier();
    uint32_t g0 = 0;
    {
      // per-stream reservation + exclusive scan of the sub-bin sizes (one per thread, tid < NSUB)
      const int lane = tid & 63, wave = tid >> 6;
      const bool mine = tid < NSUB;
      const uint32_t x0 = mine ? hist[tid] : 0u;
      // returning atomic: issued here, consumed after the sort (its latency flies under it)
      if (x0 && !(v.dbg & CFRK_ABL_P2_NO_ATOMIC)) g0 = atomicAdd(&v.cnt2[b1 * NSUB + tid], x0);
      const uint32_t incl = dev_wave_scan_incl(x0);
      if (lane == 63) wtot[wave] = incl;
      lds_barrier();
      uint32_t bs = 0;
      for (int w = 0; w < wave; ++w) bs += wtot[w];
      if (mine) {
        loff[tid] = bs + incl - x0;
        hist[tid] = 0;                                     // from here on: fill cursor
      }
      lds_barrier();
    }
    // counting sort of the tile by sub-bin, in LDS
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t idx = (uint32_t)i * P2_THREADS + tid;
      if (idx < nt) {
        const uint32_t sb = sub_of(r[i].w);
        sorted[loff[sb] + atomicAdd(&hist[sb], 1u)] = r[i];
      }
    }
    if (tid < NSUB) {
      const uint64_t leaf = ((uint64_t)b1 * NSUB + tid) >> 1;
      const uint32_t cls = tid & 1u;
      uint64_t cap = cls ? v.cap2c : v.cap2t;
      uint64_t at = (leaf >> v.sel_bits) * (v.cap2c + v.cap2t) + (cls ? 0 : v.cap2c);
      if (exact2) { cap = v.lcap[b1 * NSUB + tid]; at = v.lbase[b1 * NSUB + tid]; }
      const uint32_t lo = loff[tid];
      dabs[tid] = at + g0 - lo;
      const uint64_t room = cap > (uint64_t)g0 ? cap - g0 : 0;
      plim[tid] = (uint16_t)min((uint64_t)lo + room, (uint64_t)0xFFFFu);
    }
    lds_barrier();
    // copy out: consecutive lanes -> consecutive records of the same stream
    for (uint32_t p = tid; p < nt; p += P2_THREADS) {
      const uint4 rec = sorted[p];
      const uint32_t sb = sub_of(rec.w);
      if (v.dbg & CFRK_ABL_P2_LINEAR_OUT) {                // timing ablation: the tile goes out as it lies in LDS, bin after bin
        v.rec2[(uint64_t)b1 * (NSUB / 2) * (v.cap2c + v.cap2t) + r0 + p] = rec;
        continue;
      }
      if (p < (uint32_t)plim[sb]) { v.rec2[dabs[sb] + p] = rec; continue; }
      // the stream is full: park the record or raise the flag (below)
      const uint64_t leaf = ((uint64_t)b1 * NSUB + sb) >> 1;
      const uint32_t cls = sb & 1u;
      uint64_t cap = cls ? v.cap2c : v.cap2t;
      uint64_t at = (leaf >> v.sel_bits) * (v.cap2c + v.cap2t) + (cls ? 0 : v.cap2c);
      if (exact2) { cap = v.lcap[b1 * NSUB + sb]; at = v.lbase[b1 * NSUB + sb]; }
      const uint64_t dst = dabs[sb] + p - at;              // = the stream's reserved base + rank inside the segment
      // a stream position beyond 2^31: a single-key flood is on its way to wrapping the 32-bit cursor -- the host counts
      // the add through the HBM table instead (checked HERE, on the full-stream path only: a flood fills its stream first)
      if (dst >= 0x80000000ull) v.stats[ST_CWRAP] = 1;
      if (dst < cap) v.rec2[at + dst] = rec;
      else if (exact2) spill_record(rec, k, canon != 0, t);      // cannot happen: cap is the exact count
      else {
        // too small by a little: park the record (the host counts the parked ones through the
        // HBM table); by a lot: the cursors keep counting and the host redoes P2 with exact sizes
        // (once the flag is up nobody takes a number any more: 10^7 atomics on one word are slow)
        if (*(volatile uint64_t *)&v.stats[ST_L2OVF] == 0) {
          const unsigned long long o = atomicAdd((unsigned long long *)&v.stats[ST_OVFN], 1ull);
          if (o < v.ovf_cap) v.ovf[o] = rec;
          else v.stats[ST_L2OVF] = 1;
        }
      }
    }
    lds_barrier();                                       // sorted/loff/dabs are reused by the next tile
  }
}

// PERSISTENT form (stand-alone kernel): nwg workgroups (a multiple of 8); workgroup w belongs to XCD
// group w % 8 and that group walks ITS bins one after the other -- the group's items are numbered
// bin after bin from the records the bins really hold (no empty items), and workgroup j of the group
// takes items j, j + nwg/8, ...
template <int PER, bool EX, bool PF>
__device__ __forceinline__ void p2_role(uint4 *pool, uint32_t wg, uint32_t nwg, int k, int canon, const MspView &v, const TableView &t) {
  constexpr int TILE = P2_THREADS * PER;
  uint32_t *const gpre = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(pool) + P2Lds<PER>::GPRE);   // exclusive prefix of the group's items per bin
  constexpr int BPG = B1 / NXCD;                 // bins per XCD group
  static_assert(P2_THREADS == BPG * 16, "sixteen threads add up a bin's sub-regions");
  const int tid = threadIdx.x;
  const bool exact1 = EX && v.exact1;
  const uint32_t xg = wg & (NXCD - 1), j0 = wg / NXCD, nj = nwg / NXCD;
  // items per bin of this group: sixteen threads per bin add the sub-region sizes
  {
    const uint32_t i = (uint32_t)tid >> 4, part = (uint32_t)tid & 15u, b = xg + NXCD * i;
    uint32_t sum = 0;
    for (uint32_t sr = part; sr < v.nxg; sr += 16u) {
      const uint32_t reg = l1_reg(b, sr);
      sum += (uint32_t)min((uint64_t)v.cnt1[reg], exact1 ? (uint64_t)v.rcap[reg] : v.cap1);
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d);
    if (part == 0) gpre[i + 1] = (sum + (uint32_t)(TILE * P2_GROUP) - 1u) / (uint32_t)(TILE * P2_GROUP);
    if (tid == 0) gpre[0] = 0;
    lds_barrier();
    if (tid == 0) for (int q = 1; q <= BPG; ++q) gpre[q] += gpre[q - 1];
    lds_barrier();
  }
  const uint32_t nitems = gpre[BPG];
  uint32_t bi = 0;                               // bin of the current item (items grow: the cursor only moves forward)
  for (uint32_t item = j0; item < nitems; item += nj) {
    while (gpre[bi + 1] <= item) ++bi;
    p2_item<PER, EX, PF, P2_GROUP>(pool, xg + NXCD * bi, item - gpre[bi], k, canon, v, t);
  }
}

__global__ __launch_bounds__(P2_THREADS, 4) void msp_p2_kernel(int k, int canon, MspView v, TableView t) {
  __shared__ uint4 pool[(P2Lds<P2_PER>::BYTES + 15) / 16];
  p2_role<P2_PER, true, true>(pool, blockIdx.x, gridDim.x, k, canon, v, t);
}

// ---------------------------------------------------------------------------------------- P3
// rare path, kept out of line so that the hot loop stays small
__device__ __noinline__ void spill_kmer(TableView t, uint64_t key, uint32_t add) {
  if (key == CFRK_EMPTY_KEY) {   // k = 32, all T, forward strand
    atomicAdd((unsigned long long *)&t.stats[ST_ONES], (unsigned long long)add);
    return;
  }
  t.stats[ST_SPILLED] = 1;
  dev_count_event(&t.stats[ST_AUX1]);
  table_add1(t, key, add);
}

// The leaf's k-mer table is TS/2 buckets of two keys (one ds_read_b128 fetches a bucket), linear
// probing over buckets.
constexpr int NBUCKET = TS / 2;
__device__ __forceinline__ uint32_t lds_bucket(uint64_t key) {
  return (((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u) >> (32 - (TS_LOG - 1));
}

// Every instruction, vector or scalar, costs an issue slot, and a wave runs a probe loop for as
// long as its slowest lane: the step below is written for instruction count (one CAS-or-match
// decision per bucket, no per-lane probe counter), and callers feed waves with records of equal
// length.
constexpr int KT_TRIPS = 64;
constexpr int KT_TRIPS_SPLIT = 16;          // (buckets of two slots)

// One bucket attempt for one key per lane: count on a match, claim an empty slot, else move on.
// The lane's state is its bucket index b with KT_DONE or-ed in once the key is counted (or the
// lane had none): written to keep lane masks out of loop-carried values and control-flow merges
// (each costs the compiler three scalar instructions) -- the only branch is the rare claim, its
// outcome lands in a VGPR, and lanes with nothing to count add 0.
// SAT: the count saturates at CFRK_COUNT_MAX (table.h: sat_add) -- the merge of lists whose counts nothing bounds;
// the leaf kernel's own counts cannot overflow (HUGE_LEAF)
constexpr uint32_t KT_DONE = 0x80000000u;
template <bool SAT = false>
__device__ __forceinline__ void kt_try(unsigned long long *keys, uint32_t *cnts, uint64_t key, uint32_t &b,
                                       uint32_t add) {
  const bool p = (int32_t)b >= 0;
  const uint32_t bb = b & (NBUCKET - 1);
  const ulonglong2 q = reinterpret_cast<const ulonglong2 *>(keys)[bb];
  const bool m0 = q.x == key, m1 = q.y == key;
  const bool e0 = q.x == CFRK_EMPTY_KEY, e1 = q.y == CFRK_EMPTY_KEY;
  const bool hit = m0 || m1;
  const uint32_t s = 2 * bb + ((m0 || (!m1 && e0)) ? 0u : 1u);  // matching slot, else first empty, else any
  uint32_t won = 0u;
  if (p && !hit && (e0 || e1)) {
    const unsigned long long old = atomicCAS(&keys[s], (unsigned long long)CFRK_EMPTY_KEY, (unsigned long long)key);
    won = (old == CFRK_EMPTY_KEY || old == key) ? 1u : 0u;      // lost to another key: same bucket again
  }
  const bool ok = hit || won != 0u;
  if (SAT) {
    if (p && ok) {
      const uint32_t old = atomicAdd(&cnts[s], add);
      if (old > CFRK_COUNT_MAX - add) atomicExch(&cnts[s], CFRK_COUNT_MAX);
    }
  } else atomicAdd(&cnts[s], (p && ok) ? add : 0u);
  const bool full = !(hit || e0 || e1);
  const uint32_t nb = full ? ((bb + 1) & (NBUCKET - 1)) : bb;
  b = (p && !ok) ? nb : (b | KT_DONE);
}

// count one key per lane (valid lanes) `add` times, saturating; every lane of the wave must call
__device__ __forceinline__ void kt_count(unsigned long long *keys, uint32_t *cnts, uint64_t key, uint32_t add,
                                         bool valid, const TableView &t) {
  uint32_t b = lds_bucket(key) | (valid ? 0u : KT_DONE);
  add = min(add, CFRK_COUNT_MAX);
  for (int it = 0; it < KT_TRIPS && __ballot((int32_t)b >= 0); ++it) kt_try<true>(keys, cnts, key, b, add);
  if ((int32_t)b >= 0) spill_kmer(t, key, add);
}

// Subset of a leaf's key space (a leaf with more distinct k-mers than its LDS table holds is
// counted in several passes, each over the keys whose selector bits equal `val`): the selector
// comes from the same product as the bucket index, but from bits the bucket does not use.
struct KeySubset { uint32_t mask, val; };
__device__ __forceinline__ bool in_subset(uint64_t key, KeySubset ss) {
  return (((((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u) >> 8) & ss.mask) == ss.val;
}

// expand one record per lane (valid lanes), two k-mers per step; every lane of the wave must call.
// Only keys of subset `ss` are counted.  A key that finds no room is counted in the HBM table when
// ovf == nullptr; otherwise *ovf (an LDS flag) is raised and the caller redoes the subset in
// halves -- and waves that see the flag up stop working on a pass that is lost anyway.
template <bool CANON>
__device__ __forceinline__ void count_record_v2(unsigned long long *keys, uint32_t *cnts, uint4 rec, uint32_t add,
                                                bool valid, int k, uint64_t kmask, int rcsh,
                                                const TableView &t, KeySubset ss = KeySubset{0u, 0u},
                                                uint32_t *ovf = nullptr, int part = 0, int parts = 1,
                                                const uint8_t *tb = nullptr, uint32_t tb_n = 0u) {
  // tb[0 .. tb_n): lengths (in k-mers) of the truncated runs that are prefixes of this record: k-mer J
  // of the record is counted once more for every one of them that is longer than J
  if (ovf && *(volatile uint32_t *)ovf) return;
  // a record may be shared by `parts` lanes, each expanding a contiguous share of its k-mers
  const int nall = valid ? (int)(rec.w & 63u) + 1 : 0;
  const int per = (parts == 1) ? nall : (parts == 2) ? ((nall + 1) >> 1) : ((nall + 2) / 3);
  const int j0 = part * per;
  const int nk = max(min(nall, j0 + per) - j0, 0);
  uint64_t hi = ((uint64_t)rec.x << 32) | rec.y;
  uint64_t lo = (uint64_t)rec.z << 32;
  if (parts != 1 && j0) {                          // start at k-mer j0: 1 <= j0 <= 31
    hi = (hi << (2 * j0)) | (lo >> (64 - 2 * j0));
    lo <<= 2 * j0;
  }
  uint64_t fwd = hi >> (64 - 2 * k);
  uint64_t rc = CANON ? dev_revcomp64(fwd, k) : 0;
  uint64_t T = (k == 32) ? lo : ((hi << (2 * k)) | (lo >> (64 - 2 * k)));
  // (canonical counting) the lists are walked ONCE when every lane of the wave has at most 8 k-mers
  // to expand and at most 127 noted runs: a lane counts its list's entries by d = clamp(length - j0,
  // 0, 8) in nine 7-bit fields of one 64-bit register; k-mer j0 + q is counted once more for every
  // entry with d > q, i.e. the fields' suffix sums, kept as eight bytes (msp2.hip: count_record2).
  // The forward-strand instantiation keeps the loop per k-mer pair: it is out of registers.
  unsigned long long exq = 0ull;
  const bool tb_fast = CANON && tb && !__ballot(valid && (nk > 8 || tb_n > 127u));
  if (tb_fast) exq = noted_counts8(tb, tb_n, j0);
  for (int j = 0; __ballot(j < nk); j += 2) {
    uint32_t add0 = add, add1 = add;
    if (tb_fast) {
      add0 += (uint32_t)(exq >> (8 * (j & 7))) & 255u;
      add1 += (uint32_t)(exq >> (8 * ((j + 1) & 7))) & 255u;
    } else if (tb) {
      const uint32_t J = (uint32_t)(j0 + j);
      for (uint32_t e = 0; __ballot(e < tb_n); ++e) {       // (as many steps as the wave's longest list)
        const uint32_t tv = (e < tb_n) ? (uint32_t)tb[e] : 0u;
        add0 += (tv > J) ? 1u : 0u;
        add1 += (tv > J + 1u) ? 1u : 0u;
      }
    }
    const uint64_t key0 = (CANON && rc < fwd) ? rc : fwd;
    uint32_t nb = (uint32_t)(T >> 62);
    T <<= 2;
    fwd = ((fwd << 2) | nb) & kmask;
    if (CANON) rc = (rc >> 2) | ((uint64_t)(3u - nb) << rcsh);
    const uint64_t key1 = (CANON && rc < fwd) ? rc : fwd;
    nb = (uint32_t)(T >> 62);
    T <<= 2;
    fwd = ((fwd << 2) | nb) & kmask;
    if (CANON) rc = (rc >> 2) | ((uint64_t)(3u - nb) << rcsh);
    bool p0 = j < nk && in_subset(key0, ss), p1 = j + 1 < nk && in_subset(key1, ss);
    if (!CANON && k == 32) {
      // forward-strand all-T 32-mer collides with the EMPTY marker: side counter
      if (p0 && key0 == CFRK_EMPTY_KEY) { spill_kmer(t, key0, add0); p0 = false; }
      if (p1 && key1 == CFRK_EMPTY_KEY) { spill_kmer(t, key1, add1); p1 = false; }
    }
    uint32_t b0 = lds_bucket(key0) | (p0 ? 0u : KT_DONE), b1 = lds_bucket(key1) | (p1 ? 0u : KT_DONE);
    // (a pass that may still be split gives up early: probing a nearly full table is the slow way
    //  to find out that it is full)
    // (... but not the pass over the WHOLE leaf: an unlucky cluster in a table that is a third full must not
    //  split a leaf that fits -- a split leaf has two segments in the result list and no entry in the leaf
    //  index, which the export by leaf needs: at C3's size a handful of the 65 536 leaves were split)
    const int trips = ovf ? (ss.mask == 0u ? KT_TRIPS : KT_TRIPS_SPLIT) : KT_TRIPS;
    for (int it = 0; it < trips && __ballot((int32_t)(b0 & b1) >= 0); ++it) {
      kt_try(keys, cnts, key0, b0, add0);
      kt_try(keys, cnts, key1, b1, add1);
    }
    if (ovf) {
      if ((int32_t)(b0 & b1) >= 0) *ovf = 1u;
    } else {
      if ((int32_t)b0 >= 0) spill_kmer(t, key0, add0);
      if ((int32_t)b1 >= 0) spill_kmer(t, key1, add1);
    }
  }
}

// Record table: at high coverage most complete super-k-mer records of a leaf are byte-identical
// (the same genome locus seen by many reads).  Entries are {R0,R1,R2,meta},
// meta = count << 6 | (n-1); meta == RT_EMPTY empty, RT_LOCK while the claiming lane writes the
// bases.
constexpr int RT_LOG = 10, RT = 1 << RT_LOG;
constexpr int TL_PER = 4, TL_CAP = TL_PER * 1024;     // length-sorted list of truncated runs (indices)
constexpr int FL_CAP = 1024;                          // truncated runs without a complete twin, when the others are anchored
static_assert(TL_CAP <= (1 << 14), "list positions are 14 bits");
static_assert(RT == P3_THREADS, "phase 2 lists the record table with one slot per thread");
constexpr uint32_t RT_LOCK = 0xFFFFFFFFu;

// (rotations decorrelate the three base words, one multiply spreads them into the top bits: the
// bases of a record are sequence, already close to uniform)
__device__ __forceinline__ uint32_t rtab_slot(uint4 rec, int log_slots = RT_LOG) {
  const uint32_t t = rec.x ^ __builtin_amdgcn_alignbit(rec.y, rec.y, 11) ^ __builtin_amdgcn_alignbit(rec.z, rec.z, 21) ^
                     ((rec.w & 63u) << 26);
  return (t * 0x9E3779B1u) >> (32 - log_slots);
}

// Every instruction -- vector or scalar -- costs an issue slot here, so the probe step is written
// for instruction count: identity is one OR-reduction of XORs (EMPTY and LOCK carry low header
// bits no record has, so they never compare equal), and the loop is bounded by a wave-uniform
// trip counter instead of per-lane probe counts.
constexpr uint32_t RT_EMPTY = 62u;                 // n-1 = 62 does not occur (n <= 32)
constexpr int RT_TRIPS = 96;

__device__ __forceinline__ uint32_t rtab_diff(uint4 e, uint4 rec) {
  return (e.x ^ rec.x) | (e.y ^ rec.y) | (e.z ^ rec.z) | ((e.w ^ rec.w) & 63u);
}

// The record table of the ordinary path is keyed by the record's FIRST k-mer: a truncated run that
// is a prefix of a complete run shares it, so it finds its complete twin by probing from the same
// slot (a suffix is a prefix of the other strand's twin after a reverse complement).
__device__ __forceinline__ uint32_t rtab_slot_k(uint4 rec, int k, int log_slots) {
  const uint32_t y = (k >= 32) ? rec.y : (rec.y & ~(0xFFFFFFFFu >> (2 * k - 32)));   // 16 <= k: the top 2k-32 bits of y
  const uint32_t t = rec.x ^ __builtin_amdgcn_alignbit(y, y, 11);
  return (t * 0x9E3779B1u) >> (32 - log_slots);
}
// do the first `len` bases (32 <= 2*len <= 96 bits) of two records agree?
__device__ __forceinline__ bool rec_prefix_equal(uint4 e, uint4 r, int len) {
  const int rb = 2 * len - 32;                     // bits beyond the first word: 0 .. 64
  const uint32_t my = (rb >= 32) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> rb);
  const uint32_t mz = (rb <= 32) ? 0u : ((rb >= 64) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (rb - 32)));
  return ((e.x ^ r.x) | ((e.y ^ r.y) & my) | ((e.z ^ r.z) & mz)) == 0u;
}
// reverse complement of a record's run (len bases, 16 <= len <= 48); header word unchanged
__device__ __forceinline__ uint4 revcomp_record(uint4 rec, int len) {
  auto rcw = [](uint32_t x) {                      // reverse the 16 bases of a word and complement them
    x = __brev(x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return ~x;
  };
  // the 48-base field reversed: the reverse complement preceded by 48 - len complemented pad bases
  const uint64_t A = ((uint64_t)rcw(rec.z) << 32) | rcw(rec.y);
  const uint64_t B = (uint64_t)rcw(rec.x) << 32;
  const int s = 2 * (48 - len);                    // 0 .. 64: shift the pad out at the top
  uint64_t hi, lo;
  if (s == 0) { hi = A; lo = B; }
  else if (s < 64) { hi = (A << s) | (B >> (64 - s)); lo = B << s; }
  else { hi = B; lo = 0; }
  uint4 out = rec;
  out.x = (uint32_t)(hi >> 32); out.y = (uint32_t)hi; out.z = (uint32_t)(lo >> 32);
  return out;
}

// insert-or-count one record per lane; the lane's state is its slot h with RT_DONE or-ed in once
// the record is placed (lanes without a record start that way).  On return lanes still without
// RT_DONE found no place.
constexpr uint32_t RT_DONE = 0x80000000u;
// inc = the record's weight << 6: 1 << 6 for a record straight from a read, its multiplicity when
// the stream already holds deduplicated runs (the owner side of the multi-GPU exchange)
__device__ __forceinline__ void rtab_insert_loop(uint4 *rtab, uint4 rec, uint32_t &h, uint32_t mask = RT - 1,
                                                 int trips = RT_TRIPS, uint32_t inc = 1u << 6) {
  uint32_t *rmeta = reinterpret_cast<uint32_t *>(rtab);
  const uint32_t nm1 = rec.w & 63u;
  for (int it = 0; it < trips && __ballot((int32_t)h >= 0); ++it) {
    const bool p = (int32_t)h >= 0;
    const uint32_t hh = h & mask;
    const uint4 e = rtab[hh];
    const bool match = rtab_diff(e, rec) == 0u;
    const bool empty = e.w == RT_EMPTY;
    uint32_t won = 0u;
    if (p && empty) {
      if (atomicCAS(&rmeta[4 * hh + 3], RT_EMPTY, RT_LOCK) == RT_EMPTY) {
        rmeta[4 * hh + 0] = rec.x; rmeta[4 * hh + 1] = rec.y; rmeta[4 * hh + 2] = rec.z;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicExch(&rmeta[4 * hh + 3], inc | nm1);
        won = 1u;
      }
    }
    atomicAdd(&rmeta[4 * hh + 3], (p && match) ? inc : 0u);
    // an empty slot lost to another lane, or a locked one, is read again; a slot holding another
    // record sends the lane on
    const bool stay = match || empty || e.w == RT_LOCK;
    const uint32_t nh = stay ? hh : ((hh + 1) & mask);
    h = (p && !match && won == 0u) ? nh : (h | RT_DONE);
  }
}

// Second chance for a leaf whose complete runs do not fit the record table (a small k has short
// windows, hence ~2/(W+1) distinct runs per locus and strand: ~600 per leaf at k = 21 where k = 31
// has ~300, and heavy leaves several times that).  Before phase 2 the k-mer table is still empty,
// so the whole LDS pool (k-mer table + record table, 64 KB) serves as one 4096-slot record table;
// the distinct runs with their multiplicities then overwrite the head of the leaf's complete-run
// stream in HBM -- this workgroup is its only reader and has read all of it -- and phase 2 expands
// them from there.  Returns the number of distinct runs, or ~0 if even that table is too small
// (the leaf is then counted k-mer by k-mer from its streams).  Out of line: a rare path that would
// otherwise cost the hot path registers.
constexpr int BT_LOG = 12, BT = 1 << BT_LOG, BT_TRIPS = 192;
constexpr int BT_MIN_RUNS = 2 * BT;
// occupied slots of the pool-wide record table -> head of the stream; returns their number
__device__ __noinline__ uint32_t p3_compact(uint4 *pool, uint4 *stream, uint32_t *wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t *meta = reinterpret_cast<const uint32_t *>(pool);
  // a thread owns BT / P3_THREADS consecutive slots (two passes over LDS: an array of entries
  // would live in scratch memory)
  constexpr int PER = BT / P3_THREADS;
  uint32_t mine = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) mine += (meta[4 * (PER * tid + i) + 3] != RT_EMPTY) ? 1u : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < P3_THREADS / 64; ++w) {
    const uint32_t x = wsum[w];
    base += (w < wave) ? x : 0u;
    total += x;
  }
  uint32_t at = base + incl - mine;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint4 e = pool[PER * tid + i];
    if (e.w != RT_EMPTY) stream[at++] = e;
  }
  // the list is read back by the waves of this workgroup only: workgroup scope (one CU, whose
  // vector cache its own stores keep current) -- an agent-scope release would write back the L2
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return total;
}
// occupied slots of the ordinary record table (one per thread) -> head of the stream; returns their number
__device__ __noinline__ uint32_t p3_dump_rtab(const uint4 *rtab, uint4 *stream, uint32_t *wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint4 e = rtab[tid];
  const bool occ = e.w != RT_EMPTY;
  const unsigned long long m = __ballot(occ);
  if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < P3_THREADS / 64; ++w) {
    const uint32_t x = wsum[w];
    base += (w < wave) ? x : 0u;
    total += x;
  }
  if (occ) stream[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = e;
  return total;
}

__device__ __noinline__ uint32_t p3_big_dedupe(uint4 *pool, uint4 *stream, uint64_t n1, uint32_t *wsum, uint32_t *fail, bool weighted) {
  const int tid = threadIdx.x;
  __syncthreads();                               // every thread has read *fail (that is why we are here)
  for (int s = tid; s < BT; s += P3_THREADS) pool[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
  if (tid == 0) *fail = 0u;
  __syncthreads();
  for (uint64_t r = tid; r < ((n1 + 63) & ~63ull); r += P3_THREADS) {
    const bool valid = r < n1;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (valid) rec = stream[r];
    uint32_t h = valid ? rtab_slot(rec, BT_LOG) : RT_DONE;
    rtab_insert_loop(pool, rec, h, BT - 1, BT_TRIPS, weighted ? ((rec.w >> 6) << 6) : (1u << 6));
    if ((int32_t)h >= 0) *fail = 1u;
  }
  __syncthreads();
  if (*fail) return 0xFFFFFFFFu;
  return p3_compact(pool, stream, wsum);
}

// HUGE LEAF: 2^25 records or more in one leaf (half a gigabyte: a single-key flood -- a homopolymer, one amplicon
// sequenced 10^8 times).  The record table keeps a run's multiplicity in 26 bits and the LDS k-mer counts are 32-bit:
// below this size neither can overflow (2^25 records x 32 k-mers < 2^31); at or above it the leaf is not counted in
// LDS at all but k-mer by k-mer in the HBM table, whose adds SATURATE (table.h: sat_add).  Slow and exact.
// (A sender of the runs exchange does not deduplicate a leaf of 2^19 complete runs or more, so that the multiplicities
// an owner adds up for one run stay below 64 ranks x 2^19 + 2^25 = 2^26.)
constexpr uint64_t HUGE_LEAF = 1ull << 25;
constexpr uint64_t HUGE_LEAF_SENDER = 1ull << 19;
// (a kernel of its own, launched behind the leaf kernel: every thread looks at ONE leaf's sizes and normally finds
//  nothing -- a few microseconds; inside the leaf kernel the same loop cost the canonical instantiation four more
//  spilled dwords)
__global__ __launch_bounds__(256) void msp_huge_leaves_kernel(int k, int canon, int weighted, uint32_t nl /* leaves of this pass / owner */, MspView v, TableView t) {
  __shared__ uint32_t nhuge, huge[256];
  if (threadIdx.x == 0) nhuge = 0;
  __syncthreads();
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
  if (i0 < nl) {
    const uint32_t leaf = (i0 << v.sel_bits) | v.sel_val;
    const uint64_t nt = min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
    const uint64_t n1 = min((uint64_t)v.cnt2[NCLS * leaf + 1], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 1] : v.cap2c);
    if (nt + n1 >= HUGE_LEAF) huge[atomicAdd(&nhuge, 1u)] = i0;
  }
  __syncthreads();
  for (uint32_t q = 0; q < nhuge; ++q) {
    const uint32_t i = huge[q], leaf = (i << v.sel_bits) | v.sel_val;
    const uint64_t nt = min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
    const uint64_t n1 = min((uint64_t)v.cnt2[NCLS * leaf + 1], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 1] : v.cap2c);
    const uint4 *c1 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 1] : v.rec2 + (uint64_t)i * (v.cap2c + v.cap2t);
    const uint4 *c0 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 0] : c1 + v.cap2c;
    for (uint64_t j = threadIdx.x; j < n1 + nt; j += 256) {
      const uint4 rec = (j < n1) ? c1[j] : c0[j - n1];
      spill_record(rec, k, canon != 0, t, (weighted && j < n1) ? (rec.w >> 6) : 1u);
    }
  }
}

// mode: P3_WEIGHTED = the complete stream holds DISTINCT runs, header word = multiplicity << 6 | n-1
// (what msp_dedupe_export_kernel leaves behind; from several ranks): an owner counting the runs it received.
// SHARED: 2^sub_bits workgroups per leaf, each taking the records whose sub-value -- the extra
// minimizer-hash bits in the header's top byte, written by msp_p1b_kernel<.., true> -- names it.  Every
// occurrence of a k-mer has the same minimizer, so the workgroups' key sets are disjoint and every record
// is expanded once, where key-subset passes expand every record in every pass (the c5 shape at k = 31:
// 15 000 distinct k-mers per leaf).  The workgroups of a leaf run next to each other on one XCD, whose L2
// serves all but the first read of the leaf's streams (msp2.hip: msp2_p3_kernel<.., true>).
// P3_EXPORT = stop after the deduplication and leave the leaf's distinct complete runs at the head of its
// stream (leaf_n = their number).  The host no longer asks for it (msp_dedupe_export_kernel does that job
// with eight workgroups per CU), but the compiler allocates the hot loops' registers better with the
// branch in place: without it the canonical kernel spills six dwords instead of two and P3 takes 7.10
// instead of 6.80 ms on C3.
constexpr uint32_t P3_EXPORT = 1u, P3_WEIGHTED = 2u;
// LISTS (round 5, the owner side of the PIPELINED runs exchange): the leaf's runs are not two streams in rec2 but the
// N lists its ranks sent, read IN PLACE from the receive buffer -- rank r's segment of this group starts with a header
// (row 0: {rows used, leaves, first local leaf, magic}, then one uint4 {row offset, distinct, truncated, noted} per local
// leaf), and a leaf's rows are [distinct complete runs, header = multiplicity << 6 | n-1][truncated runs][notes, eight per
// row].  A note is turned back into the run it stands for (the first n k-mers of its twin) where it is read.  No layout
// kernels, no scatter pass, no host synchronisation before the leaf kernel.  An instantiation of its own
// (msp_p3_lists_kernel): the one-GPU kernel's code and registers are what they were.
template <bool CANON, bool SHARED, bool LISTS>
__device__ __forceinline__ void p3_body(int k, uint32_t mode, const MspView &v, const TableView &t, const P3ListsT<LISTS> &lx) {
  // k-mer table (keys, counts) and record table in one allocation: p3_big_dedupe uses all of it
  __shared__ uint4 pool[BT];
  static_assert(BT * 16 == TS * 12 + RT * 16, "the pool is exactly the k-mer table plus the record table");
  unsigned long long *const keys = reinterpret_cast<unsigned long long *>(pool);
  uint32_t *const cnts = reinterpret_cast<uint32_t *>(pool + TS / 2);
  uint4 *const rtab = pool + TS / 2 + TS / 4;
  __shared__ uint16_t occ_list[RT];
  // truncated runs: either a length-sorted list of stream positions (tlist), or -- with the record
  // table resident -- their lengths grouped by the complete twin they are a prefix of (th: per
  // record-table slot the group's start, tbytes: the lengths) plus the few without a twin (flist)
  __shared__ uint32_t tmem[RT + 1 + TL_CAP / 4 + 2];
  uint16_t *const tlist = reinterpret_cast<uint16_t *>(tmem);
  uint32_t *const th = tmem;
  uint8_t *const tbytes = reinterpret_cast<uint8_t *>(tmem + RT + 1);
  static_assert(sizeof(uint16_t) * TL_CAP <= sizeof(uint32_t) * (RT + 1 + TL_CAP / 4), "tlist fits the same memory");
  __shared__ uint16_t flist[FL_CAP];
  __shared__ uint32_t nfb, nfl;
  __shared__ uint32_t nhist[32], thist[32];
  __shared__ uint32_t wsum[P3_THREADS / 64];
  __shared__ uint32_t nocc;
  __shared__ uint32_t wg_total;
  __shared__ unsigned long long wg_base;
  __shared__ uint32_t rt_fail;                   // the record table ran out of room: second chance, then count from the streams
  __shared__ uint32_t kovf;                      // the k-mer table ran out of room in this pass
  __shared__ uint32_t stk[40];                   // key subsets still to count: bits << 16 | value
  __shared__ int sp;
  __shared__ uint32_t leaf_total, nseg;
  const int tid = threadIdx.x, lane = tid & 63;
  // (the grid holds the leaves of this pass only: were the others launched and left at once, the
  //  pass's leaves -- equal low bits -- would all sit on 8 / 2^sel_bits of the 8 XCDs)
  // (shared leaves: workgroup b goes to XCD b % 8 -- eight leaves side by side, a leaf's workgroups one after the other)
  const uint32_t sub_bits = SHARED ? v.sub_bits : 0u, smask = (1u << sub_bits) - 1u;
  const uint32_t vq = blockIdx.x >> 3, rsel = vq & smask;
  const uint32_t leaf = ((SHARED ? (((vq >> sub_bits) << 3) | (blockIdx.x & 7u)) : blockIdx.x) << v.sel_bits) | v.sel_val;
  auto mine = [&](uint32_t w) { return !SHARED || ((w >> 24) & smask) == rsel; };
  // LISTS: rank r's lists of this leaf -- l_base[r] = its first row, complete runs [l_cpre[r], l_cpre[r+1]) of the leaf's
  // complete "stream", truncated runs + notes [l_tpre[r], l_tpre[r+1]) of its truncated one (l_nu[r] records, then the notes)
  __shared__ uint32_t l_cpre[LISTS ? 65 : 1], l_tpre[LISTS ? 65 : 1], l_nu[LISTS ? 64 : 1];
  __shared__ const uint4 *l_base[LISTS ? 64 : 1];
  uint64_t nt_ = 0, n1_ = 0;
  if constexpr (LISTS) {
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
          const uint4 e = hdr[1u + blockIdx.x];
          const uint64_t tot = (uint64_t)e.y + e.z + (e.w + (uint32_t)NOTES_PER_ROW - 1u) / (uint32_t)NOTES_PER_ROW;
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
    n1_ = l_cpre[64]; nt_ = l_tpre[64];
  } else {
    nt_ = min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);  // truncated runs
    n1_ = min((uint64_t)v.cnt2[NCLS * leaf + 1], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 1] : v.cap2c);  // complete runs
  }
  const uint64_t nt = nt_, n1 = n1_;
  const uint4 *leaf_rec = LISTS ? nullptr : (v.exact ? v.rec2 + v.lbase[NCLS * leaf + 1] : v.rec2 + (uint64_t)(leaf >> v.sel_bits) * (v.cap2c + v.cap2t));
  // record i of the leaf's complete runs / of its truncated runs, wherever they lie
  auto l_find = [&](const uint32_t *pre, uint32_t i) {        // rank r with pre[r] <= i < pre[r + 1] (LISTS)
    uint32_t lo = 0, hi = 64;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= i) lo = mid; else hi = mid; }
    return lo;
  };
  auto ld_c = [&](uint64_t i) -> uint4 {
    if constexpr (LISTS) { const uint32_t r = l_find(l_cpre, (uint32_t)i); return l_base[r][(uint32_t)i - l_cpre[r]]; }
    else return leaf_rec[i];
  };
  if constexpr (!LISTS) {
    if (nt + n1 - 1ull >= HUGE_LEAF - 1ull) return;     // nothing to count -- or a HUGE leaf: msp_huge_leaves_kernel counts it
  }
  // Short windows (k < 28) mean more distinct runs per leaf than the record table holds (~600 at
  // k = 21, heavy leaves several times that): the complete runs are then deduplicated in a table
  // over the whole pool right away and listed in the stream (see p3_big_dedupe / p3_compact).
  // Both uses of the pool-wide table need duplicates to pay for the extra pass: a leaf with
  // fewer than BT_MIN_RUNS complete runs cannot hold many copies of more than ~10^3 distinct ones
  // (low coverage of a large genome: the stream path is the better fallback there).
  // (a shared leaf never rewrites its stream: the other workgroups are reading it)
  const bool big_first = !SHARED && !LISTS && k < 28 && n1 >= (uint64_t)BT_MIN_RUNS;   // (lists are read in place: nothing to rewrite)
  uint4 *const tab = big_first ? pool : rtab;
  const int tab_log = big_first ? BT_LOG : RT_LOG;
  const uint32_t tab_mask = (1u << tab_log) - 1u;
  const int tab_trips = big_first ? BT_TRIPS : RT_TRIPS;
  if (!SHARED && (mode & P3_EXPORT) && n1 == 0) return;      // nothing to deduplicate: the truncated runs leave as they are
  // DISPLACED-RUN CACHE.  At load 0.3 a sixth of the distinct runs do not sit in their home slot of the
  // record table, and every copy of such a run (~50 each at C3's depth) took the slow path: compaction
  // across the wave plus the probe loop -- there is such a record in nearly every wave-step.  While the
  // stream is scanned the k-mer table's memory is idle: it holds a direct-mapped cache of DC entries
  // (same entry format, own hash) in which a displaced run is installed by the probe loop that placed or
  // found it.  A record that misses its home slot looks there next (one more LDS read) and counts in the
  // cache entry; after the scan the cache's counts are added to the table's entries and the memory
  // becomes the k-mer table.  Only first sightings and cache conflicts still take the slow path.
  constexpr int DC = TS / 2 + TS / 4;            // uint4 entries in the k-mer table's memory (3072)
  const bool use_cache = !big_first;
  uint4 *const dcache = pool;
  if (big_first) {
    for (int s = tid; s < BT; s += P3_THREADS) pool[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
  } else {
    for (int s = tid; s < DC; s += P3_THREADS) dcache[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
    for (int s = tid; s < RT; s += P3_THREADS) rtab[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
  }
  if (tid == 0) { wg_total = 0; nocc = 0; rt_fail = 0; kovf = 0; sp = 0; leaf_total = 0; nseg = 0; nfb = 0; nfl = 0; }
  if (tid < 32) { nhist[tid] = 0; thist[tid] = 0; }
  for (int s = tid; s < RT + 1; s += P3_THREADS) th[s] = 0;
  __syncthreads();

  const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1ull);
  const int rcsh = 2 * k - 2;
  // truncated runs that go through the length-sorted list: as many as fit
  const uint4 *trunc = LISTS ? nullptr : (v.exact ? v.rec2 + v.lbase[NCLS * leaf + 0] : leaf_rec + v.cap2c);
  const uint32_t tl = (uint32_t)min(nt, (uint64_t)TL_CAP);
  auto ld_t = [&](uint64_t i) -> uint4 {
    if constexpr (LISTS) {
      const uint32_t r = l_find(l_tpre, (uint32_t)i), j = (uint32_t)i - l_tpre[r];
      const uint4 *b = l_base[r];
      const uint32_t nd_r = l_cpre[r + 1] - l_cpre[r], nu_r = l_nu[r];
      if (j < nu_r) return b[nd_r + j];
      // a note: the first n k-mers of its twin, closed on the left only (msp_runs_scatter_kernel did this in a pass of its own)
      const uint32_t note = reinterpret_cast<const uint16_t *>(b + nd_r + nu_r)[j - nu_r];
      const uint4 twin = b[min(note >> 5, nd_r - 1u)];                    // a position outside the list is not followed
      const uint32_t nm1 = min(note & 31u, twin.w & 31u);
      const int rb = 2 * ((int)nm1 + k) - 32;                           // bits of the run beyond the first word: 0 .. 64
      const uint32_t my = (rb >= 32) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> rb);
      const uint32_t mz = (rb <= 32) ? 0u : ((rb >= 64) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (rb - 32)));
      return make_uint4(twin.x, twin.y & my, twin.z & mz, 64u | nm1);
    } else return trunc[i];
  };
  if constexpr (LISTS) {
    if (nt + n1 == 0) return;
    if (nt + n1 >= HUGE_LEAF) {                        // (see HUGE_LEAF: counted in the HBM table, whose adds saturate)
      for (uint64_t i = tid; i < n1 + nt; i += P3_THREADS) {
        const uint4 rec = (i < n1) ? ld_c(i) : ld_t(i - n1);
        spill_record(rec, k, CANON, t, (i < n1) ? (rec.w >> 6) : 1u);
      }
      return;
    }
  }

  // ---- phase 1a: complete runs, one record-table update per record.  Most records find their
  //      twin in the home slot: that case is one LDS read + one LDS add with every lane busy.
  //      The rest (first sightings, displaced entries) are compacted into a wave-private
  //      leftover set, 64 lanes wide, and only a full set runs the probe loop -- the loop's trip
  //      count is the longest chain among its lanes, so it must not run for a handful of them.
  {
    const uint4 *src = leaf_rec;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const uint64_t n1a = n1;
    uint32_t *rmeta = reinterpret_cast<uint32_t *>(tab);
    uint4 L = zero4;                 // leftover records, lanes [0, c)
    uint32_t Lh = 0;
    int c = 0;                       // wave-uniform
    const bool weighted = !SHARED && (mode & P3_WEIGHTED) != 0u;   // (an owner's leaves are never shared)
    auto dc_slot = [&](const uint4 rec) {
      const uint32_t t_ = rec.y ^ __builtin_amdgcn_alignbit(rec.x, rec.x, 7) ^ __builtin_amdgcn_alignbit(rec.z, rec.z, 19) ^ ((rec.w & 63u) << 25);
      return __umulhi(t_ * 0x85EBCA77u, (uint32_t)DC);
    };
    auto drain = [&](int cnt) {
      uint32_t h = Lh | ((lane < cnt) ? 0u : RT_DONE);
      rtab_insert_loop(tab, L, h, tab_mask, tab_trips, weighted ? ((L.w >> 6) << 6) : (1u << 6));
      // no room in the record table: a leaf with more distinct runs than it holds (low coverage
      // of a large genome).  Dedupe is pointless there: the whole leaf is counted from its streams.
      if ((int32_t)h >= 0) rt_fail = 1u;
      // a record that ended up away from its home slot goes into the cache (count 0: what the table's entry
      // holds stays there), so that its next copies find it with one look instead of coming through here
      if (use_cache && lane < cnt && (int32_t)h < 0 && ((h ^ Lh) & tab_mask) != 0u) {
        uint32_t *dmeta = reinterpret_cast<uint32_t *>(dcache);
        const uint32_t cs = dc_slot(L);
        if (atomicCAS(&dmeta[4 * cs + 3], RT_EMPTY, RT_LOCK) == RT_EMPTY) {
          dmeta[4 * cs + 0] = L.x; dmeta[4 * cs + 1] = L.y; dmeta[4 * cs + 2] = L.z;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          atomicExch(&dmeta[4 * cs + 3], L.w & 63u);
        }
      }
    };
    auto home = [&](const uint4 rec, bool valid) {
      const uint32_t h = big_first ? rtab_slot(rec, tab_log) : rtab_slot_k(rec, k, tab_log);
      const uint4 e = tab[h];
      const bool match = valid && rtab_diff(e, rec) == 0u;
      if (match) atomicAdd(&rmeta[4 * h + 3], weighted ? ((rec.w >> 6) << 6) : (1u << 6));
      bool left = valid && !match;
      if (use_cache && __ballot(left)) {
        uint32_t *dmeta = reinterpret_cast<uint32_t *>(dcache);
        const uint32_t cs = dc_slot(rec);
        const uint4 ce = dcache[cs];
        const bool chit = left && rtab_diff(ce, rec) == 0u;
        if (chit) atomicAdd(&dmeta[4 * cs + 3], weighted ? ((rec.w >> 6) << 6) : (1u << 6));
        left = left && !chit;
      }
      const unsigned long long mask = __ballot(left);
      if (mask == 0ull) return;
      const int n = __popcll(mask);
      if (c + n > 64) { drain(c); c = 0; }
      uint32_t set[5] = {L.x, L.y, L.z, L.w, Lh};
      const uint32_t mine_[5] = {rec.x, rec.y, rec.z, rec.w, h};
      wave_append<5>(set, mine_, left, mask, c, n);
      L = make_uint4(set[0], set[1], set[2], set[3]); Lh = set[4];
      c += n;
    };
    if (!SHARED) {
      for (uint64_t r = tid; r < ((n1a + 63) & ~63ull) && !(v.dbg & CFRK_ABL_P3_NO_RTAB); r += 2ull * P3_THREADS) {
        const uint64_t r1 = r + P3_THREADS;
        const bool v0 = r < n1a, v1 = r1 < n1a;
        uint4 rec0 = zero4, rec1 = zero4;
        if (v0) rec0 = LISTS ? ld_c(r) : src[r];
        if (v1) rec1 = LISTS ? ld_c(r1) : src[r1];
        home(rec0, v0);
        home(rec1, v1);
      }
    } else {
      // a shared leaf: most records read here are another workgroup's.  The one lane in 2^sub_bits that
      // holds a record of this workgroup is gathered into full sets of 64 before the table look-up:
      // home() costs the wave ~100 instructions however few lanes take part (msp2.hip)
      uint4 Cr = zero4;
      int cc = 0;                        // wave-uniform
      auto cfeed = [&](const uint4 rec, bool keep) {
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) return;
        const int n = __popcll(mask);
        if (cc + n > 64) { home(Cr, lane < cc); cc = 0; }
        uint32_t set[4] = {Cr.x, Cr.y, Cr.z, Cr.w};
        const uint32_t mine_[4] = {rec.x, rec.y, rec.z, rec.w};
        wave_append<4>(set, mine_, keep, mask, cc, n);
        Cr = make_uint4(set[0], set[1], set[2], set[3]);
        cc += n;
      };
      for (uint64_t r = tid; r < ((n1a + 63) & ~63ull); r += 2ull * P3_THREADS) {
        const uint64_t r1 = r + P3_THREADS;
        const bool v0 = r < n1a, v1 = r1 < n1a;
        uint4 rec0 = zero4, rec1 = zero4;
        if (v0) rec0 = src[r];
        if (v1) rec1 = src[r1];
        cfeed(rec0, v0 && mine(rec0.w));
        if (r1 < ((n1a + 63) & ~63ull)) cfeed(rec1, v1 && mine(rec1.w));       // (wave-uniform)
      }
      if (cc) home(Cr, lane < cc);
    }
    if (c) drain(c);
    if ((v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW) && !big_first) rt_fail = 1u;
  }
  __syncthreads();
  if (use_cache) {
    // the cache's counts go to the table's entries (the run is there: it was placed before it was cached;
    // a table that overflowed is not used at all), then the memory becomes the k-mer table
    uint32_t *rmeta = reinterpret_cast<uint32_t *>(rtab);
    for (int s = tid; s < DC && rt_fail == 0u; s += P3_THREADS) {
      const uint4 ce = dcache[s];
      if (ce.w == RT_EMPTY || (ce.w >> 6) == 0u) continue;
      uint32_t hh = rtab_slot_k(ce, k, RT_LOG);
      for (int it = 0; it < RT; ++it) {
        const uint4 e = rtab[hh];
        if (rtab_diff(e, ce) == 0u) { atomicAdd(&rmeta[4 * hh + 3], (ce.w >> 6) << 6); break; }
        if (e.w == RT_EMPTY) break;                // (cannot happen)
        hh = (hh + 1u) & (uint32_t)(RT - 1);
      }
    }
    __syncthreads();
    if (SHARED || !(mode & P3_EXPORT))             // (an exporting leaf never touches its k-mer table)
      for (int s = tid; s < TS; s += P3_THREADS) { keys[s] = CFRK_EMPTY_KEY; cnts[s] = 0; }
    __syncthreads();
  }
  // ---- phase 2: k-mer by k-mer -- every distinct complete record of the record table (weight =
  //      its multiplicity) and the truncated runs (weight 1).  Both are first listed SORTED BY
  //      LENGTH (counting sort of 16-bit indices in LDS): a wave expands 64 records in lock-step
  //      for as many steps as its longest one, so equal lengths keep every lane busy.
  uint32_t nd = 0xFFFFFFFFu;                     // distinct runs listed in the stream (second chance), ~0: none
  if (!SHARED && (mode & P3_EXPORT)) {
    uint4 *const stream = const_cast<uint4 *>(leaf_rec);
    const bool weighted = (mode & P3_WEIGHTED) != 0u;
    if (big_first) { if (rt_fail == 0u) nd = p3_compact(pool, stream, wsum); }
    else if (rt_fail == 0u) nd = p3_dump_rtab(rtab, stream, wsum);
    else if (n1 >= (uint64_t)BT_MIN_RUNS || (v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW)) nd = p3_big_dedupe(pool, stream, n1, wsum, &rt_fail, weighted);
    if (nd == 0xFFFFFFFFu) {
      // nothing to merge (more distinct runs than any table here holds): the runs leave as they are, multiplicity 1
      if (!weighted)
        for (uint64_t i = tid; i < n1; i += P3_THREADS) stream[i].w = (1u << 6) | (stream[i].w & 63u);
      nd = (uint32_t)n1;
    }
    if (tid == 0) v.leaf_n[leaf] = nd;
    return;
  }
  if (big_first || rt_fail != 0u) {              // (rt_fail read after the barrier above: uniform)
    if (!SHARED && !LISTS && !big_first && (n1 >= (uint64_t)BT_MIN_RUNS || (v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW))) nd = p3_big_dedupe(pool, const_cast<uint4 *>(leaf_rec), n1, wsum, &rt_fail, (mode & P3_WEIGHTED) != 0u);
    else if (!big_first) __syncthreads();        // (as below)
    else if (rt_fail == 0u) nd = p3_compact(pool, const_cast<uint4 *>(leaf_rec), wsum);
    else __syncthreads();                        // (the pool is cleared below: everybody has read rt_fail and the table)
    for (int s = tid; s < TS; s += P3_THREADS) { keys[s] = CFRK_EMPTY_KEY; cnts[s] = 0; }
    for (int s = tid; s < RT; s += P3_THREADS) rtab[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
    __syncthreads();
  }
  const bool listed = nd != 0xFFFFFFFFu;
  const bool big = !listed && rt_fail != 0u;
  // Truncated runs (read ends) are most of the leaf's k-mer insertions, one by one with weight 1.
  // But a run that is cut on ONE side is a prefix -- or, read on the other strand, a suffix -- of
  // the complete run of its locus, which sits in the record table: it is looked up there by its
  // first k-mer, and only its LENGTH is noted with its twin.  The twin's expansion then counts its
  // first t k-mers once more: one table lookup per truncated run instead of one per k-mer.
  // Runs without a twin (cut on both sides, shallow loci, forward-strand suffixes) stay k-mer by k-mer.
  const bool anchors_on = !big && !listed && !(v.dbg & CFRK_DEBUG_NO_ANCHORS);
  bool use_anchors = false;
  {
    // (a) record table: occupied slots, longest first (not when the leaf is counted from its streams)
    const uint4 e = rtab[tid];                       // RT == P3_THREADS
    const bool occ = !big && e.w != RT_EMPTY;
    uint32_t rank = 0;
    if (occ) rank = atomicAdd(&nhist[e.w & 31u], 1u);
    // (b) truncated runs: the first TL_CAP of the stream, by position
    constexpr uint32_t TW_NONE = 0xFFFFFFFFu, TW_TWIN = 0x80000000u;
    uint32_t tw[TL_PER], trank[TL_PER];              // n-1, or TW_TWIN | twin slot << 8 | n
    // (all of a thread's records are asked for before the first is looked up: one HBM round trip instead of
    //  TL_PER -- the compiler put a full wait behind every load, 4 x ~1.5 us per leaf with two leaves per CU)
    uint4 trec[TL_PER];
#pragma unroll
    for (int i = 0; i < TL_PER; ++i) trec[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tl) {                                        // (uniform; lanes past the end load the last record again: no
#pragma unroll                                       //  per-lane branch, so the four loads are in flight together)
      for (int i = 0; i < TL_PER; ++i) trec[i] = ld_t(min((uint32_t)(i * P3_THREADS + tid), tl - 1u));
    }
#pragma unroll
    for (int i = 0; i < TL_PER; ++i) {
      const uint32_t g = (uint32_t)(i * P3_THREADS + tid);
      bool valid = g < tl;
      tw[i] = TW_NONE; trank[i] = 0u;
      uint4 rec = trec[i];
      if (SHARED) valid = valid && mine(rec.w);
      const uint32_t nm1 = rec.w & 31u;
      if (valid) tw[i] = nm1;
      if (anchors_on && __ballot(valid)) {
        const bool lc = (rec.w & 64u) != 0u, rc_ = (rec.w & 128u) != 0u;
        const bool suf = CANON && valid && !lc && rc_;
        if (suf) rec = revcomp_record(rec, (int)nm1 + k);
        const bool anchored = suf || (valid && lc && !rc_);
        uint32_t h = anchored ? rtab_slot_k(rec, k, RT_LOG) : RT_DONE;
        uint32_t found = TW_NONE;
        for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
          const bool p = (int32_t)h >= 0;
          const uint32_t hh = h & (uint32_t)(RT - 1);
          const uint4 e2 = rtab[hh];
          const bool empty = e2.w == RT_EMPTY;
          // the twin holds at least as many k-mers and starts with the same nm1 + k bases
          const bool hit = p && !empty && (e2.w & 31u) >= nm1 && rec_prefix_equal(e2, rec, (int)nm1 + k);
          found = hit ? hh : found;
          h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(RT - 1)) : (h | RT_DONE);
        }
        if (found != TW_NONE) tw[i] = TW_TWIN | (found << 8) | (nm1 + 1u);
        const unsigned long long fb = __ballot(valid && found == TW_NONE);
        if (lane == 0 && fb) atomicAdd(&nfb, (uint32_t)__popcll(fb));
      }
    }
    __syncthreads();
    use_anchors = anchors_on && nfb <= (uint32_t)FL_CAP;
    if (use_anchors) {
#pragma unroll
      for (int i = 0; i < TL_PER; ++i) {
        if (tw[i] == TW_NONE) continue;
        if (tw[i] & TW_TWIN) trank[i] = atomicAdd(&th[(tw[i] >> 8) & (uint32_t)(RT - 1)], 1u);
        else flist[atomicAdd(&nfl, 1u)] = (uint16_t)(i * P3_THREADS + tid);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TL_PER; ++i) {
        if (tw[i] == TW_NONE) continue;
        const uint32_t g = (uint32_t)(i * P3_THREADS + tid);
        tw[i] = ld_t(g).w & 31u;                     // (a twin that was found is not used)
        trank[i] = atomicAdd(&thist[tw[i]], 1u);
      }
    }
    __syncthreads();
    if (tid < 64) {
      // exclusive prefix over DEscending length, both histograms (lanes 0..31 / 32..63)
      const uint32_t own = (tid < 32) ? nhist[31 - tid] : thist[63 - tid];
      uint32_t incl = own;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d, 32);
        if ((tid & 31) >= d) incl += y;
      }
      if (tid < 32) nhist[31 - tid] = incl - own; else thist[63 - tid] = incl - own;
      if (tid == 31) nocc = incl;
      if (tid == 63 && !use_anchors) nfl = incl;       // entries of the length-sorted list (a shared leaf: this workgroup's only)
    }
    uint32_t goff = 0;
    if (use_anchors) {
      // group starts: exclusive prefix of the groups' sizes over the record table's slots
      const uint32_t own = th[tid];
      uint32_t incl = own;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      uint32_t base = 0;
      for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
      goff = base + incl - own;
      __syncthreads();                               // (everybody has read its group's size)
      th[tid] = goff;
      if (tid == P3_THREADS - 1) th[RT] = goff + own;
    }
    __syncthreads();
    if (occ) occ_list[nhist[e.w & 31u] + rank] = (uint16_t)tid;
    if (use_anchors) {
#pragma unroll
      for (int i = 0; i < TL_PER; ++i)
        if (tw[i] != TW_NONE && (tw[i] & TW_TWIN)) tbytes[th[(tw[i] >> 8) & (uint32_t)(RT - 1)] + trank[i]] = (uint8_t)(tw[i] & 63u);
    } else {
#pragma unroll
      for (int i = 0; i < TL_PER; ++i)
        if (tw[i] != TW_NONE) tlist[thist[tw[i]] + trank[i]] = (uint16_t)(i * P3_THREADS + tid);
    }
    // Key subsets to count: the whole leaf in one pass -- or, when the record table overflowed (it
    // then held ~10^3 distinct runs, i.e. twice as many distinct k-mers as the k-mer table takes),
    // four quarters to begin with.
    if (tid == 0) {
      if (big) {
        // as many key subsets to begin with as the leaf's size suggests (one per ~6000 records, 4 .. 32)
        uint32_t b0 = 2u;
        while (b0 < 5u && ((nt + n1) >> b0) > 6000ull) ++b0;
        for (uint32_t q = 0; q < (1u << b0); ++q) stk[q] = (b0 << 16) | q;
        sp = (int)(1u << b0);
      } else { stk[0] = 0u; sp = 1; }
    }
  }
  __syncthreads();
  constexpr uint32_t SUBSET_BITS_MAX = 8;        // 256 passes at most; beyond: the HBM table
  bool first_pass = true;
  while (true) {
    const int depth = sp;                        // (every thread reads the same value: barriers around)
    if (depth == 0) break;
    const uint32_t item = stk[depth - 1];
    __syncthreads();
    if (tid == 0) { sp = depth - 1; kovf = 0; wg_total = 0; }
    const uint32_t bits = item >> 16;
    const KeySubset ss{(1u << bits) - 1u, item & 0xFFFFu};
    uint32_t *ovf = (bits < SUBSET_BITS_MAX) ? &kovf : nullptr;
    if (!first_pass)
      for (int s = tid; s < TS; s += P3_THREADS) { keys[s] = CFRK_EMPTY_KEY; cnts[s] = 0; }
    first_pass = false;
    __syncthreads();
    {
      if (listed) {
        // distinct complete runs with their multiplicities, from the head of the stream
        for (uint32_t i = tid; i < ((nd + 63u) & ~63u); i += P3_THREADS) {
          const bool valid = i < nd;
          uint4 rec = make_uint4(0u, 0u, 0u, 0u);
          if (valid) rec = leaf_rec[i];
          count_record_v2<CANON>(keys, cnts, rec, rec.w >> 6, valid, k, kmask, rcsh, t, ss, ovf);
        }
      } else if (!big) {
        // ~320 distinct runs for 1024 lanes, and a wave works for as many steps as its longest run
        // has k-mers: up to three lanes share a record, each expanding a share of its k-mers
        const uint32_t nlist = (v.dbg & CFRK_ABL_P3_NO_CEXP) ? 0u : nocc;
        const int parts = (nlist * 3u <= (uint32_t)P3_THREADS) ? 3 : (nlist * 2u <= (uint32_t)P3_THREADS) ? 2 : 1;
        const uint32_t nitems = nlist * (uint32_t)parts;
        for (uint32_t i = tid; i < ((nitems + 63u) & ~63u); i += P3_THREADS) {
          const bool valid = i < nitems;
          const uint32_t ri = (parts == 1) ? i : (parts == 2) ? (i >> 1) : (i / 3u);
          uint4 rec = make_uint4(0u, 0u, 0u, 0u);
          uint32_t slot = 0;
          if (valid) { slot = occ_list[ri]; rec = rtab[slot]; }
          if (use_anchors) {
            // ... plus one for every truncated run of this locus that reaches the k-mer
            const uint32_t g0 = th[slot];
            count_record_v2<CANON>(keys, cnts, rec, rec.w >> 6, valid, k, kmask, rcsh, t, ss, ovf,
                                   (int)(i - ri * (uint32_t)parts), parts, tbytes + g0, valid ? th[slot + 1] - g0 : 0u);
          } else {
            count_record_v2<CANON>(keys, cnts, rec, rec.w >> 6, valid, k, kmask, rcsh, t, ss, ovf,
                                   (int)(i - ri * (uint32_t)parts), parts);
          }
        }
      } else {
        // the complete runs straight from their stream, weight 1 each
        for (uint64_t i = tid; i < ((n1 + 63) & ~63ull); i += P3_THREADS) {
          const bool valid = i < n1;
          uint4 rec = make_uint4(0u, 0u, 0u, 0u);
          if (valid) rec = ld_c(i);
          count_record_v2<CANON>(keys, cnts, rec, (!SHARED && (mode & P3_WEIGHTED)) ? (rec.w >> 6) : 1u, valid && mine(rec.w), k, kmask, rcsh, t, ss, ovf);
        }
      }
      // truncated runs: those without a twin (anchored leaf), or all of the sorted list
      const uint32_t ntr = nfl;
      for (uint32_t i = tid; i < ((ntr + 63u) & ~63u) && !(v.dbg & CFRK_ABL_P3_NO_TRUNC); i += P3_THREADS) {
        const bool valid = i < ntr;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (valid) rec = ld_t(use_anchors ? flist[i] : tlist[i]);
        count_record_v2<CANON>(keys, cnts, rec, 1u, valid, k, kmask, rcsh, t, ss, ovf);
      }
      // truncated runs beyond the sorted list (very large leaves): in stream order
      for (uint64_t i = (uint64_t)tl + tid; i < ((nt + 63) & ~63ull) && tl < nt; i += P3_THREADS) {
        const bool valid = i < nt;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (valid) rec = ld_t(i);
        count_record_v2<CANON>(keys, cnts, rec, 1u, valid && mine(rec.w), k, kmask, rcsh, t, ss, ovf);
      }
    }
    __syncthreads();
    if (kovf) {
      // this subset does not fit the table either: its two halves replace it (the table is discarded)
      if (tid == 0) {
        const int d0 = sp;
        stk[d0] = ((bits + 1u) << 16) | ss.val;
        stk[d0 + 1] = ((bits + 1u) << 16) | ss.val | (1u << bits);
        sp = d0 + 2;
      }
      __syncthreads();
      continue;
    }

    // ---- compact occupied slots to the output list: ONE cursor atomic per workgroup and pass (a
    //      global atomic per wave on the single cursor word serialises the whole grid)
    constexpr int NIT = TS / P3_THREADS;
    uint32_t wbase[NIT];
    if (v.dbg & CFRK_ABL_P3_NO_OUT) continue;
    auto occupied = [&](int s) { return keys[s] != CFRK_EMPTY_KEY; };
    wg_rank_slots<NIT, P3_THREADS>(wbase, &wg_total, occupied);
    __syncthreads();
    if (tid == 0) {
      wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)wg_total);
      // a leaf counted in several passes has several segments in the list: no per-leaf index then
      // (a shared leaf: one index entry per sub-value, (leaf << sub_bits) | sub-value)
      uint32_t sg = leaf;
      if (SHARED) sg = (leaf << sub_bits) | rsel;
      if (!LISTS) {                                  // (an owner's result is not kept in leaf form)
        if (nseg == 0) v.leaf_off[sg] = wg_base;
        else if (wg_total) v.stats[ST_MULTISEG] = 1;
      }
      if (wg_total) nseg = nseg + 1;
      leaf_total += wg_total;
      if (!LISTS) v.leaf_n[sg] = leaf_total;
    }
    __syncthreads();
    wg_emit_slots<NIT, P3_THREADS>(wbase, wg_base, occupied, [&](int s, unsigned long long dst) {
      if (dst < v.out_cap) { v.out_keys[dst] = keys[s]; v.out_cnt[dst] = cnts[s]; }
      else v.stats[ST_OVERFLOW] = 1;
    });
    __syncthreads();
  }
}

template <bool CANON, bool SHARED>
__global__ __launch_bounds__(P3_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void msp_p3_kernel(int k, uint32_t mode, MspView v, TableView t) {
  p3_body<CANON, SHARED, false>(k, mode, v, t, P3ListsT<false>{});
}
// the owner of the pipelined runs exchange: workgroup b counts local leaf lx.ll0 + b from the lists its ranks sent
template <bool CANON>
__global__ __launch_bounds__(P3_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void msp_p3_lists_kernel(int k, MspView v, TableView t, P3ListsT<true> lx) {
  p3_body<CANON, false, true>(k, P3_WEIGHTED, v, t, lx);
}

// ---------------------------------------------------------------------------- multi-GPU by leaf
// Leaves are disjoint in key space on EVERY rank (same k -> same minimizer -> same leaf), so the
// owner of a key can be the owner of its leaf: owner(leaf) = leaf % parts.  Each rank ships its
// leaves' (key,count) lists to their owners, and the owner adds the lists of one leaf in an LDS
// table exactly like P3 does -- no HBM atomics in the merge.

// copy every leaf's entries into owner-major order (dst_off from the host's prefix sum)
// (out_hi: high key words of a two-word result, msp2.hip; NULL otherwise)
__global__ __launch_bounds__(256) void msp_gather_kernel(MspView v, const uint64_t *__restrict__ dst_off,
                                                         uint64_t *__restrict__ out_keys,
                                                         uint64_t *__restrict__ out_hi,
                                                         uint32_t *__restrict__ out_cnt) {
  const uint32_t leaf = blockIdx.x;
  uint64_t dof = dst_off[leaf];
  for (uint32_t j = 0; j < (1u << v.seg_bits); ++j) {          // (one segment unless the leaf was shared by record)
    const uint32_t sg = (leaf << v.seg_bits) | j;
    const uint32_t n = v.leaf_n[sg];
    const uint64_t so = v.leaf_off[sg];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      out_keys[dof + i] = v.out_keys[so + i];
      if (out_hi) out_hi[dof + i] = v.out_hi[so + i];
      out_cnt[dof + i] = v.out_cnt[so + i];
    }
    dof += n;
  }
}

// one workgroup per owned leaf: add the `parts` incoming lists of that leaf in LDS
__global__ __launch_bounds__(P3_THREADS) void msp_merge_kernel(const uint64_t *__restrict__ in_keys,
                                                               const uint32_t *__restrict__ in_cnt,
                                                               const uint64_t *__restrict__ seg_off,
                                                               const uint32_t *__restrict__ seg_n,
                                                               int parts, int leaves_per_part,
                                                               MspView v, TableView t) {
  __shared__ unsigned long long keys[TS];
  __shared__ uint32_t cnts[TS];
  __shared__ uint32_t wg_total;
  __shared__ unsigned long long wg_base;
  const int tid = threadIdx.x;
  const uint32_t ll = blockIdx.x;                       // local leaf index on this owner
  uint32_t total = 0;
  for (int p = 0; p < parts; ++p) total += seg_n[(size_t)p * leaves_per_part + ll];
  if (total == 0) return;
  for (int s = tid; s < TS; s += P3_THREADS) { keys[s] = CFRK_EMPTY_KEY; cnts[s] = 0; }
  if (tid == 0) wg_total = 0;
  __syncthreads();
  for (int p = 0; p < parts; ++p) {
    const uint32_t n = seg_n[(size_t)p * leaves_per_part + ll];
    const uint64_t off = seg_off[(size_t)p * leaves_per_part + ll];
    for (uint32_t i = tid; i < ((n + 63u) & ~63u); i += P3_THREADS) {
      const bool valid = i < n;
      const uint64_t key = valid ? in_keys[off + i] : 0ull;
      const uint32_t c = valid ? in_cnt[off + i] : 0u;
      kt_count(keys, cnts, key, c, valid && c != 0u, t);
    }
  }
  __syncthreads();
  constexpr int NIT = TS / P3_THREADS;
  uint32_t wbase[NIT];
  auto occupied = [&](int s) { return keys[s] != CFRK_EMPTY_KEY; };
  wg_rank_slots<NIT, P3_THREADS>(wbase, &wg_total, occupied);
  __syncthreads();
  if (tid == 0) wg_base = atomicAdd((unsigned long long *)&v.stats[ST_CURSOR], (unsigned long long)wg_total);
  __syncthreads();
  wg_emit_slots<NIT, P3_THREADS>(wbase, wg_base, occupied, [&](int s, unsigned long long dst) {
    if (dst < v.out_cap) { v.out_keys[dst] = keys[s]; v.out_cnt[dst] = cnts[s]; }
    else v.stats[ST_OVERFLOW] = 1;
  });
}

// The exporting side of the runs exchange needs none of the leaf kernel's k-mer machinery: a small
// workgroup per leaf (256 threads, 16 KB of LDS: eight or more per CU) deduplicates the leaf's
// complete runs in the 1024-entry record table and leaves the distinct ones, header word =
// multiplicity << 6 | n-1, at the head of the leaf's own stream (leaf_n = their number).  A shard
// of a multi-GPU job holds ~2000 complete runs per leaf; the 65536 workgroups are mostly fixed cost.
// A leaf with more distinct runs than the table holds leaves as it is (multiplicity 1 each): the
// owner's table merges what can be merged.
// Truncated runs (read ends: two per read, each unique -- most of what a rank would ship) are then
// looked up in the same table, exactly as the leaf kernel anchors them: one that is a prefix of a
// distinct complete run of this rank (a suffix, read on the other strand; canonical counting only)
// is replaced by a 16-bit NOTE, position of that run in the leaf's list << 5 | n-1.  The owner
// rebuilds the run from its twin.  A noted record is marked in place (w = RUN_NOTED, x = the note);
// leaf_off[leaf] = how many were noted.
constexpr int DX_THREADS = 256, DX_INFL = 4;
__global__ __launch_bounds__(DX_THREADS) void msp_dedupe_export_kernel(int k, int canon, MspView v) {
  __shared__ uint4 rtab[RT];
  __shared__ uint16_t sidx[RT];                    // record-table slot -> position in the leaf's list
  __shared__ uint32_t wsum[DX_THREADS / 64];
  __shared__ uint32_t rt_fail, noted;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t leaf = blockIdx.x;
  const uint64_t n1 = min((uint64_t)v.cnt2[NCLS * leaf + 1], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 1] : v.cap2c);
  if (n1 == 0) return;
  uint4 *const stream = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 1] : v.rec2 + (uint64_t)(leaf >> v.sel_bits) * (v.cap2c + v.cap2t);
  for (int s = tid; s < RT; s += DX_THREADS) rtab[s] = make_uint4(0u, 0u, 0u, RT_EMPTY);
  // (a leaf of 2^19 complete runs or more leaves undeduplicated: see HUGE_LEAF)
  const bool too_many = n1 >= HUGE_LEAF_SENDER;
  if (tid == 0) { rt_fail = ((v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW) || too_many) ? 1u : 0u; noted = 0u; }
  __syncthreads();
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  if (!too_many) {
    uint32_t *rmeta = reinterpret_cast<uint32_t *>(rtab);
    uint4 L = zero4;                 // leftover records, lanes [0, c) (see the leaf kernel's phase 1a)
    uint32_t Lh = 0;
    int c = 0;                       // wave-uniform
    auto drain = [&](int cnt) {
      uint32_t h = Lh | ((lane < cnt) ? 0u : RT_DONE);
      rtab_insert_loop(rtab, L, h, RT - 1, RT_TRIPS);
      if ((int32_t)h >= 0) rt_fail = 1u;
    };
    // DX_INFL records per thread are asked for before the first is looked up: a shard's leaf holds a few
    // thousand records, i.e. ~10 trips of one dependent load each -- the kernel was waiting for HBM
    for (uint64_t r0 = 0; r0 < n1; r0 += (uint64_t)DX_INFL * DX_THREADS) {
      uint4 recs[DX_INFL];
#pragma unroll
      for (int u = 0; u < DX_INFL; ++u) {
        const uint64_t r = r0 + (uint64_t)u * DX_THREADS + tid;
        recs[u] = zero4;
        if (r < n1) recs[u] = stream[r];
      }
#pragma unroll
      for (int u = 0; u < DX_INFL; ++u) {
      const uint64_t r = r0 + (uint64_t)u * DX_THREADS + tid;
      const bool valid = r < n1;
      const uint4 rec = recs[u];
      const uint32_t h = rtab_slot_k(rec, k, RT_LOG);
      const uint4 e = rtab[h];
      const bool match = valid && rtab_diff(e, rec) == 0u;
      if (match) atomicAdd(&rmeta[4 * h + 3], 1u << 6);
      const bool left = valid && !match;
      const unsigned long long mask = __ballot(left);
      if (mask == 0ull) continue;
      const int n = __popcll(mask);
      if (c + n > 64) { drain(c); c = 0; }
      uint32_t set[5] = {L.x, L.y, L.z, L.w, Lh};
      const uint32_t mine_[5] = {rec.x, rec.y, rec.z, rec.w, h};
      wave_append<5>(set, mine_, left, mask, c, n);
      L = make_uint4(set[0], set[1], set[2], set[3]); Lh = set[4];
      c += n;
      }
    }
    if (c) drain(c);
  }
  __syncthreads();
  uint32_t nd;
  if (rt_fail) {
    // (every record was read before the barrier; the rewrite touches the header word only)
    for (uint64_t i = tid; i < n1; i += DX_THREADS) stream[i].w = (1u << 6) | (stream[i].w & 63u);
    nd = (uint32_t)n1;
  } else {
    // occupied slots -> head of the stream (four slots per thread)
    // (two passes over LDS: an array of entries would live in scratch memory)
    constexpr int PER = RT / DX_THREADS;
    const uint32_t *meta = reinterpret_cast<const uint32_t *>(rtab);
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mine += (meta[4 * (PER * tid + i) + 3] != RT_EMPTY) ? 1u : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < DX_THREADS / 64; ++w) { const uint32_t x = wsum[w]; base += (w < wave) ? x : 0u; total += x; }
    uint32_t at = base + incl - mine;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint4 e = rtab[PER * tid + i];
      sidx[PER * tid + i] = (uint16_t)at;
      if (e.w != RT_EMPTY) stream[at++] = e;
    }
    nd = total;
    __syncthreads();
    // truncated runs -> notes (the lookup of the leaf kernel's anchoring, msp_p3_kernel)
    const uint64_t nt = min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
    uint4 *const trunc = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 0] : stream + v.cap2c;
    for (uint64_t g0 = 0; g0 < nt && !(v.dbg & CFRK_DEBUG_NO_ANCHORS); g0 += (uint64_t)DX_INFL * DX_THREADS) {
      uint4 recs[DX_INFL];
#pragma unroll
      for (int u = 0; u < DX_INFL; ++u) {
        const uint64_t g = g0 + (uint64_t)u * DX_THREADS + tid;
        recs[u] = zero4;
        if (g < nt) recs[u] = trunc[g];
      }
#pragma unroll
      for (int u = 0; u < DX_INFL; ++u) {
      const uint64_t g = g0 + (uint64_t)u * DX_THREADS + tid;
      const bool valid = g < nt;
      uint4 rec = recs[u];
      const uint32_t nm1 = rec.w & 31u;
      const bool lc = (rec.w & 64u) != 0u, rc_ = (rec.w & 128u) != 0u;
      const bool suf = canon && valid && !lc && rc_;
      if (suf) rec = revcomp_record(rec, (int)nm1 + k);
      const bool anchored = suf || (valid && lc && !rc_);
      uint32_t h = anchored ? rtab_slot_k(rec, k, RT_LOG) : RT_DONE;
      uint32_t found = 0xFFFFFFFFu;
      for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
        const bool p = (int32_t)h >= 0;
        const uint32_t hh = h & (uint32_t)(RT - 1);
        const uint4 e2 = rtab[hh];
        const bool empty = e2.w == RT_EMPTY;
        const bool hit = p && !empty && (e2.w & 31u) >= nm1 && rec_prefix_equal(e2, rec, (int)nm1 + k);
        found = hit ? hh : found;
        h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(RT - 1)) : (h | RT_DONE);
      }
      const bool hit = found != 0xFFFFFFFFu;
      if (hit) { trunc[g].x = ((uint32_t)sidx[found] << 5) | nm1; trunc[g].w = RUN_NOTED; }
      const unsigned long long hb = __ballot(hit);
      if (lane == 0 && hb) atomicAdd(&noted, (uint32_t)__popcll(hb));
      }
    }
    __syncthreads();
  }
  if (tid == 0) { v.leaf_n[leaf] = nd; v.leaf_off[leaf] = noted; }
}

// ------------------------------------------------------------------- multi-GPU by runs, PIPELINED (round 5)
// The sender's half of the pipelined exchange: deduplication AND packing in one kernel, one group of leaves at a
// time, straight into the send buffer -- no rewrite of the streams, no sizes / plan / gather kernels, no host
// round trip (the gather alone was 0.21 ms of a 5.1 ms critical path at N = 8).
// The leaves of an owner are cut into `ngroups` ranges of local indices; group g of owner p has its segment at
// rows [(g * parts + p) * seg_cap, ...) of the send buffer: header (row 0 = {rows used, leaves, first local leaf,
// magic}, written by msp_runs_group_finish_kernel; then one uint4 {row offset, distinct, truncated, noted} per local
// leaf) followed by the leaves' rows in the order their workgroups CLAIM them (one atomic on the segment's cursor
// per leaf; the header says where each leaf went).  Only the used rows of a segment travel.
// A leaf's rows: [distinct complete runs, header word = multiplicity << 6 | n-1][truncated runs that found no twin]
// [notes: 16 bits each, position of the twin in the leaf's list << 5 | n-1, eight per row, padded with 0xFFFF].
// One workgroup per leaf, everything a thread needs of the complete stream in flight at once (DS_INFL loads: a
// shard's leaf is ~2700 records); the truncated runs are looked up once, their verdict (note or record) waits in LDS
// for the leaf's row count, then they are written from a second read that hits the L2.
// SPLIT RECORD TABLE.  With 16-byte entries {bases x 3, meta} the count word of every slot sits in LDS bank 3 mod 4: a
// wave's 64 ds_add on it land on 8 of the 32 banks and serialise 8 ways -- 62 % of this kernel's LDS-active cycles were
// bank conflicts, the LDS was busy for 76 % of its time (profiles/r05/dedupe_send_pmc_before_split_table.txt).  Here the
// bases (12 bytes per slot) and the meta words (4 bytes per slot) are two arrays: the adds spread over all banks.
constexpr int DS_THREADS = 256, DS_INFL = 12, DS_TCAP = 4096;
// insert-or-count one record per lane in the split table (rtab_insert_loop's logic); lanes still without RT_DONE found no place
__device__ __forceinline__ void ds_insert_loop(uint32_t *rb, uint32_t *rm, uint4 rec, uint32_t &h) {
  const uint32_t nm1 = rec.w & 63u;
  for (int it = 0; it < RT_TRIPS && __ballot((int32_t)h >= 0); ++it) {
    const bool p = (int32_t)h >= 0;
    const uint32_t hh = h & (uint32_t)(RT - 1);
    const uint32_t m = rm[hh];
    const uint32_t e0 = rb[3 * hh], e1 = rb[3 * hh + 1], e2 = rb[3 * hh + 2];
    const bool match = ((e0 ^ rec.x) | (e1 ^ rec.y) | (e2 ^ rec.z) | ((m ^ rec.w) & 63u)) == 0u;
    const bool empty = m == RT_EMPTY;
    uint32_t won = 0u;
    if (p && empty) {
      if (atomicCAS(&rm[hh], RT_EMPTY, RT_LOCK) == RT_EMPTY) {
        rb[3 * hh] = rec.x; rb[3 * hh + 1] = rec.y; rb[3 * hh + 2] = rec.z;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicExch(&rm[hh], (1u << 6) | nm1);
        won = 1u;
      }
    }
    atomicAdd(&rm[hh], (p && match) ? (1u << 6) : 0u);
    const bool stay = match || empty || m == RT_LOCK;
    const uint32_t nh = stay ? hh : ((hh + 1) & (uint32_t)(RT - 1));
    h = (p && !match && won == 0u) ? nh : (h | RT_DONE);
  }
}
__global__ __launch_bounds__(DS_THREADS) void msp_dedupe_send_kernel(int k, int canon, MspView v, RunsSend sg) {
  __shared__ uint32_t rb[3 * RT];                  // record table: the bases of slot s at rb[3 s .. 3 s + 2] ...
  __shared__ uint32_t rm[RT];                      // ... its meta word (count << 6 | n-1; RT_EMPTY, RT_LOCK) at rm[s]
  __shared__ uint16_t sidx[RT];                    // record-table slot -> position in the leaf's list
  __shared__ uint16_t tres[DS_TCAP];               // truncated run g: its note, or 0xFFFF = travels as a record
  __shared__ uint32_t wsum[DS_THREADS / 64];
  __shared__ uint32_t rt_fail, noted, cu, cn, row0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t leaf = sg.leaf0 + blockIdx.x;
  const uint32_t own = leaf % (uint32_t)sg.parts, ll = leaf / (uint32_t)sg.parts;
  uint4 *const seg = sg.packed + (uint64_t)own * sg.seg_cap;
  const uint32_t hrows = 1u + sg.lcount;
  uint4 *const entry = seg + 1u + (ll - sg.ll0);
  const uint64_t n1 = min((uint64_t)v.cnt2[NCLS * leaf + 1], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 1] : v.cap2c);
  const uint64_t nt64 = min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
  if (n1 + nt64 == 0) { if (tid == 0) *entry = make_uint4(0u, 0u, 0u, 0u); return; }
  const uint4 *const c1 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 1] : v.rec2 + (uint64_t)(leaf >> v.sel_bits) * (v.cap2c + v.cap2t);
  const uint4 *const c0 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 0] : c1 + v.cap2c;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  // (a leaf of 2^19 complete runs or more leaves undeduplicated: see HUGE_LEAF; a row count beyond 32 bits cannot be claimed)
  const bool too_many = n1 >= HUGE_LEAF_SENDER;
  const bool unclaimable = n1 + nt64 >= 0xFFFFFFFFull;
  // the first round of the complete stream is asked for before anything else is done
  uint4 recs[DS_INFL];
#pragma unroll
  for (int u = 0; u < DS_INFL; ++u) {
    const uint64_t r = (uint64_t)u * DS_THREADS + tid;
    recs[u] = zero4;
    if (r < n1 && !too_many) recs[u] = c1[r];
  }
  for (int s2 = tid; s2 < RT; s2 += DS_THREADS) rm[s2] = RT_EMPTY;
  if (tid == 0) { rt_fail = ((v.dbg & CFRK_DEBUG_FORCE_RT_OVERFLOW) || too_many) ? 1u : 0u; noted = 0u; cu = 0u; cn = 0u; }
  __syncthreads();
  if (!too_many) {
    uint4 L = zero4;                 // leftover records, lanes [0, c) (see the leaf kernel's phase 1a)
    uint32_t Lh = 0;
    int c = 0;                       // wave-uniform
    auto drain = [&](int cnt) {
      uint32_t h = Lh | ((lane < cnt) ? 0u : RT_DONE);
      ds_insert_loop(rb, rm, L, h);
      if ((int32_t)h >= 0) rt_fail = 1u;
    };
    for (uint64_t r0 = 0; r0 < n1; r0 += (uint64_t)DS_INFL * DS_THREADS) {
      if (r0) {
#pragma unroll
        for (int u = 0; u < DS_INFL; ++u) {
          const uint64_t r = r0 + (uint64_t)u * DS_THREADS + tid;
          recs[u] = zero4;
          if (r < n1) recs[u] = c1[r];
        }
      }
#pragma unroll
      for (int u = 0; u < DS_INFL; ++u) {
        const uint64_t r = r0 + (uint64_t)u * DS_THREADS + tid;
        if (r0 + (uint64_t)u * DS_THREADS >= n1) break;          // (wave-uniform)
        const bool valid = r < n1;
        const uint4 rec = recs[u];
        const uint32_t h = rtab_slot_k(rec, k, RT_LOG);
        const uint32_t m = rm[h];
        const uint32_t e0 = rb[3 * h], e1 = rb[3 * h + 1], e2 = rb[3 * h + 2];
        // (EMPTY and LOCK carry low bits no record has: they never compare equal; stale bases of an empty slot do not matter)
        const bool match = valid && ((e0 ^ rec.x) | (e1 ^ rec.y) | (e2 ^ rec.z) | ((m ^ rec.w) & 63u)) == 0u;
        if (match) atomicAdd(&rm[h], 1u << 6);
        const bool left = valid && !match;
        const unsigned long long mask = __ballot(left);
        if (mask == 0ull) continue;
        const int n = __popcll(mask);
        if (c + n > 64) { drain(c); c = 0; }
        uint32_t set[5] = {L.x, L.y, L.z, L.w, Lh};
        const uint32_t mine_[5] = {rec.x, rec.y, rec.z, rec.w, h};
        wave_append<5>(set, mine_, left, mask, c, n);
        L = make_uint4(set[0], set[1], set[2], set[3]); Lh = set[4];
        c += n;
      }
    }
    if (c) drain(c);
  }
  __syncthreads();
  const bool plain = rt_fail != 0u;                 // no deduplication: every complete run leaves with multiplicity 1, no notes
  const uint32_t nt = (uint32_t)min(nt64, (uint64_t)0xFFFFFFFFull);
  uint32_t nd, at0 = 0;
  constexpr int PER = RT / DS_THREADS;
  if (plain) {
    nd = (uint32_t)min(n1, (uint64_t)0xFFFFFFFFull);
  } else {
    // occupied slots -> positions in the leaf's list (four slots per thread)
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) mine += (rm[PER * tid + i] != RT_EMPTY) ? 1u : 0u;
    const uint32_t incl = dev_wave_scan_incl(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < DS_THREADS / 64; ++w) { const uint32_t x = wsum[w]; base += (w < wave) ? x : 0u; total += x; }
    at0 = base + incl - mine;
    uint32_t at = at0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      sidx[PER * tid + i] = (uint16_t)at;
      if (rm[PER * tid + i] != RT_EMPTY) ++at;
    }
    nd = total;
    __syncthreads();
    // truncated runs: which of the first DS_TCAP are a prefix of a distinct complete run of this rank (a suffix, read on
    // the other strand; canonical counting only)?  The verdict waits in LDS (the lookup of the leaf kernel's anchoring)
    const uint32_t tlook = (v.dbg & CFRK_DEBUG_NO_ANCHORS) ? 0u : min(nt, (uint32_t)DS_TCAP);
    for (uint32_t g0 = 0; g0 < tlook; g0 += DS_THREADS) {
      const uint32_t g = g0 + tid;
      const bool valid = g < tlook;
      uint4 rec = zero4;
      if (valid) rec = c0[g];
      const uint32_t nm1 = rec.w & 31u;
      const bool lc = (rec.w & 64u) != 0u, rc_ = (rec.w & 128u) != 0u;
      const bool suf = canon && valid && !lc && rc_;
      if (suf) rec = revcomp_record(rec, (int)nm1 + k);
      const bool anchored = suf || (valid && lc && !rc_);
      uint32_t h = anchored ? rtab_slot_k(rec, k, RT_LOG) : RT_DONE;
      uint32_t found = 0xFFFFFFFFu;
      for (int it = 0; it < 32 && __ballot((int32_t)h >= 0); ++it) {
        const bool p = (int32_t)h >= 0;
        const uint32_t hh = h & (uint32_t)(RT - 1);
        const uint4 e2 = make_uint4(rb[3 * hh], rb[3 * hh + 1], rb[3 * hh + 2], rm[hh]);
        const bool empty = e2.w == RT_EMPTY;
        const bool hit = p && !empty && (e2.w & 31u) >= nm1 && rec_prefix_equal(e2, rec, (int)nm1 + k);
        found = hit ? hh : found;
        h = (p && !hit && !empty) ? ((hh + 1u) & (uint32_t)(RT - 1)) : (h | RT_DONE);
      }
      const bool hit = found != 0xFFFFFFFFu;
      if (valid) tres[g] = hit ? (uint16_t)(((uint32_t)sidx[found] << 5) | nm1) : (uint16_t)0xFFFFu;
      const unsigned long long hb = __ballot(hit);
      if (lane == 0 && hb) atomicAdd(&noted, (uint32_t)__popcll(hb));
    }
    __syncthreads();
  }
  const uint32_t na = plain ? 0u : noted, nu = nt - na;
  const uint64_t rows = (uint64_t)nd + nu + (na + NOTES_PER_ROW - 1) / NOTES_PER_ROW;
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
    for (uint64_t i = tid; i < n1; i += DS_THREADS) { uint4 r = c1[i]; r.w = (1u << 6) | (r.w & 63u); dst[i] = r; }
    for (uint32_t i = tid; i < nt; i += DS_THREADS) dst[nd + i] = c0[i];
    return;
  }
  {
    uint32_t at = at0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t sl = PER * tid + i, m = rm[sl];
      if (m != RT_EMPTY) dst[at++] = make_uint4(rb[3 * sl], rb[3 * sl + 1], rb[3 * sl + 2], m);
    }
  }
  uint16_t *const notes = reinterpret_cast<uint16_t *>(dst + nd + nu);
  for (uint32_t i = tid; i < ((nt + 63u) & ~63u); i += DS_THREADS) {
    const bool valid = i < nt;
    const uint32_t note = (valid && i < (uint32_t)DS_TCAP && !(v.dbg & CFRK_DEBUG_NO_ANCHORS)) ? (uint32_t)tres[i] : 0xFFFFu;
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
    else if (valid) { const uint32_t at = bu + (uint32_t)__popcll(mu & below); if (at < nu) dst[nd + at] = c0[i]; }
  }
  const uint32_t pad = (NOTES_PER_ROW - na % NOTES_PER_ROW) % NOTES_PER_ROW;
  if ((uint32_t)tid < pad) notes[na + tid] = 0xFFFFu;
}

// ---------------------------------------------------------------------------- multi-GPU by runs
// Strong scaling: with the reads split over N ranks every rank still meets almost every locus, so
// counting on every rank and exchanging (key, count) lists (above) makes each rank expand every
// distinct run and ship ~D entries -- neither shrinks with N.  Here a rank only partitions and
// deduplicates (P1, P2, the leaf kernel in P3_EXPORT mode) and ships, per leaf, its distinct
// complete runs with multiplicities plus its truncated runs to the leaf's owner
// (owner(leaf) = leaf % parts); the owner lines the N lists of each of its leaves up as that
// leaf's streams and runs the ordinary leaf kernel on them (P3_WEIGHTED): expansion and counting
// happen once, for 1/N of the leaves.

// sender: what every leaf contributes -- n1 distinct complete runs, nt truncated runs as records, na as
// notes, rows in all -- one thread per leaf, coalesced (the plan kernel below used to gather these
// four words per leaf itself, three times over, 64 dependent strided loads per thread each time:
// 0.50 ms of a 5.9 ms critical path at N = 8)
__global__ __launch_bounds__(256) void msp_runs_sizes_kernel(MspView v, uint4 *__restrict__ sz, unsigned long long *__restrict__ plan_sync) {
  const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
  if (leaf < 72u) plan_sync[leaf] = 0ull;                      // (the plan kernel's look-back words)
  if (leaf >= (uint32_t)NLEAF) return;
  uint32_t n1 = 0, na = 0;
  uint32_t nt = (uint32_t)min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
  if (v.cnt2[NCLS * leaf + 1]) {                               // (a leaf without complete runs never wrote its counts)
    n1 = v.leaf_n[leaf];
    na = min((uint32_t)v.leaf_off[leaf], nt);
  }
  nt -= na;
  sz[leaf] = make_uint4(n1, nt, na, n1 + nt + (na + NOTES_PER_ROW - 1) / NOTES_PER_ROW);
}

// sender: leaf -> [nd distinct complete runs][nu truncated runs][na notes, 8 per row] at record
// dst_off[leaf] of the send buffer (the truncated stream holds records and noted records mixed)
__global__ __launch_bounds__(256) void msp_runs_gather_kernel(MspView v, const uint64_t *__restrict__ dst_off, uint4 *__restrict__ out,
                                                              const uint64_t *__restrict__ plan_rows, const uint64_t *__restrict__ seg_start, int parts, uint64_t cap_rows) {
  __shared__ uint32_t cu, cn;
  if (plan_rows[parts] > cap_rows) return;         // the buffer is too small: nothing was planned
  const uint32_t leaf = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const bool has1 = v.cnt2[NCLS * leaf + 1] != 0u;
  const uint32_t nd = has1 ? v.leaf_n[leaf] : 0u;
  const uint32_t nt = (uint32_t)min((uint64_t)v.cnt2[NCLS * leaf + 0], v.exact ? (uint64_t)v.lcap[NCLS * leaf + 0] : v.cap2t);
  const uint32_t na = has1 ? min((uint32_t)v.leaf_off[leaf], nt) : 0u;
  const uint32_t nu = nt - na;
  if (threadIdx.x == 0) runs_write_header(out, seg_start, parts, blockIdx.x, nd, nu, na);
  const uint4 *c1 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 1] : v.rec2 + (uint64_t)(leaf >> v.sel_bits) * (v.cap2c + v.cap2t);
  const uint4 *c0 = v.exact ? v.rec2 + v.lbase[NCLS * leaf + 0] : c1 + v.cap2c;
  uint4 *dst = out + dst_off[leaf];
  for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = c1[i];
  if (na == 0u) {
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) dst[nd + i] = c0[i];
    return;
  }
  if (threadIdx.x == 0) { cu = 0u; cn = 0u; }
  __syncthreads();
  uint16_t *notes = reinterpret_cast<uint16_t *>(dst + nd + nu);
  for (uint32_t i = threadIdx.x; i < ((nt + 63u) & ~63u); i += blockDim.x) {
    const bool valid = i < nt;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (valid) rec = c0[i];
    const bool isn = valid && rec.w == RUN_NOTED;
    const unsigned long long mn = __ballot(isn), mu = __ballot(valid && !isn);
    uint32_t bn = 0, bu = 0;
    if (lane == 0) {
      if (mn) bn = atomicAdd(&cn, (uint32_t)__popcll(mn));
      if (mu) bu = atomicAdd(&cu, (uint32_t)__popcll(mu));
    }
    bn = __shfl(bn, 0); bu = __shfl(bu, 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    // (the counts of the plan bound both: a stream that changed under us cannot write outside the leaf's rows)
    if (isn) { const uint32_t at = bn + (uint32_t)__popcll(mn & below); if (at < na) notes[at] = (uint16_t)rec.x; }
    else if (valid) { const uint32_t at = bu + (uint32_t)__popcll(mu & below); if (at < nu) dst[nd + at] = rec; }
  }
  const uint32_t pad = (NOTES_PER_ROW - na % NOTES_PER_ROW) % NOTES_PER_ROW;
  if (threadIdx.x < pad) notes[na + threadIdx.x] = 0xFFFFu;
}

// owner: segment (source rank, local leaf) of the received buffer -> its place in the leaf's two streams
// A note becomes the run it stands for: the first n k-mers of its twin, closed on the left only (a
// prefix -- of the twin as the sender stored it, whichever strand the read showed).
__global__ __launch_bounds__(256) void msp_runs_scatter_kernel(const uint4 *__restrict__ in, RunsRecv rr, int lpp, int k,
                                                               const uint64_t *__restrict__ src_off,
                                                               const uint64_t *__restrict__ dst1, const uint64_t *__restrict__ dst0,
                                                               uint4 *__restrict__ rec2) {
  const uint32_t seg = blockIdx.x;
  const uint32_t r = seg / (uint32_t)lpp, ll = seg - r * (uint32_t)lpp;
  const uint32_t *hdr = reinterpret_cast<const uint32_t *>(in + rr.rstart[r]);
  const uint32_t nd = hdr[3 * ll], nt = hdr[3 * ll + 1], na = hdr[3 * ll + 2];
  const uint4 *src = in + src_off[seg];
  for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) rec2[dst1[seg] + i] = src[i];
  for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) rec2[dst0[seg] + i] = src[nd + i];
  const uint16_t *notes = reinterpret_cast<const uint16_t *>(src + nd + nt);
  for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) {          // (nd > 0: the layout kernel checked)
    const uint32_t note = notes[i];
    const uint4 twin = src[min(note >> 5, nd - 1u)];                  // a position outside the list is not followed
    const uint32_t nm1 = min(note & 31u, twin.w & 31u);
    const int rb = 2 * ((int)nm1 + k) - 32;                           // bits of the run beyond the first word: 0 .. 64
    const uint32_t my = (rb >= 32) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> rb);
    const uint32_t mz = (rb <= 32) ? 0u : ((rb >= 64) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (rb - 32)));
    rec2[dst0[seg] + nt + i] = make_uint4(twin.x, twin.y & my, twin.z & mz, 64u | nm1);
  }
}

// exact layout of a level from the demand the first attempt counted: base = exclusive prefix sum
// of the n cursors, cap = the cursors themselves (single workgroup, 1024 threads)
// (slack: room beyond the counted demand per region -- the chunked path runs P1 again, and which records its
//  first level PARKED last time depended on the order of atomics: a stream's demand may differ by a few)
__global__ __launch_bounds__(1024) void msp_layout_kernel(const uint32_t *__restrict__ cnt, uint32_t n,
                                                          uint64_t *__restrict__ base, uint32_t *__restrict__ cap, uint32_t slack = 0u) {
  __shared__ unsigned long long part[1024];
  const uint32_t per = (n + 1023u) / 1024u;
  const uint32_t tid = threadIdx.x;
  unsigned long long s = 0;
  for (uint32_t i = 0; i < per; ++i) { const uint32_t l = tid * per + i; if (l < n) s += (unsigned long long)cnt[l] + slack; }
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
    if (l < n) { const uint32_t c = cnt[l] + slack; base[l] = run; cap[l] = c; run += c; }
  }
}

// the few records that did not fit their leaf stream: counted k-mer by k-mer in the HBM table
__global__ __launch_bounds__(256) void msp_spill_list_kernel(const uint4 *__restrict__ recs, uint32_t n, int k,
                                                             int canon, TableView t) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) spill_record(recs[i], k, canon != 0, t);
}

// sum of n cursors (one workgroup): how many records the first chunk of a batch made
__global__ __launch_bounds__(1024) void msp_sum_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *out) {
  __shared__ unsigned long long tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  unsigned long long mine = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) mine += cnt[i];
  atomicAdd(&tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = tot; out[1] = 0; out[2] = 0; }
}

// ... and what share of them are truncated runs: the records of sub-region 0 of every level-1 bin (1/64
// of the chunk, every bin in it) -- out[1] += records looked at, out[2] += truncated ones among them
__global__ __launch_bounds__(256) void msp_class_sample_kernel(MspView v, unsigned long long *out) {
  const uint32_t reg = l1_reg(blockIdx.x, 0u);
  const uint32_t n = (uint32_t)min((uint64_t)v.cnt1[reg], v.cap1);
  const uint32_t *w = reinterpret_cast<const uint32_t *>(v.rec1 + (uint64_t)reg * v.cap1) + 3;
  uint32_t tr = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 256u) tr += ((w[4 * (size_t)i] & 192u) != 192u) ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d);
  if ((threadIdx.x & 63) == 0 && tr) atomicAdd(&out[2], (unsigned long long)tr);
  if (threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)n);
}

__global__ void msp_info_kernel(MspView v, uint64_t *out) {
  // diagnostics: record totals / maxima per level (single block)
  __shared__ unsigned long long tot1, max1, tot2, max2;
  if (threadIdx.x == 0) { tot1 = max1 = tot2 = max2 = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < B1 * (int)v.nxg; i += blockDim.x) {
    atomicAdd(&tot1, (unsigned long long)v.cnt1[i]);
    atomicMax(&max1, (unsigned long long)v.cnt1[i]);
  }
  for (int i = threadIdx.x; i < NCLS * NLEAF; i += blockDim.x) {
    atomicAdd(&tot2, (unsigned long long)v.cnt2[i]);
    atomicMax(&max2, (unsigned long long)v.cnt2[i]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = tot1; out[1] = max1; out[2] = v.cap1; out[3] = tot2; out[4] = max2; out[5] = v.cap2c;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------ host
cfrk_msp *cfrk_msp_get(cfrk_ctx *ctx) {
  if (!ctx->msp) {
    ctx->msp = new (std::nothrow) cfrk_msp();
    if (ctx->msp) memset(ctx->msp, 0, sizeof(cfrk_msp));
  }
  return ctx->msp;
}

bool cfrk_msp_usable(const cfrk_ctx *ctx) {
  return !ctx->g_two && ctx->g_k >= 16 && ctx->g_k <= 32 && !(ctx->g_flags & CFRK_FORCE_HASH);
}

void cfrk_msp_reset(cfrk_ctx *ctx) {
  if (ctx->msp) ctx->msp->view.seg_bits = 0;
  if (ctx->msp) { ctx->msp->pending = false; ctx->msp->table_dirty = false; ctx->msp->list_n_valid = false; ctx->msp->runs_ready = false;
                  ctx->msp->runs_deduped = false; ctx->msp->runs_unchecked = false; ctx->msp->lists_group = 0; }
}

void cfrk_msp_note_table_write(cfrk_ctx *ctx) {
  cfrk_msp *m = cfrk_msp_get(ctx);
  if (m) m->table_dirty = true;
}

bool cfrk_msp_table_written(const cfrk_ctx *ctx) {
  // without msp state nobody tracked the writes: assume written
  return !ctx->msp || ctx->msp->table_dirty;
}

void cfrk_msp_destroy(cfrk_ctx *ctx) {
  delete ctx->msp;
  ctx->msp = nullptr;
}

static void msp_params(int k, int *W, int *m) {
  // The longer the window, the longer the runs (fewer, fuller records): W = k - 12 keeps
  // m = k - W + 1 = 13, the smallest m-mer whose canonical values (3.4e7) rarely repeat across
  // the loci of a genome -- with fewer distinct minimizers the leaves get lumpy (at k = 31: m = 11 gave a
  // 10x heavier leaf, m = 12 2.3x).  W is capped by what a 48-base record holds (49 - k k-mers),
  // which raises m to 14 / 16 for k = 31 / 32.
  // k <= 26 (round 5): m = 12.  There the window is what is short (5 .. 14 k-mers at m = 13: one record per three k-mers
  // at k = 17), and one more k-mer per window is 7 .. 14 % fewer records through all three kernels (10 M reads:
  // k = 17 9.03 -> 8.30 ms, 19 7.62 -> 6.79, 22 6.30 -> 5.98, 25 5.41 -> 5.21; 100 M reads of a 10^8-base genome:
  // k = 20 51.5 -> 48.3 ms, k = 25 35.6 -> 35.0, k = 28 no change; heaviest stream / mean unchanged at 3.3).  m = 11
  // was measured too: k = 17 7.85 ms, but the heaviest stream grows to 4.1 x the mean and outgrows its room.
  *W = std::min(k <= 26 ? k - 11 : k - 12, 49 - k);
  *m = k - *W + 1;
}

// Sub-regions (cursors) per level-1 bin, from the expected number of records.  Many cursors keep
// the chains of returning atomics on one address short and spread the write frontiers (msp_dev.h:
// NXG); P2 reads a bin's sub-regions as one stream, so they cost it nothing.  Tiny batches keep 8
// (every sub-region reserves 2048 records of slack).
static int msp_nxg(double expect_records) { return expect_records >= 2.5e7 ? NXG : NXCD; }

// bytes of pool memory one pipeline pass over `span` base positions needs (the caps below)
static size_t msp_need(const cfrk_ctx *ctx, int64_t span) {
  int W0, m0;
  msp_params(ctx->g_k, &W0, &m0);
  const double expect0 = (double)span * (2.0 / (W0 + 1) + 1.0 / 64.0);
  return (size_t)(expect0 * 1.35 * 16) + (size_t)B1 * msp_nxg(expect0) * 2048 * 16 + (size_t)(expect0 * 2.7 * 16) +
         (size_t)NLEAF * 192 * 16 + (size_t)ctx->g_cap * 12;
}

// one pass of P1 -> P2 -> P3 over the P1 tiles [tile0, tile0 + ntiles): the k-mers that START in
// those tiles (P1 reads its neighbours' bases from the whole buffer, so a tile range produces
// exactly the records it produces in a full launch)
// slack >= 1 widens the per-leaf streams beyond what msp_need() accounts for (memory permitting)
// sel_bits / sel_val: only the leaves with (leaf & (2^sel_bits - 1)) == sel_val are emitted and counted;
// first: the first pass of an add (later passes append to the result list and keep the leaf index)
static int msp_count_tiles(cfrk_ctx *ctx, cfrk_msp *ms, const int8_t *d_data, int64_t nN, int64_t tile0,
                           int64_t ntiles, double slack, int sel_bits = 0, uint32_t sel_val = 0, bool first = true) {
  int rc;
  const int k = ctx->g_k;
  int W, m;
  msp_params(k, &W, &m);
  const int canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  const int64_t span = std::min(nN, ntiles * (int64_t)P1_WAVES * P1_OWN * 32);

  // expected records: one per minimizer change (2/(W+1) per position) plus read ends
  const double dens = 2.0 / (W + 1) + 1.0 / 64.0;
  const double expect_all = (double)span * dens;                 // records of the whole batch
  const double expect = expect_all / (double)(1u << sel_bits);   // ... of this pass (1 / 2^sel_bits of the leaves)
  const int nxg = msp_nxg(expect);
  const uint64_t cap1 = (uint64_t)(expect / (B1 * nxg) * 1.35) + 2048;   // per sub-region
  // per leaf: complete runs dominate at depth; truncated ones are ~2 per read plus invalid bases
  // (a leaf of the pass holds ALL its records: the pass has fewer leaves, not lighter ones)
  const uint64_t cap2c = (uint64_t)(expect_all / NLEAF * 2.1 * slack) + 96;
  const uint64_t cap2t = (uint64_t)(expect_all / NLEAF * 0.6 * slack) + 96;   // truncated runs: ~15 % of the records
  // Large batches are counted in CHUNKS of tiles: P1 on chunk c, then P2 on chunk c, with one
  // level-1 buffer that only ever holds one chunk (C3: 2.7 GB instead of 40 GB); the leaf streams
  // accumulate over the chunks.  Same kernels, same work; anything that overflows a region starts
  // over on the one-chunk path below, whose cursors add up to the batch's exact demand.
  // (What the chunks were built for did not pay: running P1 on chunk c+1 BESIDE P2 on chunk c in one
  // launch -- persistent roles or an interleaved grid -- takes as long as the two one after the other,
  // profiles/r03/persistent_p1_is_slower.txt: a CU is full with three partition workgroups.)
  const bool small_pipe = (ctx->dbg_flags & CFRK_DEBUG_SMALL_PIPELINE) != 0;
  const int64_t chunk_min = small_pipe ? 3 : 16384;            // tiles (16384 = 256 MB of input)
  // as many chunks as keep the level-1 buffer under ~6 GB (each costs ~30 us of launches and kernel
  // tails: 16 chunks +0.5 ms on C3, 64 +2.7 ms -- and nothing comes back from the Infinity Cache
  // even at 512 chunks of 43 MB, profiles/r03/chunk_count_sweep_c3.txt)
  // (a batch whose level-1 records fit ~6 GB anyway -- a shard of a multi-GPU job -- stays in one chunk)
  int nchunks = (int)std::min<int64_t>(small_pipe ? 5 : (int64_t)((double)B1 * nxg * cap1 * 16.0 / 6e9) + 1, ntiles / chunk_min);
  if (ctx->dbg_param[CFRK_PARAM_MSP_CHUNKS] > 0) nchunks = std::max(1, std::min((int)ctx->dbg_param[CFRK_PARAM_MSP_CHUNKS], (int)std::min<int64_t>(4096, ntiles / 8)));   // (experiments)
  const bool pipelined = (nxg == NXG || small_pipe) && nchunks >= 2 && !(ctx->dbg_flags & CFRK_DEBUG_NO_PIPELINE);
  if (!pipelined) nchunks = 1;
  const int64_t chunk_tiles = (ntiles + nchunks - 1) / nchunks;
  // per sub-region of ONE chunk (the cursors of a chunk see 1 / nchunks of the records)
  const uint64_t cap1c = pipelined ? (uint64_t)(expect * ((double)chunk_tiles / (double)ntiles) / (B1 * nxg) * 1.35) + 2048 : cap1;

  void *p;
  MspView &v = ms->view;
  const size_t nreg = (size_t)B1 * nxg;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, nreg * (pipelined ? cap1c : cap1) * sizeof(uint4), &p))) return rc;
  v.rec1 = (uint4 *)p; v.cap1 = cap1; v.nxg = (uint32_t)nxg; v.dbg = ctx->dbg_flags;
  v.sel_mask = (1u << sel_bits) - 1u; v.sel_val = sel_val; v.sel_bits = (uint32_t)sel_bits;
  // (a chunked batch sizes its leaf streams after the first chunk: whatever the pool already holds will do until then)
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (pipelined && !small_pipe) ? 16 : (size_t)(NLEAF >> sel_bits) * (cap2c + cap2t) * sizeof(uint4), &p))) return rc;
  v.rec2 = (uint4 *)p; v.cap2c = cap2c; v.cap2t = cap2t;
  // far more distinct k-mers expected than the leaf tables hold (65536 x ~2500): records carry extra
  // minimizer-hash bits and 2^sub_bits workgroups share a leaf (~2000 distinct k-mers each; msp2.hip)
  uint32_t sub_bits = 0;
  while (sub_bits < (uint32_t)SUB_BITS && ((ctx->g_cap / NLEAF) >> sub_bits) > 2048u) ++sub_bits;
  if (ctx->g_cap / NLEAF <= 4096u) sub_bits = 0;
  if ((ctx->dbg_flags & CFRK_DEBUG_RECORD_SUBSETS) && sub_bits < 2u) sub_bits = 2u;
  if (ctx->g_flags & CFRK_RUNS_ONLY) sub_bits = 0;       // (the exchange by runs has no room for the bits in a distinct run's header)
  const bool sub = sub_bits != 0u;
  v.sub_bits = sub_bits; v.seg_bits = sub_bits;
  // (leaf index: one entry per leaf, or per (leaf, sub-value) when leaves are shared)
  const size_t nseg = (size_t)NLEAF << sub_bits;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, nseg * 8 + ((size_t)B1 * nxg + (size_t)NCLS * NLEAF + nseg) * sizeof(uint32_t), &p))) return rc;
  v.leaf_off = (uint64_t *)p;
  v.cnt1 = (uint32_t *)(v.leaf_off + nseg); v.cnt2 = v.cnt1 + B1 * nxg; v.leaf_n = v.cnt2 + NCLS * NLEAF;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_keys = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats;
  TableView t = cfrk_table_view(ctx);

  // (cnt1, cnt2 and -- first pass only -- the leaf index and the list cursor)
  HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, ((size_t)B1 * nxg + (size_t)NCLS * NLEAF + (first ? nseg : 0)) * sizeof(uint32_t), ctx->stream));
  if (first) HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));

  // Both levels are laid out for an input that spreads evenly over the minimizer space.  One that
  // does not -- deep coverage of a small genome puts tens of thousands of records into a handful of
  // leaves, a single amplicon into a handful of level-1 bins -- overflows its regions; the cursors
  // keep counting past the capacity, so after P2 the host knows the exact demand of both levels
  // (one 40-byte D2H + stream sync per add).  A few overflowing records (lumpy leaves) were
  // parked in a small buffer and are counted through the HBM table; more than that and the
  // level is laid out again back to back with exactly the room each region needs -- all records
  // together never exceed what the buffers already hold -- and its kernel runs again.
  constexpr uint32_t OVF_CAP = 1u << 20;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OVF, (size_t)OVF_CAP * sizeof(uint4), &p))) return rc;
  v.ovf = (uint4 *)p; v.ovf_cap = (uint32_t)std::min<double>((double)OVF_CAP, expect / 256.0);   // "a few": < 0.4 %
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OVF1, (size_t)OVF_CAP * sizeof(uint4), &p))) return rc;
  v.ovf1 = (uint4 *)p; v.ovf1_cap = v.ovf_cap;
  // a job that only partitions (CFRK_RUNS_ONLY) has no table of its own for parked records: any
  // overflow goes straight to the exact re-layout
  if (ctx->g_flags & CFRK_RUNS_ONLY) v.ovf_cap = v.ovf1_cap = 0;
  v.exact = 0; v.lbase = nullptr; v.lcap = nullptr;
  v.exact1 = 0; v.rbase = nullptr; v.rcap = nullptr;
  // stand-alone P1 over the tiles [t0, t1), one workgroup each
  auto launch_p1 = [&](int64_t t0, int64_t t1, const MspView &vv) -> int {
    const dim3 g1((unsigned)(t1 - t0)), b1(P1_THREADS);
#define CFRK_P1B_CASE(WW) \
    case WW: \
      if (sub) hipLaunchKernelGGL((msp_p1b_kernel<WW, (WW >= 16 ? 4 : WW >= 12 ? 6 : WW >= 8 ? 8 : WW >= 6 ? 10 : 12), true>), g1, b1, 0, ctx->stream, d_data, nN, k, m, canon, t0, vv, t); \
      else hipLaunchKernelGGL((msp_p1b_kernel<WW, (WW >= 16 ? 4 : WW >= 12 ? 6 : WW >= 8 ? 8 : WW >= 6 ? 10 : 12), false>), g1, b1, 0, ctx->stream, d_data, nN, k, m, canon, t0, vv, t); \
      break;
    switch (W) {
      CFRK_P1B_CASE(4) CFRK_P1B_CASE(5) CFRK_P1B_CASE(6) CFRK_P1B_CASE(7) CFRK_P1B_CASE(8) CFRK_P1B_CASE(9)
      CFRK_P1B_CASE(10) CFRK_P1B_CASE(11) CFRK_P1B_CASE(12) CFRK_P1B_CASE(13) CFRK_P1B_CASE(14)
      CFRK_P1B_CASE(15) CFRK_P1B_CASE(16) CFRK_P1B_CASE(17) CFRK_P1B_CASE(18)
      default: return cfrk_fail(ctx, CFRK_ERR_ARG, "no partition kernel for W=%d", W);
    }
#undef CFRK_P1B_CASE
    HIP_TRY(ctx, hipGetLastError());
    return CFRK_OK;
  };
  // exact layout of the leaf streams from the demand their cursors counted (they count on past a
  // stream's capacity): streams back to back, each with exactly the room it needs; the buffer grows
  // when the batch holds more records than it was sized for (a chunked batch's streams were sized from
  // its first chunk)
  auto layout_l2 = [&](uint32_t slack) -> int {
    int rc2;
    void *q;
    if ((rc2 = cfrk_pool_get(ctx, BUF_MSP_LAYOUT, (size_t)NCLS * NLEAF * (sizeof(uint64_t) + sizeof(uint32_t)), &q))) return rc2;
    uint64_t *lbase = (uint64_t *)q;
    uint32_t *lcap = (uint32_t *)(lbase + NCLS * NLEAF);
    hipLaunchKernelGGL(msp_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt2, (uint32_t)(NCLS * NLEAF), lbase, lcap, slack);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc2 = cfrk_pool_get(ctx, BUF_SCRATCH, 64 * sizeof(uint64_t), &q))) return rc2;
    hipLaunchKernelGGL(msp_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt2, (uint32_t)(NCLS * NLEAF), (uint64_t *)q);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t all = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&all, q, sizeof all, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    all += (uint64_t)slack * NCLS * NLEAF;
    if ((size_t)all * sizeof(uint4) > ctx->pool[BUF_MSP_L2].cap) {
      if ((rc2 = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)all * sizeof(uint4), &q))) return rc2;
      v.rec2 = (uint4 *)q;
    }
    HIP_TRY(ctx, hipMemsetAsync(v.cnt2, 0, (size_t)NCLS * NLEAF * sizeof(uint32_t), ctx->stream));
    v.exact = 1; v.lbase = lbase; v.lcap = lcap;
    return CFRK_OK;
  };
  const unsigned p2_grid = 2u * (unsigned)std::max(8, ctx->num_cus / 8 * 8);   // P2: persistent, two workgroups per CU, a multiple of 8
  const bool defer = (ctx->g_flags & CFRK_RUNS_ONLY) && (ctx->g_flags & CFRK_RUNS_DEFER);
  bool run_p1 = true, settled = false, piped = pipelined;
  uint64_t parked1 = 0, parked2 = 0;
  for (int attempt = 0; attempt < 5; ++attempt) {
    if (piped) {
      // ---- chunks: P1(c), P2(c) on the one level-1 buffer
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L1OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L1OVF, ST_OVFN1
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L2OVF, ST_OVFN
      MspView vc = v;
      vc.cap1 = cap1c;
      for (int c = 0; c < nchunks; ++c) {
        const int64_t t0 = tile0 + (int64_t)c * chunk_tiles, t1 = std::min(tile0 + ntiles, t0 + chunk_tiles);
        if (t0 >= t1) break;
        if (c > 0 || attempt > 0) HIP_TRY(ctx, hipMemsetAsync(vc.cnt1, 0, nreg * sizeof(uint32_t), ctx->stream));   // (chunk 0 of the first attempt: cleared above)
        if ((rc = launch_p1(t0, t1, vc))) return rc;
        if (c == 0 && !small_pipe && !v.exact) {
          // The leaf streams are sized from what the first chunk really made, not from the worst
          // density a batch of unknown read length could have (the estimate above allows for reads
          // as short as 2k: 1.34 x the records 150-base reads make, times 2.7 for lumpy leaves --
          // 79 GB for C3's 21.9 GB of records).  One 24-byte read-back + sync per add.
          if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, 64 * sizeof(uint64_t), &p))) return rc;
          hipLaunchKernelGGL(msp_sum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)vc.cnt1, (uint32_t)nreg, (uint64_t *)p);
          HIP_TRY(ctx, hipGetLastError());
          hipLaunchKernelGGL(msp_class_sample_kernel, dim3(B1), dim3(256), 0, ctx->stream, vc, (unsigned long long *)p);
          HIP_TRY(ctx, hipGetLastError());
          uint64_t made[3] = {0, 0, 0};               // records of the chunk; records sampled, truncated ones among them
          HIP_TRY(ctx, hipMemcpyAsync(made, p, sizeof made, hipMemcpyDeviceToHost, ctx->stream));
          HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
          // records of this pass's leaves in the whole batch (+3 %: chunks differ a little), per leaf
          const double per_leaf = (double)made[0] * ((double)ntiles / (double)(t1 - t0)) * 1.03 / (double)(NLEAF >> sel_bits);
          // ... split into complete and truncated runs by the sample (truncated: two per read + invalid
          // bases -- 16 % of a deep batch of 150-base reads, half of the records of 50-base reads at
          // k = 31); the heaviest complete stream of a uniform batch is ~1.35 x the mean one
          const double ft = made[1] >= 4096 ? std::min(1.0, (double)made[2] / (double)made[1] + 0.01) : 0.5;
          // (measured on C3: with 1.3 x a few leaves' complete streams overflow and their records are parked --
          //  counted through the HBM table, +0.5 ms and a merge at finish; 1.4 x has none)
          double fc_slack = 1.45, ft_slack = 1.6;
          {
            // ... and a leaf's load scatters with the number of distinct runs it holds (the copies of a run come
            // and go together): lambda = distinct k-mers per leaf x 4 / (W + 1) -- 320 for C3, 64 for 20 M reads
            // of a 20 Mb genome, whose heaviest complete stream is 2.0 x the mean one (5448 records parked at 1.45 x)
            // (the table holds 2 .. 4 x the announced distinct k-mers: a third of it stands for the hint)
            const double lambda = std::max(4.0, (double)ctx->g_cap / 3.0 / (double)NLEAF * 4.0 / (double)(W + 1));
            fc_slack = std::min(4.0, std::max(fc_slack, 1.2 + 6.5 / std::sqrt(lambda)));
            ft_slack = std::min(4.0, std::max(ft_slack, 1.3 + 6.5 / std::sqrt(lambda)));
          }
          if (ctx->dbg_param[CFRK_PARAM_L2_SLACK_COMPLETE] > 0) fc_slack = ctx->dbg_param[CFRK_PARAM_L2_SLACK_COMPLETE];   // (experiments)
          if (ctx->dbg_param[CFRK_PARAM_L2_SLACK_TRUNCATED] > 0) ft_slack = ctx->dbg_param[CFRK_PARAM_L2_SLACK_TRUNCATED];
          const uint64_t m2c = (uint64_t)(per_leaf * (1.0 - ft) * fc_slack + per_leaf * 0.02) + 512, m2t = (uint64_t)(per_leaf * ft * ft_slack) + 256;
          if (m2c + m2t < v.cap2c + v.cap2t) { v.cap2c = m2c; v.cap2t = m2t; }
          if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(NLEAF >> sel_bits) * (v.cap2c + v.cap2t) * sizeof(uint4), &p))) return rc;
          v.rec2 = (uint4 *)p;
          vc.cap2c = v.cap2c; vc.cap2t = v.cap2t; vc.rec2 = v.rec2;
        }
        hipLaunchKernelGGL(msp_p2_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, ctx->stream, k, canon, vc, t);
        HIP_TRY(ctx, hipGetLastError());
      }
    } else {
    if (run_p1) {
      HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L1OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L1OVF, ST_OVFN1
      if ((rc = launch_p1(tile0, tile0 + ntiles, v))) return rc;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_L2OVF, 0, 2 * sizeof(uint64_t), ctx->stream));   // ST_L2OVF, ST_OVFN
    hipLaunchKernelGGL(msp_p2_kernel, dim3(p2_grid), dim3(P2_THREADS), 0, ctx->stream, k, canon, v, t);
    HIP_TRY(ctx, hipGetLastError());
    }
    // CFRK_RUNS_DEFER: the add ends here, unsynchronised -- whether a region overflowed is looked at by the export
    // (the group epilogue of the pipelined one folds the flags into what the host waits for anyway)
    if (defer) { settled = true; break; }
    uint64_t st[ST_NWORDS];
    HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st[ST_CWRAP]) return CFRK_INTERNAL_FLOOD;      // (nothing of this pass has been counted yet)
    // (level-1 records PARKED in a chunked attempt whose leaf streams overflowed: which records are parked depends on
    //  the order of atomics, so the exact stream sizes counted now need not hold when P1 runs again -- same remedy)
    if (piped && (st[ST_L1OVF] || (st[ST_L2OVF] && st[ST_OVFN1]))) {
      // a level-1 region overflowed by more than the parking buffer takes: the level-1 cursors of a chunk
      // do not add up to the batch's demand (the buffer is reused), so the batch starts over on the
      // one-chunk path, which lays the overflowing level out exactly -- with the leaf streams the
      // density estimate gives (what the first chunk suggested is void)
      piped = false;
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, nreg * cap1 * sizeof(uint4), &p))) return rc;
      v.rec1 = (uint4 *)p;
      v.cap2c = cap2c; v.cap2t = cap2t;
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(NLEAF >> sel_bits) * (cap2c + cap2t) * sizeof(uint4), &p))) return rc;
      v.rec2 = (uint4 *)p;
      HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, (nreg + (size_t)NCLS * NLEAF) * sizeof(uint32_t), ctx->stream));   // cnt1 and cnt2
      run_p1 = true;
      continue;
    }
    if (piped && st[ST_L2OVF]) {
      // only leaf streams overflowed (the first chunk was no measure of the batch: reads sorted or
      // clustered, a short-read or N-rich prefix, concatenated libraries).  Their cursors counted on
      // over ALL chunks, so the batch's exact demand is known: lay the streams out back to back and
      // run the chunks again (one more P1 + P2 over the input; the level-1 buffer stays one chunk).
      // (nothing was parked on level 1 -- see above -- so P1 makes the same records again; eight to spare per stream)
      if ((rc = layout_l2(8u))) return rc;
      continue;
    }
    if (st[ST_L1OVF]) {
      // exact level-1 layout; P2 ran on an incomplete level 1 and is redone as well
      if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT1, nreg * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
      uint64_t *rbase = (uint64_t *)p;
      uint32_t *rcap = (uint32_t *)(rbase + nreg);
      hipLaunchKernelGGL(msp_layout_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)v.cnt1, (uint32_t)nreg, rbase, rcap);
      HIP_TRY(ctx, hipGetLastError());
      {
        // all records of the batch: the buffer must hold them back to back
        std::vector<uint32_t> c1(nreg);
        HIP_TRY(ctx, hipMemcpyAsync(c1.data(), v.cnt1, nreg * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        uint64_t maxbin = 0, all = 0;
        for (int b = 0; b < B1; ++b) {
          uint64_t sum = 0;
          for (int r = 0; r < nxg; ++r) sum += c1[l1_reg((uint32_t)b, (uint32_t)r)];
          maxbin = std::max(maxbin, sum);
          all += sum;
        }
        if (all > (uint64_t)B1 * nxg * cap1) {           // more records than the density estimate allowed for
          if ((rc = cfrk_pool_get(ctx, BUF_MSP_L1, (size_t)all * sizeof(uint4), &p))) return rc;
          v.rec1 = (uint4 *)p;
        }
        (void)maxbin;                                  // (P2 is persistent: it sizes its work from the cursors)
      }
      HIP_TRY(ctx, hipMemsetAsync(v.cnt1, 0, (nreg + (size_t)NCLS * NLEAF) * sizeof(uint32_t), ctx->stream));   // cnt1 and cnt2
      v.exact1 = 1; v.rbase = rbase; v.rcap = rcap;
      run_p1 = true;
      continue;
    }
    parked1 = st[ST_OVFN1];
    if (st[ST_L2OVF]) {
      if ((rc = layout_l2(0u))) return rc;
      run_p1 = false;
      continue;
    }
    parked2 = st[ST_OVFN];
    settled = true;
    break;
  }
  if (!settled) return cfrk_fail(ctx, CFRK_ERR_STATE, "the record regions did not settle after an exact layout");
  if (parked1 && v.ovf1_cap) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(parked1, v.ovf1_cap);
    hipLaunchKernelGGL(msp_spill_list_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, (const uint4 *)v.ovf1, n, k, canon, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (parked2 && v.ovf_cap) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(parked2, v.ovf_cap);
    hipLaunchKernelGGL(msp_spill_list_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, (const uint4 *)v.ovf, n, k, canon, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  const bool runs_only = (ctx->g_flags & CFRK_RUNS_ONLY) != 0;
  if (runs_only) { if (!defer) hipLaunchKernelGGL(msp_dedupe_export_kernel, dim3(NLEAF), dim3(DX_THREADS), 0, ctx->stream, k, canon, v); }
  else if (sub) {
    const dim3 g3(((unsigned)NLEAF >> sel_bits) << v.sub_bits);
    if (canon) hipLaunchKernelGGL((msp_p3_kernel<true, true>), g3, dim3(P3_THREADS), 0, ctx->stream, k, 0u, v, t);
    else hipLaunchKernelGGL((msp_p3_kernel<false, true>), g3, dim3(P3_THREADS), 0, ctx->stream, k, 0u, v, t);
  }
  else if (canon) hipLaunchKernelGGL((msp_p3_kernel<true, false>), dim3((unsigned)NLEAF >> sel_bits), dim3(P3_THREADS), 0, ctx->stream, k, 0u, v, t);
  else hipLaunchKernelGGL((msp_p3_kernel<false, false>), dim3((unsigned)NLEAF >> sel_bits), dim3(P3_THREADS), 0, ctx->stream, k, 0u, v, t);
  HIP_TRY(ctx, hipGetLastError());
  if (!runs_only) {
    hipLaunchKernelGGL(msp_huge_leaves_kernel, dim3((((unsigned)NLEAF >> sel_bits) + 255u) / 256u), dim3(256), 0, ctx->stream, k, canon, 0, (uint32_t)NLEAF >> sel_bits, v, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  ms->pending = !runs_only;
  ms->runs_ready = runs_only;
  ms->runs_deduped = runs_only && !defer;
  ms->runs_unchecked = defer;
  ms->leaf_form = !runs_only;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// How many passes does a batch of `ntiles` tiles (tile_span base positions each) take so that one
// pass's buffers fit the memory the pool may use?  0 = not even a minimal pass fits.
int cfrk_msp_plan_groups(cfrk_ctx *ctx, int64_t nN, int64_t ntiles, int64_t tile_span,
                         size_t (*need_fn)(const cfrk_ctx *, int64_t), size_t acc_bytes, size_t have,
                         int *groups) {
  (void)ntiles; (void)tile_span; (void)acc_bytes;
  *groups = 1;
  if (need_fn(ctx, nN) <= have && !ctx->mem_budget) return CFRK_OK;   // fits what the pool already holds
  size_t free_b = 0, total_b = 0;
  HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
  size_t budget = have + free_b;
  if (ctx->mem_budget) budget = std::min(budget, ctx->mem_budget);
  // a pass over 1/g of the leaves needs the record buffers of 1/g of the batch (power-of-two g:
  // the subset is picked by the leaf id's low bits)
  int g = 1;
  while (need_fn(ctx, (nN + g - 1) / g) > budget) {
    if (g >= 256) {
      *groups = 0;
      return CFRK_OK;
    }
    g *= 2;
  }
  *groups = g;
  return CFRK_OK;
}

int cfrk_msp_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN) {
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  int rc;
  const int64_t nchunks = (nN + 31) / 32;
  const int64_t nwaves = (nchunks + P1_OWN - 1) / P1_OWN;
  const int64_t ntiles = (nwaves + P1_WAVES - 1) / P1_WAVES;
  if (ntiles > 0x7FFFFFFF) return cfrk_fail(ctx, CFRK_ERR_ARG, "batch too large for one add");
  // every buffer a pass needs must fit: decide before touching the pool so that a refusal leaves
  // the context usable for the fallback path
  const size_t have = ctx->pool[BUF_MSP_L1].cap + ctx->pool[BUF_MSP_L2].cap + ctx->pool[BUF_MSP_OUTK].cap +
                      ctx->pool[BUF_MSP_OUTC].cap;
  int groups = 1;
  if ((rc = cfrk_msp_plan_groups(ctx, nN, ntiles, (int64_t)P1_WAVES * P1_OWN * 32, msp_need, (size_t)ctx->g_cap * 12,
                                 have, &groups))) return rc;
  if (groups == 0) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "partitioned path does not fit device memory");
  const int passes = groups;
  ctx->last_passes = passes;
  if (ctx->g_flags & CFRK_RUNS_ONLY) {
    if (ms->runs_ready) return cfrk_fail(ctx, CFRK_ERR_STATE, "a CFRK_RUNS_ONLY job takes one add");
    if (passes != 1) return cfrk_fail(ctx, CFRK_ERR_RUNS_REFUSED, "a CFRK_RUNS_ONLY job must fit device memory in one pass");
  }
  if (ms->pending && (rc = cfrk_msp_flush_to_table(ctx))) return rc;
  if (passes == 1) {
    // Leaves are lumpy when the genome is small (few distinct runs per leaf, each repeated by
    // every read over it): with memory to spare the leaf streams get up to twice the room, so
    // that an ordinary imbalance does not end in the spill path.
    double slack = 1.0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      int W0, m0;
      msp_params(ctx->g_k, &W0, &m0);
      const double l2 = (double)nN * (2.0 / (W0 + 1) + 1.0 / 64.0) * 2.7 * 16;
      size_t budget = have + free_b;
      if (ctx->mem_budget) budget = std::min(budget, ctx->mem_budget);
      const double room = 0.5 * (double)budget - (double)msp_need(ctx, nN);
      if (room > 0 && l2 > 0) slack = std::min(2.0, 1.0 + room / l2);
    }
    return msp_count_tiles(ctx, ms, d_data, nN, 0, ntiles, slack);
  }
  // A batch whose records do not fit beside the caller's data is counted in `passes` passes over
  // the WHOLE input, each emitting only the runs of 1/passes of the leaves (leaf id low bits): the
  // partition kernel's front end runs again every pass, but every pass produces the FINAL counts
  // of its leaves -- nothing to merge afterwards, and the result stays in per-leaf list form.
  int sel_bits = 0;
  while ((1 << sel_bits) < passes) ++sel_bits;
  for (int pass = 0; pass < passes; ++pass) {
    if ((rc = msp_count_tiles(ctx, ms, d_data, nN, 0, ntiles, 1.0, sel_bits, (uint32_t)pass, pass == 0))) {
      // a refusal after the first pass must not reach the caller's fallback (it would count
      // the finished passes twice)
      if (pass > 0 && (rc == CFRK_ERR_NOMEM || rc == CFRK_INTERNAL_FLOOD)) return cfrk_fail(ctx, CFRK_ERR_STATE, "out of device memory in pass %d of a multi-pass add", pass);
      return rc;
    }
  }
  return CFRK_OK;
}

int cfrk_msp_sync_stats(cfrk_ctx *ctx, uint64_t st[ST_NWORDS]) {
  HIP_TRY(ctx, hipMemcpyAsync(st, ctx->g_stats, ST_NWORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CFRK_OK;
}

int cfrk_msp_flush_to_table(cfrk_ctx *ctx) {
  cfrk_msp *ms = ctx->msp;
  if (!ms || !ms->pending) return CFRK_OK;
  uint64_t st[ST_NWORDS];
  int rc = cfrk_msp_sync_stats(ctx, st);
  if (rc) return rc;
  if (st[ST_OVERFLOW]) return cfrk_fail(ctx, CFRK_ERR_TABLE_FULL, "result list of %llu entries overflowed", (unsigned long long)ms->view.out_cap);
  const uint64_t n = st[ST_CURSOR];
  ms->pending = false;
  ms->table_dirty = true;
  if (n == 0) return CFRK_OK;
  return cfrk_hash_merge(ctx, ms->view.out_keys, ctx->g_two ? ms->view.out_hi : nullptr, ms->view.out_cnt, (int64_t)n);
}

int cfrk_msp_resolve(cfrk_ctx *ctx, ResultSrc *src, bool *use_list) {
  *use_list = false;
  if (ctx->g_flags & CFRK_RUNS_ONLY)
    return cfrk_fail(ctx, CFRK_ERR_STATE, "a CFRK_RUNS_ONLY job holds runs, not counts: cfrk_global_export_runs_device");
  cfrk_msp *ms = ctx->msp;
  if (!ms || !ms->pending) return CFRK_OK;
  uint64_t st[ST_NWORDS];
  int rc = cfrk_msp_sync_stats(ctx, st);
  if (rc) return rc;
  if (st[ST_OVERFLOW]) return cfrk_fail(ctx, CFRK_ERR_TABLE_FULL, "result list of %llu entries overflowed", (unsigned long long)ms->view.out_cap);
  if (st[ST_SPILLED] || ms->table_dirty) return cfrk_msp_flush_to_table(ctx);
  src->lo = ms->view.out_keys; src->hi = ctx->g_two ? ms->view.out_hi : nullptr; src->cnt = ms->view.out_cnt;
  src->n = st[ST_CURSOR]; src->kind = ctx->g_two ? 3 : 2; src->stats = ctx->g_stats;
  *use_list = true;
  return CFRK_OK;
}

extern "C" int cfrk_debug_device_bytes(cfrk_ctx *ctx, uint64_t *out_bytes) {
  if (!ctx || !out_bytes) return CFRK_ERR_ARG;
  uint64_t n = 0;
  for (int i = 0; i < BUF_NSLOTS; ++i) n += ctx->pool[i].cap;
  if (ctx->g_keys_lo) n += ctx->g_cap * 8;
  if (ctx->g_keys_hi) n += ctx->g_cap * 8;
  if (ctx->g_counts) n += ctx->g_cap * 4;
  *out_bytes = n;
  return CFRK_OK;
}

extern "C" int cfrk_debug_set_mem_budget(cfrk_ctx *ctx, uint64_t bytes) {
  if (!ctx) return CFRK_ERR_ARG;
  ctx->mem_budget = (size_t)bytes;
  return CFRK_OK;
}

extern "C" int cfrk_debug_set_param(cfrk_ctx *ctx, int which, double value) {
  if (!ctx) return CFRK_ERR_ARG;
  if (which < 0 || which >= 4 || !(value >= 0)) return cfrk_fail(ctx, CFRK_ERR_ARG, "cfrk_debug_set_param: no such parameter / negative value");
  if (value != 0) {
    if (which == CFRK_PARAM_MSP_CHUNKS && !(value >= 1 && value <= 4096)) return cfrk_fail(ctx, CFRK_ERR_ARG, "chunks: 1 .. 4096");
    if ((which == CFRK_PARAM_L2_SLACK_COMPLETE || which == CFRK_PARAM_L2_SLACK_TRUNCATED) && !(value >= 1.0 && value <= 16.0))
      return cfrk_fail(ctx, CFRK_ERR_ARG, "leaf-stream slack: 1 .. 16");
    if (which == CFRK_PARAM_MSP2_SUBVALUE_BITS && !(value >= 1 && value <= 3)) return cfrk_fail(ctx, CFRK_ERR_ARG, "sub-value bits + 1: 1 .. 3");
  }
  ctx->dbg_param[which] = value;
  return CFRK_OK;
}
extern "C" int cfrk_debug_set_flags(cfrk_ctx *ctx, uint32_t flags) {
  if (!ctx) return CFRK_ERR_ARG;
  // (the timing ablations of msp.h exist in an ablation build only: in the product they are refused, not ignored)
  if (flags & ~CFRK_DEBUG_KNOWN_BITS) return cfrk_fail(ctx, CFRK_ERR_ARG, "cfrk_debug_set_flags: unknown bits 0x%x", flags & ~CFRK_DEBUG_KNOWN_BITS);
  ctx->dbg_flags = flags;
  return CFRK_OK;
}

extern "C" int cfrk_debug_last_add_passes(cfrk_ctx *ctx, int *out_passes) {
  if (!ctx || !out_passes) return CFRK_ERR_ARG;
  *out_passes = ctx->last_passes;
  return CFRK_OK;
}

// diagnostics (not part of the counting path): out[0..5] = L1 records total / max bin / bin
// capacity, L2 records total / max leaf / leaf capacity; out[6] = records spilled to the HBM
// table, out[7] = k-mers spilled by full leaf tables, out[8] = entries in the result list
extern "C" int cfrk_debug_msp_info(cfrk_ctx *ctx, uint64_t out[9]) {
  if (!ctx || !out) return CFRK_ERR_ARG;
  cfrk_msp *ms = ctx->msp;
  for (int i = 0; i < 9; ++i) out[i] = 0;
  if (!ms) return CFRK_OK;
  int rc;
  if (ms->view.cnt1) {                                   // (the two-word path keeps no view of its record levels)
    void *p;
    if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, 64 * sizeof(uint64_t), &p))) return rc;
    hipLaunchKernelGGL(msp_info_kernel, dim3(1), dim3(1024), 0, ctx->stream, ms->view, (uint64_t *)p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, p, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  uint64_t st[ST_NWORDS];
  rc = cfrk_msp_sync_stats(ctx, st);
  if (rc) return rc;
  out[6] = st[ST_AUX0]; out[7] = st[ST_AUX1]; out[8] = st[ST_CURSOR];
  return CFRK_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange by leaf
extern "C" int cfrk_global_export_leaves_device(cfrk_ctx *ctx, uint64_t *d_keys, uint64_t *d_keys_hi,
                                                uint32_t *d_counts, uint64_t cap, int parts,
                                                uint64_t *part_counts, uint32_t *d_leaf_counts) {
  if (!ctx || !part_counts || parts < 1 || parts > NLEAF) return CFRK_ERR_ARG;
  if (ctx->g_active && ctx->g_two && !d_keys_hi && cap) return cfrk_fail(ctx, CFRK_ERR_ARG, "two-word keys need d_keys_hi");
  cfrk_msp *ms = ctx->msp;
  if (!ctx->g_active || !ms || !ms->pending || !ms->leaf_form || ms->table_dirty)
    return cfrk_fail(ctx, CFRK_ERR_STATE, "result is not in per-leaf list form");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint64_t st[ST_NWORDS];
  int rc = cfrk_msp_sync_stats(ctx, st);
  if (rc) return rc;
  if (st[ST_OVERFLOW]) return cfrk_fail(ctx, CFRK_ERR_TABLE_FULL, "result list overflowed");
  if (st[ST_SPILLED] || st[ST_ONES]) return cfrk_fail(ctx, CFRK_ERR_STATE, "part of the result lives in the HBM table");
  if (st[ST_MULTISEG]) return cfrk_fail(ctx, CFRK_ERR_STATE, "leaves were counted in several passes: no per-leaf index");
  const int lpp = (NLEAF + parts - 1) / parts;           // leaves per part (owner p: leaves p, p+parts, ...)
  std::vector<uint32_t> ln(NLEAF), ordered((size_t)parts * lpp, 0u);
  std::vector<uint64_t> doff(NLEAF);
  const uint32_t sgb = ms->view.seg_bits;
  if (sgb == 0) {
    HIP_TRY(ctx, hipMemcpyAsync(ln.data(), ms->view.leaf_n, NLEAF * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    // a leaf shared by record: its entries sit in 2^seg_bits segments (one per sub-value)
    std::vector<uint32_t> sn((size_t)NLEAF << sgb);
    HIP_TRY(ctx, hipMemcpyAsync(sn.data(), ms->view.leaf_n, sn.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t l = 0; l < (size_t)NLEAF; ++l) {
      uint32_t t = 0;
      for (uint32_t j = 0; j < (1u << sgb); ++j) t += sn[(l << sgb) | j];
      ln[l] = t;
    }
  }
  uint64_t run = 0;
  for (int p = 0; p < parts; ++p) {
    uint64_t pc = 0;
    for (int j = 0; j < lpp; ++j) {
      const int leaf = p + j * parts;
      if (leaf >= NLEAF) break;
      doff[leaf] = run;
      run += ln[leaf]; pc += ln[leaf];
      ordered[(size_t)p * lpp + j] = ln[leaf];
    }
    part_counts[p] = pc;
  }
  if (run != st[ST_CURSOR]) return cfrk_fail(ctx, CFRK_ERR_STATE, "leaf index does not cover the list");
  if (run > cap) return cfrk_fail(ctx, CFRK_ERR_SMALL_BUF, "%llu entries, room for %llu", (unsigned long long)run, (unsigned long long)cap);
  void *p;
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, NLEAF * sizeof(uint64_t), &p))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(p, doff.data(), NLEAF * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  if (d_leaf_counts)
    HIP_TRY(ctx, hipMemcpyAsync(d_leaf_counts, ordered.data(), ordered.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  if (run) {
    hipLaunchKernelGGL(msp_gather_kernel, dim3(NLEAF), dim3(256), 0, ctx->stream, ms->view, (const uint64_t *)p, d_keys,
                       ctx->g_two ? d_keys_hi : (uint64_t *)nullptr, d_counts);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // doff / ordered are host temporaries
  return CFRK_OK;
}

extern "C" int cfrk_global_leaves_per_part(int parts) { return parts >= 1 ? (NLEAF + parts - 1) / parts : 0; }

extern "C" int cfrk_global_merge_leaves_device(cfrk_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_keys_hi,
                                               const uint32_t *d_counts, const uint64_t *recv_counts,
                                               const uint32_t *d_leaf_counts, int parts) {
  if (!ctx || parts < 1 || parts > NLEAF || !recv_counts || !d_leaf_counts) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_leaves needs an active job");
  if (ctx->g_two && !d_keys_hi) return cfrk_fail(ctx, CFRK_ERR_ARG, "two-word keys need d_keys_hi");
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  if (ms->pending || ms->table_dirty) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_leaves needs an empty job (call cfrk_global_begin first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int lpp = (NLEAF + parts - 1) / parts;
  int rc;
  void *p;
  MspView &v = ms->view;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_keys = (uint64_t *)p;
  if (ctx->g_two) {
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTH, (size_t)ctx->g_cap * 8, &p))) return rc;
    v.out_hi = (uint64_t *)p;
  }
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats;
  // segment (source rank, local leaf): counts arrive as [parts][lpp]; data as rank-major runs
  const size_t nseg = (size_t)parts * lpp;
  std::vector<uint32_t> sn(nseg);
  std::vector<uint64_t> so(nseg);
  HIP_TRY(ctx, hipMemcpyAsync(sn.data(), d_leaf_counts, nseg * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t run = 0;
  for (int r = 0; r < parts; ++r) {
    uint64_t got = 0;
    for (int j = 0; j < lpp; ++j) { so[(size_t)r * lpp + j] = run; run += sn[(size_t)r * lpp + j]; got += sn[(size_t)r * lpp + j]; }
    if (got != recv_counts[r]) return cfrk_fail(ctx, CFRK_ERR_ARG, "rank %d sent %llu entries but its leaf counts add up to %llu", r, (unsigned long long)recv_counts[r], (unsigned long long)got);
  }
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, nseg * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
  uint64_t *d_so = (uint64_t *)p;
  uint32_t *d_sn = (uint32_t *)(d_so + nseg);
  HIP_TRY(ctx, hipMemcpyAsync(d_so, so.data(), nseg * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_sn, sn.data(), nseg * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
  TableView t = cfrk_table_view(ctx);
  if (run && ctx->g_two) {
    if ((rc = cfrk_msp2_merge_lists(ctx, d_keys, d_keys_hi, d_counts, d_so, d_sn, parts, lpp))) return rc;
  } else if (run) {
    hipLaunchKernelGGL(msp_merge_kernel, dim3(lpp), dim3(P3_THREADS), 0, ctx->stream, d_keys, d_counts,
                       (const uint64_t *)d_so, (const uint32_t *)d_sn, parts, lpp, v, t);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // so / sn are host temporaries
  ctx->h_stats_valid = false;                          // this kernel may have spilled into the table
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange by runs

extern "C" int cfrk_global_export_runs_device(cfrk_ctx *ctx, void *d_packed, uint64_t cap_rows, int parts,
                                              uint64_t *part_rows) {
  if (!ctx || !part_rows || parts < 1 || parts > 64) return CFRK_ERR_ARG;
  cfrk_msp *ms = ctx->msp;
  if (!ctx->g_active || !ms || !ms->runs_ready)
    return cfrk_fail(ctx, CFRK_ERR_STATE, "no deduplicated runs to export (begin with CFRK_RUNS_ONLY, then one add)");
  if (!d_packed) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->g_two) return cfrk_msp2_export_runs(ctx, d_packed, cap_rows, parts, part_rows);
  const MspView &v = ms->view;
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
    hipLaunchKernelGGL(msp_dedupe_export_kernel, dim3(NLEAF), dim3(DX_THREADS), 0, ctx->stream, ctx->g_k, (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0, v);
    HIP_TRY(ctx, hipGetLastError());
    ms->runs_deduped = true;
  }
  const int lpp = (NLEAF + parts - 1) / parts;           // leaves per part (owner p: leaves p, p+parts, ...)
  const int hrows = runs_header_rows(lpp);
  int rc;
  void *p;
  // offsets, headers and segment sizes are worked out on the device (one workgroup); the host
  // only learns the segment sizes -- together with the job's flags, in ONE copy
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, (NLEAF + 65 + ST_NWORDS + 1 + 64 + 72) * sizeof(uint64_t) + (size_t)NLEAF * sizeof(uint4), &p))) return rc;
  uint64_t *d_off = (uint64_t *)p, *d_rows = d_off + NLEAF;
  uint64_t *d_seg = d_rows + 65 + ST_NWORDS + 1;
  unsigned long long *d_sync = (unsigned long long *)(d_seg + 64);
  uint4 *d_sz = (uint4 *)(d_sync + 72);      // (16-byte aligned: the pool is, and NLEAF + 65 + ST_NWORDS + 1 is even)
  static_assert((NLEAF + 65 + ST_NWORDS + 1 + 64 + 72) % 2 == 0, "d_sz is 16-byte aligned");
  hipLaunchKernelGGL(msp_runs_sizes_kernel, dim3(NLEAF / 256), dim3(256), 0, ctx->stream, v, d_sz, d_sync);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(msp_runs_plan_kernel, dim3(runs_plan_grid(parts, lpp)), dim3(1024), 0, ctx->stream, (const uint4 *)d_sz, parts, lpp, hrows, d_off,
                     d_rows + parts, d_seg, d_sync);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(msp_runs_gather_kernel, dim3(NLEAF), dim3(256), 0, ctx->stream, v, (const uint64_t *)d_off, (uint4 *)d_packed,
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

extern "C" int cfrk_global_merge_runs_device(cfrk_ctx *ctx, const void *d_packed, const uint64_t *recv_rows, int parts) {
  if (!ctx || parts < 1 || parts > 64 || !recv_rows || !d_packed) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_runs needs an active job");
  if (cfrk_msp2_usable(ctx) && !(ctx->g_flags & CFRK_RUNS_ONLY)) return cfrk_msp2_merge_runs(ctx, d_packed, recv_rows, parts);
  if (!cfrk_msp_usable(ctx) || (ctx->g_flags & CFRK_RUNS_ONLY)) return cfrk_fail(ctx, CFRK_ERR_ARG, "merge_runs needs a counting job with 16 <= k <= 64");
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
  MspView &v = ms->view;
  memset(&v, 0, sizeof v);
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_AUX, (size_t)NLEAF * 8 + (size_t)(B1 * NXG + (NCLS + 1) * NLEAF) * sizeof(uint32_t), &p))) return rc;
  v.leaf_off = (uint64_t *)p;
  v.cnt1 = (uint32_t *)(v.leaf_off + NLEAF); v.nxg = NXG; v.cnt2 = v.cnt1 + B1 * NXG; v.leaf_n = v.cnt2 + NCLS * NLEAF;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_LAYOUT, (size_t)NCLS * NLEAF * (sizeof(uint64_t) + sizeof(uint32_t)), &p))) return rc;
  uint64_t *d_lbase = (uint64_t *)p;
  uint32_t *d_lcap = (uint32_t *)(d_lbase + NCLS * NLEAF);
  v.exact = 1; v.lbase = d_lbase; v.lcap = d_lcap;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
  v.out_keys = (uint64_t *)p;
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
  v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
  v.stats = ctx->g_stats; v.dbg = ctx->dbg_flags;
  if ((rc = cfrk_pool_get(ctx, BUF_SCRATCH, (nseg * 3 + 2) * sizeof(uint64_t) + nseg * sizeof(uint32_t), &p))) return rc;
  uint64_t *d_src = (uint64_t *)p, *d_d1 = d_src + nseg, *d_d0 = d_d1 + nseg, *d_out = d_d0 + nseg;
  uint32_t *d_segrows = (uint32_t *)(d_out + 2);
  HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 2 * sizeof(uint64_t), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
  TableView t = cfrk_table_view(ctx);
  // segment (source rank, local leaf): the ranks' headers say how large; all offsets on the device
  hipLaunchKernelGGL((msp_runs_layout1_kernel<NCLS, 1, 0, 1>), dim3((unsigned)(lpp + 255) / 256), dim3(256), 0, ctx->stream, (const uint4 *)d_packed, rr, parts, lpp,
                     d_segrows, d_d1, d_d0, d_lcap, d_out);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL((msp_runs_layout_kernel<NCLS, 1, 0>), dim3((unsigned)parts + 1u), dim3(1024), 0, ctx->stream, rr, parts, lpp, hrows, (const uint32_t *)d_segrows,
                     d_src, d_d1, d_d0, d_lbase, (const uint32_t *)d_lcap, v.cnt2, d_out);
  HIP_TRY(ctx, hipGetLastError());
  // the headers are checked before anything is copied by them: sizes that add up to the rows each
  // rank sent keep every segment inside its rank's part of the buffer and every stream inside rec2
  uint64_t h[2];
  HIP_TRY(ctx, hipMemcpyAsync(h, d_out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (h[1]) return cfrk_fail(ctx, CFRK_ERR_ARG, "a rank's header does not add up to the rows it sent");
  // the leaf streams: what arrived, every note a record again (h[0] records; at most eight per row)
  if (h[0] > rows_all * NOTES_PER_ROW) return cfrk_fail(ctx, CFRK_ERR_ARG, "the headers announce more records than the rows can hold");
  if ((rc = cfrk_pool_get(ctx, BUF_MSP_L2, (size_t)(h[0] ? h[0] : 1) * sizeof(uint4), &p))) return rc;
  v.rec2 = (uint4 *)p;
  hipLaunchKernelGGL(msp_runs_scatter_kernel, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, (const uint4 *)d_packed, rr, lpp, k,
                     (const uint64_t *)d_src, (const uint64_t *)d_d1, (const uint64_t *)d_d0, v.rec2);
  HIP_TRY(ctx, hipGetLastError());
  if (canon) hipLaunchKernelGGL((msp_p3_kernel<true, false>), dim3(lpp), dim3(P3_THREADS), 0, ctx->stream, k, P3_WEIGHTED, v, t);
  else hipLaunchKernelGGL((msp_p3_kernel<false, false>), dim3(lpp), dim3(P3_THREADS), 0, ctx->stream, k, P3_WEIGHTED, v, t);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(msp_huge_leaves_kernel, dim3(((unsigned)lpp + 255u) / 256u), dim3(256), 0, ctx->stream, k, canon, 1, (uint32_t)lpp, v, t);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->ev_valid = true;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->h_stats_valid = false;                          // the leaf kernel may have spilled into the table
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange by runs, pipelined

extern "C" int cfrk_global_export_runs_async(cfrk_ctx *ctx, void *d_packed, uint64_t seg_cap_rows, int parts, int ngroups) {
  if (!ctx || parts < 1 || parts > 64 || ngroups < 1 || ngroups > CFRK_RUNS_MAX_GROUPS) return CFRK_ERR_ARG;
  cfrk_msp *ms = ctx->msp;
  if (!ctx->g_active || !ms || !ms->runs_ready)
    return cfrk_fail(ctx, CFRK_ERR_STATE, "no runs to export (begin with CFRK_RUNS_ONLY, then one add)");
  if (ms->runs_deduped) return cfrk_fail(ctx, CFRK_ERR_STATE, "the leaf streams were deduplicated in place already (add with CFRK_RUNS_DEFER for the pipelined export)");
  if (!d_packed) return cfrk_fail(ctx, CFRK_ERR_ARG, "NULL buffer");
  if (ctx->g_two) return cfrk_msp2_export_runs_async(ctx, d_packed, seg_cap_rows, parts, ngroups);
  const MspView &v = ms->view;
  const int k = ctx->g_k, canon = (ctx->g_flags & CFRK_CANONICAL) ? 1 : 0;
  return runs_export_async_host(ctx, d_packed, seg_cap_rows, parts, ngroups, [&](const RunsSend &sg) {
    hipLaunchKernelGGL(msp_dedupe_send_kernel, dim3(sg.nleaf), dim3(DS_THREADS), 0, ctx->stream, k, canon, v, sg);
  });
}

extern "C" int cfrk_global_export_runs_wait(cfrk_ctx *ctx, int group, uint64_t *part_rows) {
  if (!ctx || !part_rows) return CFRK_ERR_ARG;
  if (group < 0 || group >= ctx->runs_groups || !ctx->h_runs) return cfrk_fail(ctx, CFRK_ERR_STATE, "no such group in flight (cfrk_global_export_runs_async first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ctx->runs_ev[group]));
  const uint64_t *u = ctx->h_runs + (size_t)group * 65;
  const int parts = ctx->runs_parts;
  if (u[parts]) return cfrk_fail(ctx, CFRK_ERR_STATE, "the add overflowed a record region or spilled: the shard's runs are not all in the leaf streams");
  if (ctx->msp) ctx->msp->runs_unchecked = false;      // (the flags were clean when this group finished: the add held)
  for (int q = 0; q < parts; ++q)
    if (u[q] > ctx->runs_seg_cap) return cfrk_fail(ctx, CFRK_ERR_SMALL_BUF, "segment (group %d, owner %d) needs more than %llu rows", group, q, (unsigned long long)ctx->runs_seg_cap);
  for (int q = 0; q < parts; ++q) part_rows[q] = u[q];
  return CFRK_OK;
}

extern "C" int cfrk_global_merge_runs_group_device(cfrk_ctx *ctx, const void *d_recv, const uint64_t *recv_rows, int parts, int group, int ngroups) {
  if (!ctx || parts < 1 || parts > 64 || !recv_rows || !d_recv || ngroups < 1 || ngroups > CFRK_RUNS_MAX_GROUPS || group < 0 || group >= ngroups) return CFRK_ERR_ARG;
  if (!ctx->g_active) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_runs_group needs an active job");
  if (ctx->g_two && !(ctx->g_flags & (CFRK_RUNS_ONLY | CFRK_FORCE_HASH))) return cfrk_msp2_merge_runs_group(ctx, d_recv, recv_rows, parts, group, ngroups);
  if (!cfrk_msp_usable(ctx) || (ctx->g_flags & CFRK_RUNS_ONLY)) return cfrk_fail(ctx, CFRK_ERR_ARG, "merge_runs_group needs a counting job with 16 <= k <= 64");
  cfrk_msp *ms = cfrk_msp_get(ctx);
  if (!ms) return cfrk_fail(ctx, CFRK_ERR_NOMEM, "host allocation failed");
  if (group != ms->lists_group) return cfrk_fail(ctx, CFRK_ERR_STATE, "groups are merged in order: expected group %d", ms->lists_group);
  if (group == 0 && (ms->pending || ms->table_dirty)) return cfrk_fail(ctx, CFRK_ERR_STATE, "merge_runs_group needs an empty job (call cfrk_global_begin first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int lpp = (NLEAF + parts - 1) / parts;
  if (ngroups > lpp) return cfrk_fail(ctx, CFRK_ERR_ARG, "more groups than leaves per owner");
  int rc;
  void *p;
  MspView &v = ms->view;
  if (group == 0) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    memset(&v, 0, sizeof v);
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTK, (size_t)ctx->g_cap * 8, &p))) return rc;
    v.out_keys = (uint64_t *)p;
    if ((rc = cfrk_pool_get(ctx, BUF_MSP_OUTC, (size_t)ctx->g_cap * 4, &p))) return rc;
    v.out_cnt = (uint32_t *)p; v.out_cap = ctx->g_cap;
    v.stats = ctx->g_stats; v.dbg = ctx->dbg_flags;
    HIP_TRY(ctx, hipMemsetAsync(ctx->g_stats + ST_CURSOR, 0, sizeof(uint64_t), ctx->stream));
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
    if (ctx->g_flags & CFRK_CANONICAL) hipLaunchKernelGGL((msp_p3_lists_kernel<true>), dim3(lx.lcount), dim3(P3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
    else hipLaunchKernelGGL((msp_p3_lists_kernel<false>), dim3(lx.lcount), dim3(P3_THREADS), 0, ctx->stream, ctx->g_k, v, t, lx);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ctx->ev_valid = true;
  ctx->h_stats_valid = false;                          // the leaf kernel may have spilled into the table
  ms->lists_group = group + 1;
  ms->pending = true;
  ms->leaf_form = false;
  ms->list_n_valid = false;
  return CFRK_OK;
}

// Device time (ms) from the start of the job's add (its first kernel) to the end of group `group` of the pipelined export
// (HIP events on the context stream): the rehearsal tool's clock (tools/scale_emul.py).  Synchronises on that group.
extern "C" int cfrk_global_runs_group_ms(cfrk_ctx *ctx, int group, float *ms) {
  if (!ctx || !ms) return CFRK_ERR_ARG;
  if (group < 0 || group >= ctx->runs_groups || !ctx->ev_valid) return cfrk_fail(ctx, CFRK_ERR_STATE, "no such group in flight");
  HIP_TRY(ctx, hipEventSynchronize(ctx->runs_ev[group]));
  HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->runs_ev[group]));
  return CFRK_OK;
}
