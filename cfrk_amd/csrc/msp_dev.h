// msp_dev.h -- device helpers shared by the partitioned counting kernels (msp.hip: one-word keys,
// msp2.hip: two-word keys).
#pragma once
#include "common.h"

constexpr int B1_LOG = 8, B2_LOG = 8;
constexpr int B1 = 1 << B1_LOG, B2 = 1 << B2_LOG;
// Workgroups b and b+8 land on the same XCD (observed round-robin dispatch; speed only, never
// correctness).  A leaf stream that is appended to from ONE XCD has its partially written
// 64-byte sectors merged in that XCD's L2 before they reach HBM: the second-level kernels walk the
// level-1 bins in NXCD groups.  A level-1 bin is split into NXG sub-regions, each with its own
// cursor: a P1 workgroup appends to sub-region blockIdx % NXG (its XCD's, NXG / NXCD of them per
// XCD).  Measured on C3 (P1 + P2 ms): NXG = 8: 19.0 + 8.4, 16: 19.3 + 8.5, 32: 17.4 + 9.2,
// 64: 17.5 + 9.1, 128: 18.1 + 8.7 -- more cursors shorten the chains of returning atomics on one
// address and spread the write frontiers over more HBM channels; P2 pays a little for the
// additional partly filled last tiles.
constexpr int NXCD = 8;
constexpr int NXG = 64;
constexpr int NLEAF = B1 * B2;


// ordering hash of a canonical m-mer (bijective: odd multiplier).  No xorshift on top: the window
// minimum looks at the product's top bits, which already mix every bit of c, and the leaf id is
// bits 8..23 -- folding the top half down would only add the winner's (small, skewed) top bits.
__device__ __forceinline__ uint32_t hash_mmer(uint32_t c) { return c * 0x9E3779B1u; }
// minimizer -> leaf id (16 bits).  The window minimum is skewed low in its TOP bits only (that is
// what the comparison looks at); bits 8..23 of the winning hash stay uniform (simulated: same
// leaf balance as a full re-mix), and the low bits hold the position tag.  Byte-aligned so that
// one v_perm_b32 packs the leaf ids of two positions.
__device__ __forceinline__ uint32_t leaf_of(uint32_t wmin) { return (wmin >> 8) & 0xFFFFu; }

// Leaf ids of a lane's 32 window minima, two per word (even position in the low half).
struct LeafPack { uint32_t w[16]; };
template <int N>
__device__ __forceinline__ LeafPack leaf_pack(const uint32_t (&H)[N]) {
  LeafPack L;
#pragma unroll
  for (int i = 0; i < 16; ++i)   // bytes 1,2 of H[2i] -> bytes 0,1; bytes 1,2 of H[2i+1] -> bytes 2,3
    L.w[i] = __builtin_amdgcn_perm(H[2 * i + 1], H[2 * i], 0x06050201u);
  return L;
}
// leaf id at a lane-varying position a (0..31): a 4-level select tree on 16 VALUES plus a half
// pick (19 selects/shifts; selecting between array elements directly makes clang select pointers
// and park the arrays in scratch/LDS, and a tree over the 32 unpacked minima is twice as deep).
__device__ __forceinline__ uint32_t leaf_pick(const LeafPack &L, int a) {
  const bool c16 = (a & 16) != 0, c8 = (a & 8) != 0, c4 = (a & 4) != 0, c2 = (a & 2) != 0;
#define CFRK_SEL(c, hi_, lo_) ({ const uint32_t x_ = (lo_), y_ = (hi_); (c) ? y_ : x_; })
  const uint32_t s0 = CFRK_SEL(c16, L.w[8], L.w[0]), s1 = CFRK_SEL(c16, L.w[9], L.w[1]);
  const uint32_t s2 = CFRK_SEL(c16, L.w[10], L.w[2]), s3 = CFRK_SEL(c16, L.w[11], L.w[3]);
  const uint32_t s4 = CFRK_SEL(c16, L.w[12], L.w[4]), s5 = CFRK_SEL(c16, L.w[13], L.w[5]);
  const uint32_t s6 = CFRK_SEL(c16, L.w[14], L.w[6]), s7 = CFRK_SEL(c16, L.w[15], L.w[7]);
  const uint32_t e0 = CFRK_SEL(c8, s4, s0), e1 = CFRK_SEL(c8, s5, s1), e2 = CFRK_SEL(c8, s6, s2);
  const uint32_t e3 = CFRK_SEL(c8, s7, s3);
  const uint32_t f0 = CFRK_SEL(c4, e2, e0), f1 = CFRK_SEL(c4, e3, e1);
  const uint32_t g = CFRK_SEL(c2, f1, f0);
#undef CFRK_SEL
  return (g >> ((a & 1) << 4)) & 0xFFFFu;
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a full fence: the compiler
// puts s_waitcnt vmcnt(0) in front of it, so a persistent workgroup would wait at the end of every
// tile until that tile's global STORES (and returning atomics) have completed -- 2-3 us of a 10 us
// tile in the partition kernel.  The kernels below exchange data between threads through LDS only;
// global stores drain in the background.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the same within one wave (LDS written by some lanes, read by others)
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Wave-level compaction, used wherever the leaf kernels collect the few lanes that hold work into full
// sets of 64 (records that missed their home slot, a shared leaf's own records, ...): the lanes with
// `keep` set hand their NW words, in lane order, to the lanes [c, c + n) of the wave's set, n = the number
// of set bits in mask = __ballot(keep).  The caller empties a set that would overflow (c + n > 64) first.
template <int NW>
__device__ __forceinline__ void wave_append(uint32_t (&set)[NW], const uint32_t (&mine)[NW], bool keep, unsigned long long mask, int c, int n) {
  const int lane = threadIdx.x & 63;
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  const int dst = keep ? (c + rank) : ((c + n) & 63);      // the others aim at a lane nobody keeps
  const int da = dst << 2;
  const bool take = lane >= c && lane < c + n;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t p = __builtin_amdgcn_ds_permute(da, mine[i]);
    set[i] = take ? p : set[i];
  }
}

// Noted truncated runs of a record (leaf kernels: the lengths, in k-mers, of the truncated runs that are a
// prefix of the record a lane expands).  k-mer j0 + q of the record is counted once more for every noted
// length > j0 + q; for q = 0..7 those eight counts come back as the bytes of the result.  One walk over the
// list: a lane counts its entries by d = clamp(length - j0, 0, 8) in nine 7-bit fields of one 64-bit
// register, the counts are the fields' suffix sums.  Every lane of the wave must call (the walk takes as
// many steps as the wave's longest list); needs tb_n <= 127.
__device__ __forceinline__ unsigned long long noted_counts8(const uint8_t *tb, uint32_t tb_n, int j0) {
  unsigned long long hist = 0ull;
  for (uint32_t e = 0; __ballot(e < tb_n); ++e) {
    if (e < tb_n) {
      const int d = min(max((int)tb[e] - j0, 0), 8);
      hist += 1ull << (7 * d);
    }
  }
  unsigned long long exq = 0ull;
  uint32_t run = 0;
#pragma unroll
  for (int d = 8; d >= 1; --d) {
    run += (uint32_t)(hist >> (7 * d)) & 127u;
    exq |= (unsigned long long)run << (8 * (d - 1));
  }
  return exq;
}

// Epilogue of the leaf and merge kernels: the occupied slots of a workgroup's LDS table (slot s = i * NT +
// tid, i < NIT, per thread) go to consecutive places of the result list with ONE cursor atomic per
// workgroup (an atomic per wave on the single cursor word serialises the grid).  Part 1 ranks the waves'
// occupied slots inside the workgroup (*wg_total must be 0; barrier afterwards); between the parts one
// thread takes `*wg_total` places from the list's cursor; part 2 writes: emit(slot, place in the list).
template <int NIT, int NT, class Occ>
__device__ __forceinline__ void wg_rank_slots(uint32_t (&wbase)[NIT], uint32_t *wg_total, Occ occupied) {
  const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const unsigned long long m = __ballot(occupied(i * NT + tid));
    uint32_t b = 0;
    if (lane == 0 && m) b = atomicAdd(wg_total, (uint32_t)__popcll(m));
    wbase[i] = __shfl(b, 0);
  }
}
template <int NIT, int NT, class Occ, class Emit>
__device__ __forceinline__ void wg_emit_slots(const uint32_t (&wbase)[NIT], unsigned long long list_base, Occ occupied, Emit emit) {
  const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int s = i * NT + tid;
    const bool occ = occupied(s);
    const unsigned long long m = __ballot(occ);
    if (occ) emit(s, list_base + wbase[i] + __popcll(m & ((1ull << lane) - 1ull)));
  }
}

// exclusive prefix sum of cnt[0..NB) into off[0..NB); every thread of the block must call it
// (blockDim >= NB, NB a multiple of 64, NB <= 512); wtot is NB/64 words of LDS scratch
template <int NB, bool LDS_ONLY = false>
__device__ __forceinline__ void block_scan(const uint32_t *cnt, uint32_t *off, uint32_t *wtot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t x = 0, incl = 0;
  if (tid < NB) {
    x = cnt[tid];
    incl = dev_wave_scan_incl(x);
    if (lane == 63) wtot[wave] = incl;
  }
  if (LDS_ONLY) lds_barrier(); else __syncthreads();
  if (tid < NB) {
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    off[tid] = base + incl - x;
  }
  if (LDS_ONLY) lds_barrier(); else __syncthreads();
}

// Minimizers of one lane's 32 window starts (shared by msp.hip and msp2.hip).
// hi / mid = the lane's own 32 bases and the next lane's 32 bases, 2 bits each, first base in the
// top bits; chunk = index of the lane's 32-byte chunk (for the absolute-position tag).
// On return H[0..31] = packed minimizer (hash & ~127 | position tag) of the windows starting at
// the own positions (H[32..] is scratch); the result is the change mask:
// bit(63-p) set when the minimizer OCCURRENCE differs between positions p-1 and p
// (p = 0: against the previous lane's last position).
// (Measured in round 3 and not kept, profiles/r03/raw_order_minimizers_ab.txt: ordering the RAW
// canonical m-mers, top-aligned in their windows -- four instructions per position instead of seven
// -- makes 9.4 % more records (lexicographic minimizers are denser) and clumps the leaves: P1
// unchanged, P2 +1.0 ms, P3 +3.0 ms; keeping the multiply but extracting both strands top-aligned
// (six instructions) makes 2.2 % more records for no gain in P1.)
template <int W>
__device__ __forceinline__ uint64_t msp_minimizers(uint64_t hi, uint64_t mid, int64_t chunk, int m,
                                                   uint32_t (&H)[32 + W - 1]) {
  constexpr int NH = 32 + W - 1;
  constexpr int P = (W >= 16) ? 16 : (W >= 8) ? 8 : (W >= 4) ? 4 : 2;
  static_assert(W < 32, "position tags are 5 bits");
  // canonical m-mer hashes of the own 32 positions, rolled one base at a time; the low bits carry
  // the position mod 32 so that equal packed values mean the SAME m-mer occurrence (a window is
  // shorter than 32 positions, and only minima of overlapping windows are ever compared): a run
  // then never exceeds W k-mers
  {
    // Both strands by extraction instead of rolling: the 16 bases from position j are one
    // v_alignbit of two string words (static shift), the m-mer is their top 2m bits; its
    // reverse complement is the low 2m bits of a 32-bit window of the reverse-complemented string
    // ending where the m-mer starts.  Four instructions per position for the two m-mers.
    (void)chunk;
    const uint32_t D[3] = {(uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(mid >> 32)};
    uint32_t R[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      uint32_t x = __brev(D[i]);
      x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
      R[3 - i] = ~x;
    }
    R[0] = 0;
    const uint32_t mmask = (m == 16) ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    const int fsh = 32 - 2 * m;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int o = 2 * j, q = o >> 5, r = o & 31;
      const uint32_t X = r ? __builtin_amdgcn_alignbit(D[q], D[q + 1], 32 - r) : D[q];
      const int o2 = 96 - 2 * j, q2 = o2 >> 5, r2 = o2 & 31;
      const uint32_t Y = r2 ? __builtin_amdgcn_alignbit(R[q2], R[q2 + 1], 32 - r2) : R[q2];
      const uint32_t fm = X >> fsh, rm = Y & mmask;
      H[j] = (hash_mmer(min(fm, rm)) & ~127u) | (uint32_t)j;
    }
  }
#pragma unroll
  for (int j = 0; j < W - 1; ++j) H[32 + j] = dev_lane_next(H[j]);   // next lane's first W-1 hashes
  if (W >= 8) {
    // Sliding-window minimum, van Herk / Gil-Werman: cut the positions into blocks of W; a window
    // of W is a block suffix followed by a block prefix, so suffix minima S and prefix minima
    // Pm within blocks give every window minimum with one more min -- about three minima per
    // position whatever W is (doubling needs log2(W) + 1).
    uint32_t Pm[32];                       // Pm[i] = min of H[block start .. q], q = i + W - 1 (i >= 1)
#pragma unroll
    for (int i = 1; i < 32; ++i) {
      const int q = i + W - 1;
      Pm[i] = (q % W == 0) ? H[q] : min(Pm[(q % W == 0) ? i : i - 1], H[q]);
    }
    // Pm[1] starts at q = W, a block start, so the chain never reads Pm[0]; window 0 is block 0.
    // S in place: H[j] <- min of H[j .. end of j's block], for j <= 31 (descending inside blocks)
    constexpr int LASTB = (31 / W) * W + W - 1;        // end of the block that holds position 31
#pragma unroll
    for (int j = (LASTB < NH ? LASTB : NH - 1) - 1; j >= 0; --j)
      if ((j + 1) % W != 0) H[j] = min(H[j], H[j + 1]);
#pragma unroll
    for (int i = 1; i < 32; ++i) H[i] = min(H[i], Pm[i]);
  } else {
    // sliding-window minimum over W by doubling: H[j] <- min H[j .. j+P), then one combine
    // (three-input v_min3_u32 steps need fewer instructions but measured slower)
#pragma unroll
    for (int s = 1; s < P; s <<= 1) {
#pragma unroll
      for (int j = 0; j + s < NH; ++j) H[j] = min(H[j], H[j + s]);
    }
    if (W > P) {
#pragma unroll
      for (int i = 0; i < 32; ++i) H[i] = min(H[i], H[i + W - P]);
    }
  }
  // H[0..31] = minimizers of the own k-mers.
  // Cx bit(63-p): minimizer occurrence changes between positions p-1 and p (p = 0..NH-1)
  // Two VALU instructions per position: a compare into VCC and an add-with-carry that doubles
  // the accumulator and takes the compare bit in (the first position ends up in the top bit).
  // Inline asm: from C the compiler builds compare + select + or with a constant move each.
  auto push = [](uint32_t &acc, uint32_t a, uint32_t b) {
    asm("v_cmp_ne_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  };
  const uint32_t prevW = dev_lane_prev(H[31]);
  uint32_t ch = 0;
  push(ch, H[0], prevW);
#pragma unroll
  for (int p = 1; p < 32; ++p) push(ch, H[p], H[p - 1]);
  // positions 32 .. NH-1 are the next lane's first W-1 (its position 0 compares against this
  // lane's last one): its change bits, not a second set of minima and compares
  const uint32_t cl = dev_lane_next(ch) & ~(0xFFFFFFFFu >> (W - 1));
  return ((uint64_t)ch << 32) | cl;
}
