// table.h -- device-side view of the global open-addressing table (see global_hash.hip) and
// of a compact (key,count) list; shared by global_hash.hip and msp.hip.
#pragma once
#include "common.h"

constexpr uint32_t LOCKED = 0xFFFFFFFFu;

template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct TableView {
  uint64_t *lo, *hi;
  uint32_t *cnt;
  uint64_t *stats;
  uint64_t mask;
  int shift;  // 64 - log2(cap)
};

// Counts are 32-bit and SATURATE at CFRK_COUNT_MAX = 2^32 - 2 (the reference's rows are `int`,
// /root/reference/src/tipos.h:28, and simply wrap).  sat_add: a returning add; the one add in 2^32 that carries
// the word past the maximum (old > MAX - add) puts it back to MAX and raises ST_SAT.  Between its wrapped add and its
// exchange other adds may land on the small wrapped value: they are overwritten by the exchange, and any add after
// the LAST exchange finds old = MAX and exchanges again -- the word ends at MAX whatever the order.  (One-word table
// only: its count word is not the slot state, so the transient wrapped value harms nobody.)
__device__ __forceinline__ void sat_add(uint32_t *cnt, uint32_t add, uint64_t *stats) {
  const uint32_t old = atomicAdd(cnt, add);
  if (old > CFRK_COUNT_MAX - add) {
    atomicExch(cnt, CFRK_COUNT_MAX);
    stats[ST_SAT] = 1;
  }
}
// ... where the count word is also the slot state (two-word table: 0 = empty, 0xFFFFFFFF = locked) a wrapped value
// must never be visible: compare-and-swap from the value just read
__device__ __forceinline__ void sat_add_cas(uint32_t *cnt, uint32_t cur, uint32_t add, uint64_t *stats) {
  for (;;) {
    const bool over = cur > CFRK_COUNT_MAX - add;
    const uint32_t want = over ? CFRK_COUNT_MAX : cur + add;
    const uint32_t old = atomicCAS(cnt, cur, want);
    if (old == cur) { if (over) stats[ST_SAT] = 1; return; }
    cur = old;                                       // (never 0 or LOCKED again: the slot holds this key)
  }
}
// the k = 32 all-T forward key lives in a 64-bit side word (ST_ONES): clamped where it is read
__device__ __forceinline__ uint32_t sat_ones(uint64_t ones) { return ones > (uint64_t)CFRK_COUNT_MAX ? CFRK_COUNT_MAX : (uint32_t)ones; }

__device__ __forceinline__ void table_add1(const TableView &t, uint64_t key, uint32_t add) {
  if (key == CFRK_EMPTY_KEY) {
    atomicAdd((unsigned long long *)&t.stats[ST_ONES], (unsigned long long)add);
    return;
  }
  uint64_t h = dev_mix64(key) >> t.shift;
  for (uint32_t probe = 0; probe < CFRK_MAX_PROBE; ++probe) {
    uint64_t cur = ld_agent(&t.lo[h]);   // a stale read can only show EMPTY; the CAS decides
    if (cur == CFRK_EMPTY_KEY) {
      cur = atomicCAS((unsigned long long *)&t.lo[h], (unsigned long long)CFRK_EMPTY_KEY,
                      (unsigned long long)key);
      if (cur == CFRK_EMPTY_KEY) cur = key;
    }
    if (cur == key) {
      sat_add(&t.cnt[h], add, t.stats);
      return;
    }
    h = (h + 1) & t.mask;
  }
  t.stats[ST_OVERFLOW] = 1;
}

__device__ __forceinline__ void table_add2(const TableView &t, uint64_t lo, uint64_t hi, uint32_t add) {
  uint64_t h = dev_mix64(lo ^ dev_mix64(hi)) >> t.shift;
  uint32_t probe = 0, spins = 0;
  if (add > CFRK_COUNT_MAX) add = CFRK_COUNT_MAX;    // (a published count is never the LOCKED pattern)
  while (probe < CFRK_MAX_PROBE && spins < (1u << 24)) {
    uint32_t c = ld_agent(&t.cnt[h]);
    if (c == 0) {
      uint32_t old = atomicCAS(&t.cnt[h], 0u, LOCKED);
      if (old == 0) {
        __hip_atomic_store(&t.lo[h], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&t.hi[h], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&t.cnt[h], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      ++spins;
      continue;  // somebody else took the slot: look at it again
    }
    if (c == LOCKED) { ++spins; continue; }
    // the count word vouches for the key words: they are read after it (pairs with the claimer's release)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint64_t klo = ld_agent(&t.lo[h]);
    const uint64_t khi = ld_agent(&t.hi[h]);
    if (klo == lo && khi == hi) {
      // in the lower half of the range the plain add is safe (2^31 adds of one cannot land on one word between this
      // lane's read of c and its add: same-address atomics take ~5 ns each); above it, or with a weight of its own
      // (merges), compare-and-swap
      if (add == 1u && c < 0x80000000u) atomicAdd(&t.cnt[h], 1u);
      else sat_add_cas(&t.cnt[h], c, add, t.stats);
      return;
    }
    h = (h + 1) & t.mask;
    ++probe;
  }
  t.stats[ST_OVERFLOW] = 1;
}


// where digest / export read the result from
struct ResultSrc {
  const uint64_t *lo, *hi;
  const uint32_t *cnt;
  uint64_t n;        // slots (table) or entries (list)
  int kind;          // 0: one-word table, 1: two-word table, 2: one-word list, 3: two-word list
  uint64_t *stats;
};
