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
      atomicAdd(&t.cnt[h], add);
      return;
    }
    h = (h + 1) & t.mask;
  }
  t.stats[ST_OVERFLOW] = 1;
}

__device__ __forceinline__ void table_add2(const TableView &t, uint64_t lo, uint64_t hi, uint32_t add) {
  uint64_t h = dev_mix64(lo ^ dev_mix64(hi)) >> t.shift;
  uint32_t probe = 0, spins = 0;
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
    const uint64_t klo = ld_agent(&t.lo[h]);
    const uint64_t khi = ld_agent(&t.hi[h]);
    if (klo == lo && khi == hi) {
      atomicAdd(&t.cnt[h], add);
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
