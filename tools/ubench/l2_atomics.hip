// Decision experiment (round 3): how fast are atomics on a table that stays resident in ONE XCD's
// L2?  If a level-1 bin's distinct complete runs (~82 k x 20 B for C3) could be deduplicated by
// the second-level kernel in an L2-resident table, the leaf streams would carry distinct runs +
// multiplicities instead of every record (-36 GB of HBM traffic per C3 step).  That only pays if
// the chip sustains >~ 2e11 table operations per second.
//
// Grid: 2 workgroups of 512 threads per CU.  "affine": workgroup b only touches table b % 8 (one
// table per XCD, observed round-robin dispatch; the XCC_ID register is tallied to confirm), so a
// table of <= 4 MB can stay in that XCD's L2; "shared": every workgroup touches all tables.
// Ops (random 32-bit-hashed indices, ITERS per thread, 4 independent in flight):
//   add_nr_agent  non-returning atomicAdd u32, agent scope      (what the cursors use today)
//   add_r_agent   returning atomicAdd u32, agent scope
//   add_nr_wg     non-returning atomicAdd u32, workgroup scope  (executes in the XCD's L2)
//   add_r_wg      returning atomicAdd u32, workgroup scope
//   cas64_wg      64-bit compare-and-swap, workgroup scope
//   ld16          plain 16-byte load (the table-lookup half of a dedupe step)
//   ld16_add_wg   16-byte load of a key, compare, workgroup-scope add on a hit (the dedupe step)
// build: hipcc -O3 --offload-arch=gfx950 l2_atomics.hip -o l2_atomics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

enum Op { ADD_NR_AGENT, ADD_R_AGENT, ADD_NR_WG, ADD_R_WG, CAS64_WG, LD16, LD16_ADD_WG, NOPS };
static const char *op_name[] = {"add_nr_agent", "add_r_agent", "add_nr_wg", "add_r_wg", "cas64_wg", "ld16", "ld16_add_wg"};

__device__ __forceinline__ uint32_t mixu(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int OP>
__global__ __launch_bounds__(512) void k(uint8_t *tables, uint64_t table_bytes, int affine, int iters, uint32_t *sink,
                                         unsigned long long *tally) {
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  if (threadIdx.x == 0) atomicAdd(&tally[(blockIdx.x & 7) * 16 + xcc], 1ull);
  const uint32_t nslot16 = (uint32_t)(table_bytes / 16);       // 16-byte slots per table
  uint32_t x = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x = x * 1664525u + 1013904223u;
      const uint32_t h = mixu(x);
      const uint32_t tb = affine ? (blockIdx.x & 7u) : (h >> 29);
      const uint32_t slot = (uint32_t)(((uint64_t)(h & 0x1FFFFFFFu) * nslot16) >> 29);
      uint8_t *p = tables + (uint64_t)tb * table_bytes + (uint64_t)slot * 16;
      if (OP == ADD_NR_AGENT) __hip_atomic_fetch_add((uint32_t *)p + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (OP == ADD_R_AGENT) acc += __hip_atomic_fetch_add((uint32_t *)p + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (OP == ADD_NR_WG) __hip_atomic_fetch_add((uint32_t *)p + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == ADD_R_WG) acc += __hip_atomic_fetch_add((uint32_t *)p + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (OP == CAS64_WG) {
        unsigned long long expect = (unsigned long long)h;
        __hip_atomic_compare_exchange_strong((unsigned long long *)p, &expect, (unsigned long long)h + 1ull, __ATOMIC_RELAXED,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc += (uint32_t)expect;
      }
      if (OP == LD16) {
        const uint4 q = *(const uint4 *)p;
        acc += q.x ^ q.w;
      }
      if (OP == LD16_ADD_WG) {
        const uint4 q = *(const uint4 *)p;
        // keys are all zero: the compare succeeds (and depends on the load)
        if ((q.x | q.y | q.z) == 0u) __hip_atomic_fetch_add((uint32_t *)p + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

template <int OP>
static void run(uint8_t *d_tab, uint64_t table_bytes, int affine, uint32_t *d_sink, unsigned long long *d_tally, int ncu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = ncu * 2, iters = 256;
  hipMemset(d_tab, 0, 8 * table_bytes);
  hipMemset(d_tally, 0, 8 * 16 * 8);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d_tab, table_bytes, affine, 32, d_sink, d_tally);   // warm the L2
  hipMemset(d_tally, 0, 8 * 16 * 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d_tab, table_bytes, affine, iters, d_sink, d_tally);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 512 * iters * 4;
  unsigned long long h[8 * 16];
  hipMemcpy(h, d_tally, sizeof h, hipMemcpyDeviceToHost);
  unsigned long long on = 0, all = 0;
  for (int b = 0; b < 8; ++b) for (int xx = 0; xx < 16; ++xx) { all += h[b * 16 + xx]; if (xx == b) on += h[b * 16 + xx]; }
  printf("%-14s table %8.2f MB x8 %-7s %9.3f ms  %8.2f e9 ops/s   (workgroup b on XCD b%%8: %llu of %llu)\n", op_name[OP],
         table_bytes / 1048576.0, affine ? "affine" : "shared", ms, ops / ms / 1e6, on, all);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs, L2 %d KB\n", prop.name, ncu, prop.l2CacheSize / 1024);
  const uint64_t max_tab = 512ull << 20;
  uint8_t *d_tab; uint32_t *d_sink; unsigned long long *d_tally;
  if (hipMalloc(&d_tab, 8 * max_tab) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&d_sink, 64); hipMalloc(&d_tally, 8 * 16 * 8);
  const uint64_t sizes[] = {256ull << 10, 1ull << 20, 2ull << 20, 3ull << 20, 4ull << 20, 16ull << 20, 512ull << 20};
  for (uint64_t tb : sizes) {
    for (int affine = 1; affine >= 0; --affine) {
      if (!affine && tb != (1ull << 20) && tb != (512ull << 20)) continue;
      run<ADD_NR_AGENT>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<ADD_R_AGENT>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<ADD_NR_WG>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<ADD_R_WG>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<CAS64_WG>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<LD16>(d_tab, tb, affine, d_sink, d_tally, ncu);
      run<LD16_ADD_WG>(d_tab, tb, affine, d_sink, d_tally, ncu);
    }
  }
  return 0;
}
