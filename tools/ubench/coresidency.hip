// Can a VALU-bound kernel and an HBM-bound kernel share every CU?  (round 3, the question behind
// overlapping the partition kernel P1 with the second-level kernel P2.)
//
// tools/overlap_probe.py showed that two ordinary grids on two streams do not overlap at all: the
// first grid owns every workgroup slot until it drains.  Here both kernels are PERSISTENT with
// exact grids -- K_valu with 2 workgroups per CU (512 threads, 50 KB of LDS each: arithmetic only,
// like P1), K_mem with 1 workgroup per CU (512 threads, 56 KB of LDS: a staged copy, like P2) --
// so that 2 + 1 fit one CU's LDS (160 KB) and wave slots together.  Measured: each alone, both on
// two streams, and one fused launch of 3 workgroups per CU whose role is blockIdx / (2 * CUs).
// Every workgroup tallies the CU it ran on (XCC_ID, HW_ID), so the mix per CU is visible.
// build: hipcc -O3 --offload-arch=gfx950 coresidency.hip -o coresidency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t cu_index() {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
  return ((xcc & 7u) << 8) | (se << 5) | (sh << 4) | cu;      // 11 bits
}

// role 0: arithmetic (n_valu work items of `spin` dependent multiply-adds on 8 accumulators)
__device__ __forceinline__ void role_valu(uint32_t wg, uint32_t nwg, uint32_t items, int spin, uint32_t *sink, uint32_t *lds) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
  for (uint32_t it = wg; it < items; it += nwg) {
    for (int s = 0; s < spin; ++s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] * 1664525u + 1013904223u;
    }
    lds[threadIdx.x] = a[0];
    __syncthreads();
    a[1] ^= lds[threadIdx.x ^ 1];
    __syncthreads();
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i];
  if (r == 0x12345u) sink[0] = r;
}

// role 1: staged copy of `tiles` tiles of 2048 uint4 (32 KB): load, LDS, barrier, store
__device__ __forceinline__ void role_mem(uint32_t wg, uint32_t nwg, uint32_t tiles, const uint4 *src, uint4 *dst, uint4 *lds) {
  uint4 nx[4];
  uint32_t t = wg;
  if (t < tiles) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nx[i] = src[(uint64_t)t * 2048 + i * 512 + threadIdx.x];
  }
  for (; t < tiles; t += nwg) {
    uint4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = nx[i];
    if (t + nwg < tiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) nx[i] = src[(uint64_t)(t + nwg) * 2048 + i * 512 + threadIdx.x];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) lds[(i * 512 + threadIdx.x) ^ 5] = r[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(uint64_t)t * 2048 + i * 512 + threadIdx.x] = lds[i * 512 + threadIdx.x];
    __syncthreads();
  }
}

constexpr int LDS_VALU = 50 * 1024, LDS_MEM = 56 * 1024;

__global__ __launch_bounds__(512) void k_valu(uint32_t items, int spin, uint32_t *sink, uint32_t *tally) {
  __shared__ uint32_t lds[LDS_VALU / 4];
  if (threadIdx.x == 0) atomicAdd(&tally[cu_index() * 2 + 0], 1u);
  role_valu(blockIdx.x, gridDim.x, items, spin, sink, lds);
}

__global__ __launch_bounds__(512) void k_mem(uint32_t tiles, const uint4 *src, uint4 *dst, uint32_t *tally, int prio) {
  __shared__ uint4 lds[LDS_MEM / 16];
  // s_setprio: a wave's priority in the SIMD's issue arbitration (0..3; waves start at 0 and the
  // arbiter otherwise favours the OLDEST wave -- which is why the launch order matters below)
  if (prio) __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0) atomicAdd(&tally[cu_index() * 2 + 1], 1u);
  role_mem(blockIdx.x, gridDim.x, tiles, src, dst, lds);
}

// fused: workgroups [0, n_valu) do arithmetic, the rest copy (every workgroup allocates the larger LDS)
__global__ __launch_bounds__(512) void k_fused(uint32_t n_valu, uint32_t items, int spin, uint32_t tiles, const uint4 *src, uint4 *dst,
                                               uint32_t *sink, uint32_t *tally, int prio) {
  __shared__ uint4 lds[LDS_VALU / 16];     // 50 KB: three per CU; the copy role stages 2048 uint4 = 32 KB
  const bool valu = blockIdx.x < n_valu;
  if (prio && !valu) __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0) atomicAdd(&tally[cu_index() * 2 + (valu ? 0 : 1)], 1u);
  if (valu) role_valu(blockIdx.x, n_valu, items, spin, sink, reinterpret_cast<uint32_t *>(lds));
  else role_mem(blockIdx.x - n_valu, gridDim.x - n_valu, tiles, src, dst, lds);
}

static void show_mix(const char *what, uint32_t *d_tally) {
  std::vector<uint32_t> h(2048 * 2);
  (void)hipMemcpy(h.data(), d_tally, h.size() * 4, hipMemcpyDeviceToHost);
  int hist[5][5] = {};
  int cus = 0;
  for (int c = 0; c < 2048; ++c) {
    const uint32_t a = h[c * 2], b = h[c * 2 + 1];
    if (!a && !b) continue;
    ++cus;
    hist[a > 4 ? 4 : a][b > 4 ? 4 : b]++;
  }
  printf("    %s: %d CUs seen; CUs by (valu workgroups, mem workgroups):", what, cus);
  for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) if (hist[a][b]) printf(" (%d,%d)x%d", a, b, hist[a][b]);
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const uint32_t tiles = 1u << 19;                     // 2^19 x 32 KB = 17.2 GB read + 17.2 GB written
  uint4 *src, *dst; uint32_t *sink, *tally;
  CHECK(hipMalloc(&src, (size_t)tiles * 32768)); CHECK(hipMalloc(&dst, (size_t)tiles * 32768));
  CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&tally, 2048 * 2 * 4));
  CHECK(hipMemset(src, 1, (size_t)tiles * 32768));
  hipStream_t sa, sb;
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, ea, eb;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&ea)); CHECK(hipEventCreate(&eb));
  const uint32_t items = 1u << 19;
  const int spin = 96;
  auto ms_of = [&](hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; };
  printf("device %s, %d CUs\n", prop.name, ncu);
  for (int rep = 0; rep < 2; ++rep) {
    for (int wpc : {2, 3}) {
      CHECK(hipMemsetAsync(tally, 0, 2048 * 2 * 4, sa));
      CHECK(hipEventRecord(e0, sa));
      hipLaunchKernelGGL(k_valu, dim3(ncu * wpc), dim3(512), 0, sa, items, spin, sink, tally);
      CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1));
      printf("valu alone, %d workgroups per CU: %.3f ms\n", wpc, ms_of(e0, e1));
      if (rep == 0) show_mix("placement", tally);
    }
    for (int wpc : {1, 2}) {
      CHECK(hipMemsetAsync(tally, 0, 2048 * 2 * 4, sa));
      CHECK(hipEventRecord(e0, sa));
      hipLaunchKernelGGL(k_mem, dim3(ncu * wpc), dim3(512), 0, sa, tiles, src, dst, tally, 0);
      CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1));
      const float ms = ms_of(e0, e1);
      printf("mem alone, %d workgroups per CU: %.3f ms = %.2f TB/s (read + write)\n", wpc, ms, 2.0 * tiles * 32768 / ms / 1e9);
      if (rep == 0) show_mix("placement", tally);
    }
    // two streams, exact grids: 2 valu + 1 mem per CU (without / with s_setprio 3 in the mem kernel)
    for (int prio = 0; prio < 2; ++prio) {
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemset(tally, 0, 2048 * 2 * 4));
      CHECK(hipEventRecord(e0, sa));
      CHECK(hipStreamWaitEvent(sb, e0, 0));
      hipLaunchKernelGGL(k_valu, dim3(ncu * 2), dim3(512), 0, sa, items, spin, sink, tally);
      hipLaunchKernelGGL(k_mem, dim3(ncu), dim3(512), 0, sb, tiles, src, dst, tally, prio);
      CHECK(hipEventRecord(ea, sa)); CHECK(hipEventRecord(eb, sb));
      CHECK(hipEventSynchronize(ea)); CHECK(hipEventSynchronize(eb));
      printf("two streams, valu first%s: valu done after %.3f ms, mem done after %.3f ms\n", prio ? ", mem waves at s_setprio 3" : "",
             ms_of(e0, ea), ms_of(e0, eb));
      show_mix("placement", tally);
    }
    // the same with the mem kernel first
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(tally, 0, 2048 * 2 * 4));
    CHECK(hipEventRecord(e0, sb));
    CHECK(hipStreamWaitEvent(sa, e0, 0));
    hipLaunchKernelGGL(k_mem, dim3(ncu), dim3(512), 0, sb, tiles, src, dst, tally, 0);
    hipLaunchKernelGGL(k_valu, dim3(ncu * 2), dim3(512), 0, sa, items, spin, sink, tally);
    CHECK(hipEventRecord(ea, sa)); CHECK(hipEventRecord(eb, sb));
    CHECK(hipEventSynchronize(ea)); CHECK(hipEventSynchronize(eb));
    printf("two streams, mem first: valu done after %.3f ms, mem done after %.3f ms\n", ms_of(e0, ea), ms_of(e0, eb));
    show_mix("placement", tally);
    // fused launch
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(tally, 0, 2048 * 2 * 4));
    CHECK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(k_fused, dim3(ncu * 3), dim3(512), 0, sa, (uint32_t)ncu * 2, items, spin, tiles, src, dst, sink, tally, 0);
    CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1));
    printf("fused launch (roles by blockIdx, 3 workgroups per CU): %.3f ms\n", ms_of(e0, e1));
    show_mix("placement", tally);
    CHECK(hipMemset(tally, 0, 2048 * 2 * 4));
    CHECK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(k_fused, dim3(ncu * 3), dim3(512), 0, sa, (uint32_t)ncu * 2, items, spin, tiles, src, dst, sink, tally, 1);
    CHECK(hipEventRecord(e1, sa)); CHECK(hipEventSynchronize(e1));
    printf("fused launch, mem waves at s_setprio 3: %.3f ms\n", ms_of(e0, e1));
  }
  return 0;
}
