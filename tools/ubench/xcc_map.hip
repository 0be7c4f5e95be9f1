// Which XCD does workgroup b run on?  Tallies (blockIdx % 8, XCC_ID) for a large grid of
// 512-thread workgroups with 52 KB of LDS and uneven run times (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 xcc_map.hip -o xcc_map ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(512) void k(unsigned long long *tally, uint32_t *sink, int spin) {
  __shared__ uint32_t pad[13000];
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  uint32_t x = blockIdx.x * 2654435761u + threadIdx.x;
  const int n = spin * (1 + (int)((blockIdx.x * 2246822519u) >> 30));     // 1..4 x spin
  for (int i = 0; i < n; ++i) x = x * 1664525u + 1013904223u;
  pad[threadIdx.x] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&tally[(blockIdx.x & 7) * 16 + xcc], 1ull);
    sink[blockIdx.x & 1023] = pad[(x >> 8) % 512];
  }
}

int main() {
  unsigned long long *d; uint32_t *s;
  hipMalloc(&d, 8 * 16 * 8); hipMalloc(&s, 4096);
  for (int spin : {0, 200, 2000}) {
    hipMemset(d, 0, 8 * 16 * 8);
    hipLaunchKernelGGL(k, dim3(200000), dim3(512), 0, 0, d, s, spin);
    unsigned long long h[8 * 16];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("spin=%d\n", spin);
    for (int b = 0; b < 8; ++b) {
      printf("  blockIdx%%8=%d:", b);
      for (int x = 0; x < 8; ++x) printf(" %7llu", h[b * 16 + x]);
      printf("\n");
    }
  }
  return 0;
}
