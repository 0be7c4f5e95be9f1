// Issue-rate micro-benchmark for the integer VALU ops the counting kernels lean on (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a[8];
  uint64_t q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; q[i] = ((uint64_t)a[i] << 32) | (a[i] * 77u); }
  const uint32_t c = seed | 1u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = a[i] + c;                                   // v_add_u32
      if (OP == 1) a[i] = a[i] * c;                                   // v_mul_lo_u32
      if (OP == 2) a[i] = __umul24(a[i], c);                          // v_mul_u32_u24
      if (OP == 3) a[i] = __umulhi(a[i], c);                          // v_mul_hi_u32
      if (OP == 4) a[i] = min(a[i], a[(i + 1) & 7] ^ c);              // v_xor + v_min
      if (OP == 5) q[i] = (q[i] << 2) | (q[i] >> 61);                 // 64-bit shifts
      if (OP == 6) a[i] = (q[i] < q[(i + 1) & 7]) ? a[i] + 1 : a[i];  // v_cmp_u64 + cndmask/add
      if (OP == 7) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 7);  // v_alignbit
      if (OP == 8) a[i] = __shfl_down(a[i], 1);                       // DPP / ds_bpermute
      if (OP == 9) q[i] = q[i] + (uint64_t)c;                         // 64-bit add
      if (OP == 10) a[i] = (a[i] & c) | (a[(i + 1) & 7] >> 3);       // and_or / shift
      if (OP == 11) a[i] = __builtin_amdgcn_ubfe(a[i] ^ c, 5, 13) + a[i];      // bfe
      if (OP == 12) a[i] = __popc(a[i]) + a[i] * 3u;                  // popc + mul small const
      if (OP == 13) a[i] = __brev(a[i]) ^ c;                          // brev
      if (OP == 14) a[i] = (a[i] * c) ^ a[(i + 1) & 7];               // v_mul_lo_u32 + v_xor (not foldable)
      if (OP == 15) a[i] = (a[i] * 0x9E3779B1u) ^ a[(i + 1) & 7];     // v_mul_lo_u32 by a literal + v_xor
      if (OP == 16) a[i] = __umulhi(a[i] & 0x80808080u, 0x10080402u) ^ a[(i + 1) & 7];  // and + mul_hi + xor
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, uint32_t *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 16;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waveops = (double)blocks * 4 * ITERS * 8;
  printf("%-28s %8.3f ms  %.3f wave-stmts/cycle/CU (2.4 GHz, 256 CUs)\n", name, ms, waveops / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
  uint32_t *d;
  hipMalloc(&d, 256 * 16 * 256 * 4);
  run<0>("add_u32", d); run<1>("mul_lo_u32", d); run<2>("mul_u32_u24", d); run<3>("mul_hi_u32", d);
  run<4>("xor+min", d); run<5>("shl64|shr64", d); run<6>("cmp_u64+sel", d); run<7>("alignbit", d);
  run<8>("shfl_down 1", d); run<9>("add_u64", d); run<10>("and,shr,or", d); run<11>("xor,bfe,add", d);
  run<12>("popc + mul3", d); run<13>("brev,xor", d);
  run<14>("mul_lo,xor", d); run<15>("mul_lo literal,xor", d); run<16>("and,mul_hi,xor", d);
  return 0;
}
