// Issue-rate micro-benchmark for the integer VALU / LDS instructions the counting kernels lean on
// (gfx950).  Every measured instruction is `asm volatile`, so the compiler can neither fold the
// loop nor merge instructions; 8 independent accumulators per lane, several waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
// Prints wave-instructions per clock per CU (clock measured with s_memtime against wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { ADD, MUL_LO, MUL_U24, MAD_U24, MUL_HI, MIN_U, XOR, AND_OR, LSHL_OR, ALIGNBIT, PERM, BFE, CNDMASK, CMP_ADDC,
          DPP_SHR, LSHL64, ADD64, MIN3, XAD, FMA, DS_READ128, DS_ADD, DS_BPERM, NOPS,
          PK_MIN_U16, PK_MUL_LO_U16, PK_ADD_U16, PK_LSHL_B16, PK_MAD_U16, PK_MAX_U16, PK_SUB_U16, MIN_DPP };

template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed, unsigned long long *clk) {
  __shared__ uint4 lds[2048];
  uint32_t a[8], b[8];
  uint64_t q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = a[i] ^ 0x5bd1e995u; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = make_uint4(i, i, i, i);
  __syncthreads();
  const uint32_t c = seed | 1u;
  const uint32_t la = (threadIdx.x * 16u) & 0x7FF0u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#define ONE(i)                                                                                                    \
    if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                     \
    if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                               \
    if (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));                             \
    if (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));              \
    if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                               \
    if (OP == MIN_U) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                \
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                                  \
    if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));                \
    if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b[i]));                       \
    if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));                     \
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                    \
    if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 5, 13" : "+v"(a[i]));                                           \
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : );                  \
    if (OP == CMP_ADDC) asm volatile("v_cmp_ne_u32_e32 vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc"); \
    if (OP == DPP_SHR) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));    \
    if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, 2, %0" : "+v"(q[i]));                                        \
    if (OP == ADD64) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(c), "v"(c) : "vcc"); \
    if (OP == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                    \
    if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                      \
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                      \
    if (OP == DS_ADD) asm volatile("ds_add_u32 %0, %1" : : "v"(la), "v"(c) : "memory");                            \
    if (OP == DS_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(la)); \
    if (OP == PK_MIN_U16) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                        \
    if (OP == PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                        \
    if (OP == PK_MUL_LO_U16) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));                     \
    if (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                        \
    if (OP == PK_SUB_U16) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));                        \
    if (OP == PK_LSHL_B16) asm volatile("v_pk_lshlrev_b16 %0, 2, %0" : "+v"(a[i]));                                \
    if (OP == PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));            \
    if (OP == MIN_DPP) asm volatile("v_min_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i])); \
    if (OP == NOPS) asm volatile("s_nop 0");
    REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
    if (OP == DS_READ128) {
      uint4 x0, x1, x2, x3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(la) : "memory");
      a[0] ^= x0.x ^ x1.y ^ x2.z ^ x3.w;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  out[blockIdx.x * 512 + threadIdx.x] = r + lds[threadIdx.x].x;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

static double g_ghz = 2.4;

template <int OP>
void run(const char *name, uint32_t *d, unsigned long long *dclk, int per_iter = 32, int instr_per_stmt = 1) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;        // 512 threads: 8 waves; 2 blocks per CU resident -> 4 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 12345u, dclk);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, d, 12345u, dclk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waveops = (double)blocks * 8 * ITERS * per_iter * instr_per_stmt;
  printf("%-34s %8.3f ms  %6.3f wave-instr/clk/CU at %.2f GHz  (%.1f per ns per CU)\n", name, ms,
         waveops / (ms * 1e-3) / (g_ghz * 1e9) / 256, g_ghz, waveops / (ms * 1e6) / 256);
}

int main() {
  uint32_t *d;
  unsigned long long *dclk, hclk = 0;
  hipMalloc(&d, 256 * 8 * 512 * 4);
  hipMalloc(&dclk, 8);
  {
    // shader clock: cycles one workgroup counts for the s_nop loop against the launch's wall time of
    // a single resident wave set (approximate; printed so that the per-clock figures can be rescaled)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NOPS>, dim3(256), dim3(512), 0, 0, d, 1u, dclk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NOPS>, dim3(256), dim3(512), 0, 0, d, 1u, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hclk, dclk, 8, hipMemcpyDeviceToHost);
    printf("s_memtime/readcyclecounter ticks in the loop: %llu over <= %.3f ms wall (tick rate >= %.3f GHz; this counter runs at a fixed 100 MHz on some parts)\n",
           hclk, ms, hclk / (ms * 1e6));
  }
  run<ADD>("v_add_u32", d, dclk); run<MUL_LO>("v_mul_lo_u32", d, dclk); run<MUL_U24>("v_mul_u32_u24", d, dclk);
  run<MAD_U24>("v_mad_u32_u24", d, dclk); run<MUL_HI>("v_mul_hi_u32", d, dclk); run<MIN_U>("v_min_u32", d, dclk);
  run<XOR>("v_xor_b32", d, dclk); run<AND_OR>("v_and_or_b32", d, dclk); run<LSHL_OR>("v_lshl_or_b32", d, dclk);
  run<ALIGNBIT>("v_alignbit_b32", d, dclk); run<PERM>("v_perm_b32", d, dclk); run<BFE>("v_bfe_u32", d, dclk);
  run<CNDMASK>("v_cndmask_b32", d, dclk); run<CMP_ADDC>("v_cmp_ne_u32 + v_addc_co_u32", d, dclk, 32, 2);
  run<DPP_SHR>("v_mov_b32_dpp wave_shr:1", d, dclk); run<LSHL64>("v_lshlrev_b64", d, dclk);
  run<ADD64>("v_add_co_u32 + v_addc_co_u32", d, dclk, 32, 2); run<MIN3>("v_min3_u32", d, dclk);
  run<XAD>("v_xad_u32", d, dclk); run<FMA>("v_fma_f32 (reference)", d, dclk);
  run<DS_READ128>("ds_read_b128 (4 + wait per iter)", d, dclk, 4); run<DS_ADD>("ds_add_u32 (no return)", d, dclk);
  run<DS_BPERM>("ds_bpermute_b32 + wait", d, dclk); run<NOPS>("s_nop 0", d, dclk);
  // round 4: the packed 16-bit forms a two-positions-per-instruction minimizer front end would need
  run<PK_MIN_U16>("v_pk_min_u16", d, dclk); run<PK_MAX_U16>("v_pk_max_u16", d, dclk); run<PK_MUL_LO_U16>("v_pk_mul_lo_u16", d, dclk);
  run<PK_ADD_U16>("v_pk_add_u16", d, dclk); run<PK_SUB_U16>("v_pk_sub_u16", d, dclk); run<PK_LSHL_B16>("v_pk_lshlrev_b16", d, dclk);
  run<PK_MAD_U16>("v_pk_mad_u16", d, dclk); run<MIN_DPP>("v_min_u32_dpp row_shr:1", d, dclk);
  return 0;
}
