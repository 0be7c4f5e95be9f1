// export_sort.hip -- order an exported (key, count) list by key on the device, for the host-side
// cfrk_global_export (SURVEY 8d "full sorted (key,count) dump").  Not part of the counting path:
// a plain library radix sort (hipCUB) replaces what used to be an indirect std::sort of 10^8
// indices on the host.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

__global__ void iota_kernel(uint32_t *idx, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (uint32_t)i;
}
__global__ void gather64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                uint64_t *__restrict__ dst, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void gather32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                uint32_t *__restrict__ dst, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

}  // namespace

// Sorts the n entries (d_lo, d_hi or NULL, d_cnt) by (hi, lo).  On return *s_lo / *s_hi / *s_cnt
// point at the sorted arrays (device memory owned by the context's pool, valid until the next
// library call that uses the scratch slot).
int cfrk_sort_export(cfrk_ctx *ctx, const uint64_t *d_lo, const uint64_t *d_hi, const uint32_t *d_cnt, uint64_t n,
                     const uint64_t **s_lo, const uint64_t **s_hi, const uint32_t **s_cnt) {
  if (n >= (1ull << 32)) return cfrk_fail(ctx, CFRK_ERR_ARG, "list of %llu entries is too long to sort", (unsigned long long)n);
  const int in = (int)n;
  size_t tmp_a = 0, tmp_b = 0;
  HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_a, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, in, 0, 64, ctx->stream));
  tmp_b = tmp_a;
  const size_t a8 = (n * 8 + 255) & ~(size_t)255, a4 = (n * 4 + 255) & ~(size_t)255;
  const size_t tmp = (std::max(tmp_a, tmp_b) + 255) & ~(size_t)255;
  // scratch layout: [tmp | k1 (8) | k2 (8) | k3 (8) | v1 (4) | v2 (4) | v3 (4)]
  void *p;
  int rc = cfrk_pool_get(ctx, BUF_SCRATCH, tmp + 3 * a8 + 3 * a4, &p);
  if (rc) return rc;
  char *base = (char *)p;
  void *d_tmp = base;
  uint64_t *k1 = (uint64_t *)(base + tmp), *k2 = (uint64_t *)(base + tmp + a8), *k3 = (uint64_t *)(base + tmp + 2 * a8);
  uint32_t *v1 = (uint32_t *)(base + tmp + 3 * a8), *v2 = (uint32_t *)(base + tmp + 3 * a8 + a4),
           *v3 = (uint32_t *)(base + tmp + 3 * a8 + 2 * a4);
  size_t tsz = tmp;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (!d_hi) {
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(d_tmp, tsz, d_lo, k1, d_cnt, v1, in, 0, 64, ctx->stream));
    *s_lo = k1; *s_hi = nullptr; *s_cnt = v1;
    return CFRK_OK;
  }
  // two words: stable LSD -- by the low word first, then by the high word, carrying indices
  hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, ctx->stream, v1, n);
  HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(d_tmp, tsz, d_lo, k1, (const uint32_t *)v1, v2, in, 0, 64, ctx->stream));
  hipLaunchKernelGGL(gather64_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_hi, (const uint32_t *)v2, k1, n);
  tsz = tmp;
  HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(d_tmp, tsz, (const uint64_t *)k1, k2, (const uint32_t *)v2, v3, in, 0, 64, ctx->stream));
  hipLaunchKernelGGL(gather64_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_lo, (const uint32_t *)v3, k3, n);
  hipLaunchKernelGGL(gather32_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_cnt, (const uint32_t *)v3, v1, n);
  HIP_TRY(ctx, hipGetLastError());
  *s_lo = k3; *s_hi = k2; *s_cnt = v1;
  return CFRK_OK;
}
