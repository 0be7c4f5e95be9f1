// dense.hip -- per-read dense k-mer histograms for gfx950: the drop-in for the reference's
// SetMatrix + ComputeIndex + ComputeFreqNew launches (/root/reference/src/kmer_main.cu:107-111,
// kernels at src/kmer_kernel.cu:6-90).
//
// One 64-lane wave per read (the reference uses a 1024-thread block per read with <= L-1 lanes
// active, src/kmer_main.cu:82-83).  Each lane owns a contiguous run of window starts and rolls
// the 2-bit index forward one base at a time (exact integers; the reference recomputes k powf
// terms per window, src/kmer_kernel.cu:33-46).  No Index[] array is materialised (the
// reference writes and re-reads 4 B per base).
//   k <= 6 : row histogram lives in LDS (4^k * 4 B <= 16 KiB per wave), ds_add_u32, then one
//            coalesced row store -- the row is written exactly once, no memset, no HBM atomics.
//   k >= 7 : HBM atomics into the zeroed row, as the reference does.
// CFRK_COMPAT reproduces ComputeFreqNew's missing -1 guard: every invalid window of read i is
// added to row i-1's last bin (dropped for i == 0, where the reference writes Freq[-1]).
#include "common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ int wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool COMPAT, bool LDS_HIST>
__global__ __launch_bounds__(256) void dense_kernel(const int8_t *__restrict__ data,
                                                    const int64_t *__restrict__ start,
                                                    const int32_t *__restrict__ length, int64_t nN,
                                                    int64_t nS, int k, int32_t *__restrict__ freq,
                                                    int32_t *__restrict__ spill) {
  extern __shared__ int32_t lds[];
  const int64_t fourk = (int64_t)1 << (2 * k);
  const uint32_t mask = (uint32_t)(fourk - 1);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t *hist = lds + (LDS_HIST ? wave * (int)fourk : 0);
  if (LDS_HIST)
    for (int b = lane; b < (int)fourk; b += 64) hist[b] = 0;

  for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < nS; i += (int64_t)gridDim.x * 4) {
    const int64_t st = start[i];
    const int L = length[i];
    // compat: threadIdx.x < length-1 with blockDim 1024 (src/kmer_kernel.cu:85); native: every
    // position of the read, validity decides (src/kmer_kernel.cu:61-68)
    int nwin = COMPAT ? min(max(L - 1, 0), 1024) : max(L, 0);
    const int per = (nwin + 63) >> 6;
    const int t0 = lane * per;
    const int t1 = min(t0 + per, nwin);
    int invalid = 0;
    int32_t *row = freq + i * fourk;
    if (t0 < t1) {
      uint32_t val = 0;
      int run = 0;
      for (int p = t0; p < t1 + k - 1; ++p) {
        const int64_t g = st + p;
        const int c = (g < nN) ? (int)data[g] : -1;
        if (c < 0 || c > 3) {
          run = 0;
        } else {
          val = ((val << 2) | (uint32_t)c) & mask;
          ++run;
        }
        if (p >= t0 + k - 1) {
          if (run >= k) {
            if (LDS_HIST) atomicAdd(&hist[val], 1);
            else atomicAdd(&row[val], 1);
          } else if (COMPAT) {
            if (LDS_HIST) ++invalid;
            else if (i > 0) atomicAdd(&row[-1], 1);   // Freq[fourk*i + (-1)]
          }
        }
      }
    }
    if (LDS_HIST) {
      for (int b = lane; b < (int)fourk; b += 64) {
        row[b] = hist[b];
        hist[b] = 0;
      }
      if (COMPAT) {
        invalid = wave_sum(invalid);
        if (lane == 0) spill[i] = invalid;
      }
    }
  }
}

// second pass of the LDS variant: row i-1's last bin += invalid windows of read i
__global__ void dense_spill_kernel(int32_t *__restrict__ freq, const int32_t *__restrict__ spill,
                                   int64_t nS, int64_t fourk) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (i < nS) {
    int s = spill[i];
    if (s) freq[i * fourk - 1] += s;
  }
}

}  // namespace

int cfrk_launch_dense(cfrk_ctx *ctx, const int8_t *d_data, const int64_t *d_start,
                      const int32_t *d_length, int64_t nN, int64_t nS, int k, int flags,
                      int32_t *d_freq) {
  const bool compat = (flags & CFRK_COMPAT) != 0;
  const int64_t fourk = (int64_t)1 << (2 * k);
  const int64_t want = (nS + 3) / 4;
  const int grid = (int)std::min<int64_t>(want, (int64_t)ctx->num_cus * 16);
  if (k <= 6) {
    void *d_spill = nullptr;
    int rc;
    if (compat && (rc = cfrk_pool_get(ctx, BUF_SPILL, (size_t)nS * 4, &d_spill))) return rc;
    const size_t lds = 4 * (size_t)fourk * sizeof(int32_t);
    if (compat) {
      hipLaunchKernelGGL((dense_kernel<true, true>), dim3(grid), dim3(256), lds, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)d_spill);
      if (nS > 1)
        hipLaunchKernelGGL(dense_spill_kernel, dim3((unsigned)((nS - 1 + 255) / 256)), dim3(256), 0,
                           ctx->stream, d_freq, (const int32_t *)d_spill, nS, fourk);
    } else {
      hipLaunchKernelGGL((dense_kernel<false, true>), dim3(grid), dim3(256), lds, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)nullptr);
    }
  } else {
    HIP_TRY(ctx, hipMemsetAsync(d_freq, 0, (size_t)nS * (size_t)fourk * 4, ctx->stream));  // SetMatrix(d_Freq, 0)
    if (compat)
      hipLaunchKernelGGL((dense_kernel<true, false>), dim3(grid), dim3(256), 0, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)nullptr);
    else
      hipLaunchKernelGGL((dense_kernel<false, false>), dim3(grid), dim3(256), 0, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)nullptr);
  }
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}
