// dense.hip -- per-read dense k-mer histograms for gfx950: the drop-in for the reference's
// SetMatrix + ComputeIndex + ComputeFreqNew launches (/root/reference/src/kmer_main.cu:107-111,
// kernels at src/kmer_kernel.cu:6-90).
//
// One 64-lane wave per read -- 16 lanes for k <= 4, four reads per wave -- (the reference uses a
// 1024-thread block per read with <= L-1 lanes active, src/kmer_main.cu:82-83).  Each lane owns a contiguous run of window starts and rolls
// the 2-bit index forward one base at a time (exact integers; the reference recomputes k powf
// terms per window, src/kmer_kernel.cu:33-46).  No Index[] array is materialised (the
// reference writes and re-reads 4 B per base).
//   k <= 6 : the read's codes are staged in LDS by coalesced dword loads; the row histogram
//            lives in LDS (4^k * 4 B <= 16 KiB per read), ds_add_u32, then one coalesced row
//            store -- the row is written exactly once, no memset, no HBM atomics.
//   k >= 7 : HBM atomics into the zeroed row, as the reference does.
// CFRK_COMPAT reproduces ComputeFreqNew's missing -1 guard: every invalid window of read i is
// added to row i-1's last bin (dropped for i == 0, where the reference writes Freq[-1]).
#include "common.h"

#include <algorithm>

namespace {

// sum over the G consecutive lanes that share a read
template <int G>
__device__ __forceinline__ int group_sum(int v) {
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int DENSE_SEG = 1024;                    // windows staged per pass (the compat cap)
constexpr int DENSE_STAGE = DENSE_SEG + 64;        // bytes of LDS per wave: segment + k-1 + alignment slack

// G lanes per read: 64 (one wave per read) or, for the small rows of k <= 4, 16 (four reads per
// wave: the per-read fixed work -- table lookups, staging, zeroing and writing the row -- is
// shared by a quarter of the lanes, and a 150-base read keeps 15 of 16 lanes busy instead of 50
// of 64).
template <bool COMPAT, bool LDS_HIST, int G>
__global__ __launch_bounds__(256) void dense_kernel(const int8_t *__restrict__ data,
                                                    const int64_t *__restrict__ start,
                                                    const int32_t *__restrict__ length, int64_t nN,
                                                    int64_t nS, int k, int32_t *__restrict__ freq,
                                                    int32_t *__restrict__ spill) {
  extern __shared__ int32_t lds[];
  const int64_t fourk = (int64_t)1 << (2 * k);
  const uint32_t mask = (uint32_t)(fourk - 1);
  constexpr int RPB = 256 / G;                      // reads per workgroup pass
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;   // `lane` = lane within the read's group
  int32_t *hist = lds + (LDS_HIST ? grp * (int)fourk : 0);
  // LDS variant: the read's codes are staged in LDS with coalesced dword loads (one load
  // instruction per 256 bytes of read instead of one byte gather per window base)
  int32_t *stage_dw = lds + RPB * (int)fourk + grp * (DENSE_STAGE / 4);
  const int8_t *stage = reinterpret_cast<const int8_t *>(stage_dw);
  if (LDS_HIST)
    for (int b = lane; b < (int)fourk; b += G) hist[b] = 0;

  for (int64_t i = (int64_t)blockIdx.x * RPB + grp; i < nS; i += (int64_t)gridDim.x * RPB) {
    const int64_t st = start[i];
    const int L = length[i];
    // compat: threadIdx.x < length-1 with blockDim 1024 (src/kmer_kernel.cu:85); native: every
    // position of the read, validity decides (src/kmer_kernel.cu:61-68)
    const int nwin_all = COMPAT ? min(max(L - 1, 0), 1024) : max(L, 0);
    int invalid = 0;
    int32_t *row = freq + i * fourk;
    for (int seg0 = 0; seg0 < (LDS_HIST ? nwin_all : 1); seg0 += DENSE_SEG) {
      const int nwin = LDS_HIST ? min(DENSE_SEG, nwin_all - seg0) : nwin_all;
      int skew = 0;
      if (LDS_HIST) {
        // bytes [A, A + nwin + k - 1) of the buffer, as the aligned dwords that cover them; a dword
        // that is not wholly inside [data, data + nN) is assembled from guarded byte loads
        const int64_t A = st + seg0;
        const int nbytes = nwin + k - 1;
        const uintptr_t abs0 = reinterpret_cast<uintptr_t>(data) + (uintptr_t)A;
        skew = (int)(abs0 & 3u);
        const int ndw = (skew + nbytes + 3) >> 2;
        for (int d = lane; d < ndw; d += G) {
          const int64_t off = A - skew + 4 * (int64_t)d;          // offset of the dword in data
          int32_t w;
          if (off >= 0 && off + 4 <= nN) {
            w = *reinterpret_cast<const int32_t *>(data + off);
          } else {
            uint32_t u = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int64_t g = off + j;
              const uint32_t c = (g >= 0 && g < nN) ? (uint32_t)(uint8_t)data[g] : 0xFFu;
              u |= c << (8 * j);
            }
            w = (int32_t)u;
          }
          stage_dw[d] = w;
        }
        // (the group's lanes sit in one wave, which executes in lock step, and LDS is in order per
        // wave: no barrier needed)
      }
      const int per = (nwin + G - 1) / G;
      const int t0 = lane * per;
      const int t1 = min(t0 + per, nwin);
      if (t0 < t1) {
        uint32_t val = 0;
        int run = 0;
        for (int p = t0; p < t1 + k - 1; ++p) {
          int c;
          if (LDS_HIST) {
            c = (int)stage[skew + p];
          } else {
            const int64_t g = st + p;
            c = (g < nN) ? (int)data[g] : -1;
          }
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
    }
    if (LDS_HIST) {
      // 16 bytes per lane and store (rows are 16-byte aligned: 4^k * 4 bytes each from an aligned base)
      {
        int4 *row4 = reinterpret_cast<int4 *>(row);
        int4 *hist4 = reinterpret_cast<int4 *>(hist);
        for (int b = lane; b < (int)fourk / 4; b += G) {
          row4[b] = hist4[b];
          hist4[b] = make_int4(0, 0, 0, 0);
        }
      }
      if (COMPAT) {
        invalid = group_sum<G>(invalid);
        if (lane == 0) spill[i] = invalid;
      }
    }
  }
}

// second pass of the LDS variant: row i-1's last bin += invalid windows of read i
// CFRK_FLOAT_INDEX (k = 13..15): the index of a window exactly as the reference computes it,
//   lint index = 0;  for i < k:  index += nuc * powf(4, (k-1)-i);      src/kmer_kernel.cu:30-46
// i.e. every step converts the running index to float, adds the (exact) product and truncates
// back.  Partial sums above 2^24 are rounded to the float grid, so for k >= 13 most windows land in
// a neighbouring bin (51 % wrong at k = 13, 94 % at k = 15) and an all-T window rounds
// UP to 4^k: the reference then adds to bin 0 of the NEXT row (beyond the last row it writes out of
// bounds; that one add is dropped here).  Float adds are IEEE round-to-nearest on both sides, the
// products are exact, so this reproduces the reference bit for bit.  One thread per window, k byte
// loads each, global atomics: a compatibility curiosity, not a fast path.
template <bool COMPAT>
__global__ __launch_bounds__(256) void dense_float_index_kernel(const int8_t *__restrict__ data,
                                                                const int64_t *__restrict__ start,
                                                                const int32_t *__restrict__ length, int64_t nN,
                                                                int64_t nS, int k, int32_t *__restrict__ freq) {
  const int64_t fourk = (int64_t)1 << (2 * k);
  const int64_t nF = nS * fourk;
  for (int64_t i = blockIdx.x; i < nS; i += gridDim.x) {
    const int64_t st = start[i];
    const int L = length[i];
    // compat: thread t < length-1 of a 1024-thread block (src/kmer_kernel.cu:85); native: every
    // position of the read incl. its terminator, guarded by Index != -1 (src/kmer_kernel.cu:61-68)
    const int nwin = COMPAT ? min(max(L - 1, 0), 1024) : max(L + 1, 0);
    for (int t = threadIdx.x; t < nwin; t += blockDim.x) {
      const int64_t id = st + t;
      if (id >= nN) continue;
      long index = 0;
      for (int j = 0; j < k; ++j) {
        const int8_t nuc = (id + j < nN) ? data[id + j] : (int8_t)-1;
        if (nuc != -1) {
          // 4^n exactly (a power of two: libm's powf returns it exactly on the host; the device's
          // powf is an exp2/log2 approximation and is not used)
          const float p4 = (float)(1u << (2 * ((k - 1) - j)));
          const float f = (float)index + (float)nuc * p4;
          index = (long)f;
        } else {
          index = -1;
          break;
        }
      }
      if (index == -1 && !COMPAT) continue;             // ComputeFreq's guard; ComputeFreqNew has none
      const int64_t pos = fourk * i + (int64_t)(int)index;   // Index[] is an int array
      if (pos >= 0 && pos < nF) atomicAdd(&freq[pos], 1);
    }
  }
}

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
  if (k <= 6) {
    void *d_spill = nullptr;
    int rc;
    if (compat && (rc = cfrk_pool_get(ctx, BUF_SPILL, (size_t)nS * 4, &d_spill))) return rc;
    const int G = (k <= 4) ? 16 : 64;                       // lanes per read
    const int rpb = 256 / G;
    const int grid = (int)std::min<int64_t>((nS + rpb - 1) / rpb, (int64_t)ctx->num_cus * 16);
    const size_t lds = (size_t)rpb * ((size_t)fourk * sizeof(int32_t) + (size_t)DENSE_STAGE);
    const void *fn = compat ? (G == 16 ? (const void *)dense_kernel<true, true, 16> : (const void *)dense_kernel<true, true, 64>)
                            : (G == 16 ? (const void *)dense_kernel<false, true, 16> : (const void *)dense_kernel<false, true, 64>);
    // more than the default 64 KiB of dynamic LDS at k = 6
    HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define CFRK_DENSE_LAUNCH(C_, G_)                                                                        \
    hipLaunchKernelGGL((dense_kernel<C_, true, G_>), dim3(grid), dim3(256), lds, ctx->stream, d_data, d_start, \
                       d_length, nN, nS, k, d_freq, (int32_t *)d_spill)
    if (compat) { if (G == 16) CFRK_DENSE_LAUNCH(true, 16); else CFRK_DENSE_LAUNCH(true, 64); }
    else { if (G == 16) CFRK_DENSE_LAUNCH(false, 16); else CFRK_DENSE_LAUNCH(false, 64); }
#undef CFRK_DENSE_LAUNCH
    if (compat && nS > 1)
      hipLaunchKernelGGL(dense_spill_kernel, dim3((unsigned)((nS - 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                         d_freq, (const int32_t *)d_spill, nS, fourk);
  } else if ((flags & CFRK_FLOAT_INDEX) && k >= 13) {
    // (k <= 12: the float accumulation is exact, the ordinary kernels already are the reference)
    const int grid = (int)std::min<int64_t>(nS, (int64_t)ctx->num_cus * 32);
    HIP_TRY(ctx, hipMemsetAsync(d_freq, 0, (size_t)nS * (size_t)fourk * 4, ctx->stream));
    if (compat) hipLaunchKernelGGL((dense_float_index_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_data, d_start, d_length, nN, nS, k, d_freq);
    else hipLaunchKernelGGL((dense_float_index_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_data, d_start, d_length, nN, nS, k, d_freq);
  } else {
    const int grid = (int)std::min<int64_t>((nS + 3) / 4, (int64_t)ctx->num_cus * 16);
    HIP_TRY(ctx, hipMemsetAsync(d_freq, 0, (size_t)nS * (size_t)fourk * 4, ctx->stream));  // SetMatrix(d_Freq, 0)
    if (compat)
      hipLaunchKernelGGL((dense_kernel<true, false, 64>), dim3(grid), dim3(256), 0, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)nullptr);
    else
      hipLaunchKernelGGL((dense_kernel<false, false, 64>), dim3(grid), dim3(256), 0, ctx->stream, d_data,
                         d_start, d_length, nN, nS, k, d_freq, (int32_t *)nullptr);
  }
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}
