// synth.hip -- deterministic synthetic reads generated on device, laid out as
// the reference's struct read (/root/reference/src/tipos.h:23-30, src/fastaIO.h:74-102).
// Not part of the timed path; it only has to be reproducible on the CPU.
#include "common.h"

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void synth_kernel(int64_t r0, int64_t R, int L, int64_t Glen,
                                                    uint64_t seedG, uint64_t seedR, uint64_t seedS,
                                                    int uniform, int8_t *__restrict__ data,
                                                    int64_t *__restrict__ start,
                                                    int32_t *__restrict__ length) {
  // one wave per read: lane j writes bases j, j+64, ...
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < R; i += nwaves) {
    const uint64_t r = (uint64_t)(r0 + i);
    int8_t *d = data + i * (int64_t)(L + 1);
    uint64_t pos = 0;
    int rcs = 0;
    if (!uniform) {
      pos = dev_splitmix64(seedR ^ r) % (uint64_t)(Glen - L + 1);
      rcs = (int)(dev_splitmix64(seedS ^ r) & 1);
    }
    for (int j = lane; j < L; j += 64) {
      int c;
      if (uniform) c = (int)(dev_splitmix64(seedR ^ (r * 256 + (uint64_t)j)) & 3);
      else if (!rcs) c = (int)(dev_splitmix64(seedG + pos + (uint64_t)j) & 3);
      else c = 3 - (int)(dev_splitmix64(seedG + pos + (uint64_t)(L - 1 - j)) & 3);
      d[j] = (int8_t)c;
    }
    if (lane == 0) {
      d[L] = -1;
      if (start) start[i] = i * (int64_t)(L + 1);
      if (length) length[i] = L;
    }
  }
}

}  // namespace

int cfrk_launch_synth(cfrk_ctx *ctx, int64_t r0, int64_t R, int L, int64_t Glen, uint64_t seedG,
                      uint64_t seedR, uint64_t seedS, int uniform, int8_t *d_data, int64_t *d_start,
                      int32_t *d_length) {
  const int64_t blocks = std::min<int64_t>((R + 3) / 4, (int64_t)ctx->num_cus * 32);
  hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, r0, R, L, Glen,
                     seedG, seedR, seedS, uniform, d_data, d_start, d_length);
  HIP_TRY(ctx, hipGetLastError());
  return CFRK_OK;
}
