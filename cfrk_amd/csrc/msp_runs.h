// msp_runs.h -- the parts of the multi-GPU exchange by runs (msp.hip: "multi-GPU by runs") that do not
// depend on the record format: where every leaf's rows go in the packed buffer (sender) and where
// every received segment goes in the owner's leaf streams.  Included by msp.hip (16-byte records,
// one row each) and msp2.hip (32-byte records, two rows each) inside their anonymous namespaces.
#pragma once

constexpr uint32_t RUN_NOTED = 0xFFFFFFFFu;        // (a record's header word has the top 8 bits clear)
constexpr int NOTES_PER_ROW = 8;
// Packed form, one segment per owner: [header: the owner's leaves_per_part x (distinct, truncated,
// noted) sizes, uint32 triples, padded to whole 16-byte rows][leaf after leaf: the distinct complete
// runs, the truncated runs, the notes (16 bits each, eight per row)].
static inline int runs_header_rows(int lpp) { return (lpp * 3 * (int)sizeof(uint32_t) + 15) / 16; }

// exclusive prefix sum over a block of 1024 threads (wtot: 16 words of LDS); returns the exclusive
// prefix of x, *total = the block's sum
__device__ __forceinline__ uint64_t block_scan_u64(uint64_t x, unsigned long long *wtot, uint64_t *total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t incl = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  __syncthreads();                               // (wtot may still be read from a previous scan)
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint64_t base = 0, all = 0;
  for (int w = 0; w < 16; ++w) { const uint64_t t = wtot[w]; base += (w < wave) ? t : 0; all += t; }
  *total = all;
  return base + incl - x;
}

// sender: where every leaf's rows go in the packed buffer (owner-major order, a header of hrows rows in
// front of every owner's segment) and where every segment starts.  One item per thread (item i =
// (owner p, j): leaf p + j * parts), ceil(items / 1024) workgroups: a workgroup scans its 1024 sizes,
// publishes their sum and adds up the sums of the workgroups BEFORE it (they were dispatched earlier
// and wait for nobody behind them, so the wait ends whatever is resident) -- one load and one
// look-back deep, where one workgroup walked 64 items per thread three times over (0.19 ms of a
// rank's 4 ms at N = 8).  The header triples are written by the gather kernel (runs_write_header).
constexpr unsigned long long RUNS_PLAN_READY = 1ull << 63;
__global__ __launch_bounds__(1024) void msp_runs_plan_kernel(const uint4 *__restrict__ sz, int parts, int lpp, int hrows,
                                                             uint64_t *__restrict__ dst_off,
                                                             uint64_t *__restrict__ all_rows /* rows of the whole buffer */,
                                                             uint64_t *__restrict__ seg_start /* [parts]: first row of every segment */,
                                                             unsigned long long *__restrict__ sync /* [gridDim.x], zeroed */) {
  __shared__ unsigned long long wtot[16];
  __shared__ unsigned long long before;
  const int tid = threadIdx.x, b = (int)blockIdx.x;
  const int n = parts * lpp;
  const int i = b * 1024 + tid, p = i / lpp, j = i - p * lpp, leaf = p + j * parts;
  uint32_t r = 0u;
  if (i < n && leaf < NLEAF) r = sz[leaf].w;
  uint64_t total;
  const uint64_t excl = block_scan_u64(r, wtot, &total);
  if (tid == 0) __hip_atomic_store(&sync[b], RUNS_PLAN_READY | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 64) {                                   // (at most 65 workgroups: parts * ceil(65536 / parts) <= 65536 + 63 items)
    unsigned long long x = 0ull;
    if (tid < b) {
      do x = __hip_atomic_load(&sync[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); while (!(x & RUNS_PLAN_READY));
      x &= ~RUNS_PLAN_READY;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if (tid == 0) before = x;
  }
  __syncthreads();
  const uint64_t run = before + excl;
  if (i < n) {
    if (j == 0) seg_start[p] = run + (uint64_t)p * hrows;         // (a segment starts with the first item of its owner)
    if (leaf < NLEAF) dst_off[leaf] = run + (uint64_t)(p + 1) * hrows;
  }
  if (b == (int)gridDim.x - 1 && tid == 0) *all_rows = before + total + (uint64_t)parts * hrows;
}
static inline unsigned runs_plan_grid(int parts, int lpp) { return (unsigned)((parts * lpp + 1023) / 1024); }

// gather kernel, one thread of the leaf's workgroup: the leaf's (distinct, truncated, noted) sizes in its owner's header
__device__ __forceinline__ void runs_write_header(uint4 *packed, const uint64_t *seg_start, int parts, uint32_t leaf,
                                                  uint32_t nd, uint32_t nu, uint32_t na) {
  const uint32_t p = leaf % (uint32_t)parts, j = leaf / (uint32_t)parts;
  uint32_t *hdr = reinterpret_cast<uint32_t *>(packed + seg_start[p]);
  hdr[3 * j] = nd; hdr[3 * j + 1] = nu; hdr[3 * j + 2] = na;
  // (parts does not divide 65536: the last entry of the last owners' headers stands for no leaf)
  const uint32_t lpp = ((uint32_t)NLEAF + (uint32_t)parts - 1u) / (uint32_t)parts;
  if (leaf + (uint32_t)parts >= (uint32_t)NLEAF && j + 1u < lpp) { hdr[3 * j + 3] = 0u; hdr[3 * j + 4] = 0u; hdr[3 * j + 5] = 0u; }
}

// (err = 1 when a rank's header does not add up to the rows it sent)
struct RunsRecv { uint64_t rstart[64]; uint64_t rows[64]; };
// owner: every rank's header says how large its leaves' segments are.  Two kernels: (1) one thread per
// local leaf reads the `parts` header triples of that leaf (coalesced across threads) and writes, per
// (rank, leaf), the segment's rows and where it goes INSIDE the leaf's two streams, and per leaf the
// stream sizes; (2) one workgroup turns those compact arrays into offsets with plain scans.  (One
// workgroup used to do all of it with dependent header loads: 0.27 ms at N = 8.)
// NC streams per leaf, CI1 / CI0: which of them take the complete / the truncated runs; RMUL rows per record
template <int NC, int CI1, int CI0, int RMUL>
__global__ __launch_bounds__(256) void msp_runs_layout1_kernel(const uint4 *__restrict__ packed, RunsRecv rr, int parts, int lpp,
                                                                uint32_t *__restrict__ rows /* [parts][lpp] */,
                                                                uint64_t *__restrict__ d1 /* relative */, uint64_t *__restrict__ d0 /* relative */,
                                                                uint32_t *__restrict__ lcap, uint64_t *__restrict__ out) {
  const int ll = blockIdx.x * 256 + threadIdx.x;
  if (ll >= lpp) return;
  uint64_t n1 = 0, n0 = 0;
  uint32_t err = 0;
  // (the complete streams of all ranks first, then the truncated ones: rank order inside both)
  for (int r = 0; r < parts; ++r) {
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(packed + rr.rstart[r]);
    const uint32_t a = hdr[3 * ll], b = hdr[3 * ll + 1], c = hdr[3 * ll + 2];
    rows[(size_t)r * lpp + ll] = (uint32_t)RMUL * (a + b) + (c + NOTES_PER_ROW - 1) / NOTES_PER_ROW;
    if (c && !a) err = 1;                                      // notes without a run they could point at
    d1[(size_t)r * lpp + ll] = n1;
    n1 += a;
  }
  for (int r = 0; r < parts; ++r) {
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(packed + rr.rstart[r]);
    d0[(size_t)r * lpp + ll] = n1 + n0;
    n0 += (uint64_t)hdr[3 * ll + 1] + hdr[3 * ll + 2];         // (a note becomes a record again)
  }
  if (n1 > 0xFFFFFFFFull || n0 > 0xFFFFFFFFull) err = 1;
  lcap[(size_t)NC * ll + CI1] = (uint32_t)n1;
  lcap[(size_t)NC * ll + CI0] = (uint32_t)n0;
  if (err) out[1] = 1;
}

// parts + 1 workgroups: workgroup r < parts scans rank r's segment sizes, the last one the leaves' stream sizes
template <int NC, int CI1, int CI0>
__global__ __launch_bounds__(1024) void msp_runs_layout_kernel(RunsRecv rr, int parts, int lpp, int hrows,
                                                               const uint32_t *__restrict__ rows,
                                                               uint64_t *__restrict__ src, uint64_t *__restrict__ d1, uint64_t *__restrict__ d0,
                                                               uint64_t *__restrict__ lbase, const uint32_t *__restrict__ lcap, uint32_t *__restrict__ cnt2,
                                                               uint64_t *__restrict__ out /* [0]: records in all, [1]: err */) {
  __shared__ unsigned long long wtot[16];
  const int tid = threadIdx.x;
  const int per = (lpp + 1023) / 1024;
  if ((int)blockIdx.x < parts) {
    // (1) rank r's segments: record offsets inside its part of the buffer
    const int r = (int)blockIdx.x;
    uint64_t mine = 0;
    for (int q = 0; q < per; ++q) {
      const int ll = tid * per + q;
      if (ll < lpp) mine += rows[(size_t)r * lpp + ll];
    }
    uint64_t total;
    uint64_t run = block_scan_u64(mine, wtot, &total);
    if (total + (uint64_t)hrows != rr.rows[r]) out[1] = 1;
    for (int q = 0; q < per; ++q) {
      const int ll = tid * per + q;
      if (ll >= lpp) break;
      src[(size_t)r * lpp + ll] = rr.rstart[r] + (uint64_t)hrows + run;
      run += rows[(size_t)r * lpp + ll];
    }
    return;
  }
  // (2) the owner's leaves = local indices: stream (ll, class) = the ranks' parts in rank order, complete stream first
  uint64_t mine = 0;
  for (int q = 0; q < per; ++q) {
    const int ll = tid * per + q;
    if (ll < lpp) mine += (uint64_t)lcap[(size_t)NC * ll + CI0] + lcap[(size_t)NC * ll + CI1];
  }
  uint64_t total;
  uint64_t run = block_scan_u64(mine, wtot, &total);
  for (int q = 0; q < per; ++q) {
    const int ll = tid * per + q;
    if (ll >= lpp) break;
    const uint32_t n1 = lcap[(size_t)NC * ll + CI1], n0 = lcap[(size_t)NC * ll + CI0];
    for (int r = 0; r < parts; ++r) {              // relative -> absolute
      d1[(size_t)r * lpp + ll] += run;
      d0[(size_t)r * lpp + ll] += run;
    }
    lbase[(size_t)NC * ll + CI1] = run; cnt2[(size_t)NC * ll + CI1] = n1;
    lbase[(size_t)NC * ll + CI0] = run + n1; cnt2[(size_t)NC * ll + CI0] = n0;
    run += (uint64_t)n1 + n0;
  }
  if (tid == 0) out[0] = total;
}

// ------------------------------------------------------------------- the PIPELINED exchange (round 5), both key widths
// Sender: the leaves of an owner are cut into `ngroups` ranges of local indices; group g of owner p has its segment at
// rows [(g * parts + p) * seg_cap, ...) of the send buffer: header (row 0 = {rows used, leaves, first local leaf, magic},
// written by msp_runs_group_finish_kernel; then one uint4 {row offset, distinct, truncated, noted} per local leaf)
// followed by the leaves' rows in the order their workgroups CLAIM them (one atomic on the segment's cursor per leaf).
struct RunsSend {
  uint4 *packed;             // this GROUP's segments: owner p's at packed + p * seg_cap
  uint64_t seg_cap;          // rows per segment (header included)
  uint32_t *cursor;          // [parts] rows claimed behind the header (zeroed before the kernel)
  uint32_t leaf0, nleaf;     // the group's leaves [leaf0, leaf0 + nleaf): all owners' local leaves [ll0, ll0 + lcount)
  uint32_t ll0, lcount;
  int parts;
};
// Owner: the leaf kernel reads the N lists of a leaf IN PLACE from the receive buffer (msp.hip: p3_body, msp2.hip: q3_body)
template <bool LISTS> struct P3ListsT {};
template <> struct P3ListsT<true> {
  const uint4 *packed;       // the receive buffer of this group: rank r's segment at row rr.rstart[r], rr.rows[r] rows
  RunsRecv rr;
  int parts;
  uint32_t ll0, lcount;      // the group's local leaves [ll0, ll0 + lcount): workgroup b counts local leaf ll0 + b
};
constexpr uint32_t RUNS2_MAGIC = 0x32535543u;   // "CUS2"

static inline uint32_t runs_ll0(int lpp, int g, int ngroups) { return (uint32_t)(((int64_t)lpp * g) / ngroups); }

// ... and the group's epilogue: row 0 of every segment's header, the rows every segment uses, the job's flags
__global__ __launch_bounds__(64) void msp_runs_group_finish_kernel(RunsSend sg, const uint64_t *__restrict__ stats, uint64_t *__restrict__ used /* [parts + 1] */) {
  const int p = threadIdx.x;
  if (p < sg.parts) {
    const uint64_t u = 1ull + sg.lcount + (uint64_t)sg.cursor[p];
    used[p] = u;
    sg.packed[(uint64_t)p * sg.seg_cap] = make_uint4((uint32_t)min(u, (uint64_t)0xFFFFFFFFull), sg.lcount, sg.ll0, RUNS2_MAGIC);
  }
  if (p == 0) used[sg.parts] = (stats[ST_SPILLED] || stats[ST_ONES] || stats[ST_L1OVF] || stats[ST_L2OVF] || stats[ST_OVFN] ||
                                stats[ST_OVFN1] || stats[ST_CWRAP] || stats[ST_OVERFLOW]) ? 1ull : 0ull;
}


// host side of cfrk_global_export_runs_async for both key widths: per group, zero headers of the leaves that stand for no
// leaf, the fused dedupe + pack kernel (launch(sg)), the epilogue, the sizes to pinned memory, the group's event
template <class Launch>
static int runs_export_async_host(cfrk_ctx *ctx, void *d_packed, uint64_t seg_cap_rows, int parts, int ngroups, Launch launch) {
  const int lpp = (NLEAF + parts - 1) / parts;
  if (ngroups > lpp) return cfrk_fail(ctx, CFRK_ERR_ARG, "more groups than leaves per owner");
  if (seg_cap_rows < (uint64_t)(lpp + ngroups - 1) / ngroups + 2 || seg_cap_rows > 0xFFFFFFF0ull)
    return cfrk_fail(ctx, CFRK_ERR_ARG, "segment capacity out of range (at least a group's header: leaves per owner / groups + 2 rows)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->h_runs) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_runs, (size_t)CFRK_RUNS_MAX_GROUPS * 65 * sizeof(uint64_t), hipHostMallocDefault));
  for (int g = 0; g < ngroups; ++g)
    if (!ctx->runs_ev[g]) HIP_TRY(ctx, hipEventCreate(&ctx->runs_ev[g]));          // (with timing: cfrk_global_runs_group_ms)
  int rc;
  void *p;
  const size_t ncur = (size_t)ngroups * parts;
  if ((rc = cfrk_pool_get(ctx, BUF_RUNS_AUX, ((ncur * sizeof(uint32_t) + 15) & ~(size_t)15) + (size_t)ngroups * 65 * sizeof(uint64_t), &p))) return rc;
  uint32_t *d_cur = (uint32_t *)p;
  uint64_t *d_used = (uint64_t *)((char *)p + ((ncur * sizeof(uint32_t) + 15) & ~(size_t)15));
  HIP_TRY(ctx, hipMemsetAsync(d_cur, 0, ncur * sizeof(uint32_t), ctx->stream));
  for (int g = 0; g < ngroups; ++g) {
    RunsSend sg;
    sg.packed = (uint4 *)d_packed + (uint64_t)g * parts * seg_cap_rows;
    sg.seg_cap = seg_cap_rows;
    sg.cursor = d_cur + (size_t)g * parts;
    sg.ll0 = runs_ll0(lpp, g, ngroups);
    sg.lcount = runs_ll0(lpp, g + 1, ngroups) - sg.ll0;
    sg.leaf0 = sg.ll0 * (uint32_t)parts;
    sg.nleaf = std::min<uint32_t>((sg.ll0 + sg.lcount) * (uint32_t)parts, (uint32_t)NLEAF) - sg.leaf0;
    sg.parts = parts;
    if (sg.nleaf < sg.lcount * (uint32_t)parts) {
      // (parts does not divide 65536: the last local leaf of the last owners stands for no leaf -- its header entry is zero)
      for (uint32_t q = sg.nleaf; q < sg.lcount * (uint32_t)parts; ++q) {
        const uint32_t leaf = sg.leaf0 + q, own = leaf % (uint32_t)parts, ll = leaf / (uint32_t)parts;
        HIP_TRY(ctx, hipMemsetAsync(sg.packed + (uint64_t)own * seg_cap_rows + 1u + (ll - sg.ll0), 0, sizeof(uint4), ctx->stream));
      }
    }
    if (sg.nleaf) {
      launch(sg);
      HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(msp_runs_group_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, sg, (const uint64_t *)ctx->g_stats, d_used + (size_t)g * 65);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_runs + (size_t)g * 65, d_used + (size_t)g * 65, (size_t)(parts + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->runs_ev[g], ctx->stream));
  }
  ctx->runs_groups = ngroups; ctx->runs_parts = parts; ctx->runs_seg_cap = seg_cap_rows;
  return CFRK_OK;
}
