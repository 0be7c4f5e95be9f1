// cfrk_cli.cpp -- the `cfrk` command: same positional interface as the reference binary
// (/root/reference/src/main.cu:232-309), counting done by libcfrk_hip.so.
//
//   cfrk dataset.fasta out.cfrk k [threads] [chunkSize] [options]
//   cfrk --batch N dataset_prefix out_prefix k [threads] [chunkSize] [options]
//
// Default = byte-exact reference behaviour: compat ingest (src/fastaIO.h quirks), chunks of
// chunkSize reads (default 8192, src/main.cu:235), ComputeFreqNew semantics, and ONLY the last
// partial chunk in the file -- the reference's second PrintFreq re-opens the file with "w"
// (src/main.cu:303-305), so a read count that is a multiple of chunkSize gives an empty file.
// chunkSize is narrowed to unsigned short where the reference narrows it (SelectChunkRemain's
// `ushort chunkSize, ushort it`, src/main.cu:110): the chunk that reaches the file starts at read
// (chunkSize mod 65536) * (nChunk mod 65536) and holds gnS - nChunk * chunkSize reads.
// Options:
//   --all-chunks     write every chunk (what the reference evidently meant to do)
//   --native         guarded per-read counting (src/kmer_kernel.cu:52-70) + clean FASTA parsing
//   --global         one sparse table over all reads, k up to 64: "key:count" lines (k <= 32),
//                    "hi:lo:count" lines (k > 32: key = hi * 2^64 + lo), ascending by key
//   --binary         (global) the CFRKGLB1 binary form instead of text (cfrk_host.h: 32-byte header,
//                    then 12-byte (k <= 32) or 20-byte records, ascending)
//   --canonical      (global) count min(kmer, reverse complement)
//   --device N       first GPU ordinal (the reference picks the GPU with most memory, src/main.cu:83-108)
//   --gpus N         chunks (per-read modes) or files (--batch) are dealt round-robin to N devices,
//                    one cfrk_ctx pair per device; replaces the reference's pthread fan-out, whose
//                    threads all use the same device (src/main.cu:208-230,277-295).  With --global
//                    (16 <= k <= 64) the reads are range-partitioned over the N devices, every device
//                    partitions and deduplicates its shard, and the owner of a leaf counts it (the
//                    runs exchange of cfrk_abi.h, staged through host memory here)
//   --timing         one line on stderr, `cfrk-timing {json}`: seconds spent parsing the FASTA, in the counting calls
//                    (H2D copy + kernels), in the export (device sort + D2H) and in formatting + writing the output --
//                    the wall-clock breakdown the reference has as commented-out printf()s (src/main.cu:259-268,303-305)
//   --parse-threads N  host threads of the FASTA parser (default min(hardware threads, 64))
//   --batch N        the Swift/T workflow's loop (swift/cfrk.swf:15-20) in one process: for i < N
//                    count <dataset_prefix>_<i>.fasta into <out_prefix>_<i>.cfrk
// Chunk pipeline: every device runs two contexts (two HIP streams), each on a host thread of its
// own, so the H2D copy of chunk c+1 overlaps the kernel / D2H / text formatting of chunk c; the
// main thread writes the formatted chunks in order.
// The threads argument is the number of host threads that format the .cfrk text (the reference
// uses it for host memcpy only, src/main.cu:137,186); like the reference, with 5 positional
// arguments the 5th is chunkSize and the 4th is not parsed (src/main.cu:246-249).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cfrk_abi.h"
#include "cfrk_host.h"

namespace {

struct Options {
  int k = 0, threads = 12;
  long chunk_size = 8192;
  bool all_chunks = false, native = false, global = false, canonical = false, same_device = false, binary = false, timing = false;
  int device = 0, gpus = 1;
};

// --timing: wall-clock seconds by phase (one file; with --batch the last file's)
struct Timing {
  double parse = 0, add_call = 0, finish_wait = 0, export_ = 0, format = 0, write = 0, per_read = 0, total = 0;
  double close = 0, free_batch = 0, open = 0;
  double contexts = 0, begin = 0, wait_parse = 0;   // context creation (beside the parse), cfrk_global_begin, the main thread's wait for the parser
  float count_kernels_ms = 0;
  int64_t fasta_bytes = 0, nN = 0, nS = 0;
  uint64_t entries = 0, out_bytes = 0;
} g_timing;
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Worker {            // one context (= one HIP stream + buffer pool) on one device
  cfrk_ctx *ctx = nullptr;
  int device = 0;
};

int die(cfrk_ctx *ctx, int rc, const char *what) {
  fprintf(stderr, "cfrk: %s: %s (%s)\n", what, cfrk_strerror(rc), ctx ? cfrk_last_error(ctx) : "");
  return 2;
}

struct ChunkRange { int64_t first, count; };

// which reads reach the file, in which chunks (src/main.cu:270-305)
std::vector<ChunkRange> plan_chunks(const Options &o, int64_t nS) {
  std::vector<ChunkRange> v;
  const int64_t C = o.chunk_size;
  const int64_t n_full = nS / C;                     // nChunk = floor(gnS/chunkSize), src/main.cu:270
  if (o.all_chunks) {
    for (int64_t c = 0; c <= n_full; ++c) {
      const int64_t first = c * C, count = (c < n_full) ? C : nS - first;
      if (count > 0) v.push_back({first, count});
    }
    return v;
  }
  if (o.native) {                                    // the remainder chunk, where it really is
    if (nS - n_full * C > 0) v.push_back({n_full * C, nS - n_full * C});
    return v;
  }
  // compat: only SelectChunkRemain's chunk is written, located with the narrowed arguments
  const int64_t count = nS - n_full * C;             // chunkRemain, src/main.cu:297
  const int64_t first = (int64_t)(uint16_t)C * (int64_t)(uint16_t)n_full;
  if (count > 0 && first + count <= nS) v.push_back({first, count});
  return v;
}

// per-read modes: the chunks of one batch through `workers` (any number of devices), text in order
int run_per_read(const Options &o, const cfrk_batch &batch, std::vector<Worker> &workers, FILE *out) {
  const std::vector<ChunkRange> chunks = plan_chunks(o, batch.nS);
  const int flags = o.native ? 0 : CFRK_COMPAT;
  const size_t fourk = (size_t)1 << (2 * (o.k > 0 && o.k < 16 ? o.k : 1));
  const size_t n = chunks.size();
  std::vector<std::string> text(n);
  std::vector<char> ready(n, 0);
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<size_t> next{0};
  size_t written = 0;                                // guarded by mu
  int failed = 0;                                    // guarded by mu
  const size_t window = 2 * workers.size() + 2;      // formatted chunks that may wait for the writer
  const int fmt_threads = std::max(1, o.threads / (int)workers.size());

  auto work = [&](Worker &w) {
    std::vector<int64_t> start;
    std::vector<int32_t> freq;
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= n) return;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || c < written + window; });
        if (failed) return;
      }
      const int8_t *data; const int32_t *length; int64_t nN;
      start.resize((size_t)chunks[c].count);
      cfrk_host_chunk(&batch, chunks[c].first, chunks[c].count, &data, start.data(), &length, &nN);
      freq.resize((size_t)chunks[c].count * fourk);
      const int rc = cfrk_per_read_dense(w.ctx, data, start.data(), length, nN, chunks[c].count, o.k, flags, freq.data());
      std::string t;
      if (!rc) {
        t.resize(cfrk_host_format_dense_mt(freq.data(), chunks[c].count, o.k, nullptr, 0, fmt_threads));
        cfrk_host_format_dense_mt(freq.data(), chunks[c].count, o.k, &t[0], t.size(), fmt_threads);
      }
      std::lock_guard<std::mutex> lk(mu);
      if (rc && !failed) failed = die(w.ctx, rc, "cfrk_per_read_dense");
      text[c].swap(t);
      ready[c] = 1;
      cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < workers.size() && i < n; ++i) th.emplace_back(work, std::ref(workers[i]));
  std::thread first_worker;
  if (n) first_worker = std::thread(work, std::ref(workers[0]));
  for (size_t c = 0; c < n; ++c) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return failed || ready[c]; });
    if (failed) break;
    std::string t;
    t.swap(text[c]);
    lk.unlock();
    if (c) fputc('\n', out);
    fwrite(t.data(), 1, t.size(), out);
    lk.lock();
    written = c + 1;
    cv.notify_all();
  }
  if (first_worker.joinable()) first_worker.join();
  for (auto &t : th) t.join();
  return failed;
}

// the global result (ascending keys) as sparse text or in the binary form
void write_global(const Options &o, const uint64_t *lo, const uint64_t *hi, const uint32_t *cnt, uint64_t n, FILE *out) {
  const uint64_t *hi2 = (o.k > 32) ? hi : nullptr;
  std::string buf;
  const double t0 = now_s();
  if (o.binary) {
    buf.resize(cfrk_host_write_binary(o.k, o.canonical ? CFRK_BIN_CANONICAL : 0, lo, hi2, cnt, n, nullptr, 0));
    cfrk_host_write_binary(o.k, o.canonical ? CFRK_BIN_CANONICAL : 0, lo, hi2, cnt, n, &buf[0], buf.size());
  } else {
    buf.resize(cfrk_host_format_sparse_mt(lo, hi2, cnt, n, nullptr, 0, o.threads));
    cfrk_host_format_sparse_mt(lo, hi2, cnt, n, &buf[0], buf.size(), o.threads);
  }
  const double t1 = now_s();
  fwrite(buf.data(), 1, buf.size(), out);
  fflush(out);
  g_timing.format = t1 - t0; g_timing.write = now_s() - t1; g_timing.entries = n; g_timing.out_bytes = buf.size();
}

// early_free: the batch is no longer needed once it has been counted -- returning 1.6 GB of pages to the kernel takes
// ~0.17 s, which then runs beside the export, the formatting and the write instead of behind them
int run_global(const Options &o, const cfrk_batch &batch, Worker &w, FILE *out, std::thread *early_free = nullptr, cfrk_batch *owned = nullptr) {
  int rc;
  cfrk_ctx *ctx = w.ctx;
  // The capacity hint sizes the result list and the spill table (12 B per slot at load <= 0.5).  Distinct k-mers cannot
  // exceed the window starts (nN), but a table for nN keys is 25 - 50 GB for a 1.5 GB batch and allocating it took
  // 5 of the 6 seconds of a k = 31 run (round 5, profiles/r05/end_to_end.txt): the first attempt announces nN / 16 keys
  // (sequencing depth is rarely below that) and an overflowing result (CFRK_ERR_TABLE_FULL) is counted again with
  // eight times the room, up to nN (capped at 2^31 keys).
  const uint64_t hint_max = std::min<uint64_t>(std::max<uint64_t>((uint64_t)(batch.nN > 0 ? batch.nN : 1), 1ull << 20), 1ull << 31);
  uint64_t hint = std::min<uint64_t>(std::max<uint64_t>(hint_max / 16, 1ull << 20), hint_max);
  uint64_t n = 0;
  double t0 = 0, t1 = 0;
  for (;;) {
    const double tb = now_s();
    if ((rc = cfrk_global_begin(ctx, o.k, o.canonical ? CFRK_CANONICAL : 0, hint))) return die(ctx, rc, "cfrk_global_begin");
    t0 = now_s();
    g_timing.begin = t0 - tb;
    if ((rc = cfrk_global_add(ctx, batch.data, batch.start, batch.length, batch.nN, batch.nS))) return die(ctx, rc, "cfrk_global_add");
    t1 = now_s();
    rc = cfrk_global_finish(ctx, &n);
    if (rc == CFRK_ERR_TABLE_FULL && hint < hint_max) { hint = std::min<uint64_t>(hint * 8, hint_max); continue; }
    break;
  }
  if (early_free && owned) *early_free = std::thread([owned] { cfrk_host_free_batch(owned); });
  // (counts are 32-bit and saturate: the result is complete, the user is told)
  if (rc == CFRK_ERR_COUNT_OVERFLOW) fprintf(stderr, "cfrk: warning: %s\n", cfrk_last_error(ctx));
  else if (rc) return die(ctx, rc, "cfrk_global_finish");
  const double t2 = now_s();
  cfrk_global_last_add_ms(ctx, &g_timing.count_kernels_ms);
  std::vector<uint64_t> keys(n), hi(n);
  std::vector<uint32_t> cnt(n);
  rc = cfrk_global_export(ctx, keys.data(), hi.data(), cnt.data(), n, &n);
  if (rc && rc != CFRK_ERR_COUNT_OVERFLOW) return die(ctx, rc, "cfrk_global_export");
  g_timing.add_call = t1 - t0; g_timing.finish_wait = t2 - t1; g_timing.export_ = now_s() - t2;
  write_global(o, keys.data(), hi.data(), cnt.data(), n, out);
  return 0;
}

// --global over several devices: shard s of the reads goes to device s (first context of the pair);
// device s then owns the leaves s, s + N, ...: it receives its segment of every shard's packed runs
// and counts them on its second context.  The owners' key sets are disjoint; their sorted lists are
// merged into one ascending output.
int run_global_multi(const Options &o, const cfrk_batch &batch, std::vector<std::vector<Worker>> &per_dev, FILE *out) {
  const int N = (int)per_dev.size();
  const int flags = o.canonical ? CFRK_CANONICAL : 0;
  // (as in run_global: nN / 16 keys announced first; an owner whose result overflows makes the job count on one device,
  //  where the hint grows)
  uint64_t hint = std::min<uint64_t>(std::max<uint64_t>((uint64_t)(batch.nN > 0 ? batch.nN : 1) / 16, 1ull << 20), 1ull << 31);
  // shard s: packed rows stay ON ITS DEVICE (16 bytes per row), rows[s][o] = rows of owner o's segment
  std::vector<void *> d_packed((size_t)N, nullptr);
  std::vector<std::vector<uint64_t>> rows((size_t)N, std::vector<uint64_t>((size_t)N, 0));
  std::vector<int> status((size_t)N, 0);
  std::vector<char> refused((size_t)N, 0);           // the shard cannot export runs: count on one device instead
  auto free_packed = [&] {
    for (int sh = 0; sh < N; ++sh)
      if (d_packed[(size_t)sh]) { cfrk_device_free(per_dev[(size_t)sh][0].ctx, d_packed[(size_t)sh]); d_packed[(size_t)sh] = nullptr; }
  };
  {
    std::vector<std::thread> th;
    for (int sh = 0; sh < N; ++sh)
      th.emplace_back([&, sh] {
        cfrk_ctx *ctx = per_dev[(size_t)sh][0].ctx;
        const int64_t r0 = batch.nS * sh / N, r1 = batch.nS * (sh + 1) / N;
        const int64_t b0 = (r0 < batch.nS) ? batch.start[r0] : batch.nN, b1 = (r1 < batch.nS) ? batch.start[r1] : batch.nN;
        int rc;
        if ((rc = cfrk_global_begin(ctx, o.k, flags | CFRK_RUNS_ONLY, hint))) { status[(size_t)sh] = die(ctx, rc, "cfrk_global_begin"); return; }
        if (b1 <= b0) return;
        if ((rc = cfrk_global_add(ctx, batch.data + b0, nullptr, nullptr, b1 - b0, 0))) {
          if (rc == CFRK_ERR_RUNS_REFUSED) { refused[(size_t)sh] = 1; return; }     // (the shard needs several passes)
          status[(size_t)sh] = die(ctx, rc, "cfrk_global_add");
          return;
        }
        // distinct runs never exceed the shard's super-k-mers (about one per 8 bases), plus the headers;
        // a buffer that is too small is retried at four times the size
        // (k > 32: a record is two rows)
        uint64_t cap = (uint64_t)(b1 - b0) / 4 * (o.k > 32 ? 2 : 1) + (uint64_t)N * 70000 + 4096;
        for (int attempt = 0; attempt < 3; ++attempt) {
          void *d = nullptr;
          if ((rc = cfrk_device_alloc(ctx, cap * 16, &d))) { refused[(size_t)sh] = 1; return; }
          rc = cfrk_global_export_runs_device(ctx, d, cap, N, rows[(size_t)sh].data());
          if (!rc) { d_packed[(size_t)sh] = d; return; }
          cfrk_device_free(ctx, d);
          if (rc == CFRK_ERR_SMALL_BUF) { cap *= 4; continue; }
          // (CFRK_ERR_STATE: something of this shard spilled into the HBM table -- its runs are not
          //  all in the leaf streams; the single-device path counts such input as well)
          if (rc == CFRK_ERR_STATE || rc == CFRK_ERR_NOMEM) { refused[(size_t)sh] = 1; return; }
          status[(size_t)sh] = die(ctx, rc, "cfrk_global_export_runs_device");
          return;
        }
        refused[(size_t)sh] = 1;
      });
    for (auto &t : th) t.join();
    for (int r : status) if (r) { free_packed(); return r; }
    for (char r : refused)
      if (r) {
        free_packed();
        fprintf(stderr, "cfrk: a shard could not export its runs; counting on one device\n");
        return run_global(o, batch, per_dev[0][0], out);
      }
  }
  std::vector<std::vector<uint64_t>> keys((size_t)N), his((size_t)N);
  std::vector<std::vector<uint32_t>> cnts((size_t)N);
  {
    // owner ow gathers its segment of every shard DEVICE TO DEVICE (xGMI peer-to-peer between the
    // devices of the node; replaces the host staging of round 2), then expands and counts its leaves
    std::vector<std::thread> th;
    for (int ow = 0; ow < N; ++ow)
      th.emplace_back([&, ow] {
        cfrk_ctx *ctx = per_dev[(size_t)ow][1].ctx;
        std::vector<uint64_t> recv((size_t)N);
        uint64_t total = 0;
        for (int sh = 0; sh < N; ++sh) { recv[(size_t)sh] = rows[(size_t)sh][(size_t)ow]; total += recv[(size_t)sh]; }
        int rc;
        if ((rc = cfrk_global_begin(ctx, o.k, flags, hint / (uint64_t)N + 1024))) { status[(size_t)ow] = die(ctx, rc, "cfrk_global_begin"); return; }
        if (total) {
          void *d = nullptr;
          if ((rc = cfrk_device_alloc(ctx, total * 16, &d))) { status[(size_t)ow] = die(ctx, rc, "cfrk_device_alloc"); return; }
          uint64_t at = 0;
          for (int sh = 0; sh < N && !rc; ++sh) {
            uint64_t off = 0;
            for (int q = 0; q < ow; ++q) off += rows[(size_t)sh][(size_t)q];
            if (recv[(size_t)sh])
              rc = cfrk_memcpy_peer(ctx, (char *)d + at * 16, per_dev[(size_t)sh][0].ctx, (const char *)d_packed[(size_t)sh] + off * 16, recv[(size_t)sh] * 16);
            at += recv[(size_t)sh];
          }
          if (!rc) rc = cfrk_global_merge_runs_device(ctx, d, recv.data(), N);
          if (!rc) rc = cfrk_ctx_sync(ctx);
          cfrk_device_free(ctx, d);
          if (rc) { status[(size_t)ow] = die(ctx, rc, "cfrk_global_merge_runs_device"); return; }
        }
        uint64_t n = 0;
        rc = cfrk_global_finish(ctx, &n);
        if (rc == CFRK_ERR_COUNT_OVERFLOW) fprintf(stderr, "cfrk: warning: %s\n", cfrk_last_error(ctx));
        else if (rc == CFRK_ERR_TABLE_FULL) { refused[(size_t)ow] = 1; return; }       // (more distinct k-mers than announced)
        else if (rc) { status[(size_t)ow] = die(ctx, rc, "cfrk_global_finish"); return; }
        keys[(size_t)ow].resize(n); cnts[(size_t)ow].resize(n); his[(size_t)ow].resize(n);
        rc = cfrk_global_export(ctx, keys[(size_t)ow].data(), his[(size_t)ow].data(), cnts[(size_t)ow].data(), n, &n);
        if (rc && rc != CFRK_ERR_COUNT_OVERFLOW) status[(size_t)ow] = die(ctx, rc, "cfrk_global_export");
      });
    for (auto &t : th) t.join();
    free_packed();
    for (int r : status) if (r) return r;
    for (char r : refused)
      if (r) {
        fprintf(stderr, "cfrk: more distinct k-mers than announced; counting on one device\n");
        return run_global(o, batch, per_dev[0][0], out);
      }
  }
  // N ascending lists with disjoint keys -> one ascending list
  size_t total = 0;
  for (auto &kk : keys) total += kk.size();
  std::vector<uint64_t> mk(total), mh(total);
  std::vector<uint32_t> mc(total);
  std::vector<size_t> at((size_t)N, 0);
  // (the high words are zero for k <= 32)
  auto less = [&](int a, int b) {
    const size_t ia = at[(size_t)a], ib = at[(size_t)b];
    const uint64_t ha = his[(size_t)a][ia], hb = his[(size_t)b][ib];
    return ha != hb ? ha < hb : keys[(size_t)a][ia] < keys[(size_t)b][ib];
  };
  for (size_t i = 0; i < total; ++i) {
    int best = -1;
    for (int q = 0; q < N; ++q)
      if (at[(size_t)q] < keys[(size_t)q].size() && (best < 0 || less(q, best))) best = q;
    const size_t ib = at[(size_t)best]++;
    mk[i] = keys[(size_t)best][ib]; mh[i] = his[(size_t)best][ib]; mc[i] = cnts[(size_t)best][ib];
  }
  write_global(o, mk.data(), mh.data(), mc.data(), total, out);
  return 0;
}

// one FASTA file -> one .cfrk file on the given workers
// a FASTA file being parsed on a thread of its own while the caller creates the device contexts (single-file mode: HIP
// start-up and two contexts are ~0.15 s, the parse of a 1.6 GB file ~0.28 s -- they need nothing from each other)
double g_contexts_s = 0, g_wait_parse = 0;
struct Parsed {
  cfrk_batch batch;
  int rc = 0;
  double t0 = 0, seconds = 0;
  std::thread th;
  void start(const Options &o, const char *in) {
    t0 = now_s();
    const unsigned flags = (o.native || o.global) ? 0 : CFRK_INGEST_COMPAT;
    th = std::thread([this, in, flags] { rc = cfrk_host_read_fasta(in, flags, &batch); seconds = now_s() - t0; });
  }
};

int run_file(const Options &o, const char *in, const char *outp, std::vector<Worker> &workers,
             std::vector<std::vector<Worker>> *per_dev = nullptr, Parsed *pre = nullptr) {
  cfrk_batch batch;
  double t0 = now_s();
  int rc;
  if (pre) {
    const double w0 = now_s();
    pre->th.join();
    g_wait_parse = now_s() - w0;
    rc = pre->rc; batch = pre->batch; t0 = pre->t0;
  } else {
    rc = cfrk_host_read_fasta(in, (o.native || o.global) ? 0 : CFRK_INGEST_COMPAT, &batch);
  }
  if (rc) { fprintf(stderr, "cfrk: cannot read %s (error %d)\n", in, rc); return 1; }
  const double t1 = pre ? t0 + pre->seconds : now_s();
  g_timing = Timing();
  g_timing.contexts = g_contexts_s; g_timing.wait_parse = g_wait_parse;
  g_timing.parse = t1 - t0; g_timing.nN = batch.nN; g_timing.nS = batch.nS;
  { FILE *f = fopen(in, "rb"); if (f) { fseek(f, 0, SEEK_END); g_timing.fasta_bytes = (int64_t)ftell(f); fclose(f); } }
  std::thread freer;                                  // (global mode: frees the batch beside the export)
  const double to0 = now_s();
  FILE *out = fopen(outp, "wb");                      // PrintFreq opens with "w" even when empty
  const double t_open = now_s() - to0;
  if (!out) { fprintf(stderr, "cfrk: cannot write %s\n", outp); cfrk_host_free_batch(&batch); return 1; }
  if (o.global && per_dev && per_dev->size() > 1 && o.k >= 16 && o.k <= 64 && batch.nS >= (int64_t)per_dev->size()) rc = run_global_multi(o, batch, *per_dev, out);
  else if (o.global) rc = run_global(o, batch, workers[0], out, &freer, &batch);
  else { const double p0 = now_s(); rc = run_per_read(o, batch, workers, out); g_timing.per_read = now_s() - p0; }
  if (!o.global) { fflush(out); g_timing.out_bytes = (uint64_t)ftell(out); }
  const double tf0 = now_s();
  fclose(out);
  const double tf1 = now_s();
  if (freer.joinable()) freer.join();
  else cfrk_host_free_batch(&batch);
  g_timing.total = now_s() - t0;
  g_timing.close = tf1 - tf0; g_timing.free_batch = now_s() - tf1; g_timing.open = t_open;
  if (o.timing)
    fprintf(stderr, "cfrk-timing {\"fasta_bytes\": %lld, \"reads\": %lld, \"code_bytes\": %lld, \"parse_s\": %.4f, \"add_call_s\": %.4f, "
            "\"finish_wait_s\": %.4f, \"count_kernels_ms\": %.3f, \"export_s\": %.4f, \"format_s\": %.4f, \"write_s\": %.4f, "
            "\"per_read_pipeline_s\": %.4f, \"entries\": %llu, \"out_bytes\": %llu, \"contexts_s\": %.4f, \"wait_for_parser_s\": %.4f, "
            "\"begin_s\": %.4f, \"open_out_s\": %.4f, \"close_out_s\": %.4f, \"free_batch_s\": %.4f, \"wall_s\": %.4f}\n",
            (long long)g_timing.fasta_bytes, (long long)g_timing.nS, (long long)g_timing.nN, g_timing.parse, g_timing.add_call,
            g_timing.finish_wait, (double)g_timing.count_kernels_ms, g_timing.export_, g_timing.format, g_timing.write,
            g_timing.per_read, (unsigned long long)g_timing.entries, (unsigned long long)g_timing.out_bytes, g_timing.contexts,
            g_timing.wait_parse, g_timing.begin, g_timing.open, g_timing.close, g_timing.free_batch, g_timing.total);
  return rc;
}

}  // namespace

int main(int argc, char **argv) {
  std::vector<const char *> pos;
  Options o;
  int batch_n = -1;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--all-chunks")) o.all_chunks = true;
    else if (!strcmp(argv[i], "--native")) o.native = true;
    else if (!strcmp(argv[i], "--global")) o.global = true;
    else if (!strcmp(argv[i], "--canonical")) o.canonical = true;
    else if (!strcmp(argv[i], "--binary")) o.binary = true;
    else if (!strcmp(argv[i], "--timing")) o.timing = true;
    else if (!strcmp(argv[i], "--parse-threads") && i + 1 < argc) cfrk_host_set_parse_threads(atoi(argv[++i]));
    else if (!strcmp(argv[i], "--same-device")) o.same_device = true;   // rehearsal: every "device" is --device
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) o.device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) o.gpus = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch_n = atoi(argv[++i]);
    else pos.push_back(argv[i]);
  }
  if (pos.size() < 3) {
    // src/main.cu:239-243
    printf("Usage: ./cfrk [dataset.fasta] [file_out.cfrk] [k] <number of threads: Default 12> <chunkSize: Default 8192>");
    return 1;
  }
  o.k = atoi(pos[2]);
  if (pos.size() >= 4) o.threads = atoi(pos[3]);
  if (o.threads < 1) o.threads = 1;
  { const unsigned hw = std::thread::hardware_concurrency(); if (hw && (unsigned)o.threads > hw) o.threads = (int)hw; }
  if (pos.size() == 5) o.chunk_size = atol(pos[4]);     // argc == 6 in the reference
  if (o.chunk_size <= 0) { fprintf(stderr, "cfrk: chunkSize must be positive\n"); return 1; }
  if (o.gpus < 1) { fprintf(stderr, "cfrk: --gpus must be positive\n"); return 1; }
  if (batch_n == 0 || batch_n < -1) { fprintf(stderr, "cfrk: --batch needs a positive file count\n"); return 1; }

  // single-file mode: the parse starts now, beside the creation of the contexts
  Parsed pre;
  if (batch_n < 0) pre.start(o, pos[0]);
  struct Joiner { Parsed &p; ~Joiner() { if (p.th.joinable()) { p.th.join(); if (!p.rc) cfrk_host_free_batch(&p.batch); } } } joiner{pre};   // (early returns)
  const double tc0 = now_s();                          // (runtime start-up + the contexts)
  int ndev = 0, rc;
  if ((rc = cfrk_device_count(&ndev))) return die(nullptr, rc, "cfrk_device_count");
  if (!o.same_device && o.device + o.gpus > ndev) {
    fprintf(stderr, "cfrk: --device %d --gpus %d but %d device(s) present\n", o.device, o.gpus, ndev);
    return 1;
  }
  // two contexts (streams) per device: chunk c+1 is copied in while chunk c is counted / copied out / formatted
  std::vector<std::vector<Worker>> per_dev((size_t)o.gpus);
  for (int g = 0; g < o.gpus; ++g)
    for (int s = 0; s < 2; ++s) {
      Worker w;
      w.device = o.same_device ? o.device : o.device + g;
      if ((rc = cfrk_ctx_create(w.device, nullptr, &w.ctx))) return die(nullptr, rc, "cfrk_ctx_create");
      per_dev[(size_t)g].push_back(w);
    }

  g_contexts_s = now_s() - tc0;
  int status = 0;
  if (batch_n < 0) {
    std::vector<Worker> all;
    for (int s = 0; s < 2; ++s)                       // device-major would put both streams of a device first
      for (int g = 0; g < o.gpus; ++g) all.push_back(per_dev[(size_t)g][(size_t)s]);
    status = run_file(o, pos[0], pos[1], all, &per_dev, &pre);
  } else {
    // file i goes to device i % gpus (swift/cfrk.swf:15-20 starts one cfrk process per file)
    std::vector<int> st((size_t)o.gpus, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < o.gpus; ++g)
      th.emplace_back([&, g] {
        for (int i = g; i < batch_n; i += o.gpus) {
          const std::string in = std::string(pos[0]) + "_" + std::to_string(i) + ".fasta";
          const std::string outp = std::string(pos[1]) + "_" + std::to_string(i) + ".cfrk";
          const int r = run_file(o, in.c_str(), outp.c_str(), per_dev[(size_t)g]);
          if (r && !st[(size_t)g]) st[(size_t)g] = r;
        }
      });
    for (auto &t : th) t.join();
    for (int r : st) if (r && !status) status = r;
  }
  for (auto &d : per_dev) for (auto &w : d) cfrk_ctx_destroy(w.ctx);
  return status;
}
