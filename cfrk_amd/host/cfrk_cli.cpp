// cfrk_cli.cpp -- the `cfrk` command: same positional interface as the reference binary
// (/root/reference/src/main.cu:232-309), counting done by libcfrk_hip.so.
//
//   cfrk dataset.fasta out.cfrk k [threads] [chunkSize] [options]
//
// Default = byte-exact reference behaviour: compat ingest (src/fastaIO.h quirks), chunks of
// chunkSize reads (default 8192, src/main.cu:235), ComputeFreqNew semantics, and ONLY the last
// partial chunk in the file -- the reference's second PrintFreq re-opens the file with "w"
// (src/main.cu:303-305), so a read count that is a multiple of chunkSize gives an empty file.
// Options:
//   --all-chunks     write every chunk (what the reference evidently meant to do)
//   --native         guarded per-read counting (src/kmer_kernel.cu:52-70) + clean FASTA parsing
//   --global         one sparse table over all reads ("key:count" lines), k up to 32
//   --canonical      (global) count min(kmer, reverse complement)
//   --device N       GPU ordinal (the reference picks the GPU with most memory, src/main.cu:83-108)
// The threads argument is accepted and ignored (the reference uses it for host memcpy only,
// src/main.cu:137,186); like the reference, with 5 positional arguments the 5th is chunkSize and
// the 4th is not parsed (src/main.cu:246-249).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/cfrk_abi.h"
#include "cfrk_host.h"

static int die(cfrk_ctx *ctx, int rc, const char *what) {
  fprintf(stderr, "cfrk: %s: %s (%s)\n", what, cfrk_strerror(rc), ctx ? cfrk_last_error(ctx) : "");
  return 2;
}

int main(int argc, char **argv) {
  std::vector<const char *> pos;
  bool all_chunks = false, native = false, global = false, canonical = false;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--all-chunks")) all_chunks = true;
    else if (!strcmp(argv[i], "--native")) native = true;
    else if (!strcmp(argv[i], "--global")) global = true;
    else if (!strcmp(argv[i], "--canonical")) canonical = true;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else pos.push_back(argv[i]);
  }
  if (pos.size() < 3) {
    // src/main.cu:239-243
    printf("Usage: ./cfrk [dataset.fasta] [file_out.cfrk] [k] <number of threads: Default 12> <chunkSize: Default 8192>");
    return 1;
  }
  const int k = atoi(pos[2]);
  // <number of threads>: the reference spreads chunks over pthreads; here it is the number of host
  // threads that format the .cfrk text (the GPU side needs none)
  int threads = 12;
  if (pos.size() >= 4) threads = atoi(pos[3]);
  if (threads < 1) threads = 1;
  { const unsigned hw = std::thread::hardware_concurrency(); if (hw && (unsigned)threads > hw) threads = (int)hw; }
  long chunk_size = 8192;
  if (pos.size() == 5) chunk_size = atol(pos[4]);     // argc == 6 in the reference
  if (chunk_size <= 0) { fprintf(stderr, "cfrk: chunkSize must be positive\n"); return 1; }

  cfrk_batch batch;
  int rc = cfrk_host_read_fasta(pos[0], (native || global) ? 0 : CFRK_INGEST_COMPAT, &batch);
  if (rc) { fprintf(stderr, "cfrk: cannot read %s (error %d)\n", pos[0], rc); return 1; }

  cfrk_ctx *ctx = nullptr;
  if ((rc = cfrk_ctx_create(device, nullptr, &ctx))) return die(nullptr, rc, "cfrk_ctx_create");

  FILE *out = fopen(pos[1], "wb");                    // PrintFreq opens with "w" even when empty
  if (!out) { fprintf(stderr, "cfrk: cannot write %s\n", pos[1]); return 1; }

  if (global) {
    // distinct k-mers cannot exceed the number of window starts; the hint sizes the result list
    // and the spill table (12 B per slot at load 0.5), so it is capped at 2^31 keys
    uint64_t hint = (uint64_t)(batch.nN > 0 ? batch.nN : 1);
    if (hint < (1ull << 20)) hint = 1ull << 20;
    if (hint > (1ull << 31)) hint = 1ull << 31;
    if ((rc = cfrk_global_begin(ctx, k, canonical ? CFRK_CANONICAL : 0, hint))) return die(ctx, rc, "cfrk_global_begin");
    if ((rc = cfrk_global_add(ctx, batch.data, batch.start, batch.length, batch.nN, batch.nS))) return die(ctx, rc, "cfrk_global_add");
    uint64_t n = 0;
    if ((rc = cfrk_global_finish(ctx, &n))) return die(ctx, rc, "cfrk_global_finish");
    std::vector<uint64_t> keys(n), hi(n);
    std::vector<uint32_t> cnt(n);
    if ((rc = cfrk_global_export(ctx, keys.data(), hi.data(), cnt.data(), n, &n))) return die(ctx, rc, "cfrk_global_export");
    if (k > 32) { fprintf(stderr, "cfrk: --global text output supports k <= 32\n"); return 1; }
    std::string buf(cfrk_host_format_sparse(keys.data(), cnt.data(), n, nullptr, 0), '\0');
    cfrk_host_format_sparse(keys.data(), cnt.data(), n, &buf[0], buf.size());
    fwrite(buf.data(), 1, buf.size(), out);
  } else {
    const int64_t n_full = batch.nS / chunk_size;     // nChunk = floor(gnS/chunkSize), src/main.cu:270
    const int64_t first_written = all_chunks ? 0 : n_full;
    const int flags = native ? 0 : CFRK_COMPAT;
    const size_t fourk = (size_t)1 << (2 * (k > 0 && k < 16 ? k : 1));
    std::vector<int64_t> start((size_t)chunk_size);
    std::vector<int32_t> freq;
    std::string text;
    bool first_row = true;
    for (int64_t c = first_written; c <= n_full; ++c) {
      const int64_t first = c * chunk_size;
      const int64_t count = (c < n_full) ? chunk_size : batch.nS - first;   // remainder chunk last
      if (count == 0) break;
      const int8_t *data; const int32_t *length; int64_t nN;
      cfrk_host_chunk(&batch, first, count, &data, start.data(), &length, &nN);
      freq.resize((size_t)count * fourk);
      if ((rc = cfrk_per_read_dense(ctx, data, start.data(), length, nN, count, k, flags, freq.data())))
        return die(ctx, rc, "cfrk_per_read_dense");
      text.resize(cfrk_host_format_dense_mt(freq.data(), count, k, nullptr, 0, threads));
      cfrk_host_format_dense_mt(freq.data(), count, k, &text[0], text.size(), threads);
      if (!first_row) fputc('\n', out);
      fwrite(text.data(), 1, text.size(), out);
      first_row = false;
    }
  }
  fclose(out);
  cfrk_ctx_destroy(ctx);
  cfrk_host_free_batch(&batch);
  return 0;
}
