// cfrk_host.cpp -- FASTA ingest, chunk views and .cfrk formatting (see cfrk_host.h).
#include "cfrk_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

namespace {

struct CodeTable {
  int8_t t[256];
  CodeTable() {
    memset(t, -1, sizeof t);                       // default: -1 (src/fastaIO.h:137-138)
    t['a'] = t['A'] = 0; t['c'] = t['C'] = 1;      // src/fastaIO.h:123-136
    t['g'] = t['G'] = 2; t['t'] = t['T'] = 3;
  }
};
const CodeTable kCodes;

// append decimal digits of v, return new end
inline char *put_u64(char *p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}
inline char *put_i32(char *p, int32_t v) {          // "%d"
  if (v < 0) { *p++ = '-'; return put_u64(p, (uint64_t)(-(int64_t)v)); }
  return put_u64(p, (uint64_t)v);
}
inline size_t len_u64(uint64_t v) { size_t n = 1; while (v >= 10) { v /= 10; ++n; } return n; }

}  // namespace

extern "C" {

void cfrk_host_free_batch(cfrk_batch *b) {
  if (!b) return;
  free(b->data); free(b->start); free(b->length);
  memset(b, 0, sizeof *b);
}

int cfrk_host_parse_fasta(const char *buf, size_t len, int flags, cfrk_batch *out) {
  if (!out || (!buf && len)) return -4;
  memset(out, 0, sizeof *out);
  const bool compat = (flags & CFRK_INGEST_COMPAT) != 0;
  // pass 1: record extents (sequence bytes as the reference would strcat them)
  struct Rec { size_t first_line; size_t seq_chars; };
  std::vector<Rec> recs;
  std::vector<std::pair<size_t, size_t>> lines;     // (begin, end incl. newline) of sequence lines
  std::vector<size_t> line_rec;
  size_t pos = 0;
  while (pos < len) {
    const char *nl = (const char *)memchr(buf + pos, '\n', len - pos);
    size_t end = nl ? (size_t)(nl - buf) + 1 : len;
    if (buf[pos] == '>') {
      recs.push_back(Rec{lines.size(), 0});
    } else {
      if (recs.empty()) return -2;
      lines.push_back({pos, end});
      line_rec.push_back(recs.size() - 1);
    }
    pos = end;
  }
  // per-record code counts
  std::vector<int64_t> rlen(recs.size(), 0);
  for (size_t li = 0; li < lines.size(); ++li) {
    size_t b = lines[li].first, e = lines[li].second;
    if (compat) {
      rlen[line_rec[li]] += (int64_t)(e - b);
    } else {
      while (e > b && (buf[e - 1] == '\n' || buf[e - 1] == '\r')) --e;
      rlen[line_rec[li]] += (int64_t)(e - b);
    }
  }
  int64_t nN = 0;
  for (size_t r = 0; r < recs.size(); ++r) {
    if (compat) {
      if (rlen[r] == 0) return -3;                  // header without a sequence line
      rlen[r] -= 1;                                 // len = strlen(read) - 1
    }
    if (rlen[r] > 0x7FFFFFFF) return -4;
    nN += rlen[r] + 1;
  }
  const int64_t nS = (int64_t)recs.size();
  out->data = (int8_t *)malloc((size_t)(nN > 0 ? nN : 1));
  out->start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nS > 0 ? nS : 1));
  out->length = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nS > 0 ? nS : 1));
  if (!out->data || !out->start || !out->length) { cfrk_host_free_batch(out); return -4; }
  out->nN = nN; out->nS = nS;
  // pass 2: encode (ProcessData, src/fastaIO.h:74-102: codes, then one -1 terminator)
  int64_t w = 0;
  size_t li = 0;
  for (int64_t r = 0; r < nS; ++r) {
    out->start[r] = w;
    out->length[r] = (int32_t)rlen[r];
    int64_t left = rlen[r];
    for (; li < lines.size() && line_rec[li] == (size_t)r; ++li) {
      size_t b = lines[li].first, e = lines[li].second;
      if (!compat) while (e > b && (buf[e - 1] == '\n' || buf[e - 1] == '\r')) --e;
      for (size_t p = b; p < e && left > 0; ++p, --left) out->data[w++] = kCodes.t[(unsigned char)buf[p]];
    }
    out->data[w++] = -1;
  }
  return 0;
}

int cfrk_host_read_fasta(const char *path, int flags, cfrk_batch *out) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;                                // the reference exits (src/fastaIO.h:36)
  std::string buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
  fclose(f);
  return cfrk_host_parse_fasta(buf.data(), buf.size(), flags, out);
}

int cfrk_host_chunk(const cfrk_batch *b, int64_t first, int64_t count, const int8_t **data,
                    int64_t *start_out, const int32_t **length, int64_t *nN) {
  if (!b || first < 0 || count < 0 || first + count > b->nS) return -1;
  if (count == 0) { *data = b->data; *length = b->length; *nN = 0; return 0; }
  const int64_t base = b->start[first];
  int64_t pos = 0;
  for (int64_t i = 0; i < count; ++i) {             // chunk-relative offsets, src/main.cu:191-200
    start_out[i] = pos;
    pos += (int64_t)b->length[first + i] + 1;
  }
  *data = b->data + base;
  *length = b->length + first;
  *nN = pos;
  return 0;
}

// rows [r0, r1) of the dense text; row i > 0 starts with the '\n' that separates it from row i-1
static size_t dense_rows_size(const int32_t *freq, int64_t r0, int64_t r1, int64_t fourk) {
  size_t n = 0;
  for (int64_t i = r0; i < r1; ++i) {
    if (i) ++n;
    for (int64_t b = 0; b < fourk; ++b) {
      int32_t v = freq[i * fourk + b];
      n += len_u64((uint64_t)b) + 2 + (v < 0 ? 1 + len_u64((uint64_t)(-(int64_t)v)) : len_u64((uint64_t)v));
    }
  }
  return n;
}
static char *dense_rows_put(const int32_t *freq, int64_t r0, int64_t r1, int64_t fourk, char *p) {
  for (int64_t i = r0; i < r1; ++i) {
    if (i) *p++ = '\n';
    for (int64_t b = 0; b < fourk; ++b) {
      p = put_u64(p, (uint64_t)b);
      *p++ = ':';
      p = put_i32(p, freq[i * fourk + b]);
      *p++ = ' ';
    }
  }
  return p;
}

size_t cfrk_host_format_dense(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap) {
  return cfrk_host_format_dense_mt(freq, nS, k, buf, cap, 1);
}

size_t cfrk_host_format_dense_mt(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap, int threads) {
  const int64_t fourk = (int64_t)1 << (2 * k);
  (void)cap;
  int T = threads < 1 ? 1 : threads;
  if ((int64_t)T > nS) T = nS > 0 ? (int)nS : 1;
  if (T == 1) {
    if (!buf) return dense_rows_size(freq, 0, nS, fourk);
    return (size_t)(dense_rows_put(freq, 0, nS, fourk, buf) - buf);
  }
  // row ranges per thread: sizes first (the text of a range starts where the previous ends)
  std::vector<size_t> sz((size_t)T);
  std::vector<std::thread> th;
  auto r_of = [&](int t) { return nS * t / T; };
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] { sz[(size_t)t] = dense_rows_size(freq, r_of(t), r_of(t + 1), fourk); });
  for (auto &x : th) x.join();
  size_t total = 0;
  std::vector<size_t> off((size_t)T);
  for (int t = 0; t < T; ++t) { off[(size_t)t] = total; total += sz[(size_t)t]; }
  if (!buf) return total;
  th.clear();
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] { dense_rows_put(freq, r_of(t), r_of(t + 1), fourk, buf + off[(size_t)t]); });
  for (auto &x : th) x.join();
  return total;
}

size_t cfrk_host_format_sparse(const uint64_t *keys, const uint32_t *counts, uint64_t n, char *buf,
                               size_t cap) {
  if (!buf) {
    size_t s = 0;
    for (uint64_t i = 0; i < n; ++i) s += len_u64(keys[i]) + 1 + len_u64(counts[i]) + 1;
    return s;
  }
  char *p = buf;
  (void)cap;
  for (uint64_t i = 0; i < n; ++i) {
    p = put_u64(p, keys[i]); *p++ = ':';
    p = put_u64(p, counts[i]); *p++ = '\n';
  }
  return (size_t)(p - buf);
}

}  // extern "C"
