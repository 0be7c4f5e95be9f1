// cfrk_host.cpp -- FASTA ingest, chunk views and .cfrk formatting (see cfrk_host.h).
#include "cfrk_host.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
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

// n bytes of sequence text -> codes (src/fastaIO.h:121-140: aA cC gG tT -> 0 1 2 3, anything else -1).
// The AVX2 form does 32 bytes per step (round 5: the byte-at-a-time table look-up was half of the parser's time on a
// 1.6 GB file): clear the case bit, ((c >> 1) & 3) is A 0, C 1, T 2, G 3 for the four letters, x ^ (x >> 1) swaps the
// last two into the reference's order, and a byte that is none of the four letters becomes -1.
static void encode_plain(const unsigned char *src, int8_t *dst, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = kCodes.t[src[i]];
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void encode_avx2(const unsigned char *src, int8_t *dst, size_t n) {
  const __m256i up = _mm256_set1_epi8((char)0xDF), three = _mm256_set1_epi8(3), one = _mm256_set1_epi8(1);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
    const __m256i u = _mm256_and_si256(c, up);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    const __m256i x = _mm256_and_si256(_mm256_srli_epi16(u, 1), three);            // (the & 3 drops what the 16-bit shift drags in)
    const __m256i code = _mm256_xor_si256(x, _mm256_and_si256(_mm256_srli_epi16(x, 1), one));
    // ok ? code : -1
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_or_si256(code, _mm256_xor_si256(ok, _mm256_set1_epi8(-1))));
  }
  encode_plain(src + i, dst + i, n - i);
}
#endif
static void encode_bytes(const unsigned char *src, int8_t *dst, size_t n) {
#if defined(__x86_64__)
  static const bool have_avx2 = __builtin_cpu_supports("avx2");
  if (have_avx2) { encode_avx2(src, dst, n); return; }
#endif
  encode_plain(src, dst, n);
}

// (huge-page advice for the batch's arrays was tried and dropped: with MADV_HUGEPAGE the first faults of a fresh process
//  took 0.7 - 1.1 s for a 150 MB array -- compaction at fault time -- against 0.03 s of ordinary first-touch faults)
static void *big_alloc(size_t bytes) {
  if (bytes < ((size_t)8 << 20)) return malloc(bytes ? bytes : 1);
  void *p = nullptr;
  const size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  return posix_memalign(&p, (size_t)2 << 20, rounded) == 0 ? p : nullptr;
}

// threads the parser may use (cfrk_host_set_parse_threads; 0 = min(hardware threads, 64): a 1.63 GB file parses at
// 4.7 / 5.1 / 7.1 GB/s with 8 / 16 / 64 threads on the GPU box's 256-thread host, profiles/r05/end_to_end.txt -- the
// 16-thread cap of rounds 1-4 was binding once the file was mapped instead of copied and the encoding vectorised)
int g_parse_threads = 0;

}  // namespace

extern "C" {

void cfrk_host_set_parse_threads(int n) { g_parse_threads = n > 0 ? n : 0; }

void cfrk_host_free_batch(cfrk_batch *b) {
  if (!b) return;
  free(b->data); free(b->start); free(b->length);
  memset(b, 0, sizeof *b);
}

int cfrk_host_parse_fasta(const char *buf, size_t len, int flags, cfrk_batch *out) {
  if (!out || (!buf && len)) return -4;
  memset(out, 0, sizeof *out);
  const bool compat = (flags & CFRK_INGEST_COMPAT) != 0;
  // pass 1: record extents (sequence bytes as the reference would strcat them).  Large inputs are
  // scanned by several threads, each over a range of whole lines; a sequence line that precedes
  // the first header of its range belongs to the last record of the ranges before it.
  struct Rec { size_t first_line; size_t seq_chars; };
  std::vector<Rec> recs;
  std::vector<std::pair<size_t, size_t>> lines;     // (begin, end incl. newline) of sequence lines
  unsigned nthr = std::thread::hardware_concurrency();
  if (g_parse_threads > 0) { if (nthr == 0 || nthr > (unsigned)g_parse_threads) nthr = (unsigned)g_parse_threads; }
  else if (nthr > 64) nthr = 64;
  if (nthr < 2 || len < ((size_t)8 << 20)) nthr = 1;
  {
    struct Part { std::vector<size_t> rec_first; std::vector<std::pair<size_t, size_t>> lines; };
    std::vector<Part> parts(nthr);
    std::vector<size_t> cut(nthr + 1, len);
    cut[0] = 0;
    for (unsigned t = 1; t < nthr; ++t) {            // cuts at line starts
      size_t c = len / nthr * t;
      if (c < cut[t - 1]) c = cut[t - 1];
      const char *nl = (c < len) ? (const char *)memchr(buf + c, '\n', len - c) : nullptr;
      cut[t] = nl ? (size_t)(nl - buf) + 1 : len;
    }
    auto scan = [&](unsigned t) {
      Part &pt = parts[t];
      size_t pos = cut[t];
      const size_t stop = cut[t + 1];
      while (pos < stop) {
        const char *nl = (const char *)memchr(buf + pos, '\n', stop - pos);
        const size_t end = nl ? (size_t)(nl - buf) + 1 : stop;
        if (buf[pos] == '>') pt.rec_first.push_back(pt.lines.size());
        else pt.lines.push_back({pos, end});
        pos = end;
      }
    };
    if (nthr == 1) {
      scan(0);
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < nthr; ++t) pool.emplace_back(scan, t);
      for (auto &th : pool) th.join();
    }
    size_t nrec = 0, nline = 0;
    for (auto &pt : parts) { nrec += pt.rec_first.size(); nline += pt.lines.size(); }
    recs.reserve(nrec);
    lines.reserve(nline);
    for (auto &pt : parts) {
      // (a sequence line before any header at all: the reference would dereference garbage)
      if (recs.empty() && pt.rec_first.empty() && !pt.lines.empty()) return -2;
      if (recs.empty() && !pt.rec_first.empty() && pt.rec_first[0] != 0) return -2;
      const size_t base = lines.size();
      for (size_t f : pt.rec_first) recs.push_back(Rec{base + f, 0});
      lines.insert(lines.end(), pt.lines.begin(), pt.lines.end());
    }
  }
  // per-record code counts
  std::vector<int64_t> rlen(recs.size(), 0);
  auto count = [&](size_t r0, size_t r1) {
    for (size_t r = r0; r < r1; ++r) {
      const size_t l1 = (r + 1 < recs.size()) ? recs[r + 1].first_line : lines.size();
      int64_t n = 0;
      for (size_t li = recs[r].first_line; li < l1; ++li) {
        size_t b = lines[li].first, e = lines[li].second;
        if (!compat) while (e > b && (buf[e - 1] == '\n' || buf[e - 1] == '\r')) --e;
        n += (int64_t)(e - b);
      }
      rlen[r] = n;
    }
  };
  if (nthr == 1) {
    count(0, recs.size());
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthr; ++t)
      pool.emplace_back(count, recs.size() * t / nthr, recs.size() * (t + 1) / nthr);
    for (auto &th : pool) th.join();
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
  out->data = (int8_t *)big_alloc((size_t)(nN > 0 ? nN : 1));
  out->start = (int64_t *)big_alloc(sizeof(int64_t) * (size_t)(nS > 0 ? nS : 1));
  out->length = (int32_t *)big_alloc(sizeof(int32_t) * (size_t)(nS > 0 ? nS : 1));
  if (!out->data || !out->start || !out->length) { cfrk_host_free_batch(out); return -4; }
  out->nN = nN; out->nS = nS;
  // pass 2: encode (ProcessData, src/fastaIO.h:74-102: codes, then one -1 terminator).  Records
  // are independent once their offsets are known: large batches are encoded by several threads.
  {
    int64_t w = 0;
    for (int64_t r = 0; r < nS; ++r) {
      out->start[r] = w;
      out->length[r] = (int32_t)rlen[r];
      w += rlen[r] + 1;
    }
  }
  auto encode = [&](int64_t r0, int64_t r1) {
    for (int64_t r = r0; r < r1; ++r) {
      int64_t w = out->start[r];
      int64_t left = rlen[r];
      const size_t l1 = (r + 1 < nS) ? recs[(size_t)r + 1].first_line : lines.size();
      for (size_t li = recs[(size_t)r].first_line; li < l1; ++li) {
        size_t b = lines[li].first, e = lines[li].second;
        if (!compat) while (e > b && (buf[e - 1] == '\n' || buf[e - 1] == '\r')) --e;
        const size_t n = (size_t)std::min<int64_t>((int64_t)(e - b), left);
        encode_bytes(reinterpret_cast<const unsigned char *>(buf) + b, out->data + w, n);
        w += (int64_t)n; left -= (int64_t)n;
      }
      out->data[w] = -1;
    }
  };
  if (nthr < 2 || nN < (int64_t)(8 << 20)) {
    encode(0, nS);
  } else {
    // split by bytes, not by records: reads may differ in length by orders of magnitude
    std::vector<std::thread> pool;
    int64_t r0 = 0;
    for (unsigned t = 0; t < nthr; ++t) {
      const int64_t target = nN / nthr * (t + 1);
      int64_t r1 = r0;
      if (t + 1 == nthr) r1 = nS;
      else while (r1 < nS && out->start[r1] < target) ++r1;
      if (r1 > r0) pool.emplace_back(encode, r0, r1);
      r0 = r1;
    }
    for (auto &th : pool) th.join();
  }
  return 0;
}

int cfrk_host_read_fasta(const char *path, int flags, cfrk_batch *out) {
  // a regular file is MAPPED (round 5: no copy out of the page cache, no zero-filled buffer -- reading a 1.6 GB file
  // into a std::string was a third of the parse); anything else (a pipe) is read chunk by chunk below
  {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;                          // the reference exits (src/fastaIO.h:36)
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
      if (m != MAP_FAILED) {
        (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        const int rc = cfrk_host_parse_fasta((const char *)m, (size_t)st.st_size, flags, out);
        munmap(m, (size_t)st.st_size);
        close(fd);
        return rc;
      }
    }
    close(fd);
  }
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  // a regular file is read in one piece into a buffer of its size (no growth copies); anything
  // that cannot be sized (a pipe) is appended chunk by chunk
  std::string buf;
  long size = -1;
  if (fseek(f, 0, SEEK_END) == 0) { size = ftell(f); if (fseek(f, 0, SEEK_SET) != 0) size = -1; }
  if (size > 0) {
    buf.resize((size_t)size);
    size_t got = 0, n;
    while (got < (size_t)size && (n = fread(&buf[got], 1, (size_t)size - got, f)) > 0) got += n;
    buf.resize(got);
  }
  char tmp[1 << 16];
  size_t n;
  while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);   // growing file / pipe
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

size_t cfrk_host_format_sparse2(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint32_t *counts,
                                uint64_t n, char *buf, size_t cap) {
  if (!keys_hi) return cfrk_host_format_sparse(keys_lo, counts, n, buf, cap);
  if (!buf) {
    size_t s = 0;
    for (uint64_t i = 0; i < n; ++i) s += len_u64(keys_hi[i]) + 1 + len_u64(keys_lo[i]) + 1 + len_u64(counts[i]) + 1;
    return s;
  }
  char *p = buf;
  (void)cap;
  for (uint64_t i = 0; i < n; ++i) {
    p = put_u64(p, keys_hi[i]); *p++ = ':';
    p = put_u64(p, keys_lo[i]); *p++ = ':';
    p = put_u64(p, counts[i]); *p++ = '\n';
  }
  return (size_t)(p - buf);
}

size_t cfrk_host_format_sparse_mt(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint32_t *counts,
                                  uint64_t n, char *buf, size_t cap, int threads) {
  int T = threads < 1 ? 1 : threads;
  if ((uint64_t)T > n / 65536 + 1) T = (int)(n / 65536 + 1);      // (a thread per 64 K entries at least)
  if (T == 1) return cfrk_host_format_sparse2(keys_lo, keys_hi, counts, n, buf, cap);
  // entry ranges per thread: sizes first (the text of a range starts where the previous one ends)
  auto e_of = [&](int t) { return n * (uint64_t)t / (uint64_t)T; };
  auto part = [&](int t, char *dst) {
    const uint64_t e0 = e_of(t), e1 = e_of(t + 1);
    return cfrk_host_format_sparse2(keys_lo + e0, keys_hi ? keys_hi + e0 : nullptr, counts + e0, e1 - e0, dst, dst ? (size_t)-1 : 0);
  };
  std::vector<size_t> sz((size_t)T), off((size_t)T);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { sz[(size_t)t] = part(t, nullptr); });
    for (auto &x : th) x.join();
  }
  size_t total = 0;
  for (int t = 0; t < T; ++t) { off[(size_t)t] = total; total += sz[(size_t)t]; }
  if (!buf) return total;
  if (cap < total) return 0;
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] { part(t, buf + off[(size_t)t]); });
  for (auto &x : th) x.join();
  return total;
}

static void put_le(char *p, uint64_t x, int bytes) { for (int i = 0; i < bytes; ++i) p[i] = (char)(x >> (8 * i)); }
static uint64_t get_le(const char *p, int bytes) {
  uint64_t x = 0;
  for (int i = 0; i < bytes; ++i) x |= (uint64_t)(unsigned char)p[i] << (8 * i);
  return x;
}

size_t cfrk_host_write_binary(int k, int flags, const uint64_t *keys_lo, const uint64_t *keys_hi,
                              const uint32_t *counts, uint64_t n, char *buf, size_t cap) {
  const bool two = k > 32;
  const size_t rec = two ? 20 : 12, need = 32 + (size_t)n * rec;
  if (!buf) return need;
  if (cap < need) return 0;
  memcpy(buf, "CFRKGLB1", 8);
  put_le(buf + 8, (uint64_t)k, 4);
  put_le(buf + 12, (uint64_t)((flags & CFRK_BIN_CANONICAL) | (two ? CFRK_BIN_TWO_WORD : 0)), 4);
  put_le(buf + 16, n, 8);
  uint64_t sum = 0;
  char *p = buf + 32;
  for (uint64_t i = 0; i < n; ++i) {
    if (two) { put_le(p, keys_hi ? keys_hi[i] : 0, 8); p += 8; }
    put_le(p, keys_lo[i], 8); p += 8;
    put_le(p, counts[i], 4); p += 4;
    sum += counts[i];
  }
  put_le(buf + 24, sum, 8);
  return need;
}

int cfrk_host_read_binary(const char *buf, size_t len, int *k, int *flags, uint64_t *n, uint64_t *keys_lo,
                          uint64_t *keys_hi, uint32_t *counts) {
  if (!buf || len < 32 || memcmp(buf, "CFRKGLB1", 8) != 0) return -1;
  const int kk = (int)get_le(buf + 8, 4), ff = (int)get_le(buf + 12, 4);
  const uint64_t nn = get_le(buf + 16, 8);
  const bool two = (ff & CFRK_BIN_TWO_WORD) != 0;
  if (kk < 1 || kk > 64 || two != (kk > 32)) return -1;
  const size_t rec = two ? 20 : 12;
  if (nn > (len - 32) / rec || len != 32 + (size_t)nn * rec) return -1;
  if (k) *k = kk;
  if (flags) *flags = ff;
  if (n) *n = nn;
  const char *p = buf + 32;
  uint64_t sum = 0;
  for (uint64_t i = 0; i < nn; ++i) {
    uint64_t hi = 0;
    if (two) { hi = get_le(p, 8); p += 8; }
    const uint64_t lo = get_le(p, 8); p += 8;
    const uint32_t c = (uint32_t)get_le(p, 4); p += 4;
    if (keys_lo) keys_lo[i] = lo;
    if (keys_hi) keys_hi[i] = hi;
    if (counts) counts[i] = c;
    sum += c;
  }
  return sum == get_le(buf + 24, 8) ? 0 : -1;
}

}  // extern "C"
