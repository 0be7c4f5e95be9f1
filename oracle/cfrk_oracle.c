/*
 * cfrk_oracle.c -- CPU restatement of the reference k-mer counting path.
 *
 * TEST INFRASTRUCTURE ONLY (see cfrk_oracle.h).  Parity: pinned by the reference's k=2
 * goldens via golden-derived FASTA pre-images; no oracle/_ref build (reference is CUDA).
 * Citations are file:line under /root/reference/.
 */
#include "cfrk_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ encoding */

/* src/fastaIO.h:121-140 */
int8_t orc_encode_base(int c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    default: return -1;
    }
}

uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* ------------------------------------------------------------------ ComputeIndex */

/* src/kmer_kernel.cu:21-49.  One "thread" per id < nN; the inner loop stops at the
 * first -1, so it never reads past the final terminator at nN-1. */
void orc_compute_index(const int8_t *seq, int64_t nN, int k, int float_index, int64_t *index)
{
    for (int64_t id = 0; id < nN; id++) {
        int64_t idx = 0;
        for (int64_t i = 0; i < k; i++) {
            /* the reference reads Seq[i+id] unguarded; a -1 always precedes nN */
            int8_t nuc = (i + id < nN) ? seq[i + id] : (int8_t)-1;
            if (nuc != -1) {
                if (float_index) {
                    /* src/kmer_kernel.cu:38: lint += char * powf -> float add, then truncation */
                    float f = (float)idx + (float)nuc * powf(4.0f, (float)((k - 1) - i));
                    idx = (int64_t)f;
                } else {
                    idx += (int64_t)nuc << (2 * ((k - 1) - i));
                }
            } else {
                idx = -1;
                break;
            }
        }
        index[id] = idx;
    }
}

/* ------------------------------------------------------------------ per-read dense */

int orc_per_read_dense(const int8_t *data, const int64_t *start, const int32_t *length,
                       int64_t nN, int64_t nS, int k, int flags, int32_t *freq)
{
    if (k < 1 || k > 15) return -1;            /* src/tipos.h:5  POW(k) = 1U << 2k in an int */
    const int64_t fourk = (int64_t)1 << (2 * k);
    memset(freq, 0, (size_t)(nS * fourk) * sizeof(int32_t));   /* src/kmer_main.cu:108 */

    int64_t *index = (int64_t *)malloc((size_t)(nN > 0 ? nN : 1) * sizeof(int64_t));
    if (!index) return -2;
    orc_compute_index(data, nN, k, (flags & ORC_FLOAT_INDEX) != 0, index);

    if (flags & ORC_COMPAT) {
        /* src/kmer_kernel.cu:73-90 launched <<<nS, 1024>>> (src/kmer_main.cu:82-83,111):
         * thread t < length[i]-1 does atomicAdd(&Freq[fourk*i + Index[start[i]+t]], 1)
         * with NO -1 guard, so an invalid window lands on Freq[fourk*i - 1]. */
        for (int64_t i = 0; i < nS; i++) {
            int64_t nwin = (int64_t)length[i] - 1;
            if (nwin > 1024) nwin = 1024;       /* blockDim.x = maxThreadsDim[0] = 1024 */
            for (int64_t t = 0; t < nwin; t++) {
                int64_t w = index[start[i] + t];
                int64_t pos = fourk * i + w;    /* w == -1 -> previous row's last bin */
                /* pos == -1 is the reference's OOB write: dropped; so is pos == nS*fourk, which the
                 * float index of an all-T window in the LAST read produces (it rounds up to 4^k) */
                if (pos >= 0 && pos < nS * fourk) freq[pos] += 1;
            }
        }
    } else {
        /* src/kmer_kernel.cu:52-70: every position of the read, guarded by Index != -1 */
        for (int64_t i = 0; i < nS; i++) {
            int64_t end = start[i] + length[i] + 1;
            for (int64_t p = start[i]; p < end && p < nN; p++) {
                if (index[p] != -1 && fourk * i + index[p] < nS * fourk) freq[fourk * i + index[p]] += 1;
            }
        }
    }
    free(index);
    return 0;
}

/* ------------------------------------------------------------------ global counting */

typedef struct {
    uint64_t *lo, *hi, *cnt;
    uint8_t *used;
    uint64_t cap, n;
    int two;
} orc_table;

static int tab_init(orc_table *t, uint64_t cap, int two)
{
    t->cap = cap; t->n = 0; t->two = two;
    t->lo = (uint64_t *)malloc(cap * 8);
    t->hi = two ? (uint64_t *)malloc(cap * 8) : NULL;
    t->cnt = (uint64_t *)malloc(cap * 8);
    t->used = (uint8_t *)calloc(cap, 1);
    return (t->lo && t->cnt && t->used && (!two || t->hi)) ? 0 : -1;
}

static void tab_free(orc_table *t)
{
    free(t->lo); free(t->hi); free(t->cnt); free(t->used);
    memset(t, 0, sizeof(*t));
}

static inline uint64_t key_hash(uint64_t lo, uint64_t hi)
{
    return orc_splitmix64(lo ^ (hi * 0x9E3779B97F4A7C15ULL));
}

static void tab_add(orc_table *t, uint64_t lo, uint64_t hi, uint64_t c);

static void tab_grow(orc_table *t)
{
    orc_table n;
    if (tab_init(&n, t->cap * 2, t->two)) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    for (uint64_t i = 0; i < t->cap; i++)
        if (t->used[i]) tab_add(&n, t->lo[i], t->two ? t->hi[i] : 0, t->cnt[i]);
    tab_free(t);
    *t = n;
}

static void tab_add(orc_table *t, uint64_t lo, uint64_t hi, uint64_t c)
{
    if ((t->n + 1) * 10 > t->cap * 7) tab_grow(t);
    uint64_t m = t->cap - 1, h = key_hash(lo, hi) & m;
    for (;;) {
        if (!t->used[h]) {
            t->used[h] = 1; t->lo[h] = lo; if (t->two) t->hi[h] = hi; t->cnt[h] = c; t->n++;
            return;
        }
        if (t->lo[h] == lo && (!t->two || t->hi[h] == hi)) { t->cnt[h] += c; return; }
        h = (h + 1) & m;
    }
}

/* Scan windows that START in [s0, s1) of the flat buffer; a window is valid iff all k codes
 * are in 0..3 (ComputeIndex's -1 rule, src/kmer_kernel.cu:36-46).  Terminators are -1, so
 * no window crosses a read boundary. */
static void scan_range(const int8_t *data, int64_t nN, int64_t s0, int64_t s1, int k, int canonical,
                       orc_table *t)
{
    const u128 mask = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    u128 fwd = 0, rc = 0;
    int run = 0;
    int64_t end = s1 + k - 1;
    if (end > nN) end = nN;
    for (int64_t p = s0; p < end; p++) {
        int8_t c = data[p];
        if (c < 0 || c > 3) { run = 0; fwd = 0; rc = 0; continue; }
        fwd = ((fwd << 2) | (u128)c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
        if (++run >= k) {
            int64_t st = p - k + 1;
            if (st >= s0 && st < s1) {
                u128 key = (canonical && rc < fwd) ? rc : fwd;
                tab_add(t, (uint64_t)key, (uint64_t)(key >> 64), 1);
            }
        }
    }
}

typedef struct { uint64_t lo, hi, cnt; } orc_ent;

static int ent_cmp(const void *a, const void *b)
{
    const orc_ent *x = (const orc_ent *)a, *y = (const orc_ent *)b;
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    return 0;
}

static int64_t tab_export_sorted(orc_table *t, uint64_t **klo, uint64_t **khi, uint64_t **cnt)
{
    uint64_t n = t->n;
    orc_ent *e = (orc_ent *)malloc((n ? n : 1) * sizeof(orc_ent));
    if (!e) return -2;
    uint64_t j = 0;
    for (uint64_t i = 0; i < t->cap; i++)
        if (t->used[i]) { e[j].lo = t->lo[i]; e[j].hi = t->two ? t->hi[i] : 0; e[j].cnt = t->cnt[i]; j++; }
    qsort(e, n, sizeof(orc_ent), ent_cmp);
    *klo = (uint64_t *)malloc((n ? n : 1) * 8);
    *cnt = (uint64_t *)malloc((n ? n : 1) * 8);
    if (khi) *khi = (uint64_t *)malloc((n ? n : 1) * 8);
    for (uint64_t i = 0; i < n; i++) {
        (*klo)[i] = e[i].lo; (*cnt)[i] = e[i].cnt;
        if (khi) (*khi)[i] = e[i].hi;
    }
    free(e);
    return (int64_t)n;
}

int64_t orc_global_count(const int8_t *data, int64_t nN, int k, int flags,
                         uint64_t **keys_lo, uint64_t **keys_hi, uint64_t **counts)
{
    if (k < 1 || k > 64) return -1;
    orc_table t;
    if (tab_init(&t, 1 << 16, k > 32)) return -2;
    scan_range(data, nN, 0, nN, k, (flags & ORC_CANONICAL) != 0, &t);
    int64_t n = tab_export_sorted(&t, keys_lo, keys_hi, counts);
    tab_free(&t);
    return n;
}

void orc_free(void *p) { free(p); }

/* ---- partition + radix-sort variant (large parity cases, bench.py cpu_baseline) ----
 * Same result as orc_global_count (tests/test_oracle.py compares the two), any k <= 64, any
 * thread count: (1) every thread scans its share of the window starts twice -- count per
 * partition (top bits of the key), then write the keys to their exact places; (2) partitions
 * are sorted (LSD radix on the remaining bits) and run-length encoded, one partition at a
 * time from a shared queue; (3) partitions are in key order, so concatenation is the sorted
 * (hi, lo) list.  The semantics are scan_range's: src/kmer_kernel.cu:36-46 window validity,
 * the guarded ComputeFreq of src/kmer_kernel.cu:52-70 summed over reads. */

typedef struct { uint64_t hi, lo; } key2;

typedef struct {
    const int8_t *data; int64_t nN; int k, canonical, two, pbits, nthreads, nparts;
    int64_t *s0, *s1;             /* per thread: window starts [s0, s1) */
    uint64_t *tcnt;               /* [nthreads][nparts] counts, then write cursors */
    uint64_t *pstart;             /* [nparts + 1] first key of every partition */
    uint64_t *k1; key2 *k2;       /* the keys (one-word / two-word) */
    uint32_t *rle;                /* run lengths, at the compacted positions */
    uint64_t *pdist;              /* [nparts] distinct keys per partition */
    uint64_t *ostart;             /* [nparts + 1] output offsets */
    uint64_t *out_lo, *out_hi, *out_cnt;
    volatile int64_t next;        /* partition queue */
    uint64_t maxpart;
    int phase, oom;
    pthread_barrier_t bar;
    /* bounded-memory mode (orc_synth_digest): only the keys of slice `slice` of `nslices` are kept,
     * and the input is not a buffer but the generator of orc_synth_reads, block by block */
    int slice, nslices;
    int synth, L, uniform;
    int64_t R, Glen;
    uint64_t seedG, seedR, seedS;
    const int8_t *genome;         /* g[j] = splitmix64(seedG + j) & 3, cached (NULL for uniform reads) */
    int64_t *r0, *r1;             /* per thread: reads [r0, r1) */
    uint64_t *dig;                /* [nthreads][4] digest terms of the partitions a thread finished */
} ps_job;

typedef struct { ps_job *j; int t; } ps_arg;

static inline uint32_t ps_part(const ps_job *j, u128 key)
{
    const int sh = 2 * j->k - j->pbits;
    return (uint32_t)(key >> sh);
}

/* which slice of the key space a key belongs to (bounded-memory mode); any fixed function of the
 * key will do: every occurrence of a key lands in the same slice */
static inline int ps_slice(const ps_job *j, uint64_t lo, uint64_t hi)
{
    uint64_t h = (lo ^ (hi * 0xD6E8FEB86659FD93ULL)) * 0x9E3779B97F4A7C15ULL;
    return (int)(((h >> 32) * (uint64_t)j->nslices) >> 32);
}

/* every window inside d[0, n), the scan starting with no base seen: pass 0 counts per partition,
 * pass 1 writes the keys (one-word arithmetic for k <= 32, the same loop on 128-bit values above) */
static void ps_scan_span(ps_job *j, uint64_t *cur, const int8_t *d, int64_t n, int pass)
{
    const int k = j->k;
    const int sh = 2 * k - j->pbits;
    const int canonical = j->canonical;
    const int sliced = j->nslices > 1;
    int run = 0;
    if (!j->two) {
        const uint64_t mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
        uint64_t fwd = 0, rc = 0;
        uint64_t *out = j->k1;
        for (int64_t p = 0; p < n; p++) {
            int8_t c = d[p];
            if (c < 0 || c > 3) { run = 0; fwd = 0; rc = 0; continue; }
            fwd = ((fwd << 2) | (uint64_t)c) & mask;
            rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
            if (++run >= k) {
                uint64_t key = (canonical && rc < fwd) ? rc : fwd;
                if (sliced && ps_slice(j, key, 0) != j->slice) continue;
                uint32_t q = (uint32_t)(key >> sh);
                if (pass == 0) cur[q]++;
                else out[cur[q]++] = key;
            }
        }
        return;
    }
    const u128 mask = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    u128 fwd = 0, rc = 0;
    for (int64_t p = 0; p < n; p++) {
        int8_t c = d[p];
        if (c < 0 || c > 3) { run = 0; fwd = 0; rc = 0; continue; }
        fwd = ((fwd << 2) | (u128)c) & mask;
        rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));
        if (++run >= k) {
            u128 key = (canonical && rc < fwd) ? rc : fwd;
            if (sliced && ps_slice(j, (uint64_t)key, (uint64_t)(key >> 64)) != j->slice) continue;
            uint32_t q = ps_part(j, key);
            if (pass == 0) cur[q]++;
            else { key2 v; v.hi = (uint64_t)(key >> 64); v.lo = (uint64_t)key; j->k2[cur[q]++] = v; }
        }
    }
}

/* reads [r, r + n) of the generator into buf, n * (L + 1) bytes: the same values as orc_synth_reads,
 * with the genome taken from the cached copy */
static void ps_synth_block(const ps_job *j, int64_t r, int64_t n, int8_t *buf)
{
    const int L = j->L;
    if (j->uniform) { orc_synth_reads(r, n, L, j->Glen, j->seedG, j->seedR, j->seedS, 1, buf, NULL, NULL); return; }
    for (int64_t i = 0; i < n; i++) {
        uint64_t rr = (uint64_t)(r + i);
        int8_t *d = buf + i * (int64_t)(L + 1);
        uint64_t pos = orc_splitmix64(j->seedR ^ rr) % (uint64_t)(j->Glen - L + 1);
        const int8_t *g = j->genome + pos;
        if (!(orc_splitmix64(j->seedS ^ rr) & 1)) memcpy(d, g, (size_t)L);
        else for (int q = 0; q < L; q++) d[q] = (int8_t)(3 - g[L - 1 - q]);
        d[L] = -1;
    }
}

#define PS_SYNTH_BLOCK 4096   /* reads generated at a time by a thread of the bounded-memory mode */

/* windows that start in [s0, s1) of the buffer (a window starts there iff it ENDS in
 * [s0 + k - 1, s1 + k - 1)), or -- bounded-memory mode -- every window of the thread's reads */
static void ps_scan(ps_job *j, int t, int pass)
{
    uint64_t *cur = j->tcnt + (size_t)t * j->nparts;
    if (j->synth) {
        int8_t *buf = (int8_t *)malloc((size_t)PS_SYNTH_BLOCK * (size_t)(j->L + 1));
        if (!buf) { j->oom = 1; return; }
        for (int64_t r = j->r0[t]; r < j->r1[t]; r += PS_SYNTH_BLOCK) {
            int64_t n = j->r1[t] - r < PS_SYNTH_BLOCK ? j->r1[t] - r : PS_SYNTH_BLOCK;
            ps_synth_block(j, r, n, buf);
            ps_scan_span(j, cur, buf, n * (int64_t)(j->L + 1), pass);
        }
        free(buf);
        return;
    }
    const int64_t s0 = j->s0[t], s1 = j->s1[t];
    int64_t end = s1 + j->k - 1;
    if (end > j->nN) end = j->nN;
    if (end > s0) ps_scan_span(j, cur, j->data + s0, end - s0, pass);
}

#define PS_RB 11
static void ps_sort1(uint64_t *a, uint64_t *tmp, uint64_t n, int bits)
{
    if (n < 2) return;
    if (n <= 32) {
        for (uint64_t i = 1; i < n; i++) { uint64_t x = a[i]; uint64_t q = i; while (q && a[q - 1] > x) { a[q] = a[q - 1]; q--; } a[q] = x; }
        return;
    }
    uint64_t *src = a, *dst = tmp;
    for (int sh = 0; sh < bits; sh += PS_RB) {
        uint64_t h[1 << PS_RB] = {0};
        const uint64_t dm = (1u << PS_RB) - 1;
        for (uint64_t i = 0; i < n; i++) h[(src[i] >> sh) & dm]++;
        int single = 0;
        for (int b = 0; b < (1 << PS_RB); b++) if (h[b] == n) single = 1;
        if (single) continue;
        uint64_t run = 0;
        for (int b = 0; b < (1 << PS_RB); b++) { uint64_t x = h[b]; h[b] = run; run += x; }
        for (uint64_t i = 0; i < n; i++) dst[h[(src[i] >> sh) & dm]++] = src[i];
        uint64_t *sw = src; src = dst; dst = sw;
    }
    if (src != a) memcpy(a, src, n * 8);
}

static inline int key2_gt(key2 x, key2 y) { return x.hi != y.hi ? x.hi > y.hi : x.lo > y.lo; }

static void ps_sort2(key2 *a, key2 *tmp, uint64_t n, int bits)
{
    if (n < 2) return;
    if (n <= 32) {
        for (uint64_t i = 1; i < n; i++) { key2 x = a[i]; uint64_t q = i; while (q && key2_gt(a[q - 1], x)) { a[q] = a[q - 1]; q--; } a[q] = x; }
        return;
    }
    key2 *src = a, *dst = tmp;
    for (int sh = 0; sh < bits; sh += 8) {
        uint64_t h[256] = {0};
#define PS_DIG(v) ((sh < 64 ? ((v).lo >> sh) | (sh > 56 ? (v).hi << (64 - sh) : 0) : (v).hi >> (sh - 64)) & 255)
        for (uint64_t i = 0; i < n; i++) h[PS_DIG(src[i])]++;
        int single = 0;
        for (int b = 0; b < 256; b++) if (h[b] == n) single = 1;
        if (single) continue;
        uint64_t run = 0;
        for (int b = 0; b < 256; b++) { uint64_t x = h[b]; h[b] = run; run += x; }
        for (uint64_t i = 0; i < n; i++) dst[h[PS_DIG(src[i])]++] = src[i];
#undef PS_DIG
        key2 *sw = src; src = dst; dst = sw;
    }
    if (src != a) memcpy(a, src, n * sizeof(key2));
}

static void *ps_worker(void *p)
{
    ps_arg *a = (ps_arg *)p;
    ps_job *j = a->j;
    const int t = a->t, P = j->nparts, T = j->nthreads;
    ps_scan(j, t, 0);
    pthread_barrier_wait(&j->bar);
    if (t == 0) {
        /* partition starts; thread cursors inside a partition in thread order */
        uint64_t run = 0, mx = 0;
        for (int q = 0; q < P; q++) {
            j->pstart[q] = run;
            for (int u = 0; u < T; u++) { uint64_t c = j->tcnt[(size_t)u * P + q]; j->tcnt[(size_t)u * P + q] = run; run += c; }
            if (run - j->pstart[q] > mx) mx = run - j->pstart[q];
        }
        j->pstart[P] = run;
        j->maxpart = mx;
        if (j->two) j->k2 = (key2 *)malloc((run ? run : 1) * sizeof(key2));
        else j->k1 = (uint64_t *)malloc((run ? run : 1) * 8);
        j->rle = (uint32_t *)malloc((run ? run : 1) * 4);
        if ((!j->k1 && !j->k2) || !j->rle) j->oom = 1;
    }
    pthread_barrier_wait(&j->bar);
    if (j->oom) return NULL;
    ps_scan(j, t, 1);
    pthread_barrier_wait(&j->bar);
    {
        void *tmp = malloc((j->maxpart ? j->maxpart : 1) * (j->two ? sizeof(key2) : 8));
        if (!tmp) j->oom = 1;
        const int bits = 2 * j->k - j->pbits;
        for (;;) {
            int64_t q = __sync_fetch_and_add(&j->next, 1);
            if (q >= P || !tmp) break;
            const uint64_t b = j->pstart[q], n = j->pstart[q + 1] - b;
            uint64_t d = 0;
            if (j->two) {
                key2 *v = j->k2 + b;
                ps_sort2(v, (key2 *)tmp, n, bits);
                for (uint64_t i = 0; i < n;) {
                    uint64_t e = i + 1;
                    while (e < n && v[e].hi == v[i].hi && v[e].lo == v[i].lo) e++;
                    v[d] = v[i]; j->rle[b + d] = (uint32_t)(e - i); d++; i = e;
                }
            } else {
                uint64_t *v = j->k1 + b;
                ps_sort1(v, (uint64_t *)tmp, n, bits);
                for (uint64_t i = 0; i < n;) {
                    uint64_t e = i + 1;
                    while (e < n && v[e] == v[i]) e++;
                    v[d] = v[i]; j->rle[b + d] = (uint32_t)(e - i); d++; i = e;
                }
            }
            j->pdist[q] = d;
            if (j->dig) {   /* bounded-memory mode: only the digest terms of the partition are kept */
                uint64_t *g = j->dig + 4 * (size_t)t;
                for (uint64_t i = 0; i < d; i++) {
                    uint64_t c = j->rle[b + i];
                    uint64_t kh = j->two ? j->k2[b + i].lo + orc_splitmix64(j->k2[b + i].hi) : j->k1[b + i];
                    g[1] += c; g[2] += c * orc_splitmix64(kh); g[3] ^= orc_splitmix64(kh ^ c);
                }
                g[0] += d;
            }
        }
        free(tmp);
    }
    pthread_barrier_wait(&j->bar);
    if (j->dig) return NULL;
    if (t == 0 && !j->oom) {
        uint64_t run = 0;
        for (int q = 0; q < P; q++) { j->ostart[q] = run; run += j->pdist[q]; }
        j->ostart[P] = run;
        j->out_lo = (uint64_t *)malloc((run ? run : 1) * 8);
        j->out_hi = (uint64_t *)calloc(run ? run : 1, 8);
        j->out_cnt = (uint64_t *)malloc((run ? run : 1) * 8);
        if (!j->out_lo || !j->out_hi || !j->out_cnt) j->oom = 1;
    }
    pthread_barrier_wait(&j->bar);
    if (j->oom) return NULL;
    for (int q = t; q < P; q += T) {
        const uint64_t b = j->pstart[q], o = j->ostart[q], d = j->pdist[q];
        for (uint64_t i = 0; i < d; i++) {
            if (j->two) { j->out_lo[o + i] = j->k2[b + i].lo; j->out_hi[o + i] = j->k2[b + i].hi; }
            else j->out_lo[o + i] = j->k1[b + i];
            j->out_cnt[o + i] = j->rle[b + i];
        }
    }
    return NULL;
}

int64_t orc_global_count_sorted(const int8_t *data, int64_t nN, int k, int flags, int nthreads,
                                uint64_t **keys_lo, uint64_t **keys_hi, uint64_t **counts)
{
    if (k < 1 || k > 64) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    ps_job j;
    memset(&j, 0, sizeof j);
    j.data = data; j.nN = nN; j.k = k; j.canonical = (flags & ORC_CANONICAL) != 0; j.two = k > 32;
    j.pbits = 2 * k < 12 ? 2 * k : 12;
    j.nparts = 1 << j.pbits;
    j.nthreads = nthreads;
    j.s0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads);
    j.s1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads);
    j.tcnt = (uint64_t *)calloc((size_t)nthreads * j.nparts, 8);
    j.pstart = (uint64_t *)calloc((size_t)j.nparts + 1, 8);
    j.pdist = (uint64_t *)calloc((size_t)j.nparts, 8);
    j.ostart = (uint64_t *)calloc((size_t)j.nparts + 1, 8);
    ps_arg *args = (ps_arg *)calloc((size_t)nthreads, sizeof(ps_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!j.s0 || !j.s1 || !j.tcnt || !j.pstart || !j.pdist || !j.ostart || !args || !th) return -2;
    pthread_barrier_init(&j.bar, NULL, (unsigned)nthreads);
    for (int t = 0; t < nthreads; t++) {
        j.s0[t] = nN * t / nthreads; j.s1[t] = nN * (t + 1) / nthreads;
        args[t].j = &j; args[t].t = t;
        pthread_create(&th[t], NULL, ps_worker, &args[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&j.bar);
    int64_t n = j.oom ? -2 : (int64_t)j.ostart[j.nparts];
    free(j.k1); free(j.k2); free(j.rle);
    free(j.s0); free(j.s1); free(j.tcnt); free(j.pstart); free(j.pdist); free(j.ostart); free(args); free(th);
    if (n < 0) { free(j.out_lo); free(j.out_hi); free(j.out_cnt); return n; }
    *keys_lo = j.out_lo; *counts = j.out_cnt;
    if (keys_hi) *keys_hi = j.out_hi; else free(j.out_hi);
    return n;
}

/* Bounded-memory count of a whole synthetic job, for the full-size parity cases: the reads of the
 * generator (orc_synth_reads) are never held in memory -- every thread makes its share of them block by
 * block -- and the key space is counted in `nslices` rounds, each keeping only the keys of its slice
 * (partition, sort, run-length encode exactly as orc_global_count_sorted does) and adding the slice's
 * digest terms.  Slices are disjoint in key space, so the sums and the xor over them are the digest of
 * the whole count.  Memory: the genome (Glen bytes) + 12 B per k-mer occurrence of one slice (28 B for
 * k > 32).  Same semantics as everything above: src/kmer_kernel.cu:36-46 window validity, the guarded
 * ComputeFreq of src/kmer_kernel.cu:52-70 summed over reads.  Returns 0, -1 bad argument, -2 memory. */
typedef struct { int8_t *g; int64_t a, b; uint64_t seedG; } gen_arg;
static void *gen_worker(void *p)
{
    gen_arg *a = (gen_arg *)p;
    for (int64_t q = a->a; q < a->b; q++) a->g[q] = (int8_t)(orc_splitmix64(a->seedG + (uint64_t)q) & 3);
    return NULL;
}

/* slices [slice0, slice1) of nslices */
int orc_synth_digest_slices(int64_t R, int L, int64_t Glen, uint64_t seedG, uint64_t seedR, uint64_t seedS,
                            int uniform, int k, int flags, int nthreads, int slice0, int slice1, int nslices,
                            uint64_t out[4])
{
    if (k < 1 || k > 64 || R < 0 || L < 1 || nslices < 1 || (!uniform && Glen < L)) return -1;
    if (slice0 < 0 || slice1 > nslices || slice0 > slice1) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    int rc = 0;
    int8_t *genome = NULL;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!th) return -2;
    if (!uniform) {
        genome = (int8_t *)malloc((size_t)Glen);
        gen_arg *ga = (gen_arg *)calloc((size_t)nthreads, sizeof(gen_arg));
        if (!genome || !ga) { free(genome); free(ga); free(th); return -2; }
        for (int t = 0; t < nthreads; t++) {
            ga[t].g = genome; ga[t].a = Glen * t / nthreads; ga[t].b = Glen * (t + 1) / nthreads; ga[t].seedG = seedG;
            pthread_create(&th[t], NULL, gen_worker, &ga[t]);
        }
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(ga);
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int s = slice0; s < slice1 && rc == 0; s++) {
        ps_job j;
        memset(&j, 0, sizeof j);
        j.k = k; j.canonical = (flags & ORC_CANONICAL) != 0; j.two = k > 32;
        j.pbits = 2 * k < 12 ? 2 * k : 12;
        j.nparts = 1 << j.pbits;
        j.nthreads = nthreads;
        j.slice = s; j.nslices = nslices;
        j.synth = 1; j.L = L; j.uniform = uniform; j.R = R; j.Glen = Glen;
        j.seedG = seedG; j.seedR = seedR; j.seedS = seedS; j.genome = genome;
        j.r0 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads);
        j.r1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads);
        j.tcnt = (uint64_t *)calloc((size_t)nthreads * j.nparts, 8);
        j.pstart = (uint64_t *)calloc((size_t)j.nparts + 1, 8);
        j.pdist = (uint64_t *)calloc((size_t)j.nparts, 8);
        j.dig = (uint64_t *)calloc((size_t)nthreads * 4, 8);
        ps_arg *args = (ps_arg *)calloc((size_t)nthreads, sizeof(ps_arg));
        if (!j.r0 || !j.r1 || !j.tcnt || !j.pstart || !j.pdist || !j.dig || !args) rc = -2;
        else {
            pthread_barrier_init(&j.bar, NULL, (unsigned)nthreads);
            for (int t = 0; t < nthreads; t++) {
                j.r0[t] = R * t / nthreads; j.r1[t] = R * (t + 1) / nthreads;
                args[t].j = &j; args[t].t = t;
                pthread_create(&th[t], NULL, ps_worker, &args[t]);
            }
            for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
            pthread_barrier_destroy(&j.bar);
            if (j.oom) rc = -2;
            else for (int t = 0; t < nthreads; t++) {
                out[0] += j.dig[4 * t]; out[1] += j.dig[4 * t + 1]; out[2] += j.dig[4 * t + 2]; out[3] ^= j.dig[4 * t + 3];
            }
        }
        free(j.k1); free(j.k2); free(j.rle);
        free(j.r0); free(j.r1); free(j.tcnt); free(j.pstart); free(j.pdist); free(j.dig); free(args);
    }
    free(genome); free(th);
    return rc;
}

int orc_synth_digest(int64_t R, int L, int64_t Glen, uint64_t seedG, uint64_t seedR, uint64_t seedS,
                     int uniform, int k, int flags, int nthreads, int nslices, uint64_t out[4])
{
    return orc_synth_digest_slices(R, L, Glen, seedG, seedR, seedS, uniform, k, flags, nthreads, 0, nslices, nslices, out);
}

/* ------------------------------------------------------------------ digest */

void orc_digest(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint64_t *counts,
                int64_t n, int two_word, uint64_t out[4])
{
    uint64_t s = 0, w = 0, x = 0;
    for (int64_t i = 0; i < n; i++) {
        uint64_t kh = two_word ? keys_lo[i] + orc_splitmix64(keys_hi[i]) : keys_lo[i];
        s += counts[i];
        w += counts[i] * orc_splitmix64(kh);
        x ^= orc_splitmix64(kh ^ counts[i]);
    }
    out[0] = (uint64_t)n; out[1] = s; out[2] = w; out[3] = x;
}

/* ------------------------------------------------------------------ synthetic reads */

void orc_synth_reads(int64_t r0, int64_t R, int L, int64_t Glen,
                     uint64_t seedG, uint64_t seedR, uint64_t seedS, int uniform,
                     int8_t *data, int64_t *start, int32_t *length)
{
    for (int64_t i = 0; i < R; i++) {
        uint64_t r = (uint64_t)(r0 + i);
        int8_t *d = data + i * (int64_t)(L + 1);
        if (uniform) {
            for (int j = 0; j < L; j++) d[j] = (int8_t)(orc_splitmix64(seedR ^ (r * 256 + (uint64_t)j)) & 3);
        } else {
            uint64_t pos = orc_splitmix64(seedR ^ r) % (uint64_t)(Glen - L + 1);
            int rcs = (int)(orc_splitmix64(seedS ^ r) & 1);
            for (int j = 0; j < L; j++) {
                if (!rcs) d[j] = (int8_t)(orc_splitmix64(seedG + pos + (uint64_t)j) & 3);
                else d[j] = (int8_t)(3 - (orc_splitmix64(seedG + pos + (uint64_t)(L - 1 - j)) & 3));
            }
        }
        d[L] = -1;
        if (start) start[i] = i * (int64_t)(L + 1);
        if (length) length[i] = L;
    }
}

/* ------------------------------------------------------------------ .cfrk text */

/* src/main.cu:26-62 PrintFreq: sprintf("%d:%d ", cont, Freq[i]); '\n' before every row but the
 * first; nothing after the last row. */
size_t orc_format_cfrk(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap)
{
    const int64_t fourk = (int64_t)1 << (2 * k);
    size_t n = 0;
    char tmp[64];
    for (int64_t i = 0; i < nS; i++) {
        if (i != 0) { if (buf && n < cap) buf[n] = '\n'; n++; }
        for (int64_t b = 0; b < fourk; b++) {
            int m = snprintf(tmp, sizeof tmp, "%d:%d ", (int)b, (int)freq[i * fourk + b]);
            if (buf && n + (size_t)m <= cap) memcpy(buf + n, tmp, (size_t)m);
            n += (size_t)m;
        }
    }
    return n;
}
