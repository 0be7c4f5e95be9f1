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
                if (pos >= 0) freq[pos] += 1;   /* pos == -1 is the reference's OOB write: dropped */
            }
        }
    } else {
        /* src/kmer_kernel.cu:52-70: every position of the read, guarded by Index != -1 */
        for (int64_t i = 0; i < nS; i++) {
            int64_t end = start[i] + length[i] + 1;
            for (int64_t p = start[i]; p < end && p < nN; p++) {
                if (index[p] != -1) freq[fourk * i + index[p]] += 1;
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

/* ---- multi-threaded variant (bench.py cpu_baseline) ---- */

typedef struct {
    const int8_t *data; int64_t nN, s0, s1; int k, canonical;
    orc_table tab;
} mt_scan_arg;

static void *mt_scan(void *p)
{
    mt_scan_arg *a = (mt_scan_arg *)p;
    scan_range(a->data, a->nN, a->s0, a->s1, a->k, a->canonical, &a->tab);
    return NULL;
}

typedef struct { mt_scan_arg *parts; int nparts, me; orc_table out; } mt_merge_arg;

static void *mt_merge(void *p)
{
    mt_merge_arg *a = (mt_merge_arg *)p;
    for (int s = 0; s < a->nparts; s++) {
        orc_table *t = &a->parts[s].tab;
        for (uint64_t i = 0; i < t->cap; i++)
            if (t->used[i] && (int)((key_hash(t->lo[i], 0) >> 40) % (uint64_t)a->nparts) == a->me)
                tab_add(&a->out, t->lo[i], 0, t->cnt[i]);
    }
    return NULL;
}

int64_t orc_global_count_mt(const int8_t *data, int64_t nN, int k, int flags, int nthreads,
                            uint64_t **keys_lo, uint64_t **counts)
{
    if (k < 1 || k > 32) return -1;
    if (nthreads < 1) nthreads = 1;
    mt_scan_arg *sa = (mt_scan_arg *)calloc((size_t)nthreads, sizeof(*sa));
    mt_merge_arg *ma = (mt_merge_arg *)calloc((size_t)nthreads, sizeof(*ma));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(*th));
    for (int t = 0; t < nthreads; t++) {
        sa[t].data = data; sa[t].nN = nN; sa[t].k = k; sa[t].canonical = (flags & ORC_CANONICAL) != 0;
        sa[t].s0 = nN * t / nthreads; sa[t].s1 = nN * (t + 1) / nthreads;
        if (tab_init(&sa[t].tab, 1 << 16, 0)) return -2;
        pthread_create(&th[t], NULL, mt_scan, &sa[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; t++) {
        ma[t].parts = sa; ma[t].nparts = nthreads; ma[t].me = t;
        if (tab_init(&ma[t].out, 1 << 16, 0)) return -2;
        pthread_create(&th[t], NULL, mt_merge, &ma[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    uint64_t n = 0;
    for (int t = 0; t < nthreads; t++) n += ma[t].out.n;
    orc_ent *e = (orc_ent *)malloc((n ? n : 1) * sizeof(orc_ent));
    uint64_t j = 0;
    for (int t = 0; t < nthreads; t++) {
        orc_table *o = &ma[t].out;
        for (uint64_t i = 0; i < o->cap; i++)
            if (o->used[i]) { e[j].lo = o->lo[i]; e[j].hi = 0; e[j].cnt = o->cnt[i]; j++; }
        tab_free(o);
        tab_free(&sa[t].tab);
    }
    qsort(e, n, sizeof(orc_ent), ent_cmp);
    *keys_lo = (uint64_t *)malloc((n ? n : 1) * 8);
    *counts = (uint64_t *)malloc((n ? n : 1) * 8);
    for (uint64_t i = 0; i < n; i++) { (*keys_lo)[i] = e[i].lo; (*counts)[i] = e[i].cnt; }
    free(e); free(sa); free(ma); free(th);
    return (int64_t)n;
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
