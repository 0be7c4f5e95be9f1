/*
 * cfrk_oracle.h -- CPU restatement of the reference k-mer counting path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker.  The product path (cfrk_amd/, libcfrk_hip.so) never links,
 * imports or calls anything in oracle/.
 *
 * Parity status: PINNED by the reference's own k=2 goldens
 * (tests/golden/out-seq{1,2}.cfrk, copied data files of
 * /root/reference/test/) through golden-derived FASTA pre-images
 * (tests/golden/derive_fasta.py).  The reference's sources are CUDA (.cu) and
 * cannot be built in this image without stand-ins for the CUDA toolkit, so no
 * oracle/_ref build exists; see DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/).
 */
#ifndef CFRK_ORACLE_H
#define CFRK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flags shared by the per-read and global entry points */
#define ORC_COMPAT        0x1  /* ComputeFreqNew semantics: spill bug, L-1 bound, 1024-window cap (src/kmer_kernel.cu:73-90) */
#define ORC_CANONICAL     0x2  /* key = min(fwd, revcomp(fwd)); not in the reference */
#define ORC_FLOAT_INDEX   0x4  /* accumulate the index through float exactly like src/kmer_kernel.cu:38 (differs from exact for k >= 13) */

/* src/fastaIO.h:121-140  ASCII -> {0,1,2,3,-1} */
int8_t orc_encode_base(int c);

/* src/kmer_kernel.cu:21-49  Index[id] for every id < nN; -1 when the k-window
 * starting at id touches a -1 code.  Exact 64-bit integers (k <= 31) unless
 * float_index != 0, which reproduces the reference's float accumulation. */
void orc_compute_index(const int8_t *seq, int64_t nN, int k, int float_index, int64_t *index);

/* src/kmer_kernel.cu:52-70 (native, the guarded ComputeFreq) or
 * src/kmer_kernel.cu:73-90 (compat, ComputeFreqNew as launched at
 * src/kmer_main.cu:111 with block = 1024).  freq is nS * 4^k int32, zeroed here
 * (src/kmer_main.cu:108).  Returns 0, or -1 if k is out of range (k < 1 || k > 15). */
int orc_per_read_dense(const int8_t *data, const int64_t *start, const int32_t *length,
                       int64_t nN, int64_t nS, int k, int flags, int32_t *freq);

/* Global (column-sum) counting, native semantics, any 1 <= k <= 64.
 * Output: distinct keys sorted ascending by (hi, lo) with their counts.
 * Keys: lo = low 64 bits, hi = bits 64..127 (0 for k <= 32); first base most
 * significant (src/kmer_kernel.cu:38: nuc * 4^(k-1-i)).
 * Arrays are malloc'ed; free with orc_free.  Returns number of distinct keys, <0 on error. */
int64_t orc_global_count(const int8_t *data, int64_t nN, int k, int flags,
                         uint64_t **keys_lo, uint64_t **keys_hi, uint64_t **counts);
void orc_free(void *p);

/* Partition + radix-sort variant for large inputs (the big parity cases and bench.py's
 * cpu_baseline): same output as orc_global_count, any 1 <= k <= 64, reads split across nthreads. */
int64_t orc_global_count_sorted(const int8_t *data, int64_t nN, int k, int flags, int nthreads,
                                uint64_t **keys_lo, uint64_t **keys_hi, uint64_t **counts);

/* Bounded-memory digest of the global count of a WHOLE synthetic job (reads [0, R) of orc_synth_reads'
 * generator), for the full-size parity cases (BASELINE configs[2] and one GPU's share of configs[4]):
 * the reads are generated block by block by the scanning threads and the key space is counted in
 * `nslices` rounds of 1/nslices of the keys each, so memory is the genome + one slice's occurrences.
 * out = the digest orc_digest would give for orc_global_count of the same reads.  0, or <0 on error. */
int orc_synth_digest(int64_t R, int L, int64_t Glen, uint64_t seedG, uint64_t seedR, uint64_t seedS,
                     int uniform, int k, int flags, int nthreads, int nslices, uint64_t out[4]);
/* the digest terms of slices [slice0, slice1) only (sums and xor of these over a cover of [0, nslices) give
 * orc_synth_digest; lets a driver report progress between slices) */
int orc_synth_digest_slices(int64_t R, int L, int64_t Glen, uint64_t seedG, uint64_t seedR, uint64_t seedS,
                            int uniform, int k, int flags, int nthreads, int slice0, int slice1, int nslices,
                            uint64_t out[4]);

/* Order-independent digest (SURVEY 8d "Parity at scale").
 * out[0]=D, out[1]=sum count, out[2]=sum count*splitmix64(kh), out[3]=xor splitmix64(kh ^ count)
 * with kh = lo for k <= 32 and lo + splitmix64(hi) otherwise. */
void orc_digest(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint64_t *counts,
                int64_t n, int two_word, uint64_t out[4]);

uint64_t orc_splitmix64(uint64_t x);

/* Synthetic reads: genome g[j] = splitmix64(seedG + j) & 3; read r starts at
 * splitmix64(seedR ^ r) % (Glen-L+1), reverse-complemented when splitmix64(seedS ^ r) & 1.
 * Generates reads [r0, r0+R) in the reference struct-read layout:
 * data[(r-r0)*(L+1)+j], terminator -1, start[r-r0] = (r-r0)*(L+1), length = L.
 * uniform != 0 selects the all-distinct variant: base = splitmix64(seedR ^ (r*256+j)) & 3. */
void orc_synth_reads(int64_t r0, int64_t R, int L, int64_t Glen,
                     uint64_t seedG, uint64_t seedR, uint64_t seedS, int uniform,
                     int8_t *data, int64_t *start, int32_t *length);

/* src/main.cu:26-62 PrintFreq: dense text.  Writes rows [0,nS) of freq (nS x 4^k) into buf
 * ("<idx>:<count> " per bin, '\n' between rows, none at the end).  Returns bytes written,
 * or the required size when buf == NULL. */
size_t orc_format_cfrk(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
