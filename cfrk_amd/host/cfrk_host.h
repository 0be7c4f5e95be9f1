// cfrk_host.h -- host side of the cfrk drop-in: FASTA ingest, chunking, .cfrk output.
//
// Mirrors the reference's host code around kmer_main() (paths under /root/reference/):
//   ReadFasta / ReadFASTASequences / ProcessData   src/fastaIO.h:24-148
//   SelectChunk / SelectChunkRemain                src/main.cu:110-206
//   PrintFreq                                      src/main.cu:26-62
// Plain C ABI so that the CPU tests (ctypes) and the CLI share one implementation.
// Pure host code: no HIP, no oracle.
#ifndef CFRK_HOST_H
#define CFRK_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CFRK_INGEST_COMPAT 0x1   /* reproduce ReadFasta's quirks (see cfrk_host_read_fasta) */

typedef struct cfrk_batch {      /* struct read (src/tipos.h:23-30) with owned storage */
  int8_t  *data;                 /* nN codes, -1 = invalid base / terminator */
  int64_t *start;                /* nS */
  int32_t *length;               /* nS */
  int64_t nN, nS;
} cfrk_batch;

/* Parse a FASTA file (or memory image) into the struct-read layout.
 * CFRK_INGEST_COMPAT (what the reference does, src/fastaIO.h:24-71,105-148):
 *   - a record starts at a line BEGINNING with '>' (fastaIO.h:40);
 *   - every other line, newline included, is appended to the record (fastaIO.h:49-66), so the
 *     newlines inside a multi-line record and trailing blank lines become -1 codes;
 *   - length = strlen - 1 (fastaIO.h:53,65): the last char is dropped, which is the final
 *     newline, or the last BASE when the file has no final newline.
 * Without the flag: sequence lines are joined without their line ends ('\n', '\r'), nothing
 * is dropped.  Both: aA cC gG tT -> 0 1 2 3, anything else -1 (fastaIO.h:121-140).
 * Returns 0, -1 (cannot open), -2 (sequence before the first header), -3 (header without
 * sequence: undefined behaviour in the reference, rejected here), -4 (out of memory). */
int  cfrk_host_read_fasta(const char *path, int flags, cfrk_batch *out);
int  cfrk_host_parse_fasta(const char *buf, size_t len, int flags, cfrk_batch *out);
void cfrk_host_free_batch(cfrk_batch *b);
/* threads the parser may use for large inputs (0 = the default, min(hardware threads, 64)) */
void cfrk_host_set_parse_threads(int n);

/* Chunk [first, first+count) of a batch with chunk-relative start[] (SelectChunk,
 * src/main.cu:160-206): views into the batch, nothing is copied; start_out needs count slots. */
int cfrk_host_chunk(const cfrk_batch *b, int64_t first, int64_t count, const int8_t **data,
                    int64_t *start_out, const int32_t **length, int64_t *nN);

/* PrintFreq (src/main.cu:26-62): "<idx>:<count> " for every bin, '\n' between rows, none at the
 * end.  Returns bytes needed/written (buf may be NULL to size). */
size_t cfrk_host_format_dense(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap);
/* the same text, formatted by `threads` host threads (row ranges) */
size_t cfrk_host_format_dense_mt(const int32_t *freq, int64_t nS, int k, char *buf, size_t cap, int threads);
/* Sparse global form (what the commented-out `if (Freq[i] != 0)` of src/main.cu:51-56 was heading
 * for): one line per distinct key, ascending.
 *   k <= 32:  "<key>:<count>\n"            key = the 2k-bit k-mer value in decimal (as PrintFreq prints indices)
 *   k  > 32:  "<hi>:<lo>:<count>\n"        key = hi * 2^64 + lo, both words in decimal (keys_hi != NULL) */
size_t cfrk_host_format_sparse(const uint64_t *keys, const uint32_t *counts, uint64_t n, char *buf,
                               size_t cap);
size_t cfrk_host_format_sparse2(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint32_t *counts,
                                uint64_t n, char *buf, size_t cap);
/* the same text, formatted by `threads` host threads (entry ranges); keys_hi may be NULL (k <= 32) */
size_t cfrk_host_format_sparse_mt(const uint64_t *keys_lo, const uint64_t *keys_hi, const uint32_t *counts,
                                  uint64_t n, char *buf, size_t cap, int threads);

/* Binary global form, little endian, everything in one file:
 *   header, 32 bytes:  char magic[8] = "CFRKGLB1"; uint32 k; uint32 flags (bit 0: canonical counting,
 *                      bit 1: two-word keys, i.e. k > 32); uint64 n (records); uint64 sum of counts
 *   n records, ascending by key:   k <= 32: { uint64 key; uint32 count }            12 bytes, packed
 *                                   k  > 32: { uint64 hi; uint64 lo; uint32 count }  20 bytes, packed
 * (12 / 20 bytes are the slot sizes S of the algorithmic-byte budget)
 * write: returns bytes needed / written (buf may be NULL to size); keys_hi may be NULL for k <= 32.
 * read:  parses a whole image; on success (0) *k, *flags, *n are set and, when the arrays are not NULL,
 *        n entries are stored (keys_hi gets zeros for k <= 32).  -1: not a CFRKGLB1 image or truncated. */
#define CFRK_BIN_CANONICAL 0x1
#define CFRK_BIN_TWO_WORD 0x2
size_t cfrk_host_write_binary(int k, int flags, const uint64_t *keys_lo, const uint64_t *keys_hi,
                              const uint32_t *counts, uint64_t n, char *buf, size_t cap);
int cfrk_host_read_binary(const char *buf, size_t len, int *k, int *flags, uint64_t *n, uint64_t *keys_lo,
                          uint64_t *keys_hi, uint32_t *counts);

#ifdef __cplusplus
}
#endif
#endif
