/*
 * cfrk_abi.h -- C ABI of libcfrk_hip.so, the MI355X (gfx950) replacement for the
 * reference's device hot path.
 *
 * What it replaces (paths under /root/reference/):
 *   void kmer_main(struct read *rd, lint nN, lint nS, int k, ushort device);
 *        declared src/kmer.cuh:6, defined src/kmer_main.cu:20-128, called from
 *        src/main.cu:222,294,300; launches SetMatrix / ComputeIndex / ComputeFreqNew
 *        (src/kmer_kernel.cu:6-90).
 *   struct read { char *data; int *length; lint *start; int *Freq; ... }   src/tipos.h:23-30
 *
 * Data contract (identical to struct read):
 *   data    int8 codes, A=0 C=1 G=2 T=3 (src/fastaIO.h:121-140); any other value (the
 *           reference uses -1) is an invalid base AND the per-read terminator
 *           (src/fastaIO.h:96).
 *   length  bases per read, terminator excluded (src/fastaIO.h:98).
 *   start   byte offset of read i in data; start[0]=0,
 *           start[i]=start[i-1]+length[i-1]+1 (src/main.cu:195-200).
 *   nN      bytes in data = sum(length)+nS (src/fastaIO.h:145).
 *   k-mer index = sum_i code[i] * 4^(k-1-i), first base most significant
 *           (src/kmer_kernel.cu:38).
 *
 * Conventions: every function returns CFRK_OK (0) or a negative CFRK_ERR_* code; the
 * library never prints and never calls exit() (the reference does both:
 * src/kmer_main.cu:51-63).  A cfrk_ctx owns one HIP stream and a device memory pool that
 * persists across calls (the reference mallocs/frees per call, src/kmer_main.cu:59-63,
 * 120-124).  Different contexts may be used concurrently from different threads; one
 * context may not.  Plain pointers and sizes only: no torch / C++ types cross this boundary.
 */
#ifndef CFRK_ABI_H
#define CFRK_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFRK_ABI_VERSION 1

/* error codes */
#define CFRK_OK              0
#define CFRK_ERR_ARG        -1   /* bad argument (k out of range, NULL pointer, negative size) */
#define CFRK_ERR_NOMEM      -2   /* device (or pinned host) allocation failed / would not fit   */
#define CFRK_ERR_HIP        -3   /* a HIP runtime call failed; see cfrk_last_error              */
#define CFRK_ERR_STATE      -4   /* call sequence violated (e.g. add before begin)              */
#define CFRK_ERR_LAYOUT     -5   /* data/start/length inconsistent with the struct-read layout  */
#define CFRK_ERR_TABLE_FULL -6   /* global table overflowed; re-run with a larger capacity_hint */
#define CFRK_ERR_ALIGN      -7   /* device data pointer not 16-byte aligned                      */
#define CFRK_ERR_NO_DEVICE  -8   /* no usable gfx950 device                                      */
#define CFRK_ERR_SMALL_BUF  -9   /* output buffer smaller than the result                        */
#define CFRK_ERR_COUNT_OVERFLOW -10 /* a key occurred 2^32 - 2 times or more: counts are 32-bit and SATURATE at
                                    2^32 - 2 (0xFFFFFFFE) instead of wrapping; finish / digest / export report it */
#define CFRK_ERR_RUNS_REFUSED -11 /* a CFRK_RUNS_ONLY add that does not fit device memory in one pass: nothing was
                                    counted, the job is still empty; count without the flag (leaf / key exchange)   */
#define CFRK_COUNT_MAX 0xFFFFFFFE   /* (an unsigned int: a hexadecimal constant that does not fit int) */

/* flags */
#define CFRK_COMPAT     0x1  /* per-read dense only: reproduce ComputeFreqNew exactly (no -1 guard ->
                                spill into the previous row's last bin, bound length-1, 1024-window
                                cap; src/kmer_kernel.cu:73-90, src/kmer_main.cu:82-83).  Without it:
                                the guarded ComputeFreq semantics (src/kmer_kernel.cu:52-70).        */
#define CFRK_CANONICAL  0x2  /* global only: key = min(kmer, reverse complement)                     */
#define CFRK_FORCE_HASH 0x4  /* global only: count with one HBM atomic per occurrence (the general
                                path) even where the minimizer-partitioned LDS path applies          */
#define CFRK_RUNS_ONLY  0x8  /* global only, 16 <= k <= 64: the job partitions and deduplicates ONE add
                                (which must fit device memory in one pass) and stops there; its result
                                leaves through cfrk_global_export_runs_device (multi-GPU exchange)   */
#define CFRK_RUNS_DEFER 0x20  /* with CFRK_RUNS_ONLY (16 <= k <= 64): the add only PARTITIONS -- no
                                deduplication kernel and, when the leaf streams have their fixed stride (a batch that fits
                                one pass that way; not a chunked or count-first add of a two-word job, which needs its
                                read-backs), no host synchronisation at its end (its overflow flags are looked at by the
                                export).  The shard then leaves through the PIPELINED export,
                                cfrk_global_export_runs_async / _wait (deduplication and packing in one kernel per group of
                                leaves), or through cfrk_global_export_runs_device, which deduplicates first.             */
#define CFRK_FLOAT_INDEX 0x10 /* per-read dense only, matters for k = 13..15: the window index is accumulated
                                through float exactly as ComputeIndex does (index += nuc * powf(4, k-1-i),
                                src/kmer_kernel.cu:38), rounding errors, the all-T window's carry into the
                                next row and all -- byte-identical to what the reference computes where the
                                reference is numerically wrong.  Without it: exact integers (for k <= 12 the
                                two agree).  Combine with CFRK_COMPAT for ComputeFreqNew's semantics.        */

typedef struct cfrk_ctx cfrk_ctx;

/* ---- context ------------------------------------------------------------------------- */

/* Replaces cudaSetDevice + GetDeviceProp + per-call cudaMalloc (src/kmer_main.cu:40-63).
 * hip_stream == NULL: the context creates its own non-blocking stream; otherwise it
 * launches on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int  cfrk_ctx_create(int device, void *hip_stream, cfrk_ctx **out);
void cfrk_ctx_destroy(cfrk_ctx *ctx);
int  cfrk_abi_version(void);
const char *cfrk_strerror(int code);
const char *cfrk_last_error(const cfrk_ctx *ctx);   /* detail of the last failure on ctx */
int  cfrk_device_count(int *count);
int  cfrk_ctx_sync(cfrk_ctx *ctx);                  /* cudaStreamSynchronize(0) at src/main.cu:223 */

/* plain device buffers on the context's device (for callers without another allocator) */
int  cfrk_device_alloc(cfrk_ctx *ctx, size_t bytes, void **dptr);
int  cfrk_device_free(cfrk_ctx *ctx, void *dptr);
int  cfrk_memcpy_h2d(cfrk_ctx *ctx, void *dst_device, const void *src_host, size_t bytes);
int  cfrk_memcpy_d2h(cfrk_ctx *ctx, void *dst_host, const void *src_device, size_t bytes);
/* Device-to-device copy between two contexts, possibly on different GPUs of the node: over xGMI
 * peer-to-peer (hipMemcpyPeerAsync; peer access is enabled on first use where the topology allows it),
 * enqueued on dst_ctx's stream.  src_ctx's stream is drained first, so whatever src_ctx was asked to
 * write is complete in the copy; the call returns once the copy is enqueued (dst_ctx's later work is
 * ordered behind it).  src_ctx is only read: several threads may copy from one source context at the same time, each
 * into its own dst_ctx, and every failure is reported on dst_ctx (cfrk_last_error(dst_ctx)).  This is the exchange
 * step of a one-process multi-GPU host (the `cfrk` CLI); one process per GPU uses RCCL instead (cfrk_amd/sharded.py). */
int  cfrk_memcpy_peer(cfrk_ctx *dst_ctx, void *dst_device, cfrk_ctx *src_ctx, const void *src_device, size_t bytes);

/* ---- per-read dense counting: the drop-in for kmer_main ------------------------------- */

/* Host buffers in, host buffer out; synchronous like kmer_main (blocking D2H at
 * src/kmer_main.cu:116).  freq_out is CALLER-allocated, nS * 4^k int32 (the reference
 * allocates rd->Freq itself and never frees it, src/kmer_main.cu:115).  1 <= k <= 15
 * (src/tipos.h:5).  Index arithmetic is exact integer for every k (the reference's float
 * accumulation, src/kmer_kernel.cu:38, is exact only for k <= 12). */
int cfrk_per_read_dense(cfrk_ctx *ctx, const int8_t *data, const int64_t *start,
                        const int32_t *length, int64_t nN, int64_t nS, int k, int flags,
                        int32_t *freq_out);

/* Same with every buffer already resident on the context's device; asynchronous on the
 * context stream. */
int cfrk_per_read_dense_device(cfrk_ctx *ctx, const int8_t *d_data, const int64_t *d_start,
                               const int32_t *d_length, int64_t nN, int64_t nS, int k, int flags,
                               int32_t *d_freq_out);

/* ---- global counting: sum over reads of the per-read rows, any 1 <= k <= 64 ------------ */

/* Semantics: the guarded ComputeFreq (src/kmer_kernel.cu:52-70) summed over all reads of all
 * cfrk_global_add calls since begin; a window counts iff its k codes are all valid, so no
 * window crosses a terminator.  Result = set of (key, count), count < 2^32.
 * capacity_hint = expected number of DISTINCT keys (0: library default).  It sizes the result list
 * and the HBM table, and the partitioned paths (k >= 16) read it as the job's shape: a hint above
 * ~2.7e8 makes them share every leaf between several workgroups (~2000 distinct k-mers each; one-
 * and two-word keys alike), and the room a chunked batch's leaf streams get beyond their measured
 * mean grows when the hint says that a leaf holds few distinct runs.  A job that holds far more
 * distinct k-mers than it announced still counts exactly, but splits overfull leaves by key. */
int cfrk_global_begin(cfrk_ctx *ctx, int k, int flags, uint64_t capacity_hint);

/* Host buffers (struct read fields).  start/length may be NULL; when given they are checked
 * against the terminators in data (CFRK_ERR_LAYOUT).  Stages through pinned memory, H2D on the
 * context stream, then counts; returns after the counting kernels are enqueued. */
int cfrk_global_add(cfrk_ctx *ctx, const int8_t *data, const int64_t *start,
                    const int32_t *length, int64_t nN, int64_t nS);

/* Device-resident data (16-byte aligned).  The kernels run on the context stream and the call does not wait for the
 * last of them (cfrk_global_finish / digest / export do), but it is NOT free of host synchronisation: the partitioned
 * paths (k <= 64 without CFRK_FORCE_HASH) read a few words back per add -- the region cursors after the second-level
 * kernel (did anything overflow?) and, for a batch large enough to be counted in chunks, the record count of the first
 * chunk, from which the leaf streams are sized -- each a small D2H copy + hipStreamSynchronize in the middle of the call.
 * A batch whose first chunk is no measure of the rest (sorted or clustered reads, an N-rich or short-read prefix) costs
 * one more partition pass over the input: its leaf streams are then laid out exactly, in a buffer that grows to fit. */
int cfrk_global_add_device(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);

/* Add pre-counted (key, count) pairs, e.g. another GPU's export received over RCCL.
 * d_keys_hi may be NULL for k <= 32. */
int cfrk_global_merge_device(cfrk_ctx *ctx, const uint64_t *d_keys_lo, const uint64_t *d_keys_hi,
                             const uint32_t *d_counts, int64_t n);

/* Wait for all adds; report the number of distinct keys.  CFRK_ERR_TABLE_FULL if the table
 * overflowed.  CFRK_ERR_COUNT_OVERFLOW if a count saturated: *n_distinct is written all the same
 * and the job can be digested / exported (the same code comes back from those calls, their outputs
 * filled, the saturated counts reading CFRK_COUNT_MAX). */
int cfrk_global_finish(cfrk_ctx *ctx, uint64_t *n_distinct);

/* Sorted ascending by (hi, lo).  keys_hi may be NULL for k <= 32.  cap = entries available.
 * CFRK_ERR_COUNT_OVERFLOW: the arrays are complete, at least one count reads CFRK_COUNT_MAX. */
int cfrk_global_export(cfrk_ctx *ctx, uint64_t *keys_lo, uint64_t *keys_hi, uint32_t *counts,
                       uint64_t cap, uint64_t *n_out);

/* Unsorted export into device buffers, grouped into `parts` contiguous segments by
 * owner(key) = (mix(key) >> 32) % parts (key-owner partition for the multi-GPU
 * merge).  part_counts (host, `parts` entries) receives the segment sizes.  Synchronises. */
int cfrk_global_export_device(cfrk_ctx *ctx, uint64_t *d_keys_lo, uint64_t *d_keys_hi,
                              uint32_t *d_counts, uint64_t cap, int parts, uint64_t *part_counts);

/* Multi-GPU exchange by LEAF (the partitioned path's unit of disjoint key space; same k => same
 * leaf on every rank): owner(leaf) = leaf % parts.  Export writes this rank's result grouped by
 * owner (part p = leaves p, p+parts, ... in that order), part_counts[p] entries each, and the
 * per-leaf entry counts in the same order to d_leaf_counts (parts * cfrk_global_leaves_per_part
 * uint32).  d_keys_hi receives / supplies the high key words for k > 32 and is NULL otherwise.
 * CFRK_ERR_STATE when the result is not in per-leaf list form (k < 16, something spilled to the
 * HBM table, several adds or passes): use cfrk_global_export_device instead.
 * Merge, on a context fresh from cfrk_global_begin: d_keys/d_counts = the received runs in rank
 * order (recv_counts[r] entries from rank r), d_leaf_counts = the received per-leaf counts
 * ([parts][leaves_per_part]); every leaf's lists are added in an LDS table (no HBM atomics). */
int cfrk_global_leaves_per_part(int parts);
int cfrk_global_export_leaves_device(cfrk_ctx *ctx, uint64_t *d_keys, uint64_t *d_keys_hi, uint32_t *d_counts,
                                     uint64_t cap, int parts, uint64_t *part_counts, uint32_t *d_leaf_counts);
int cfrk_global_merge_leaves_device(cfrk_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_keys_hi,
                                    const uint32_t *d_counts, const uint64_t *recv_counts,
                                    const uint32_t *d_leaf_counts, int parts);

/* Multi-GPU exchange at RUN granularity -- the strong-scaling form (replaces the pthread fan-out of
 * src/main.cu:277-295; the reference has no merge step).  With the reads of ONE job split over N
 * ranks every rank still meets almost every locus, so counting per rank and exchanging counted
 * k-mers makes every rank expand every distinct k-mer.  Instead a rank begins with CFRK_RUNS_ONLY,
 * adds its shard once (partition + deduplication only) and exports, per leaf, its distinct complete
 * runs with multiplicities followed by its truncated runs (read ends).  A truncated run that is a
 * prefix of one of the rank's own distinct complete runs is sent as a 16-bit NOTE (position of that
 * run in the leaf's list << 5 | k-mers - 1) instead of a 16-byte record.  The export is PACKED for
 * one all-to-all: 16-byte rows, one segment per owner (owner p = leaves p, p+parts, ...) = a header
 * (the owner's leaves_per_part x (distinct, truncated, noted) sizes as uint32 triples, padded to
 * whole rows) followed, leaf after leaf, by the distinct runs, the truncated runs and the notes
 * (eight per row); part_rows[p] = rows of segment p.  The owner, on a context fresh from
 * cfrk_global_begin (same k and strand flag, without CFRK_RUNS_ONLY), passes the received segments
 * in rank order (recv_rows[r] rows from rank r): the lists of each of its leaves become that leaf's
 * streams and are expanded and counted once.  Afterwards the owner holds the final counts of its
 * leaves' k-mers (finish / export / digest as usual).  parts <= 64.  CFRK_ERR_STATE from the export
 * when part of the shard was counted in the HBM table instead: use the leaf or key-owner exchange. */
int cfrk_global_export_runs_device(cfrk_ctx *ctx, void *d_packed, uint64_t cap_rows, int parts,
                                   uint64_t *part_rows);
int cfrk_global_merge_runs_device(cfrk_ctx *ctx, const void *d_packed, const uint64_t *recv_rows, int parts);

/* PIPELINED form of the exchange by runs (round 5; both key widths, a 32-byte record of k > 32 travels as two rows).  The owners' leaves are cut into `ngroups` ranges of
 * local leaf indices (group g = local leaves [lpp * g / ngroups, lpp * (g + 1) / ngroups), lpp = cfrk_global_leaves_per_part);
 * a rank deduplicates and packs group after group straight into the send buffer and hands every finished group to the
 * wire while the next one is in the works, the owner counts every received group while the next one is on the wire:
 *     rank:  begin(CFRK_RUNS_ONLY | CFRK_RUNS_DEFER), add_device (returns with the kernels enqueued),
 *            export_runs_async (enqueues everything, returns at once),
 *            for g: export_runs_wait(g) -> rows per owner -> all-to-all of group g (segment (g, p) to owner p)
 *     owner: begin (same k and strand flag), for g: merge_runs_group_device(g) (enqueued, no host synchronisation)
 * Send buffer: segment (g, p) = rows [(g * parts + p) * seg_cap_rows, ...), 16-byte rows; only its first part_rows[p]
 * rows travel.  Segment = header (row 0: {rows used, local leaves of the group, first local leaf, magic}; then per local
 * leaf a uint4 {row offset behind the header, distinct, truncated, noted}) + the leaves' rows in the order they were
 * claimed: [distinct complete runs with multiplicities][truncated runs][16-bit notes, eight per row].
 * export_runs_wait: waits for group g only (one event); CFRK_ERR_SMALL_BUF when a segment of the group ran out of room
 * (seg_cap_rows too small), CFRK_ERR_STATE when the add overflowed a region or spilled (the shard's runs are not all in
 * the leaf streams): in both cases nothing of the group may be sent -- add again without CFRK_RUNS_DEFER and take
 * cfrk_global_export_runs_device, or count the shard and exchange counts.  The leaf streams stay as the add left them,
 * so cfrk_global_export_runs_device may follow on the same job.
 * merge_runs_group_device: d_recv = the `parts` received segments of group g in rank order (recv_rows[r] rows from rank
 * r).  Groups must be merged in order, 0 .. ngroups-1, on a context fresh from cfrk_global_begin; the call returns with
 * the leaf kernel enqueued (it reads the lists in place: d_recv must stay untouched until the context is synchronised).
 * A segment whose header does not add up is not followed and makes finish / digest fail with CFRK_ERR_TABLE_FULL.
 * parts <= 64, ngroups <= 16.  Replaces the pthread fan-out of src/main.cu:277-295 (which has no merge step). */
int cfrk_global_export_runs_async(cfrk_ctx *ctx, void *d_packed, uint64_t seg_cap_rows, int parts, int ngroups);
int cfrk_global_export_runs_wait(cfrk_ctx *ctx, int group, uint64_t *part_rows);
int cfrk_global_merge_runs_group_device(cfrk_ctx *ctx, const void *d_recv, const uint64_t *recv_rows, int parts,
                                        int group, int ngroups);
/* Device time (ms, HIP events) from the start of the job's add to the end of group `group` of the pipelined export;
 * synchronises on that group.  (cfrk_global_last_add_ms after cfrk_global_merge_runs_group_device: the owner's kernels
 * from group 0 up to the last group merged.) */
int cfrk_global_runs_group_ms(cfrk_ctx *ctx, int group, float *ms);

/* Order-independent digest: out[0]=distinct, out[1]=sum count,
 * out[2]=sum count*splitmix64(kh) mod 2^64, out[3]=xor splitmix64(kh ^ count);
 * kh = lo (k<=32) or lo + splitmix64(hi). Synchronises.  CFRK_ERR_COUNT_OVERFLOW: out[] is
 * filled, computed over the saturated counts. */
int cfrk_global_digest(cfrk_ctx *ctx, uint64_t out[4]);

/* Device time (ms, HIP events on the context stream) of the counting kernels of the most
 * recent cfrk_global_add / cfrk_global_add_device; synchronises. */
int cfrk_global_last_add_ms(cfrk_ctx *ctx, float *ms);

/* Diagnostics of the minimizer-partitioned path after the most recent add (synchronises):
 * out[0..2] = level-1 records: total, largest bin, bin capacity; out[3..5] = level-2 records:
 * total, largest leaf, leaf capacity; out[6] = records and out[7] = k-mer insertions (each may
 * carry a multiplicity) that were counted in the HBM table instead (spill); out[8] = entries in
 * the leaf-output list. */
int cfrk_debug_msp_info(cfrk_ctx *ctx, uint64_t out[9]);

/* Cap (bytes, 0 = none) on the device memory the partitioned counting paths may use for their
 * record buffers.  A batch whose buffers exceed what is available is counted in several passes
 * over ranges of the input, each pass folded into the HBM table; this knob makes that
 * behaviour reachable with small inputs (tests) and lets a host that shares the GPU hold the
 * library to a budget.  out_passes (may be NULL) receives the passes of the most recent add. */
int cfrk_debug_set_mem_budget(cfrk_ctx *ctx, uint64_t bytes);
/* Device memory the context holds right now: its pool (record buffers, result list, staging) plus the
 * global table.  The caller's own buffers (the reads handed to cfrk_global_add_device) are not in it. */
int cfrk_debug_device_bytes(cfrk_ctx *ctx, uint64_t *out_bytes);
int cfrk_debug_last_add_passes(cfrk_ctx *ctx, int *out_passes);

/* Test switches for rarely taken device paths (0 = normal operation).  Bit 0: every leaf of the
 * one-word partitioned path (16 <= k <= 32) is treated as if its complete runs had overflowed the
 * record table, i.e. takes the second-chance deduplication over the whole LDS pool. */
#define CFRK_DEBUG_FORCE_RT_OVERFLOW 0x1
/* Bit 1: the first partition kernel (k >= 16) holds one trip's worth (64) of a wave's runs in
 * registers instead of 4..8: the rest takes the direct-append path meant for pathological waves. */
#define CFRK_DEBUG_SMALL_WAVE_CAP 0x2
/* Bit 2: the leaf kernel of the one-word partitioned path counts every truncated run k-mer by k-mer
 * instead of noting it with the complete run it is a prefix of (same result; for A/B timing and tests). */
#define CFRK_DEBUG_NO_ANCHORS 0x4
/* Bit 3: the two-word partitioned path (33 <= k <= 64) writes extra minimizer-hash bits into its
 * records and lets four workgroups share every leaf, each taking the records its bits name,
 * whatever the capacity hint (normally only for hints above ~2.7e8 distinct k-mers, 2..32
 * workgroups per leaf): makes that path reachable with small inputs (tests). */
#define CFRK_DEBUG_RECORD_SUBSETS 0x8
/* Bit 4: the partitioned paths (k >= 16) never count in chunks (a large batch is normally cut into
 * chunks of tiles -- partition kernel on chunk c, then second-level kernel on chunk c -- so that the
 * level-1 buffer holds one chunk and the leaf streams are sized from the first chunk's records); same
 * result, for A/B timing.  Bit 5: count in chunks of a few tiles whatever the batch size: makes the
 * chunked path reachable with small inputs (tests). */
#define CFRK_DEBUG_NO_PIPELINE 0x10
#define CFRK_DEBUG_SMALL_PIPELINE 0x20
/* Bit 6: k = 16 never takes the radix path (normally it does for adds of up to ~4e9 bases: the minimizer window of
 * the partitioned path is only four k-mers at k = 16); makes the partitioned path at k = 16 reachable with small
 * inputs (tests). */
#define CFRK_DEBUG_NO_RADIX16 0x40
/* Bit 7: the two-word leaf kernel keeps its 4096-slot k-mer table (one workgroup per CU) also for jobs that announce few
 * distinct k-mers per leaf, which normally take the 1024-slot instantiation (two per CU); keeps the large instantiation
 * reachable with small inputs (tests). */
#define CFRK_DEBUG_NO_SMALL_LEAVES 0x80
/* Any other bit is refused with CFRK_ERR_ARG.  (The timing ablations that skip a kernel phase and so produce WRONG
 * counts -- cfrk_amd/csrc/msp.h: CFRK_ABL_* -- are compiled into a separate ablation build only, `make -C
 * cfrk_amd/csrc abl`; the product library does not contain them.) */
int cfrk_debug_set_flags(cfrk_ctx *ctx, uint32_t flags);

/* Sizing knobs of the partitioned paths for experiments and tests (value 0 = the library's own choice;
 * out-of-range values are refused with CFRK_ERR_ARG).  The library reads NO environment variables. */
#define CFRK_PARAM_MSP_CHUNKS          0  /* 1 .. 4096: chunks a large batch is counted in (16 <= k <= 32)            */
#define CFRK_PARAM_L2_SLACK_COMPLETE   1  /* 1 .. 16: leaf-stream room for complete runs, x the measured mean share   */
#define CFRK_PARAM_L2_SLACK_TRUNCATED  2  /* 1 .. 16: the same for truncated runs                                     */
#define CFRK_PARAM_MSP2_SUBVALUE_BITS  3  /* 1 .. 3: log2(sub-values one workgroup of a shared leaf counts) + 1
                                             (33 <= k <= 64; normally chosen from the expected runs per leaf)         */
int cfrk_debug_set_param(cfrk_ctx *ctx, int which, double value);

/* ---- synthetic reads, generated on device ----------------------------------- */

/* Reads [r0, r0+R) of the deterministic generator, struct-read layout: d_data R*(L+1) bytes,
 * d_start R int64 (may be NULL), d_length R int32 (may be NULL). */
int cfrk_synth_reads_device(cfrk_ctx *ctx, int64_t r0, int64_t R, int L, int64_t Glen,
                            uint64_t seedG, uint64_t seedR, uint64_t seedS, int uniform,
                            int8_t *d_data, int64_t *d_start, int32_t *d_length);

#ifdef __cplusplus
}
#endif
#endif
