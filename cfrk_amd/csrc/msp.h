// msp.h -- minimizer-partitioned counting fast path (msp.hip): host-side interface.
#pragma once
#include "common.h"

struct ResultSrc;
struct TableView;

// Timing ablations: the kernel skips one of its phases, so the counts are WRONG -- only for measuring what a
// phase costs.  They exist only in a library built with -DCFRK_ABLATIONS (`make -C cfrk_amd/csrc abl` ->
// tools/_bin/libcfrk_hip_abl.so, which tools/ablate.sh swaps in); in the product build every bit is 0, the
// branches fold away, and cfrk_debug_set_flags refuses the bits with CFRK_ERR_ARG.  The P3 flags act on the
// leaf kernels of both msp.hip and msp2.hip.
#ifdef CFRK_ABLATIONS
#define CFRK_ABL_BIT(x) (x)
#else
#define CFRK_ABL_BIT(x) 0u
#endif
#define CFRK_ABL_P3_NO_TRUNC   CFRK_ABL_BIT(0x100u)     // leaf kernel: truncated runs are not expanded
#define CFRK_ABL_P3_NO_CEXP    CFRK_ABL_BIT(0x200u)     // leaf kernel: distinct complete runs are not expanded
#define CFRK_ABL_P3_NO_RTAB    CFRK_ABL_BIT(0x400u)     // leaf kernel: complete runs are not read / deduplicated
#define CFRK_ABL_P3_NO_OUT     CFRK_ABL_BIT(0x800u)     // leaf kernel: no compaction to the result list
#define CFRK_ABL_P1_NO_EMIT    CFRK_ABL_BIT(0x1000u)    // partition kernel: front end only, no records built
#define CFRK_ABL_P2_NO_ATOMIC  CFRK_ABL_BIT(0x2000u)    // second-level kernel: no cursor atomics (every tile's segments land at the stream starts)
#define CFRK_ABL_P2_LINEAR_OUT CFRK_ABL_BIT(0x4000u)    // second-level kernel: sorted tiles written back to back (what the scatter into 512 streams per bin costs)
#define CFRK_ABL_RX3_NO_CURSOR CFRK_ABL_BIT(0x10000u)   // radix leaf kernel: no atomic on the result cursor (workgroup b writes from entry b * 16384)
#define CFRK_ABL_RX3_NO_COUNT  CFRK_ABL_BIT(0x20000u)   // radix leaf kernel: keys are loaded but not counted
#define CFRK_ABL_RX2_NO_OUT    CFRK_ABL_BIT(0x40000u)   // radix second-level kernel: the sorted tile is not written out
#define CFRK_ABL_RX1_NO_OUT    CFRK_ABL_BIT(0x80000u)   // radix first-level kernel: the sorted tile is not written out
#define CFRK_ABL_RX1_LINEAR    CFRK_ABL_BIT(0x100000u)  // radix first-level kernel: the sorted tile goes out back to back (no scatter into regions)
// every bit cfrk_debug_set_flags accepts: the documented test switches of include/cfrk_abi.h, plus the ablations of an ablation build
#ifdef CFRK_ABLATIONS
#define CFRK_DEBUG_KNOWN_BITS (0xFFu | 0x7F00u | 0x1F0000u)
#else
#define CFRK_DEBUG_KNOWN_BITS 0xFFu
#endif

bool cfrk_msp_usable(const cfrk_ctx *ctx);                 // fast path applies to this begin()?
int  cfrk_msp_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);
// Where does the result live?  *use_list = true and *src filled when it is the compact list the
// leaf kernels wrote (nothing spilled, table untouched); otherwise any pending list has been
// folded into the table and the caller reads the table.  Synchronises.
int  cfrk_msp_resolve(cfrk_ctx *ctx, ResultSrc *src, bool *use_list);
int  cfrk_msp_flush_to_table(cfrk_ctx *ctx);               // fold a pending list into the table
void cfrk_msp_note_table_write(cfrk_ctx *ctx);             // table now holds counts of its own
bool cfrk_msp_table_written(const cfrk_ctx *ctx);          // ... since the last begin()?
void cfrk_msp_reset(cfrk_ctx *ctx);
void cfrk_msp_destroy(cfrk_ctx *ctx);

TableView cfrk_table_view(const cfrk_ctx *ctx);

// device-side view of the partitioned path's buffers (msp.hip) and of the result list that
// msp.hip and radix.hip both produce
struct MspView {
  uint4 *rec1; uint32_t *cnt1; uint64_t cap1;   // B1 x nxg sub-regions of cap1 records each
  uint32_t nxg;                                 // sub-regions (cursors) per level-1 bin: 8 or NXG
  uint4 *rec2; uint32_t *cnt2; uint64_t cap2c, cap2t;   // per leaf: complete stream (cap2c), then the truncated stream (cap2t)
  uint64_t *out_keys; uint32_t *out_cnt; uint64_t out_cap;
  uint64_t *out_hi;                           // high key words of the list (two-word keys only)
  uint64_t *leaf_off; uint32_t *leaf_n;        // where each leaf's entries sit in the result list
  uint32_t seg_bits;                             // ... in 2^seg_bits segments, entry (leaf << seg_bits) | j (leaves shared by record, msp2.hip)
  uint32_t sub_bits;                             // msp.hip: records carry so many sub-value bits in the header's top byte (0: none)
  // exact layout (after a leaf stream overflowed the fixed-stride layout): stream (leaf, class)
  // starts at record lbase[NCLS * leaf + class] and holds exactly lcap[...] records
  const uint64_t *lbase; const uint32_t *lcap; uint32_t exact;
  uint4 *ovf; uint32_t ovf_cap;                // parking for records that overflow a leaf stream
  // the same one level up: exact level-1 layout (region r starts at record rbase[r], holds rcap[r])
  const uint64_t *rbase; const uint32_t *rcap; uint32_t exact1;
  uint32_t dbg;                                  // cfrk_debug_set_flags
  // leaf subset of this pass (a batch whose records do not fit device memory is counted in several
  // passes over the input, each emitting only the runs of the leaves with (leaf & sel_mask) == sel_val)
  uint32_t sel_mask, sel_val, sel_bits;          // (the pass's leaves are laid out densely: stream index = leaf >> sel_bits)
  uint4 *ovf1; uint32_t ovf1_cap;
  uint64_t *stats;
};

struct cfrk_msp {
  bool leaf_form;      // the list is grouped by minimizer leaf (msp.hip); false for radix.hip
  bool runs_ready;     // CFRK_RUNS_ONLY job: the leaf streams hold the shard's runs, ready for cfrk_global_export_runs_*
  bool runs_deduped;   // ... deduplicated in place already (msp_dedupe_export_kernel); false after a CFRK_RUNS_DEFER add
  bool runs_unchecked; // ... and nobody has looked at the add's overflow flags yet (CFRK_RUNS_DEFER: no host sync after P2)
  int  lists_group;    // owner of the pipelined exchange: groups merged so far (cfrk_global_merge_runs_group_device)
  bool pending;        // a leaf-output list exists that has not been folded into the table
  bool table_dirty;    // the table holds counts of its own since begin()
  // expected records = positions x density x dens_scale: 1 unless a batch did not fit and its invalid
  // bytes were counted (reads much shorter than 8 k make far fewer records than positions suggest)
  double dens_scale;
  uint64_t list_n;     // entries in the list (valid after resolve)
  bool list_n_valid;
  MspView view;
  alignas(16) unsigned char view2[256];   // msp2.hip: the View2 of a CFRK_RUNS_ONLY job (its leaf streams hold the runs)
};

cfrk_msp *cfrk_msp_get(cfrk_ctx *ctx);
int cfrk_msp_sync_stats(cfrk_ctx *ctx, uint64_t st[ST_NWORDS]);

// passes needed so that one pass's buffers (need_fn(span)) fit device memory; *groups = 0: none does
// (acc_bytes = bytes a pass's result list adds to the accumulation buffers of a multi-pass add)
int cfrk_msp_plan_groups(cfrk_ctx *ctx, int64_t nN, int64_t ntiles, int64_t tile_span,
                         size_t (*need_fn)(const cfrk_ctx *, int64_t), size_t acc_bytes, size_t have,
                         int *groups);

// msp2.hip: 33 <= k <= 64
bool cfrk_msp2_usable(const cfrk_ctx *ctx);
int  cfrk_msp2_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);

// msp2.hip: add the `parts` lists of every leaf (ll < leaves_per_part) in LDS into the result list
int  cfrk_msp2_merge_lists(cfrk_ctx *ctx, const uint64_t *d_lo, const uint64_t *d_hi, const uint32_t *d_cnt,
                           const uint64_t *d_seg_off, const uint32_t *d_seg_n, int parts, int leaves_per_part);

// msp2.hip: the exchange by runs (cfrk_global_export_runs_device / cfrk_global_merge_runs_device) for two-word keys
int  cfrk_msp2_export_runs(cfrk_ctx *ctx, void *d_packed, uint64_t cap_rows, int parts, uint64_t *part_rows);
int  cfrk_msp2_merge_runs(cfrk_ctx *ctx, const void *d_packed, const uint64_t *recv_rows, int parts);
int  cfrk_msp2_export_runs_async(cfrk_ctx *ctx, void *d_packed, uint64_t seg_cap_rows, int parts, int ngroups);
int  cfrk_msp2_merge_runs_group(cfrk_ctx *ctx, const void *d_recv, const uint64_t *recv_rows, int parts, int group, int ngroups);

// radix.hip: k <= 15
bool cfrk_radix_usable(const cfrk_ctx *ctx);
bool cfrk_radix_prefers(const cfrk_ctx *ctx, int64_t nN);   // ... or k = 16 and a batch small enough for its leaves (radix.hip)
int  cfrk_radix_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);

