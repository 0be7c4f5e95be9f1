// msp.h -- minimizer-partitioned counting fast path (msp.hip): host-side interface.
#pragma once
#include "common.h"

struct ResultSrc;
struct TableView;

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
