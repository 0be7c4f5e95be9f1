// msp.hip -- minimizer-partitioned counting (placeholder until the fast path lands).
#include "msp.h"

bool cfrk_msp_usable(const cfrk_ctx *) { return false; }
int cfrk_msp_count(cfrk_ctx *ctx, const int8_t *, int64_t) { return cfrk_fail(ctx, CFRK_ERR_STATE, "msp path not built"); }
int cfrk_msp_flush_to_table(cfrk_ctx *) { return CFRK_OK; }
void cfrk_msp_reset(cfrk_ctx *) {}
void cfrk_msp_destroy(cfrk_ctx *) {}
