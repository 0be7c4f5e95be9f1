// msp.h -- minimizer-partitioned counting fast path (msp.hip): host-side interface.
#pragma once
#include "common.h"

bool cfrk_msp_usable(const cfrk_ctx *ctx);                 // fast path applies to this begin()?
int  cfrk_msp_count(cfrk_ctx *ctx, const int8_t *d_data, int64_t nN);
int  cfrk_msp_flush_to_table(cfrk_ctx *ctx);               // fold pending leaf results into the table
void cfrk_msp_reset(cfrk_ctx *ctx);
void cfrk_msp_destroy(cfrk_ctx *ctx);
