// src/kmer_main.cpp -- kmer_main() on MI355X through the cfrk C ABI
#include <stdio.h>
#include <stdlib.h>
#include "tipos.h"          // struct read, lint, ushort (src/tipos.h:5-30)
#include "cfrk_abi.h"

static thread_local cfrk_ctx *g_ctx = NULL;     // one context per calling pthread (src/main.cu:281-284)
static thread_local int g_dev = -1;

void kmer_main(struct read *rd, lint nN, lint nS, int k, ushort device)
{
   if (!g_ctx || g_dev != device) {
      if (g_ctx) cfrk_ctx_destroy(g_ctx);
      int rc = cfrk_ctx_create(device, NULL, &g_ctx);
      if (rc) { printf("\n[Error] %s\n", cfrk_strerror(rc)); exit(1); }   // src/kmer_main.cu:51-56
      g_dev = device;
   }
   // the reference allocates rd->Freq with cudaMallocHost and never frees it (src/kmer_main.cu:115)
   rd->Freq = (int *)malloc(sizeof(int) * (size_t)nS * POW(k));
   int rc = cfrk_per_read_dense(g_ctx, (const int8_t *)rd->data, (const int64_t *)rd->start,
                                (const int32_t *)rd->length, nN, nS, k, CFRK_COMPAT, rd->Freq);
   if (rc) printf("\n[Error %d] %s: %s\n", -rc, cfrk_strerror(rc), cfrk_last_error(g_ctx));  // src/kmer_main.cu:59-63
}
