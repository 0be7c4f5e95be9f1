"""cfrk_amd -- MI355X-native k-mer counting behind the reference's kmer_main() boundary.

The compute lives in libcfrk_hip.so (hand-written HIP for gfx950, C ABI in
include/cfrk_abi.h); this package is the thin host-side mirror used by tests, bench.py and
Python callers.  There is NO CPU fallback: if the HIP library is missing or no gfx950 device
is present, every compute entry point raises.
"""
from .lib import (CFRK_CANONICAL, CFRK_COMPAT, CFRK_COUNT_MAX, CFRK_ERR_COUNT_OVERFLOW, CFRK_ERR_RUNS_REFUSED, CFRK_DEBUG_FORCE_RT_OVERFLOW, CFRK_DEBUG_NO_PIPELINE, CFRK_DEBUG_NO_RADIX16, CFRK_DEBUG_SMALL_PIPELINE, CFRK_DEBUG_SMALL_WAVE_CAP, CFRK_FLOAT_INDEX, CFRK_FORCE_HASH, CFRK_RUNS_DEFER, CFRK_RUNS_ONLY, CfrkError, Context, GlobalCounter, Read,
                  abi_symbols, device_count, kmer_main, load_library, library_path)

__all__ = ["CFRK_CANONICAL", "CFRK_COMPAT", "CFRK_COUNT_MAX", "CFRK_ERR_COUNT_OVERFLOW", "CFRK_ERR_RUNS_REFUSED", "CFRK_DEBUG_FORCE_RT_OVERFLOW", "CFRK_DEBUG_NO_PIPELINE", "CFRK_DEBUG_NO_RADIX16", "CFRK_DEBUG_SMALL_PIPELINE", "CFRK_DEBUG_SMALL_WAVE_CAP", "CFRK_FLOAT_INDEX", "CFRK_FORCE_HASH", "CFRK_RUNS_DEFER", "CFRK_RUNS_ONLY", "CfrkError", "Context", "GlobalCounter", "Read",
           "abi_symbols", "device_count", "kmer_main", "load_library", "library_path"]
