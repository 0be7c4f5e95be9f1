#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (host library, oracle) with the CPU test suites that load
# them through ctypes.  GPU sanitizers are not available on this pool; the HIP side is covered by parity tests
# and tools/fuzz_paths.py.  usage (repo root, no GPU needed): tools/asan_cpu.sh
set -e
cd "$(dirname "$0")/.."
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
cp cfrk_amd/libcfrk_host.so /tmp/libcfrk_host.so.keep; cp oracle/liboracle.so /tmp/liboracle.so.keep
restore() { cp /tmp/libcfrk_host.so.keep cfrk_amd/libcfrk_host.so; cp /tmp/liboracle.so.keep oracle/liboracle.so; touch cfrk_amd/libcfrk_host.so oracle/liboracle.so; }
trap restore EXIT
g++ $SAN -std=c++17 -fPIC -Wall -Wextra -pthread -shared -o cfrk_amd/libcfrk_host.so cfrk_amd/host/cfrk_host.cpp
gcc $SAN -fPIC -Wall -Wextra -std=gnu11 -shared -o oracle/liboracle.so oracle/cfrk_oracle.c -lm -lpthread
touch cfrk_amd/libcfrk_host.so oracle/liboracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$PRE" python -m pytest tests/test_host_cpu.py tests/test_oracle.py -x -q
