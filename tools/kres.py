#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a .hip source (cross-compiled to gfx950 assembly).
usage: python tools/kres.py cfrk_amd/csrc/msp.hip [name filter]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                           "--cuda-device-only", src, "-o", f.name], stderr=subprocess.DEVNULL)
    s = open(f.name).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue

    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"
    print(f"{name[:70]:70s} vgpr {g('next_free_vgpr'):>4} sgpr {g('next_free_sgpr'):>4} lds {g('group_segment_fixed_size'):>6} "
          f"scratch {g('private_segment_fixed_size'):>4}")
