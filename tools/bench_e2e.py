#!/usr/bin/env python3
"""End to end through the PRODUCT: FASTA file -> `cfrk` (parse, H2D, count, export, format, write) -> .cfrk file.

What SURVEY 8(d) "What is timed" asks beside the headline and what the reference times with its (commented-out)
printf()s around main() (/root/reference/src/main.cu:259-268,303-305).  Never the bench `value`: the headline counts
device-resident reads; this is the whole command on a file.

  python tools/bench_e2e.py [--reads 10000000] [--L 150] [--k 15] [--dir /tmp] [--keep]

Writes a configs[1]-sized FASTA (10 M x 150 bp from the SURVEY 8d generator, 1.6 GB) and runs
  cfrk in.fasta out.cfrk k T --global --canonical --timing                (sparse text)
  ... --binary                                                            (CFRKGLB1)
  ... --parse-threads 16 / 8                                              (the parser by thread count; default min(hw, 64))
  cfrk in.fasta out4.cfrk 4 T 8192 --timing                               (the reference's own mode: compat, k = 4)
  cfrk first-1M-reads.fasta out4all.cfrk 4 T 8192 --all-chunks --timing   (every chunk written)
and prints one JSON object: per run the phases `cfrk --timing` reports, as seconds and GB/s.
`measure(reads, ...)` is what bench.py calls for the `end_to_end` object of its default line.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFRK = os.path.join(ROOT, "cfrk_amd", "cfrk")


def write_fasta(path, R, L, glen, ctx=None, r0=0):
    """reads [r0, r0 + R) of the generator as `>rNNNNNNNNN\\n<bases>\\n` records (fixed width: numpy only)"""
    import numpy as np
    import cfrk_amd
    own = ctx is None
    if own:
        ctx = cfrk_amd.Context(0)
    nN = R * (L + 1)
    d = ctx.alloc(nN + 64)
    ctx.synth_reads_device(r0, R, L, glen, d)
    codes = np.empty(nN, np.int8)
    ctx.d2h(codes, d)
    ctx.free(d)
    if own:
        ctx.close()
    W = 12 + L + 1                                           # ">r" + 9 digits + "\n" + L bases + "\n"
    rec = np.empty((R, W), np.uint8)
    rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 11] = ord("\n"); rec[:, W - 1] = ord("\n")
    idx = np.arange(r0, r0 + R, dtype=np.int64)
    for j in range(9):
        rec[:, 10 - j] = (idx % 10 + ord("0")).astype(np.uint8)
        idx //= 10
    lut = np.frombuffer(b"ACGT", np.uint8)
    rec[:, 12:12 + L] = lut[codes.reshape(R, L + 1)[:, :L]]
    with open(path, "wb") as f:
        f.write(memoryview(rec).cast("B"))
    return R * W


def run_cfrk(args, timeout=900):
    t0 = time.perf_counter()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "cfrk_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([CFRK] + args + ["--timing"], capture_output=True, text=True, timeout=timeout, env=env)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": f"cfrk exit {r.returncode}: {r.stderr[-400:]}"}
    t = None
    for ln in r.stderr.splitlines():
        if ln.startswith("cfrk-timing "):
            t = json.loads(ln[len("cfrk-timing "):])
    if t is None:
        return {"error": "no cfrk-timing line", "stderr": r.stderr[-400:]}
    out = {"argv": " ".join(["cfrk"] + [os.path.basename(a) if os.sep in a else a for a in args]),
           "process_wall_s": round(wall, 3), **t}
    gb = 1e9
    if t["parse_s"] > 0:
        out["parse_GBps"] = round(t["fasta_bytes"] / t["parse_s"] / gb, 3)
    if t["add_call_s"] > 0:
        out["h2d_GBps"] = round(t["code_bytes"] / t["add_call_s"] / gb, 2)       # (the add call = the H2D copy of the codes, pageable memory)
    fw = t["format_s"] + t["write_s"]
    if fw > 0 and t["out_bytes"]:
        out["format_write_GBps"] = round(t["out_bytes"] / fw / gb, 3)
    return out


def measure(reads=10_000_000, L=150, k=15, glen=0, tmpdir=None, threads=0, ctx=None, quick=False):
    """-> dict for bench.py's `end_to_end` / this tool's output.  quick: the global sparse-text run only."""
    tmpdir = tmpdir or os.environ.get("TMPDIR") or "/tmp"
    threads = threads or min(64, os.cpu_count() or 1)
    glen = glen or reads
    fa = os.path.join(tmpdir, f"cfrk_e2e_{os.getpid()}.fasta")
    outp = os.path.join(tmpdir, f"cfrk_e2e_{os.getpid()}.cfrk")
    res = {"what": "FASTA file -> cfrk CLI -> .cfrk file (parse + H2D + count + export + format + write); never the bench value",
           "reads": reads, "read_len": L, "genome": glen, "host_threads_for_formatting": threads}
    try:
        t0 = time.perf_counter()
        nbytes = write_fasta(fa, reads, L, glen, ctx)
        res["fasta_bytes"] = nbytes
        res["fasta_written_in_s"] = round(time.perf_counter() - t0, 2)
        T = str(threads)
        res["global_text"] = run_cfrk([fa, outp, str(k), T, "--global", "--canonical"])
        if not quick:
            res["global_binary"] = run_cfrk([fa, outp, str(k), T, "--global", "--canonical", "--binary"])
            res["global_text_parse16"] = run_cfrk([fa, outp, str(k), T, "--global", "--canonical", "--parse-threads", "16"])
            res["global_text_parse8"] = run_cfrk([fa, outp, str(k), T, "--global", "--canonical", "--parse-threads", "8"])
            res["global_text_k31"] = run_cfrk([fa, outp, "31", T, "--global", "--canonical"])
            res["compat_k4_as_the_reference_writes_it"] = run_cfrk([fa, outp, "4", T, "8192"])
            small = os.path.join(tmpdir, f"cfrk_e2e_{os.getpid()}_1m.fasta")
            n1 = min(reads, 1_000_000)
            with open(fa, "rb") as f, open(small, "wb") as g:
                g.write(f.read(n1 * (12 + L + 1)))
            res["compat_k4_all_chunks_first_1m_reads"] = run_cfrk([small, outp, "4", T, "8192", "--all-chunks"])
            os.remove(small)
    except Exception as e:            # noqa: BLE001 -- reported, never fatal for the bench line
        res["error"] = f"{type(e).__name__}: {e}"
    finally:
        for p in (fa, outp):
            try:
                os.remove(p)
            except OSError:
                pass
    return res


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--L", type=int, default=150)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    print(json.dumps(measure(a.reads, a.L, a.k, 0, a.dir, a.threads, quick=a.quick), indent=1))
