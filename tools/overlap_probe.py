#!/usr/bin/env python3
"""Do two independent counting jobs on ONE GPU overlap when they run on two HIP streams?

Two contexts (own stream, own pool), each with R reads resident; a step of one job alone against
the two jobs started together from two host threads.  If the VALU-bound partition kernel of one job
hides under the HBM-bound second-level kernel of the other, two concurrent steps take clearly less
than twice one step.  usage: python tools/overlap_probe.py [reads [k]]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cfrk_amd  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    L = 150
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs = [cfrk_amd.Context(0), cfrk_amd.Context(0)]
    nN = R * (L + 1)
    bufs = []
    for i, c in enumerate(ctxs):
        d = torch.empty(nN + 64, dtype=torch.int8, device=dev)
        c.synth_reads_device(i * R, R, L, R, d.data_ptr())
        bufs.append(d)
    torch.cuda.synchronize()
    hint = R + 1024

    def step(i):
        g = cfrk_amd.GlobalCounter(ctxs[i], k, cfrk_amd.CFRK_CANONICAL, hint)
        g.add_device(bufs[i].data_ptr(), nN)
        ctxs[i].sync()
        return g

    for i in (0, 1):
        step(i)
    t0 = time.perf_counter()
    for _ in range(3):
        g = step(0)
    alone = (time.perf_counter() - t0) / 3
    print(f"one job alone: {alone * 1e3:.2f} ms per step (kernels {g.last_add_ms():.2f} ms)")
    t0 = time.perf_counter()
    for _ in range(3):
        step(0)
        step(1)
    seq = (time.perf_counter() - t0) / 3
    print(f"two jobs one after the other: {seq * 1e3:.2f} ms")
    for stagger_ms in (0.0, 0.35 * alone * 1e3):
        t0 = time.perf_counter()
        for _ in range(3):
            def second():
                if stagger_ms:
                    time.sleep(stagger_ms / 1e3)
                step(1)
            th = threading.Thread(target=second)
            th.start()
            step(0)
            th.join()
        both = (time.perf_counter() - t0) / 3
        print(f"two jobs together (second starts {stagger_ms:.1f} ms later): {both * 1e3:.2f} ms "
              f"= {both / seq:.2f} x sequential")


if __name__ == "__main__":
    main()
