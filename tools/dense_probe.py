#!/usr/bin/env python3
"""Diagnostic: the layer kernels (cat_dense_forward / dgrad / wgrad) at the shapes of a minibatch step, HIP-event timed,
with the bytes they move.  Usage: python tools/dense_probe.py [rows per network]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import _learn_native as ln

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
G, bf, dev = 3, torch.bfloat16, "cuda"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, N in ((256, 512), (288, 256), (128, 512), (128, 256), (256, 128), (128, 128), (128, 64)):
    x = torch.randn(G, M, K, device=dev).to(bf)
    w = (0.05 * torch.randn(G, N, K, device=dev)).to(bf)
    b = torch.zeros(G, N, device=dev, dtype=bf)
    g = torch.randn(G, M, N, device=dev).to(bf)
    tf = timed(lambda: ln.dense_forward(x, w, b, 1))
    td = timed(lambda: ln.dense_dgrad(g, w))
    rd, wr = G * M * K * 2e-6, G * M * N * 2e-6
    print(f"M={M} K={K:3d} N={N:3d}: forward {tf:7.1f} us (reads {rd:6.1f} MB, writes {wr:6.1f} MB -> {(rd + wr) / tf:5.2f} TB/s, writes {wr / tf:5.2f})   "
          f"dgrad {td:7.1f} us (reads {wr:6.1f}, writes {rd:6.1f} -> {(rd + wr) / td:5.2f} TB/s, writes {rd / td:5.2f})", flush=True)
