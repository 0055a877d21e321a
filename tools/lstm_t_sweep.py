#!/usr/bin/env python3
"""Diagnostic: lstm_seq_fwd / lstm_seq_bwd at T = 1..32 (G = 3, B = 1024), to be run under
`rocprofv3 --kernel-trace --output-format csv`; tools/lstm_t_sweep_report.py then reads the per-dispatch durations."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import _learn_native as ln
G, B, H = 3, 1024, 128
bf = torch.bfloat16
w = (0.1 * torch.randn(G, 4 * H, H, device="cuda")).to(bf)
bias = torch.zeros(G, 4 * H, device="cuda", dtype=bf)
h0 = torch.zeros(G, B, H, device="cuda", dtype=bf)
for T in (1, 2, 4, 8, 16, 32):
    xp = torch.randn(G, T, B, 4 * H, device="cuda").to(bf)
    keep = torch.ones(T, B, device="cuda")
    for _ in range(10):
        out, hT, cT, (h_in, acts, cell) = ln.seq_forward(xp, w, bias, h0, h0, keep, True)
        ln.seq_backward(torch.ones_like(out), None, None, w, keep, acts, cell, (G, T, B), False, True)
    torch.cuda.synchronize()
