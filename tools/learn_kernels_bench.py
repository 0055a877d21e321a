#!/usr/bin/env python3
"""Diagnostic: the learner's HIP kernels (libcat_learn.so) one by one at the shapes of a 4096-env PPO minibatch step
(G = 3 stacked agents, 16-tick windows, 1024 sequences), timed with HIP events, beside their algorithmic HBM bytes and
MFMA flops.  The time of a call includes the host side of the binding (~20 us: allocation of the outputs, ctypes), so
kernels shorter than that read high here; inside the replayed HIP graph only the kernel counts (profiles/r02_learner_*).
Usage: python tools/learn_kernels_bench.py [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from as_cops_and_thieves_amd import _learn_native as ln

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev, bf = "cuda", torch.bfloat16
G, T, B, H, R = 3, 16, 1024, 128, 64
N = T * B


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


def line(name, us, mbytes, gflop):
    print(f"{name:34s} {us:8.1f} us   {mbytes:7.1f} MB -> {mbytes / us:5.2f} TB/s of 8   "
          f"{gflop:7.1f} GFLOP -> {gflop / us * 1e3:6.1f} TFLOP/s of 2500 (bf16 MFMA)")


# ---- LSTM window
xp = torch.randn(G, T, B, 4 * H, device=dev).to(bf)
w_hh = (0.1 * torch.randn(G, 4 * H, H, device=dev)).to(bf)
bias = torch.randn(G, 4 * H, device=dev).to(bf)
h0 = torch.zeros(G, B, H, device=dev, dtype=bf); c0 = torch.zeros_like(h0)
keep = torch.ones(T, B, device=dev)
out, hT, cT, (h_in, acts, cell) = ln.seq_forward(xp, w_hh, bias, h0, c0, keep, True)
d_out = torch.randn_like(out)
flop = 2 * G * T * B * 4 * H * H * 1e-9
mb_f = (xp.numel() + out.numel() + h_in.numel()) * 2e-6 + (acts.numel() + cell.numel()) * 1e-6
line("lstm_seq_fwd T=16 (training)", timed(lambda: ln.seq_forward(xp, w_hh, bias, h0, c0, keep, True)), mb_f, flop)
line("lstm_seq_bwd T=16", timed(lambda: ln.seq_backward(d_out, None, None, w_hh, keep, acts, cell, (G, T, B), False, True)),
     (d_out.numel() + xp.numel()) * 2e-6 + (acts.numel() + cell.numel()) * 1e-6, flop)
x1 = torch.randn(G, 1, 4096, 4 * H, device=dev).to(bf)
h1 = torch.zeros(G, 4096, H, device=dev, dtype=bf)
k1 = torch.ones(1, 4096, device=dev)
line("lstm_seq_fwd T=1 (rollout tick)", timed(lambda: ln.seq_forward(x1, w_hh, bias, h1, h1, k1, False)),
     (x1.numel() + 3 * h1.numel()) * 2e-6, 2 * G * 4096 * 4 * H * H * 1e-9)

# ---- convolutional trunk
for C in (2, 4):
    x = torch.rand(G, N, C * R, device=dev).to(bf)
    w1 = (0.3 * torch.randn(G, 64, C, 5, device=dev)).to(bf); b1 = torch.zeros(G, 64, device=dev, dtype=bf)
    w2 = (0.05 * torch.randn(G, 32, 64, 5, device=dev)).to(bf); b2 = torch.zeros(G, 32, device=dev, dtype=bf)
    y = ln.trunk_forward(x, w1, b1, w2, b2, R)
    dy = torch.randn_like(y)
    tiles = G * N / 16
    line(f"trunk_fwd C={C}", timed(lambda: ln.trunk_forward(x, w1, b1, w2, b2, R)), (x.numel() + y.numel()) * 2e-6, tiles * 300 * 16384 * 1e-9)
    line(f"trunk_bwd C={C}", timed(lambda: ln.trunk_backward(x, w1, b1, w2, b2, y, dy, R)), (x.numel() + 2 * y.numel()) * 2e-6,
         tiles * 690 * 16384 * 1e-9)

# ---- dense epilogues, loss, optimiser
for out_f, act in ((512, 0), (256, 2), (128, 1)):
    yy = torch.randn(G, N, out_f, device=dev).to(bf); bb = torch.randn(G, out_f, device=dev).to(bf); dd = torch.randn_like(yy)
    line(f"bias_act out={out_f}", timed(lambda: ln.dense_bias_act_(yy, bb, act)), 2 * yy.numel() * 2e-6, 0)
    line(f"act_grad out={out_f}", timed(lambda: ln.dense_act_grad(dd, yy, act)), (3 if act else 1) * yy.numel() * 2e-6, 0)
logits = torch.randn(G, T, B, 4, device=dev).to(bf); values = torch.randn(G, T, B, 1, device=dev).to(bf)
act_i = torch.randint(0, 4, (G, T, B), device=dev); f = torch.randn(G, T, B, device=dev)
line("ppo_loss_grad", timed(lambda: ln.ppo_loss_grad(logits, values, act_i, f, f, f, 0.15, 0.5, 0.02)), G * N * (8 + 2 + 8 + 12 + 10) * 1e-6, 0)
P = 797200
z = lambda *s: torch.zeros(*s, device=dev)
ar, col, ea, m, v, st, ma = torch.randn(G, P + 1, device=dev), torch.ones(G, P, device=dev), torch.ones(G, device=dev), z(G, P), z(G, P), z(G, P), torch.randn(G, P, device=dev)
lp = torch.zeros(G, P, device=dev, dtype=bf); kl, scr = z(G), z(G, 256)
line("adam step (2 launches)", timed(lambda: ln.ppo_adam_step(ar, col, ea, m, v, st, ma, lp, kl, scr, 1e-4, 0.9, 0.999, 1e-8, 0.5, 0.015)),
     G * P * (4 * 3 + 4 * 8 + 2) * 1e-6, 0)
