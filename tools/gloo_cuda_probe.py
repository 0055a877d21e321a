#!/usr/bin/env python3
"""Diagnostic: does the gloo backend all-reduce a GPU tensor on this build?  torchrun --nproc-per-node 2 tools/gloo_cuda_probe.py"""
import os, sys, datetime
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=40))
try:
    t = torch.full((1024,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: gloo all_reduce of a GPU tensor ok, value {float(t[0])}", flush=True)
except Exception as exc:   # noqa: BLE001
    print(f"rank {rank}: gloo on a GPU tensor failed: {repr(exc)[:200]}", flush=True)
    sys.exit(3)
