#!/usr/bin/env python3
"""Diagnostic: can two ranks on ONE GPU run an RCCL all-reduce (backend "nccl")?  usage: torchrun --nproc-per-node 2 tools/rccl_probe.py"""
import os, sys, datetime
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=40), device_id=torch.device("cuda:0"))
    t = torch.full((1024,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over RCCL ok, value {float(t[0])}", flush=True)
    dist.destroy_process_group()
except Exception as exc:   # noqa: BLE001
    print(f"rank {rank}: RCCL on one GPU failed: {repr(exc)[:300]}", flush=True)
    sys.exit(3)
