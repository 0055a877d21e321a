#!/usr/bin/env python3
"""rocprofv3 kernel traces of tools/learner_trace.py -> where the learner's GPU time is: the collect tick split into env / packing / networks / sampling,
the update by kernel, and the MFMA roofline of the update (network FLOPs from the layer sizes, bench.learner_flops, over the kernels' summed time).
usage: learner_split.py LABEL COLLECT_TRACE.csv FULL_TRACE.csv horizon rays rounds_timed warm_rounds"""
import csv, re, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
label, f_collect, f_full = sys.argv[1:4]
H, R, K, W = (int(x) for x in sys.argv[4:8])
N, A = 4096, 3


def load(path):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::|c10::", "", r["Kernel_Name"])
        name = ("GEMM " + name[:24]) if name.startswith("Cijk") else re.split(r"[<(]", name)[0]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[name][0] += 1; acc[name][1] += d
    return acc


def cat(name):
    if name.startswith(("step_kernel", "rollout_kernel", "reset_kernel", "tick_kernel")): return "env tick"
    if name.startswith(("pack_kernel", "post_kernel")): return "packing / post"
    if name.startswith("sample_kernel") or "distribution" in name or "uniform" in name: return "sampling"
    if name.startswith(("trunk_", "lstm_", "dense_", "wgrad", "act_grad", "GEMM", "sum_chunks")): return "networks"
    if name.startswith(("adam", "grad_norm", "ppo_", "gae")): return "loss / optimiser / GAE"
    return "other (copies, gathers, fills)"


col, full = load(f_collect), load(f_full)
rounds = K + W
ticks = rounds * H
print(f"# {label}: labyrinth 2v1 x{N}, horizon {H}, {R} rays; {rounds} rounds traced ({W} of them warm-up incl. graph capture)")
tot_c = sum(v[1] for v in col.values())
print(f"collect only: {tot_c / 1e3:.1f} ms of kernels in {rounds} rounds = {tot_c / ticks:.1f} us of kernel time per tick")
by = defaultdict(float)
for k, v in col.items(): by[cat(k)] += v[1]
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f"   {k:34s} {v / ticks:8.2f} us per tick  {100 * v / tot_c:5.1f} %")
for k, v in sorted(col.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"      {v[1] / ticks:8.2f} us per tick  {v[0] / ticks:6.2f} launches per tick  {k}")
tot_f = sum(v[1] for v in full.values())
upd = {k: (v[0] - col.get(k, [0, 0.0])[0], v[1] - col.get(k, [0, 0.0])[1]) for k, v in full.items()}
tot_u = sum(v[1] for v in upd.values())
print(f"collect + update: {tot_f / 1e3:.1f} ms of kernels; the update (difference to the collect-only trace): {tot_u / 1e3 / rounds:.2f} ms of kernel time per update")
for k, v in sorted(upd.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"      {v[1] / 1e3 / rounds:8.3f} ms per update  {v[0] / rounds:7.1f} launches  {k}")
fl = bench.learner_flops(R, A, 4)
tf_u = fl["update_per_env_step"] * H * N * rounds / (tot_u * 1e-6) / 1e12
tf_c = fl["collect_per_env_step"] * H * N * rounds / (by["networks"] * 1e-6) / 1e12
print(f"MFMA roofline (bf16 dense peak {bench.MFMA_BF16_PEAK_TFLOPS:.0f} TFLOP/s): update {fl['update_per_env_step'] / 1e6:.1f} MFLOP per env-step over its kernels' time = "
      f"{tf_u:.0f} TFLOP/s = {tf_u / bench.MFMA_BF16_PEAK_TFLOPS:.3f} of peak; collection's network kernels {fl['collect_per_env_step'] / 1e6:.2f} MFLOP per env-step = "
      f"{tf_c:.0f} TFLOP/s = {tf_c / bench.MFMA_BF16_PEAK_TFLOPS:.3f} of peak")
