#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its GPU time (debug aid): python tools/cpu_overhead.py [--optimizer]"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maskunet_amd
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = maskunet_amd.UNet(3, 150).to(dev)
model.set_compute_dtype(torch.float16).train()
x, labels, keeps = bench.synth(64, 150, 128, 42, dev)
model.set_keep_masks(keeps)
opt = maskunet_amd.FusedAdamW(model.parameters(), lr=5e-5) if "--optimizer" in sys.argv else None

def step():
    out = model(x)
    loss = F.cross_entropy(out, labels)
    (loss * 1024.0).backward()
    if opt is not None:
        opt.step(grad_scale=1024.0)
    model.zero_grad(set_to_none=True)

for _ in range(3):
    step()
torch.cuda.synchronize()
# host enqueue time: the queue is empty at the start of each step, so the host never blocks on the device
cpu = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    cpu.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
print(f"host enqueue per step: {1e3 * sorted(cpu)[2]:.1f} ms (median of 5);  back-to-back wall per step: {1e3 * wall:.1f} ms")
