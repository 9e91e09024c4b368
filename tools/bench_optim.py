#!/usr/bin/env python3
"""Timing of the fused AdamW step alone (debug aid): host time per call and GPU time per call for UNet(3,150)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maskunet_amd
dev = torch.device("cuda", 0)
model = maskunet_amd.UNet(3, 150).to(dev)
opt = maskunet_amd.FusedAdamW(model.parameters(), lr=5e-5, weight_decay=1e-1)
for p in model.parameters():
    p.grad = torch.randn_like(p) * 1e-3
for _ in range(3): opt.step(grad_scale=1024.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10): opt.step(grad_scale=1024.0)
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"FusedAdamW.step: host enqueue {1e3 * (t1 - t0) / 10:.3f} ms/call, GPU {e0.elapsed_time(e1) / 10:.3f} ms/call, wall {1e3 * (t2 - t0) / 10:.3f} ms/call")
ref = torch.optim.AdamW(model.parameters(), lr=5e-5, weight_decay=1e-1, fused=True)
for _ in range(3): ref.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
for _ in range(10): ref.step()
e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"torch AdamW(fused=True).step: host enqueue {1e3 * (t1 - t0) / 10:.3f} ms/call, GPU {e0.elapsed_time(e1) / 10:.3f} ms/call, wall {1e3 * (t2 - t0) / 10:.3f} ms/call")
