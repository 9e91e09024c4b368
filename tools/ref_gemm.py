#!/usr/bin/env python3
"""Reference point for the MFMA kernels: what the vendor GEMM reaches on this box with random fp16 operands (debug aid)."""
import torch
for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda", dtype=torch.float16); b = torch.randn(n, n, device="cuda", dtype=torch.float16)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fp16 GEMM {n}^3 (torch/hipBLASLt, random operands): {ms:.3f} ms, {2 * n ** 3 / ms / 1e9:.0f} TF/s")
