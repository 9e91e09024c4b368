#!/usr/bin/env python3
"""Debug aid: per-parameter gradient error of the HIP fp32 path and of the fp32 oracle against the fp64 oracle, in network order
(python tests/aids/debug_noise.py [B] [links 0/1])."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _gpu_checks as G
from oracle import maskunet_oracle as O
from maskunet_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2 and sys.argv[2] == "0":
    ops.GRAD_LINKS = ops.ATTN_FUSED_ADD = False
model, params, keeps, x, labels = G.build_unet(150, False, 310, torch.float32, True, B)
def run(dt):
    p = {k: ((v.to(dt) if v.dtype.is_floating_point else v).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)) for k, v in params.items()}
    out = O.unet_forward(p, x.to(dt), keeps, training=True, dropout_masks=None, new_stats={}, three_head=False)
    O.pixel_cross_entropy(out, labels, -100).backward()
    return p
p32, p64 = run(torch.float32), run(torch.float64)
out = model(x.cuda()); F.cross_entropy(out, labels.cuda()).backward()
g64max = max(float(v.grad.abs().max()) for v in p64.values() if v.requires_grad and v.grad is not None)
print(f"{'parameter':55s} {'hip maxrel':>10s} {'ref maxrel':>10s} {'ratio':>7s} {'hip L2rel':>10s} {'ref L2rel':>10s}")
for k, v in model.named_parameters():
    if p64[k].grad is None: continue
    g64 = p64[k].grad; den = max(float(g64.abs().max()), 1e-3 * g64max)
    eh = float((v.grad.double().cpu() - g64).abs().max()) / den
    er = float((p32[k].grad.double() - g64).abs().max()) / den
    lh = float((v.grad.double().cpu() - g64).norm() / (g64.norm() + 1e-300)); lr = float((p32[k].grad.double() - g64).norm() / (g64.norm() + 1e-300))
    print(f"{k:55s} {eh:10.2e} {er:10.2e} {eh / max(er, 1e-12):7.1f} {lh:10.2e} {lr:10.2e}")
