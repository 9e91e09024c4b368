#!/usr/bin/env python3
"""Weight-gradient-only micro run for PMC passes (debug aid): python tools/bench_wgrad_only.py [B H Cin Cout]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib
B, H, Cin, Cout = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 128, 128, 128)))
dev = "cuda"; dt = torch.float16
x = torch.randn(B, H, H, Cin, device=dev, dtype=dt); dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt)
gw = torch.empty(Cout, Cin, 3, 3, device=dev)
ws = _lib.workspace(_lib.load().mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9), torch.device(dev))
for _ in range(6):
    _lib.call("mu_conv_wgrad", x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, _lib.stream())
torch.cuda.synchronize()
print("done")
