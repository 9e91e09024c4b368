#!/usr/bin/env python3
"""Micro-benchmark of the 3x3 conv kernels (debug aid): python tools/bench_conv.py [B H Cin Cout]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib

def main():
    B, H, Cin, Cout = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 128, 128, 128)))
    dt = torch.float16
    dev = "cuda"
    x = torch.randn(B, H, H, Cin, device=dev, dtype=dt)
    w = (torch.randn(9, Cout, Cin, device=dev) * 0.05).to(dt)
    y = torch.empty(B, H, H, Cout, device=dev, dtype=dt)
    dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt)
    gw = torch.empty(Cout, Cin, 3, 3, device=dev)
    ws = _lib.workspace(_lib.load().mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9), torch.device(dev))
    st = _lib.stream()
    def fwd():
        _lib.call("mu_conv_fwd", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, 1, st)
    def wg():
        _lib.call("mu_conv_wgrad", x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, st)
    fl = 2.0 * B * H * H * Cin * Cout * 9
    for f, name in ((fwd, "conv3x3 fwd"), (wg, "conv3x3 wgrad")):
        for _ in range(2): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name} B={B} {H}x{H} {Cin}->{Cout} f16: {ms*1e3:.1f} us, {fl/ms/1e9:.0f} TF/s")
main()
