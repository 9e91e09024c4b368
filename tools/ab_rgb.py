#!/usr/bin/env python3
"""In-process A/B of the first-layer weight-gradient kernel (3 -> 64 channels, 128x128, B = 64): python tools/ab_rgb.py A B ..."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from maskunet_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    lib = ctypes.CDLL(os.path.join(ROOT, "gpurun_variants", f"libmu_{name}.so"))
    for n, (res, args) in _lib.SIGNATURES.items():
        f = getattr(lib, n, None)
        if f is not None:
            f.restype = res; f.argtypes = args
    return lib


def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


names = sys.argv[1:]
libs = {n: load(n) for n in names}
B, H, Cin, Cout = 64, 128, 32, 64
dev, dt = "cuda", torch.float16
x = torch.zeros(B, H, H, Cin, device=dev, dtype=dt); x[..., :3] = torch.rand(B, H, H, 3, device=dev).to(dt)
dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt)
gw = {n: torch.empty(Cout, 3, 3, 3, device=dev) for n in names}
ws = torch.empty(max(l.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9) for l in libs.values()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def mk(n):
    l = libs[n]
    return lambda: l.mu_conv_wgrad(x.data_ptr(), dy.data_ptr(), gw[n].data_ptr(), B, H, H, Cin, Cout, 9, 3, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, st)
res = {n: [] for n in names}
for n in names: mk(n)()
torch.cuda.synchronize()
for rnd in range(10):
    for n in names: res[n].append(timeit(mk(n), 5))
ref = gw[names[0]]
for n in names:
    print(f"{n}: median {statistics.median(res[n]) * 1e3:.1f} us  min {min(res[n]) * 1e3:.1f} us   max|dW - dW[{names[0]}]| / max|dW| = {float((gw[n] - ref).abs().max() / ref.abs().max()):.2e}")
