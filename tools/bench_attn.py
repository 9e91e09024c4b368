#!/usr/bin/env python3
"""Micro-benchmark of the attention block kernels (debug aid): python tools/bench_attn.py [B N C]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib, ops

def main():
    B, H, C = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (64, 128, 64)))
    dt = torch.float16 if os.environ.get("DT", "f16") == "f16" else torch.float32
    N = H * H
    dev = "cuda"
    torch.manual_seed(0)
    qkv = torch.randn(B, N, 3 * C, device=dev, dtype=dt)
    x = torch.randn(B, N, C, device=dev, dtype=dt)
    keep = torch.randint(0, 2, (B, N), device=dev, dtype=torch.uint8)
    kidx = torch.argsort(keep, dim=1, descending=True, stable=True).to(torch.int32).contiguous()
    kcnt = keep.sum(1, dtype=torch.int32).contiguous()
    g = torch.ones(C, device=dev); b_ = torch.zeros(C, device=dev)
    out = torch.empty_like(x); oattn = torch.empty_like(x)
    lse = torch.empty(B, N, device=dev); mean = torch.empty_like(lse); rstd = torch.empty_like(lse)
    dY = torch.empty_like(x); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    gout = torch.randn_like(x)
    ws = _lib.workspace(_lib.load().mu_attn_bwd_workspace_bytes(B, N, C), torch.device(dev))
    st = _lib.stream()
    def fwd():
        _lib.call("mu_attn_fwd", qkv.data_ptr(), x.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), g.data_ptr(), b_.data_ptr(), out.data_ptr(), oattn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, C, N, 1e-5, _lib.dt(x), st)
    def bwd():
        _lib.call("mu_attn_bwd", qkv.data_ptr(), x.data_ptr(), oattn.data_ptr(), gout.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dY.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dg.data_ptr(), db.data_ptr(), B, N, C, N, ws.data_ptr(), ws.numel(), _lib.dt(x), st)
    for f, name, mult in ((fwd, "fwd", 1.0), (bwd, "bwd", 3.5)):
        for _ in range(2): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        nk = float(kcnt.float().mean())
        fl = 4.0 * N * nk * C * B * mult
        print(f"attn {name} B={B} N={N} C={C} {dt}: {ms:.3f} ms, executed {fl/ms/1e9:.1f} TF/s (kept keys {nk:.0f})")
main()
