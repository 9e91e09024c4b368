#!/usr/bin/env python3
"""Race screen at the bench shapes (debug aid, GPU): every 3x3 conv shape of the model at B = 64 -- forward, data-gradient, weight-gradient --
and the attention kernels at the model's (N, C) pairs are launched REPS times on the same inputs; every output must be bit-identical to the
first launch (a hand-placed vmcnt / barrier schedule that is one phase short shows up as rare differing tiles, cdna guide section 5).
python tools/stress_determinism.py [REPS]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib
from tools.bench_layers import LAYERS  # noqa: E402  (importing runs nothing: main() is guarded below)

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B, dev, dt = 64, "cuda", torch.float16
    st = _lib.stream(); lib = _lib.load()
    bad = 0
    for H, Cin, Cout, _ in LAYERS:
        x = torch.randn(B, H, H, Cin, device=dev, dtype=dt); dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt)
        w = (torch.randn(9, Cout, Cin, device=dev) * 0.05).to(dt); wt = (torch.randn(9, Cin, Cout, device=dev) * 0.05).to(dt)
        cin_v = 3 if Cin == 32 else Cin
        ws = _lib.workspace(lib.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9), torch.device(dev))
        def fwd():
            y = torch.empty(B, H, H, Cout, device=dev, dtype=dt)
            _lib.call("mu_conv_fwd", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, 1, st); return y
        def dg():
            dx = torch.empty(B, H, H, Cin, device=dev, dtype=dt)
            _lib.call("mu_conv_fwd", dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), B, H, H, Cout, Cin, 9, Cout, Cin, 1, st); return dx
        def wg():
            gw = torch.empty(Cout, cin_v, 3, 3, device=dev)
            _lib.call("mu_conv_wgrad", x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, cin_v, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, st); return gw
        for name, f in (("fwd", fwd), ("dgrad", dg), ("wgrad", wg)):
            if name == "dgrad" and Cin == 32: continue
            ref = f(); torch.cuda.synchronize()
            diff = sum(0 if torch.equal(f(), ref) else 1 for _ in range(reps))
            if diff or not torch.isfinite(ref.float()).all():
                bad += 1; print(f"NONDETERMINISTIC conv {name} {H}x{H} {Cin}->{Cout}: {diff}/{reps} launches differ")
    for (Hs, C) in ((128, 64), (64, 64), (64, 128), (32, 128), (32, 256), (16, 256)):
        N = Hs * Hs
        qkv = torch.randn(B, N, 3 * C, device=dev, dtype=dt); x = torch.randn(B, N, C, device=dev, dtype=dt); gout = torch.randn(B, N, C, device=dev, dtype=dt)
        keep = torch.randint(0, 2, (B, N), device=dev, dtype=torch.uint8)
        kidx = torch.argsort(keep, dim=1, descending=True, stable=True).to(torch.int32).contiguous(); kcnt = keep.sum(1, dtype=torch.int32).contiguous()
        g = torch.ones(C, device=dev); b_ = torch.zeros(C, device=dev)
        wsb = _lib.workspace(lib.mu_attn_bwd_workspace_bytes(B, N, C), torch.device(dev))
        def run():
            out = torch.empty_like(x); oattn = torch.empty_like(x); lse = torch.empty(B, N, device=dev); mean = torch.empty_like(lse); rstd = torch.empty_like(lse)
            _lib.call("mu_attn_fwd", qkv.data_ptr(), x.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), g.data_ptr(), b_.data_ptr(), out.data_ptr(), oattn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, C, N, 1e-5, 1, st)
            dY = torch.empty_like(x); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse); dgm = torch.empty(C, device=dev); dbt = torch.empty(C, device=dev)
            for ph in (1, 2, 4):
                _lib.call("mu_attn_bwd_phases", qkv.data_ptr(), x.data_ptr(), oattn.data_ptr(), gout.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dY.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dgm.data_ptr(), dbt.data_ptr(), B, N, C, N, wsb.data_ptr(), wsb.numel(), 1, ph | 8, st)
            return out, dqkv, dY
        ref = run(); torch.cuda.synchronize()
        r2 = max(2, reps // 4)
        diff = sum(0 if all(torch.equal(a, b) for a, b in zip(run(), ref)) else 1 for _ in range(r2))
        if diff or not all(torch.isfinite(t.float()).all() for t in ref):
            bad += 1; print(f"NONDETERMINISTIC attention N={N} C={C}: {diff}/{r2} launches differ")
    print("stress_determinism:", "FAILED" if bad else "all launches bit-identical", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
