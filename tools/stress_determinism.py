#!/usr/bin/env python3
"""Race screen at the bench shapes (debug aid, GPU): every 3x3 conv shape of the model at B = 64 -- forward, data-gradient, weight-gradient --
and the attention kernels at the model's (N, C) pairs are launched REPS times on the same inputs; every output must be bit-identical to the
first launch (a hand-placed vmcnt / barrier schedule that is one phase short shows up as rare differing tiles, cdna guide section 5).
python tools/stress_determinism.py [REPS] [--fp32x]     (--fp32x, round 6: the fp16-pair forward kernels, the two-term HL data-gradient
kernels and the pair-reduced weight gradient of the fp32x mode, plus its attention sweeps)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib
from tools.bench_layers import LAYERS  # noqa: E402  (importing runs nothing: main() is guarded below)

def main_fp32x(reps):
    B, dev = 64, "cuda"
    st = _lib.stream(); lib = _lib.load()
    bad = 0
    for H, Cin, Cout, _ in LAYERS:
        if Cin == 32:
            continue                       # (the 3-channel stem runs plain-FMA kernels in this mode)
        x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev) * 1e-6
        woihw = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        both = torch.empty(2 * 9 * Cout * Cin, device=dev)
        _lib.call("mu_prep_weight", woihw.data_ptr(), both.data_ptr(), 2, Cout, Cin, 9, Cout, Cin, 2, st)
        w, wt = both[:9 * Cout * Cin], both[9 * Cout * Cin:]
        _lib.call("mu_split_encode_h4", x.data_ptr(), x.data_ptr(), x.numel(), st)
        dyh = torch.empty(dy.numel(), dtype=torch.float16, device=dev); sc = torch.empty(2, device=dev)
        ws0 = torch.empty(lib.mu_dy_encode_h_workspace_bytes(), dtype=torch.uint8, device=dev)
        _lib.call("mu_dy_encode_h", dy.data_ptr(), dyh.data_ptr(), sc.data_ptr(), dy.numel(), ws0.data_ptr(), ws0.numel(), st)
        ws = _lib.workspace(lib.mu_conv_wgrad_h_workspace_bytes(B, H, H, Cin, Cout), torch.device(dev))
        def fwd():
            y = torch.empty(B, H, H, Cout, device=dev)
            _lib.call("mu_conv_fwd", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, 2, st); return y
        def dg():
            dx = torch.empty(B, H, H, Cin, device=dev)
            _lib.call("mu_conv_dgrad_h", dyh.data_ptr(), wt.data_ptr(), sc.data_ptr(), dx.data_ptr(), B, H, H, Cout, Cin, Cout, Cin, st); return dx
        def wg():
            gw = torch.empty(Cout, Cin, 3, 3, device=dev)
            _lib.call("mu_conv_wgrad_h", x.data_ptr(), dyh.data_ptr(), sc.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, Cin, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), st); return gw
        for name, f in (("fwd", fwd), ("dgrad", dg), ("wgrad", wg)):
            ref = f(); torch.cuda.synchronize()
            diff = sum(0 if torch.equal(f(), ref) else 1 for _ in range(reps))
            if diff or not torch.isfinite(ref).all():
                bad += 1; print(f"NONDETERMINISTIC fp32x conv {name} {H}x{H} {Cin}->{Cout}: {diff}/{reps} launches differ")
    print("stress_determinism --fp32x (3x3 forward / two-term data gradient / pair-reduced weight gradient):", "FAILED" if bad else "all launches bit-identical", flush=True)
    sys.exit(1 if bad else 0)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(args[0]) if args else 20
    if "--fp32x" in sys.argv:
        return main_fp32x(reps)
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
