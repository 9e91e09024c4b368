#!/usr/bin/env python3
"""In-process interleaved A/B timing of library variants (cdna guide rule 24): python tools/ab_bench.py attn|bn|conv|wgrad A B [C ...]
Variants are gpurun_variants/libmu_<NAME>.so built by tools/build_variant.sh.  Debug aid."""
import ctypes, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskunet_amd import _lib


def load(name):
    lib = ctypes.CDLL(os.path.join(ROOT, "gpurun_variants", f"libmu_{name}.so"))
    for n, (res, args) in _lib.SIGNATURES.items():
        f = getattr(lib, n, None)
        if f is None: continue          # older variant without this entry point
        f.restype = res; f.argtypes = args
    return lib


def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    what, names = sys.argv[1], sys.argv[2:]
    libs = {n: load(n) for n in names}
    dev = "cuda"; dt = torch.float16; st = torch.cuda.current_stream().cuda_stream
    XF = os.environ.get("MU_AB_FP32X") == "1"          # attn: fp32 storage, chunk-encoded qkv, split-bf16 products (dtype code 2)
    if XF: dt = torch.float32
    code = 2 if XF else 1
    if what == "attn":
        B, N, C = (int(v) for v in os.environ.get("MU_ATTN_SHAPE", "64,16384,64").split(","))
        qkv = torch.randn(B, N, 3 * C, device=dev, dtype=dt); x = torch.randn(B, N, C, device=dev, dtype=dt)
        keep = torch.randint(0, 2, (B, N), device=dev, dtype=torch.uint8)
        kidx = torch.argsort(keep, dim=1, descending=True, stable=True).to(torch.int32).contiguous(); kcnt = keep.sum(1, dtype=torch.int32).contiguous()
        g = torch.ones(C, device=dev); b_ = torch.zeros(C, device=dev)
        out = torch.empty_like(x); oattn = torch.empty_like(x); lse = torch.empty(B, N, device=dev); mean = torch.empty_like(lse); rstd = torch.empty_like(lse)
        dY = torch.empty_like(x); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
        gout = torch.randn_like(x)
        qkv_h = qkv
        if XF:          # round-5 libraries read qkv as fp16 pairs (mu_split_encode_h), round-4 ones as bf16 chunks (mu_split_encode)
            qkv_h = qkv.clone()
            _lib.call("mu_split_encode_h", qkv_h.data_ptr(), qkv_h.data_ptr(), qkv_h.numel(), st)
            _lib.call("mu_split_encode", qkv.data_ptr(), qkv.data_ptr(), qkv.numel(), st)
        ws = torch.empty(max(l.mu_attn_bwd_workspace_bytes(B, N, C) for l in libs.values()), dtype=torch.uint8, device=dev)
        def mk(lib, phase, qkv_b=qkv):
            qkv = qkv_h if hasattr(lib, "mu_split_encode_h") else qkv_b
            if phase == 0:
                return lambda: lib.mu_attn_fwd(qkv.data_ptr(), x.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), g.data_ptr(), b_.data_ptr(), out.data_ptr(), oattn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, C, N, 1e-5, code, st)
            return lambda: lib.mu_attn_bwd_phases(qkv.data_ptr(), x.data_ptr(), oattn.data_ptr(), gout.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dY.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), dg.data_ptr(), db.data_ptr(), B, N, C, N, ws.data_ptr(), ws.numel(), code, phase, st)
        cases = [("fwd", 0), ("dq", 2), ("dkv", 4)]
        first = libs[names[0]]; mk(first, 0)(); mk(first, 1)()
    elif what == "conv1":
        shapes = [(64, 128, 64, 192), (64, 64, 128, 384), (64, 128, 192, 64), (64, 128, 64, 160), (64, 32, 256, 768), (64, 64, 384, 128), (64, 32, 768, 256),
                  (64, 64, 64, 192), (64, 64, 192, 64), (64, 128, 160, 64)]
        bufs = [(torch.randn(B, H, H, Cin, device=dev, dtype=dt), (torch.randn(1, Cout, Cin, device=dev) * 0.05).to(dt), torch.empty(B, H, H, Cout, device=dev, dtype=dt)) for (B, H, Cin, Cout) in shapes]
        def mk(lib, phase):
            (B, H, Cin, Cout), (x, w, y) = shapes[phase], bufs[phase]
            return lambda: lib.mu_conv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 1, Cin, Cout, 1, st)
        cases = [(f"1x1 {shapes[p]} {2e-6 * shapes[p][0] * shapes[p][1] ** 2 * (shapes[p][2] + shapes[p][3]):.0f} MB", p) for p in range(len(shapes))]
    elif what == "wgrad1":
        shapes = [(64, 128, 64, 192), (64, 128, 64, 160), (64, 64, 128, 384), (64, 64, 64, 192), (64, 32, 256, 768), (64, 32, 128, 384)]
        bufs = [(torch.randn(B, H, H, Cin, device=dev, dtype=dt), torch.randn(B, H, H, Cout, device=dev, dtype=dt), torch.empty(Cout, Cin, 1, 1, device=dev),
                 torch.empty(max(l.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 1) for l in libs.values()), dtype=torch.uint8, device=dev)) for (B, H, Cin, Cout) in shapes]
        def mk(lib, phase):
            (B, H, Cin, Cout), (x, dy, gw, ws) = shapes[phase], bufs[phase]
            return lambda: lib.mu_conv_wgrad(x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 1, Cin, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, st)
        cases = [(f"1x1 wgrad {shapes[p]} {2e-6 * shapes[p][0] * shapes[p][1] ** 2 * (shapes[p][2] + shapes[p][3]):.0f} MB", p) for p in range(len(shapes))]
    elif what == "rgb":
        B, H, Cin, Cout = 64, 128, 32, 64
        x = torch.randn(B, H, H, Cin, device=dev, dtype=dt); dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt); gw = torch.empty(Cout, 3, 3, 3, device=dev)
        ws = torch.empty(max(l.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9) for l in libs.values()), dtype=torch.uint8, device=dev)
        def mk(lib, phase):
            return lambda: lib.mu_conv_wgrad(x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, 3, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), 1, st)
        cases = [("first-layer wgrad 3->64 @128^2", 0)]
    elif what == "bn":
        M, C = (int(v) for v in os.environ.get("MU_BN_SHAPE", "1048576,128").split(","))
        x = torch.randn(M, C, device=dev, dtype=dt); y = torch.empty_like(x); gy = torch.randn_like(x); dx = torch.empty_like(x)
        mean = torch.zeros(C, device=dev); rstd = torch.ones(C, device=dev); gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
        dgam = torch.empty(C, device=dev); dbet = torch.empty(C, device=dev)
        ws = torch.empty(_lib.load().mu_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)
        def mk(lib, phase):
            if phase == 0:
                return lambda: lib.mu_bn_train_stats(x.data_ptr(), M, C, C, mean.data_ptr(), rstd.data_ptr(), None, None, None, C, 0.1, 1e-5, ws.data_ptr(), ws.numel(), 1, st)
            if phase == 1:
                return lambda: lib.mu_bn_act_fwd(x.data_ptr(), None, y.data_ptr(), M, C, C, mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1, 1, st)
            return lambda: lib.mu_bn_act_bwd(x.data_ptr(), None, gy.data_ptr(), dx.data_ptr(), None, M, C, C, mean.data_ptr(), rstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1, 1, dgam.data_ptr(), dbet.data_ptr(), ws.data_ptr(), ws.numel(), 1, st)
        cases = [(f"bn_stats {M * C * 2 >> 20} MiB", 0), ("bn_act_fwd", 1), ("bn_act_bwd", 2)]
    else:
        shapes = [(64, 128, 128, 128), (64, 64, 256, 256), (64, 32, 512, 512), (64, 16, 512, 512), (64, 128, 64, 128), (64, 16, 256, 256), (64, 16, 256, 512), (64, 32, 256, 256), (64, 64, 128, 128), (64, 128, 128, 64), (64, 128, 64, 64), (64, 64, 64, 64)]
        bufs = []
        for (B, H, Cin, Cout) in shapes:
            bufs.append((torch.randn(B, H, H, Cin, device=dev, dtype=dt), (torch.randn(9, Cout, Cin, device=dev) * 0.05).to(dt), torch.empty(B, H, H, Cout, device=dev, dtype=dt),
                         torch.randn(B, H, H, Cout, device=dev, dtype=dt), torch.empty(Cout, Cin, 3, 3, device=dev),
                         torch.empty(max(l.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9) for l in libs.values()), dtype=torch.uint8, device=dev)))
            if XF:          # MU_AB_FP32X=1: chunk-encoded operands (x, w, dy), dtype code 2
                for t in (bufs[-1][0], bufs[-1][1], bufs[-1][3]):
                    _lib.call("mu_split_encode", t.data_ptr(), t.data_ptr(), t.numel(), st)
        def mk(lib, phase):
            (B, H, Cin, Cout), (x, w, y, dy, gw, ws) = shapes[phase // 2], bufs[phase // 2]
            if phase % 2 == 0:
                return lambda: lib.mu_conv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, code, st)
            return lambda: lib.mu_conv_wgrad(x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, Cin, Cout, ws.data_ptr(), ws.numel(), code, st)
        sel = tuple(range(0, 2 * len(shapes), 2)) if what == "conv" else tuple(range(1, 2 * len(shapes), 2))
        cases = [(f"{'fwd' if p % 2 == 0 else 'wgrad'} {shapes[p // 2]} {2e-9 * 9 * shapes[p // 2][0] * shapes[p // 2][1] ** 2 * shapes[p // 2][2] * shapes[p // 2][3]:.0f} GF", p) for p in sel]
    for cname, phase in cases:
        res = {n: [] for n in names}
        for n in names: timeit(mk(libs[n], phase), 2)
        for rnd in range(7):
            for n in names: res[n].append(timeit(mk(libs[n], phase)))
        print(cname, {n: f"med {statistics.median(v):.3f} min {min(v):.3f} ms" for n, v in res.items()})

main()
