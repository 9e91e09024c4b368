#!/usr/bin/env python3
"""Per-layer table of the 3x3 conv kernels at the bench configuration (debug aid / profiles evidence):
python tools/bench_layers.py [B] [--fp32x]
Every 3x3 shape of UNet(3, c_out, hw=128) at batch B: forward, data-gradient (the same kernel family with Cin/Cout swapped) and
weight-gradient, in-process HIP-event timings over 20 launches, algorithmic TFLOP/s = 2*B*H*W*Cin*Cout*9 / t."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib

# (H, Cin, Cout, layers per forward) -- modules.py UNet.__init__, hw = 128
LAYERS = [
    (128, 32, 64, 1), (128, 64, 64, 2), (128, 128, 128, 2), (128, 128, 64, 1),
    (64, 64, 64, 2), (64, 64, 128, 1), (64, 128, 128, 1), (64, 256, 256, 2), (64, 256, 128, 1), (64, 128, 64, 1),
    (32, 128, 128, 2), (32, 128, 256, 1), (32, 256, 256, 1), (32, 512, 512, 2), (32, 512, 256, 1), (32, 256, 128, 1),
    (16, 256, 256, 5), (16, 256, 512, 1), (16, 512, 512, 3), (16, 512, 256, 1),
]

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 64
    X = "--fp32x" in sys.argv                                # fp32 storage, fp16-pair operands: 3-term forward, 2-term data gradient, 1-term weight gradient (round 6)
    dev, dt = "cuda", (torch.float32 if X else torch.float16)
    code = 2 if X else 1
    st = _lib.stream()
    lib = _lib.load()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}; totfl = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("| H=W | Cin | Cout | layers | fwd us | fwd TF/s | dgrad us | dgrad TF/s | wgrad us | wgrad TF/s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for H, Cin, Cout, cnt in LAYERS:
        x = torch.randn(B, H, H, Cin, device=dev, dtype=dt)
        dy = torch.randn(B, H, H, Cout, device=dev, dtype=dt)
        w = (torch.randn(9, Cout, Cin, device=dev) * 0.05).to(dt)
        wt = (torch.randn(9, Cin, Cout, device=dev) * 0.05).to(dt)
        y = torch.empty(B, H, H, Cout, device=dev, dtype=dt)
        dx = torch.empty(B, H, H, Cin, device=dev, dtype=dt)
        if X:
            # round 6: fp16-pair operands for the forward, dy as ONE scaled fp16 operand + HL weight rows for the two-term backward
            woihw = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
            both = torch.empty(2 * 9 * Cout * Cin, device=dev)
            _lib.call("mu_prep_weight", woihw.data_ptr(), both.data_ptr(), 2, Cout, Cin, 9, Cout, Cin, 2, st)
            w, wt = both[:9 * Cout * Cin], both[9 * Cout * Cin:]
            _lib.call("mu_split_encode_h4", x.data_ptr(), x.data_ptr(), x.numel(), st)
            dyh = torch.empty(dy.numel(), dtype=torch.float16, device=dev)
            sc = torch.empty(2, device=dev)
            ws0 = torch.empty(lib.mu_dy_encode_h_workspace_bytes(), dtype=torch.uint8, device=dev)
            _lib.call("mu_dy_encode_h", dy.data_ptr(), dyh.data_ptr(), sc.data_ptr(), dy.numel(), ws0.data_ptr(), ws0.numel(), st)
            # the weight gradient as the model runs it: ONE term on the fp16 rounding of the input (mu_conv_wgrad_h1)
            x16 = torch.randn(B, H, H, Cin, device=dev, dtype=torch.float16)
        cin_v = 3 if Cin == 32 else Cin
        gw = torch.empty(Cout, cin_v, 3, 3, device=dev)
        ws = _lib.workspace(lib.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9), torch.device(dev))
        fl = 2.0 * B * H * H * Cin * Cout * 9
        def fwd(): _lib.call("mu_conv_fwd", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, 9, Cin, Cout, code, st)
        def dg():
            if X: _lib.call("mu_conv_dgrad_h", dyh.data_ptr(), wt.data_ptr(), sc.data_ptr(), dx.data_ptr(), B, H, H, Cout, Cin, Cout, Cin, st)
            else: _lib.call("mu_conv_fwd", dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), B, H, H, Cout, Cin, 9, Cout, Cin, code, st)
        def wg():
            if X and Cin > 32:
                _lib.call("mu_conv_wgrad_h1", x16.data_ptr(), dyh.data_ptr(), sc.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, cin_v, Cout, Cin, Cout,
                          ws.data_ptr(), ws.numel(), st)
            else:       # (fp32x: the <= 3-channel first layer is the plain-FMA kernel on plain operands)
                _lib.call("mu_conv_wgrad", x.data_ptr(), dy.data_ptr(), gw.data_ptr(), B, H, H, Cin, Cout, 9, cin_v, Cout, Cin, Cout,
                          ws.data_ptr(), ws.numel(), code, st)
        row = []
        for name, f in (("fwd", fwd), ("dgrad", dg), ("wgrad", wg)):
            if name == "dgrad" and Cin == 32:
                row += ["-", "-"]; continue
            ms = timeit(f)
            tot[name] += ms * cnt; totfl[name] += fl * cnt
            row += [f"{ms * 1e3:.1f}", f"{fl / ms / 1e9:.0f}"]
        print(f"| {H} | {Cin} | {Cout} | {cnt} | " + " | ".join(row) + " |")
    print()
    for k in tot:
        print(f"{k}: {tot[k]:.3f} ms per step over all layers, {totfl[k] / tot[k] / 1e9:.0f} TF/s aggregate")


if __name__ == "__main__":
    main()
