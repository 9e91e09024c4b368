#!/usr/bin/env python3
"""Does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs?  (round 6: the fp16-pair conv operands put the lo halves of small values
into the subnormal range; tests/aids/numerics_conv_bwd_two_term.py shows the scheme needs them kept.)  Uses the shipped fp16 conv
kernels through the C ABI: y = sum_ci x * w with one operand subnormal, the other large enough for a normal fp16 result."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskunet_amd import _lib


def run(taps, xval, wval, Cin=64, Cout=128, H=16):
    dev = "cuda"
    x = torch.full((1, H, H, Cin), xval, device=dev, dtype=torch.float16)
    w = torch.full((taps, Cout, Cin), wval, device=dev, dtype=torch.float16)
    y = torch.empty((1, H, H, Cout), device=dev, dtype=torch.float16)
    _lib.call("mu_conv_fwd", x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 1, H, H, Cin, Cout, taps, Cin, Cout, _lib.MU_F16, _lib.stream())
    torch.cuda.synchronize()
    centre = float(y[0, H // 2, H // 2, 0])
    expect = float(x[0, 0, 0, 0]) * float(w[0, 0, 0]) * Cin * taps
    return centre, expect


for taps in (1, 9):
    for name, xv, wv in (("B operand (pixels) subnormal", 2.0 ** -20, 1024.0), ("A operand (weights) subnormal", 1024.0, 2.0 ** -20),
                         ("both normal", 2.0 ** -10, 1.0)):
        got, exp = run(taps, xv, wv)
        print(f"taps {taps}: {name}: got {got:.6g} expected {exp:.6g} -> {'KEPT' if abs(got - exp) <= 1e-3 * abs(exp) else 'FLUSHED / wrong'}")
