"""CPU experiment (round 6): sizing of the fp32x 3x3-conv arithmetic BEFORE building it (VERDICT r5 "Next round" #1).

Question: if the 3x3 convolutions of the fp32x mode move from [bf16 hi | bf16 lo] operands (three bf16 MFMAs per product, forward AND
backward) to
  * forward:  x and w as fp16 (hi | lo) pairs, three fp16 MFMAs per product (hi*hi + hi*lo + lo*hi), weights under a static
              power-of-two shift so that their lo halves stay fp16-normal;
  * backward: dy as ONE fp16 operand under a per-tensor power-of-two scale, against the two-term fp16 pair of the partner (w for the
              data gradient, the saved x for the weight gradient): two MFMAs per product,
do the reference-generated whole-model goldens keep the fp32 gates (outputs 1e-3, worst parameter gradient 5e-2, 1 - cos 1e-4)?

Emulates the arithmetic inside the oracle's ConvBlock (everything else of the oracle stays exact fp32) and prints the metrics of
tests/_gpu_checks.check_unet_golden.  Not a test, not product code: sizing evidence quoted in NOTES_r06.md.

usage: python tests/aids/numerics_conv_bwd_two_term.py <fwd> <bwd> [wshift] [ftz] [dyscale_slack]
   fwd: exact | bf3 (round 5) | h3        bwd: exact | bf3 (round 5) | h2 (built) | b1 (dy as ONE bf16: the judge's "too coarse" case)
        | h1x (h2 with the saved input of the WEIGHT gradient as one fp16 term too) | h1w (h2 with the weights of the DATA gradient as one
        term) | h1 (both): probes of a one-MFMA backward, not built
   wshift: log2 of the static weight shift (default 6); ftz: 1 = flush fp16 subnormals to zero (worst case for the matrix core);
   dyscale_slack: log2 of how far BELOW the ideal scale the dy scale sits (the device-side bound is loose by (2 + max|xhat|): ~3 bits)
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import maskunet_oracle as O  # noqa: E402

FWD = sys.argv[1] if len(sys.argv) > 1 else "h3"
BWD = sys.argv[2] if len(sys.argv) > 2 else "h2"
WSHIFT = int(sys.argv[3]) if len(sys.argv) > 3 else 6
FTZ = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
SLACK = int(sys.argv[5]) if len(sys.argv) > 5 else 3
GOLDEN = os.environ.get("GOLDEN", "unet1_c150_b2_train")


def h(x):
    y = x.half().float()
    if FTZ:
        y = torch.where(y.abs() < 2.0 ** -14, torch.zeros_like(y), y)
    return y


def b(x):
    return x.bfloat16().float()


def pair(x, r):
    hi = r(x)
    return hi, r(x - hi)


def conv3(x, w, r):
    """three-term product of two pairs: hi*hi + hi*lo + lo*hi (fp32 accumulate)"""
    xh, xl = pair(x, r)
    wh, wl = pair(w, r)
    return F.conv2d(xh, wh + wl, None, padding=1) + F.conv2d(xl, wh, None, padding=1)


class Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if FWD == "exact":
            return F.conv2d(x, w, None, padding=1)
        if FWD == "bf3":
            return conv3(x, w, b)
        s = 2.0 ** WSHIFT
        return conv3(x, w * s, h) / s

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if BWD == "exact":
            return torch.nn.grad.conv2d_input(x.shape, w, dy, padding=1), torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
        if BWD == "bf3":
            dh, dl = pair(dy, b)
            wh, wl = pair(w, b)
            xh, xl = pair(x, b)
            dx = torch.nn.grad.conv2d_input(x.shape, wh + wl, dh, padding=1) + torch.nn.grad.conv2d_input(x.shape, wh, dl, padding=1)
            dw = torch.nn.grad.conv2d_weight(xh + xl, w.shape, dh, padding=1) + torch.nn.grad.conv2d_weight(xh, w.shape, dl, padding=1)
            return dx, dw
        amax = float(dy.abs().max())
        if BWD == "b1":
            d1 = b(dy)
            wh, wl = pair(w, b)
            xh, xl = pair(x, b)
            return (torch.nn.grad.conv2d_input(x.shape, wh + wl, d1, padding=1), torch.nn.grad.conv2d_weight(xh + xl, w.shape, d1, padding=1))
        # h2: one scaled fp16 dy against the fp16 pair of the partner
        gs = 2.0 ** (12 - SLACK - math.floor(math.log2(amax))) if amax > 0 else 1.0       # gs * max|dy| in [2^(12-slack), 2^(13-slack))
        d1 = h(dy * gs)
        s = 2.0 ** WSHIFT
        wh, wl = pair(w * s, h)
        xh, xl = pair(x, h)
        wd = wh if BWD in ("h1w", "h1") else wh + wl
        xw = xh if BWD in ("h1x", "h1") else xh + xl
        dx = torch.nn.grad.conv2d_input(x.shape, wd, d1, padding=1) / (gs * s)
        dw = torch.nn.grad.conv2d_weight(xw, w.shape, d1, padding=1) / gs
        return dx, dw


def conv_block(x, p, prefix, residual, training, new_stats=None):
    w0, w3 = p[prefix + ".conv_block.0.weight"], p[prefix + ".conv_block.3.weight"]
    y = Conv.apply(x, w0) if w0.shape[1] > 3 else F.conv2d(x, w0, None, padding=1)      # (the 3-channel stem is a plain-FMA layer)
    y = O.batchnorm2d(y, p, prefix + ".conv_block.1", training, new_stats)
    y = O.gelu(y)
    y = Conv.apply(y, w3)
    y = O.batchnorm2d(y, p, prefix + ".conv_block.4", training, new_stats)
    return O.gelu(x + y) if residual else y


def main():
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", GOLDEN + ".npz"))
    rec = {k: z[k] for k in z.files}
    B, c_out, seed = int(rec["B"]), int(rec["c_out"]), int(rec["seed"])
    three = GOLDEN.startswith("unet3")
    p = O.make_params(O.unet_state_shapes(3, c_out, three), seed)
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    keeps = O.make_keeps(seed + 1, B)
    x, labels = O.make_inputs(seed + 2, B, c_out)
    O.conv_block = conv_block
    out = O.unet_forward(p, x, keeps, training=True, new_stats={}, three_head=three)
    out0 = out[0] if isinstance(out, (tuple, list)) else out
    ref = torch.from_numpy(rec["out0_slice"])
    eo = float((out0[:, :, ::16, ::16].detach() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    loss = O.pixel_cross_entropy(out0, labels, -100)
    loss.backward()
    gmax = max(float(v) for k, v in rec.items() if k.startswith("gnorm/"))
    floor = 1e-4 * gmax
    worst, wcos = (0.0, ""), (0.0, "")
    for k, v in p.items():
        if "gnorm/" + k not in rec or not bool(rec["ghas/" + k]) or v.grad is None:
            continue
        gn = float(v.grad.double().norm())
        r = float(rec["gnorm/" + k])
        e = abs(gn - r) / max(r, floor)
        if e > worst[0]:
            worst = (e, k)
        if "g/" + k in rec:
            rr = torch.from_numpy(rec["g/" + k])
            e2 = float((v.grad - rr).abs().max()) / max(float(rr.abs().max()), floor)
            if e2 > worst[0]:
                worst = (e2, k + " (full)")
            if float(rr.double().norm()) > floor:
                c = 1.0 - float((v.grad.double() * rr.double()).sum() / (v.grad.double().norm() * rr.double().norm()))
                if c > wcos[0]:
                    wcos = (c, k)
    print(f"{GOLDEN} fwd {FWD} bwd {BWD} wshift {WSHIFT} ftz {int(FTZ)} slack {SLACK}: out err {eo:.3e} (gate 1e-3)  "
          f"loss err {abs(loss.item() - float(rec['loss'])):.3e}  worst grad {worst[0]:.3e} [{worst[1]}] (gate 5e-2)  "
          f"worst 1-cos {wcos[0]:.2e} [{wcos[1]}] (gate 1e-4, full-tensor goldens only)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
