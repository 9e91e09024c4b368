import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _gpu_checks as G
from maskunet_amd import ops
model, params, keeps, x, labels = G.build_unet(150, False, 310, torch.float32, True, 2)
orig = ops._BNAct.backward
cap = []
def bw(ctx, gy):
    outs = orig(ctx, gy)
    xs, res, mean, rstd, g_p, b_p = ctx.saved_tensors
    cap.append((tuple(xs.shape), ctx.act, res is not None, xs.detach().clone(), gy.detach().clone(), mean.clone(), rstd.clone(), g_p.clone(), b_p.clone(), outs[0].clone(),
                None if outs[1] is None else outs[1].clone()))
    return outs
ops._BNAct.backward = staticmethod(bw)
out = model(x.cuda()); F.cross_entropy(out, labels.cuda()).backward()
print(len(cap), "BN backward calls")
for (shape, act, hasres, xs, gy, mean, rstd, g_p, b_p, dx, dres) in cap:
    C = shape[-1]
    X = xs.double().view(-1, C); Gy = gy.double().view(-1, C)
    xh = (X - mean.double()) * rstd.double()
    pre = xh * g_p.double() + b_p.double()
    if hasres:
        continue
    if act == 1:
        dz = Gy * (0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * np.pi) ** 0.5)
    elif act == 2:
        dz = Gy * (pre > 0)
    else:
        dz = Gy
    s1 = dz.mean(0); s2 = (dz * xh).mean(0)
    ref = g_p.double() * rstd.double() * (dz - s1 - xh * s2)
    e = (dx.double().view(-1, C) - ref)
    ratio = float((mean.abs() * rstd).max())
    proj = float(((dz - s1 - xh * s2).norm()) / dz.norm())
    print(f"{str(shape):24s} act {act} |dz - proj|/|dz| {proj:.3f} L2rel {float(e.norm() / ref.norm()):.2e} maxrel {float(e.abs().max() / ref.abs().max()):.2e}  max|mean|*rstd {ratio:.1f}  |s2|/rms(dz) {float((s2.abs() / dz.pow(2).mean(0).sqrt()).max()):.2e}")
