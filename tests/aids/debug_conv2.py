import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _gpu_checks as G
from maskunet_amd import ops
model, params, keeps, x, labels = G.build_unet(150, False, 310, torch.float32, True, 2)
orig = ops._Conv.backward
cap = []
def bw(ctx, gy, gpart=None):
    outs = orig(ctx, gy, gpart)
    xs, w = ctx.saved_tensors
    if w.shape[-1] == 3 and xs.shape[1] >= 64:
        cap.append((xs.detach().clone(), w.detach().clone(), gy.detach().clone(), None if outs[0] is None else outs[0].clone(), outs[1].clone()))
    return outs
ops._Conv.backward = staticmethod(bw)
out = model(x.cuda()); F.cross_entropy(out, labels.cuda()).backward()
for (xs, w, gy, gx, gw) in cap:
    O_, I_ = w.shape[:2]
    X = xs.double().permute(0, 3, 1, 2)[:, :I_].contiguous().requires_grad_(True)
    Wd = w.double().clone().requires_grad_(True)
    Y = F.conv2d(X, Wd, padding=1)
    Y.backward(gy.double().permute(0, 3, 1, 2)[:, :O_].contiguous())
    egw = float((gw.double() - Wd.grad).norm() / Wd.grad.norm())
    egx = float((gx.double().permute(0, 3, 1, 2)[:, :I_] - X.grad).norm() / X.grad.norm()) if gx is not None else -1
    print(f"x {tuple(xs.shape)} w {tuple(w.shape)}: dW L2rel {egw:.2e}  dX L2rel {egx:.2e}  |gy|max {float(gy.abs().max()):.2e} rms {float(gy.pow(2).mean().sqrt()):.2e}")
