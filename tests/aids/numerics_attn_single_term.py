"""CPU experiment (round 5): does the fp32x attention keep the fp32 gates when P and dS enter the matrix core as ONE fp16
operand and q/k/v/dY are stored as fp16 (hi | lo) pairs?  Emulates the planned kernel arithmetic inside the oracle's attention
and runs the reference-generated whole-model golden `unet1_c150_b2_train` through it (same metrics as
tests/_gpu_checks.check_unet_golden).  Not a test, not product code: sizing evidence quoted in NOTES_r05.md.

usage: python tests/aids/numerics_attn_single_term.py [exact|pair16|p16|p16ds16|p16ds16_noscale]
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import maskunet_oracle as O  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "p16ds16"
JSHIFT = 12
# round 6 probes (env): KSINGLE=1 -- the key operand of S = Q K^T as ONE fp16 term (two MFMAs instead of three), forward and recompute alike;
# VSINGLE=1 -- the value operand of dP = dO V^T as one fp16 term
KSINGLE = os.environ.get("KSINGLE", "0") == "1"
VSINGLE = os.environ.get("VSINGLE", "0") == "1"
# further probes (not built): the SECOND operand of the gradient products as one fp16 term too -- dV = P^T dO with dO single (DV1), dK = dS^T Q
# with Q single (DK1), dQ = dS K with K single (DQ1), dP = dO V^T with dO single as well (DP1): one MFMA per product
DV1, DK1, DQ1, DP1 = (os.environ.get(k, "0") == "1" for k in ("DV1", "DK1", "DQ1", "DP1"))
SB1 = os.environ.get("SB1", "0") == "1"      # the BACKWARD's recomputed scores with the query as one fp16 term too (the forward keeps two terms)


def h(x):
    return x.half().float()


def pair16(x):
    hi = h(x)
    return hi + h(x - hi)


class Attn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, V, keep):
        B, N, C = Q.shape
        scale = 1.0 / math.sqrt(C)
        sl2 = 1.4426950408889634 / math.sqrt(C)
        O_ = torch.empty_like(Q)
        lse2 = torch.empty(B, N)
        pair = MODE != "exact"
        Qe, Ke, Ve = (pair16(Q), pair16(K), pair16(V)) if pair else (Q, K, V)
        for b in range(B):
            idx = keep[b].nonzero()[:, 0]
            qs = pair16(Qe[b] * sl2) if pair else Qe[b] * sl2
            s2 = qs @ (h(Ke[b, idx]) if KSINGLE else Ke[b, idx]).T
            m = s2.max(-1, keepdim=True).values
            p = torch.exp2(s2 - m)
            if MODE.startswith("p16"):
                p = h(p)
            l = p.sum(-1, keepdim=True)
            O_[b] = (p @ Ve[b, idx]) / l
            lse2[b] = (m + torch.log2(l))[:, 0]
        ctx.save_for_backward(Qe, Ke, Ve, keep, O_, lse2)
        return O_

    @staticmethod
    def backward(ctx, dY):
        Qe, Ke, Ve, keep, O_, lse2 = ctx.saved_tensors
        B, N, C = Qe.shape
        scale = 1.0 / math.sqrt(C)
        sl2 = 1.4426950408889634 / math.sqrt(C)
        pair = MODE != "exact"
        dQ, dK, dV = torch.zeros_like(Qe), torch.zeros_like(Qe), torch.zeros_like(Qe)
        amax = float(dY.abs().max())
        gs = 1.0
        if MODE == "p16ds16" and amax > 0:
            gs = 2.0 ** (-3 - math.floor(math.log2(amax)))
        j = JSHIFT if MODE == "p16ds16" else 0
        dYe = pair16(dY * gs) if pair else dY
        delta = (dY * O_).sum(-1)
        for b in range(B):
            idx = keep[b].nonzero()[:, 0]
            qs = pair16(Qe[b] * sl2) if pair else Qe[b] * sl2
            if SB1:
                qs = h(qs)
            s2 = qs @ (h(Ke[b, idx]) if KSINGLE else Ke[b, idx]).T
            P = torch.exp2(s2 - lse2[b][:, None] + j)
            vs = pair16(Ve[b, idx] * scale) if pair else Ve[b, idx] * scale
            if VSINGLE:
                vs = h(vs)
            dP = (h(dYe[b]) if DP1 else dYe[b]) @ vs.T - (delta[b] * scale * gs)[:, None]
            if MODE.startswith("p16"):
                P = h(P)
            if MODE.startswith("p16ds16"):
                dS = h(P * h(dP))
            else:
                dS = P * dP
            un = 1.0 / (gs * 2.0 ** j)
            Pk, dSk = P, dS
            if KSINGLE and os.environ.get("KVSWEEP_SCALED_K", "0") == "1":
                # the dK/dV sweep keeps the SCALED keys resident (one fp16 term of c K) against the streamed unscaled query pair: its scores
                # differ from the forward's (c Q pair x one term of K) by the two different roundings -- emulate that sweep's own P / dS
                s2k = (h(Qe[b]) if SB1 else Qe[b]) @ h(Ke[b, idx] * sl2).T
                Pk = torch.exp2(s2k - lse2[b][:, None] + j)
                dPk = (h(dYe[b]) if DP1 else dYe[b]) @ (h(pair16(Ve[b, idx] * scale)) if VSINGLE else pair16(Ve[b, idx] * scale)).T - (delta[b] * scale * gs)[:, None]
                Pk = h(Pk)
                dSk = h(Pk * h(dPk))
            dV[b, idx] = (Pk.T @ (h(dYe[b]) if DV1 else dYe[b])) * un
            dK[b, idx] = (dSk.T @ (h(Qe[b]) if DK1 else Qe[b])) * un
            dQ[b] = (dS @ (h(Ke[b, idx]) if DQ1 else Ke[b, idx])) * un
        return dQ, dK, dV, None


def mask_attention(x, p, prefix, keep, q_block=None):
    B, C, H, W = x.shape
    N = H * W
    xs = x.reshape(B, C, N).permute(0, 2, 1)
    Q = F.linear(xs, p[prefix + ".query.weight"], p[prefix + ".query.bias"])
    K = F.linear(xs, p[prefix + ".key.weight"], p[prefix + ".key.bias"])
    V = F.linear(xs, p[prefix + ".value.weight"], p[prefix + ".value.bias"])
    out = Attn.apply(Q, K, V, keep) + xs
    out = F.layer_norm(out, (C,), p[prefix + ".norm.weight"], p[prefix + ".norm.bias"], O.LN_EPS)
    return out.reshape(B, C, H, W)


def main():
    name = "unet1_c150_b2_train"
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", name + ".npz"))
    rec = {k: z[k] for k in z.files}
    B, c_out, seed = int(rec["B"]), int(rec["c_out"]), int(rec["seed"])
    p = O.make_params(O.unet_state_shapes(3, c_out, False), seed)
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    keeps = O.make_keeps(seed + 1, B)
    x, labels = O.make_inputs(seed + 2, B, c_out)
    O.mask_attention = mask_attention
    out = O.unet_forward(p, x, keeps, training=True, new_stats={}, three_head=False)
    ref = torch.from_numpy(rec["out0_slice"])
    eo = float((out[:, :, ::16, ::16].detach() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    loss = O.pixel_cross_entropy(out, labels, -100)
    loss.backward()
    gmax = max(float(v) for k, v in rec.items() if k.startswith("gnorm/"))
    floor = 1e-4 * gmax
    worst = (0.0, "")
    for k, v in p.items():
        if "gnorm/" + k not in rec or not bool(rec["ghas/" + k]):
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
    gsl = p["norm.weight"].grad[:, ::16, ::16]
    rs = torch.from_numpy(rec["g_slice/norm.weight"])
    e3 = float((gsl - rs).abs().max()) / float(rs.abs().max())
    print(f"mode {MODE} ksingle {int(KSINGLE)} vsingle {int(VSINGLE)} dv1 {int(DV1)} dk1 {int(DK1)} dq1 {int(DQ1)} dp1 {int(DP1)} sb1 {int(SB1)}: out err {eo:.3e} (gate 1e-3)  loss err {abs(loss.item() - float(rec['loss'])):.3e}  "
          f"worst grad {worst[0]:.3e} [{worst[1]}] (gate 5e-2)  d norm.weight slice {e3:.3e} (gate 1e-1)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
