"""fused loss and metric kernels for the step right after the forward path.

``pixel_cross_entropy_nhwc`` consumes the model's NHWC channel-padded logits directly (``UNet.forward_nhwc``), so the training
step needs neither the NHWC->NCHW fp32 output conversion nor torch's log-softmax / nll kernels.  Same semantics as
``nn.CrossEntropyLoss(ignore_index=...)`` in the reference scripts (ade_semantic.py:377,399; city_semantic.py:341).
"""
from __future__ import annotations

import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call, dt, ptr, stream, workspace


class _PixelCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, n_classes, ignore_index, grad_scale):
        logits = logits.contiguous()
        labels = labels.contiguous().view(-1)
        Cp = logits.shape[-1]
        M = logits.numel() // Cp
        if labels.numel() != M or labels.dtype != torch.int64:
            raise RuntimeError("pixel_cross_entropy_nhwc: labels must be int64 with one entry per pixel")
        lse = torch.empty(M, dtype=torch.float32, device=logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        count = torch.empty(1, dtype=torch.float32, device=logits.device)
        ws = workspace(_lib.load().mu_ce_workspace_bytes(), logits.device)
        call("mu_ce_fwd", ptr(logits), ptr(labels), M, Cp, n_classes, ignore_index, ptr(lse), ptr(loss), ptr(count), ptr(ws),
             ws.numel(), dt(logits), stream())
        ctx.save_for_backward(logits, labels, lse, count)
        ctx.meta = (M, Cp, n_classes, ignore_index, grad_scale)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, lse, count = ctx.saved_tensors
        M, Cp, C, ignore_index, grad_scale = ctx.meta
        dl = torch.empty_like(logits)
        g = g.contiguous().float().view(1)
        call("mu_ce_bwd", ptr(logits), ptr(labels), ptr(lse), ptr(count), ptr(g), float(grad_scale), M, Cp, C, ignore_index, ptr(dl),
             dt(logits), stream())
        return dl, None, None, None, None


def pixel_cross_entropy_nhwc(logits_nhwc, labels, n_classes, ignore_index=-100, grad_scale=1.0):
    """Mean cross-entropy over non-ignored pixels.  logits_nhwc: [B,H,W,Cp] (fp16/fp32, first n_classes channels valid);
    labels: int64 [B,H,W].  grad_scale multiplies the backward only (static fp16 loss scale without touching the loss value)."""
    return _PixelCE.apply(logits_nhwc, labels, int(n_classes), int(ignore_index), float(grad_scale))


class _CrossEntropyNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, grad_scale):
        logits = logits.contiguous()
        labels = labels.contiguous()
        if logits.dim() < 2 or labels.dtype != torch.int64:
            raise RuntimeError("CrossEntropyLoss expects logits [B,C,...] and int64 class-index labels [B,...]")
        B, C = logits.shape[0], logits.shape[1]
        HW = logits.numel() // max(B * C, 1)
        if tuple(labels.shape) != (B,) + tuple(logits.shape[2:]):
            raise RuntimeError(f"CrossEntropyLoss: labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        lse = torch.empty(B * HW, dtype=torch.float32, device=logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        count = torch.empty(1, dtype=torch.float32, device=logits.device)
        ws = workspace(_lib.load().mu_ce_workspace_bytes(), logits.device)
        call("mu_ce_nchw_fwd", ptr(logits), ptr(labels), B, C, HW, ignore_index, ptr(lse), ptr(loss), ptr(count), ptr(ws), ws.numel(),
             dt(logits), stream())
        ctx.save_for_backward(logits, labels, lse, count)
        ctx.meta = (B, C, HW, ignore_index, grad_scale)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, lse, count = ctx.saved_tensors
        B, C, HW, ignore_index, grad_scale = ctx.meta
        dl = torch.empty_like(logits)
        g = g.contiguous().float().view(1)
        call("mu_ce_nchw_bwd", ptr(logits), ptr(labels), ptr(lse), ptr(count), ptr(g), float(grad_scale), B, C, HW, ignore_index, ptr(dl),
             dt(logits), stream())
        return dl, None, None, None


NHWC_SOURCE = os.environ.get("MU_CE_NHWC_SOURCE", "1") != "0"     # debug switch: 0 = always read the NCHW tensor


def _nhwc_source(t):
    """The NHWC tensor a module output was converted from (ops.to_nchw), if `t` is still that untouched output."""
    src = getattr(t, "_mu_nhwc", None)
    if src is None or not NHWC_SOURCE or t.dim() != 4:
        return None
    x, C, version = src
    if t._version != version or x.shape[0] != t.shape[0] or C != t.shape[1] or tuple(x.shape[1:3]) != tuple(t.shape[2:]):
        return None
    return x, C


def cross_entropy(logits, labels, ignore_index=-100, grad_scale=1.0):
    """F.cross_entropy(logits, labels, ignore_index=...) (mean reduction) on the module output as it is: logits [B,C,H,W]
    (fp32 or fp16, NCHW), labels int64 [B,H,W].  grad_scale multiplies the backward only (static fp16 loss scale).
    When `logits` is the untouched output of a maskunet_amd module, the loss reads the NHWC tensor that output was converted from
    (identical values: the NCHW fp32 output is an exact widening of it) and the gradient flows into that tensor directly -- the
    training step then skips the NCHW gradient and its transposition."""
    src = _nhwc_source(logits)
    if src is not None and labels.dim() == 3 and labels.dtype == torch.int64:
        return pixel_cross_entropy_nhwc(src[0], labels, src[1], ignore_index, grad_scale)
    return _CrossEntropyNCHW.apply(logits, labels, int(ignore_index), float(grad_scale))


class CrossEntropyLoss(torch.nn.Module):
    """The reference's criterion, nn.CrossEntropyLoss() / nn.CrossEntropyLoss(ignore_index=255) (ade_semantic.py:377,399;
    city_semantic.py:341,362), as one HIP sweep forward and one backward over the NCHW logits (class weights, label smoothing
    and probability targets are not used by the reference and are not supported)."""

    def __init__(self, ignore_index: int = -100, grad_scale: float = 1.0):
        super().__init__()
        self.ignore_index, self.grad_scale = int(ignore_index), float(grad_scale)

    def forward(self, input, target):
        return cross_entropy(input, target, self.ignore_index, self.grad_scale)


def mean_iou(y_pred, y_true, num_classes, smooth=1e-6):
    """mean_iou of the reference (ade_semantic.py:128-146) on device, no host synchronisation.
    y_pred: NCHW [B,C,H,W] (the module output) or NHWC channel-padded [B,H,W,Cp]; y_true: int64 [B,H,W]."""
    y_pred = y_pred.contiguous()
    labels = y_true.contiguous().view(-1)
    M = labels.numel()
    if y_pred.dim() != 4:
        raise RuntimeError("mean_iou expects a 4-D prediction tensor")
    if y_pred.shape[1] == num_classes and y_pred.shape[0] * y_pred.shape[2] * y_pred.shape[3] == M:   # NCHW
        hw = y_pred.shape[2] * y_pred.shape[3]
        inner, outer, cs, ps = hw, num_classes * hw, hw, 1
    else:                                                                                              # NHWC padded
        Cp = y_pred.shape[-1]
        inner, outer, cs, ps = M, 0, 1, Cp
    counts = torch.empty(3 * num_classes, dtype=torch.int32, device=y_pred.device)
    out = torch.empty(1, dtype=torch.float32, device=y_pred.device)
    call("mu_mean_iou", ptr(y_pred), ptr(labels), M, num_classes, inner, outer, cs, ps, float(smooth), ptr(counts), ptr(out),
         dt(y_pred), stream())
    return out.view(())


class _InstanceTriplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, mask, u, margin, ignore_index, id_cap, max_inst):
        features = features.contiguous()
        mask = mask.contiguous()
        if features.dtype != torch.float32 or features.dim() != 4 or not features.is_cuda:
            raise RuntimeError("InstanceContrastiveLoss expects fp32 CUDA features [B,C,H,W] (the model's NCHW output)")
        B, C, H, W = features.shape
        if mask.dtype != torch.int64 or tuple(mask.shape) != (B, H, W):
            raise RuntimeError("InstanceContrastiveLoss: instance_mask must be int64 [B,H,W]")
        ws = torch.empty(_lib.load().mu_inst_triplet_workspace_bytes(id_cap, max_inst), dtype=torch.uint8, device=features.device)
        loss = torch.empty(1, dtype=torch.float32, device=features.device)
        call("mu_inst_triplet_fwd", ptr(features), ptr(mask), B, C, H, W, -1 if ignore_index is None else int(ignore_index), float(margin),
             ptr(u), id_cap, max_inst, ptr(ws), ws.numel(), ptr(loss), stream())
        ctx.save_for_backward(features, ws)
        ctx.meta = (id_cap, max_inst)
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        features, ws = ctx.saved_tensors
        id_cap, max_inst = ctx.meta
        B, C, H, W = features.shape
        g = g.contiguous().float().view(1)
        df = torch.empty_like(features)
        call("mu_inst_triplet_bwd", ptr(features), B, C, H, W, ptr(ws), id_cap, max_inst, ptr(g), ptr(df), stream())
        return df, None, None, None, None, None, None


class InstanceContrastiveLoss(torch.nn.Module):
    """Device-side InstanceContrastiveLoss (ade_panoptic.py:390-418; `ignore_index=255`: city_instance.py:279-307): same
    constructor and call signature as the reference class, no torch.unique / nonzero host round trips.

    The negative pixel of the k-th instance is floor(u[k] * n_neg) with u ~ U[0,1) drawn on the device per call (pass `u` to fix it);
    the reference draws torch.randint on the host.  Instance ids must lie in [0, id_cap)."""

    def __init__(self, margin=1.0, ignore_index=None, id_cap=65536, max_instances=1024):
        super().__init__()
        self.margin, self.ignore_index, self.id_cap, self.max_instances = margin, ignore_index, id_cap, max_instances

    def forward(self, features, instance_mask, u=None):
        if u is None:
            u = torch.rand(self.max_instances, dtype=torch.float32, device=features.device)
        elif u.numel() < self.max_instances:
            u = torch.cat([u.float().to(features.device), torch.zeros(self.max_instances - u.numel(), device=features.device)])
        return _InstanceTriplet.apply(features, instance_mask, u.contiguous(), self.margin, self.ignore_index, self.id_cap,
                                      self.max_instances)
