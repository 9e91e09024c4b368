"""CPU oracle for the MaskAttn-UNet forward/backward path.  TEST INFRASTRUCTURE ONLY.

This file is a functional CPU restatement (stock torch ops, fp32 or fp64) of the
reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it -- it is the checker, never the
product path (``maskunet_amd`` never imports anything from ``oracle/``).

Parity status: PINNED.  ``tests/golden/make_golden.py`` (run in the build container,
where ``/root/reference`` is mounted) AST-extracts the reference's own classes
(SURVEY.md appendix A), runs them on seeded inputs and commits the inputs/outputs/grads
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below
against those vectors.  The reference ships no tests / golden vectors of its own
(SURVEY.md section 4), so vectors generated from the reference itself are the pin.

Every function cites the reference lines it restates (paths under /root/reference/).
All tensors are NCHW exactly as in the reference.  Backward is torch autograd over
these functions, which is precisely how the reference defines its backward
(``loss.backward()``, code/ade20k/ade_semantic.py:400).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BN_EPS = 1e-5       # nn.BatchNorm2d default, ade_semantic.py:200
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default
LN_EPS = 1e-5       # nn.LayerNorm default, ade_semantic.py:161,281


# --------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------
def batchnorm2d(x, p: Params, prefix: str, training: bool, new_stats: Optional[dict] = None):
    """nn.BatchNorm2d (ade_semantic.py:200,204,219,240,285).

    training: biased batch variance normalises; the running_var update uses the unbiased
    variance, momentum 0.1.  eval: running statistics.
    """
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
                new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
    else:
        mean, var = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def gelu(x):
    """nn.GELU() / F.gelu exact erf form (ade_semantic.py:201,208)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def conv_block(x, p: Params, prefix: str, residual: bool, training: bool, new_stats=None):
    """ConvBlock.forward (ade_semantic.py:192-210): conv3x3(no bias) BN GELU conv3x3 BN,
    and ``gelu(x + block(x))`` when residual (the add is BEFORE the final GELU, :208)."""
    y = F.conv2d(x, p[prefix + ".conv_block.0.weight"], None, padding=1)
    y = batchnorm2d(y, p, prefix + ".conv_block.1", training, new_stats)
    y = gelu(y)
    y = F.conv2d(y, p[prefix + ".conv_block.3.weight"], None, padding=1)
    y = batchnorm2d(y, p, prefix + ".conv_block.4", training, new_stats)
    if residual:
        return gelu(x + y)
    return y


def downsample(x, p: Params, prefix: str, training: bool, new_stats=None):
    """DownSample.forward (ade_semantic.py:212-229): MaxPool2d(2) -> ConvBlock(in,in,residual)
    -> ConvBlock(in,out) -> BatchNorm2d(out).  emb_layer is never called (:222-225)."""
    x = F.max_pool2d(x, 2)
    x = conv_block(x, p, prefix + ".maxpool_conv.1", True, training, new_stats)
    x = conv_block(x, p, prefix + ".maxpool_conv.2", False, training, new_stats)
    return batchnorm2d(x, p, prefix + ".maxpool_conv.3", training, new_stats)


def upsample_bilinear2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (ade_semantic.py:235):
    src = dst * (in-1)/(out-1), separable linear interpolation."""
    B, C, H, W = x.shape

    def axis(n_in):
        n_out = 2 * n_in
        if n_in == 1:
            z = torch.zeros(n_out, dtype=torch.long)
            return z, z, torch.zeros(n_out, dtype=x.dtype)
        src = torch.arange(n_out, dtype=torch.float64) * ((n_in - 1) / (n_out - 1))
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, (src - i0).to(x.dtype)

    h0, h1, fh = axis(H)
    w0, w1, fw = axis(W)
    rows = x[:, :, h0, :] * (1 - fh)[None, None, :, None] + x[:, :, h1, :] * fh[None, None, :, None]
    return rows[:, :, :, w0] * (1 - fw) + rows[:, :, :, w1] * fw


def upsample(x, skip, p: Params, prefix: str, training: bool, new_stats=None):
    """UpSample.forward (ade_semantic.py:231-256): bilinear x2 -> cat([skip, up], dim=1) (:253)
    -> ConvBlock(in,in,residual) -> ConvBlock(in,out,mid=in//2) -> BatchNorm2d(out)."""
    x = upsample_bilinear2x(x)
    x = torch.cat([skip, x], dim=1)
    x = conv_block(x, p, prefix + ".conv.0", True, training, new_stats)
    x = conv_block(x, p, prefix + ".conv.1", False, training, new_stats)
    return batchnorm2d(x, p, prefix + ".conv.2", training, new_stats)


def additive_mask(keep):
    """keep [B,N] (1 = key visible) -> additive key mask {0,-inf} [B,1,N]
    (ade_semantic.py:178-181: where(mask > 0.5, 0, -inf).unsqueeze(1).expand(-1,N,-1))."""
    z = torch.zeros((), dtype=torch.float32)
    ninf = torch.full((), -float("inf"), dtype=torch.float32)
    return torch.where(keep > 0, z, ninf).unsqueeze(1)


def mask_attention(x, p: Params, prefix: str, keep, q_block: Optional[int] = None):
    """Mask2FormerAttention.forward (ade_semantic.py:163-190).

    x [B,C,H,W] -> tokens [B,N,C] (:168); Q,K,V = Linear(C,C) with bias (:170-172);
    scores = QK^T / sqrt(C) (:174-175) + additive key mask (:183); softmax over keys (:185);
    PV (:186) + tokens (:187); LayerNorm([C]) (:188); then ``.view(B,C,H,W)`` of the
    [B,N,C] tensor WITHOUT permuting back (:190) -- the layout scramble is reproduced.

    q_block: if given, queries are processed in chunks of that size (same math, O(q_block*N)
    memory) -- used where the N x N matrix does not fit (256x256 inputs).
    """
    B, C, H, W = x.shape
    if C != p[prefix + ".query.weight"].shape[1]:
        raise ValueError("Input channel size does not match initialized channel size.")
    N = H * W
    xs = x.reshape(B, C, N).permute(0, 2, 1)
    Q = F.linear(xs, p[prefix + ".query.weight"], p[prefix + ".query.bias"])
    K = F.linear(xs, p[prefix + ".key.weight"], p[prefix + ".key.bias"])
    V = F.linear(xs, p[prefix + ".value.weight"], p[prefix + ".value.bias"])
    add = additive_mask(keep).to(x.dtype)
    if q_block is None or q_block >= N:
        scores = torch.matmul(Q, K.transpose(-2, -1)) / (C ** 0.5)
        attn = torch.softmax(scores + add, dim=-1)
        out = torch.matmul(attn, V)
    else:
        outs = []
        Kt = K.transpose(-2, -1)
        for s in range(0, N, q_block):
            sc = torch.matmul(Q[:, s:s + q_block], Kt) / (C ** 0.5)
            outs.append(torch.matmul(torch.softmax(sc + add, dim=-1), V))
        out = torch.cat(outs, dim=1)
    out = out + xs
    out = F.layer_norm(out, (C,), p[prefix + ".norm.weight"], p[prefix + ".norm.bias"], LN_EPS)
    return out.reshape(B, C, H, W)          # == .view on the contiguous [B,N,C] result


def head_1x1_bn_relu(x, p: Params, prefix: str, training: bool, new_stats=None):
    """final_layer / embedding_head: Conv2d 1x1 (bias) -> BatchNorm2d -> ReLU
    (ade_semantic.py:283-287; city_instance.py:248-252)."""
    y = F.conv2d(x, p[prefix + ".0.weight"], p[prefix + ".0.bias"])
    y = batchnorm2d(y, p, prefix + ".1", training, new_stats)
    return torch.relu(y)


def boundary_head(x, p: Params, prefix: str, training: bool, new_stats=None):
    """boundary_head: Conv3x3(c_out->32,bias,pad1) BN ReLU Conv1x1(32->1,bias)
    (city_instance.py:242-247)."""
    y = F.conv2d(x, p[prefix + ".0.weight"], p[prefix + ".0.bias"], padding=1)
    y = batchnorm2d(y, p, prefix + ".1", training, new_stats)
    y = torch.relu(y)
    return F.conv2d(y, p[prefix + ".3.weight"], p[prefix + ".3.bias"])


# --------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------
def unet_forward(p: Params, x, keeps: Sequence[torch.Tensor], *, training: bool,
                 dropout_masks: Optional[Sequence[torch.Tensor]] = None, p_drop: float = 0.3,
                 new_stats: Optional[dict] = None, q_block: Optional[int] = None,
                 three_head: bool = False):
    """UNet.forward, 1-head (ade_semantic.py:289-314) or 3-head (city_instance.py:253-276).

    keeps: six keep-masks [B,N_k] for self_attention1..6 (the reference draws them with
    torch.randint on first use, :177-181; the oracle takes them explicitly).
    dropout_masks: two {0,1} tensors shaped like the outputs of upsample1 / upsample2; the
    reference's nn.Dropout(0.3) (:273,304,307) is ``x * mask / (1-p)`` in training and the
    identity in eval.  None = dropout disabled.
    """
    def drop(t, i):
        if not training or dropout_masks is None:
            return t
        return t * dropout_masks[i].to(t.dtype) / (1.0 - p_drop)

    x1 = conv_block(x, p, "initial_conv", False, training, new_stats)
    x2 = downsample(x1, p, "downsample1", training, new_stats)
    x2 = mask_attention(x2, p, "self_attention1", keeps[0], q_block)
    x3 = downsample(x2, p, "downsample2", training, new_stats)
    x3 = mask_attention(x3, p, "self_attention2", keeps[1], q_block)
    x4 = downsample(x3, p, "downsample3", training, new_stats)
    x4 = mask_attention(x4, p, "self_attention3", keeps[2], q_block)

    x4 = conv_block(x4, p, "bottom1", False, training, new_stats)
    x4 = conv_block(x4, p, "bottom2", False, training, new_stats)
    x4 = conv_block(x4, p, "bottom3", False, training, new_stats)

    y = upsample(x4, x3, p, "upsample1", training, new_stats)
    y = drop(y, 0)
    y = mask_attention(y, p, "self_attention4", keeps[3], q_block)
    y = upsample(y, x2, p, "upsample2", training, new_stats)
    y = drop(y, 1)
    y = mask_attention(y, p, "self_attention5", keeps[4], q_block)
    y = upsample(y, x1, p, "upsample3", training, new_stats)
    y = mask_attention(y, p, "self_attention6", keeps[5], q_block)
    # LayerNorm([64,H,W]) per sample with full-shape affine (ade_semantic.py:281,311)
    y = F.layer_norm(y, tuple(y.shape[1:]), p["norm.weight"], p["norm.bias"], LN_EPS)
    if not three_head:
        return head_1x1_bn_relu(y, p, "final_layer", training, new_stats)
    emb = head_1x1_bn_relu(y, p, "embedding_head", training, new_stats)
    sem = head_1x1_bn_relu(y, p, "final_layer", training, new_stats)
    bnd = boundary_head(sem, p, "boundary_head", training, new_stats)
    return sem, bnd, emb


def pixel_cross_entropy(logits, labels, ignore_index: int = -100):
    """nn.CrossEntropyLoss() on [B,c_out,H,W] logits vs [B,H,W] int64 labels, mean over
    non-ignored pixels (ade_semantic.py:377,399; ignore_index=255 in city_semantic.py:341)."""
    return F.cross_entropy(logits, labels, ignore_index=ignore_index)


def mean_iou(y_pred, y_true, num_classes, smooth=1e-6):
    """mean_iou (ade_semantic.py:128-146): softmax(y/0.5) -> argmax -> per-class (I+s)/(U+s), classes with U == 0 skipped."""
    pred = torch.argmax(torch.softmax(y_pred / 0.5, dim=1), dim=1)
    ious = []
    for c in range(num_classes):
        inter = torch.sum((pred == c) & (y_true == c))
        union = torch.sum((pred == c) | (y_true == c))
        if union == 0:
            continue
        ious.append((inter.float() + smooth) / (union.float() + smooth))
    return torch.mean(torch.stack(ious))


def instance_contrastive_loss(features, instance_mask, u, margin: float = 1.0, ignore_index=None):
    """InstanceContrastiveLoss (ade_panoptic.py:390-418 and coco_panoptic.py:482-521: every id of the mask, `ignore_index=None`;
    city_instance.py:279-307: `valid_mask = instance_mask != 255`, i.e. `ignore_index=255`): for every instance id != 0 with at least two pixels, a triplet margin loss between the feature
    columns addressed by the FIRST two pixels of the instance and by one random pixel outside it, averaged over instances.

    Faithful to the reference's indexing: `nonzero(as_tuple=True)` of the [B,H,W] mask yields (batch, row, col) index
    vectors and the reference uses entries [0] and [1] -- i.e. (batch, row) of a pixel -- as (h, w) into
    features[:, :, h, w], so each "pixel feature" is the [B*C] column features[:, :, b_pix, row_pix].
    The reference draws the negative with torch.randint(0, n_neg); here the k-th instance that reaches the draw takes
    index floor(u[k] * n_neg) (the golden generator patches torch.randint the same way)."""
    if ignore_index is None:
        ids = torch.unique(instance_mask)
    else:
        ids = torch.unique(instance_mask[instance_mask != ignore_index])
    loss = features.new_zeros(())
    count = 0
    k = 0
    for inst in ids.tolist():
        if inst == 0:
            continue
        pb, ph, _ = (instance_mask == inst).nonzero(as_tuple=True)
        if pb.numel() < 2:
            continue
        nb, nh, _ = (instance_mask != inst).nonzero(as_tuple=True)
        if nb.numel() == 0:
            continue
        j = min(int(float(u[k]) * nb.numel()), nb.numel() - 1)
        k += 1
        anchor = features[:, :, pb[0], ph[0]].reshape(1, -1)
        positive = features[:, :, pb[1], ph[1]].reshape(1, -1)
        negative = features[:, :, nb[j], nh[j]].reshape(1, -1)
        d_ap = F.pairwise_distance(anchor, positive, p=2.0, eps=1e-6)
        d_an = F.pairwise_distance(anchor, negative, p=2.0, eps=1e-6)
        loss = loss + torch.clamp(d_ap - d_an + margin, min=0.0).mean()
        count += 1
    return loss / count if count > 0 else features.new_zeros(())


# --------------------------------------------------------------------------------------
# deterministic parameter / input recipe shared by the golden generator and the tests
# --------------------------------------------------------------------------------------
def _conv_block_shapes(prefix, cin, cout, mid=None):
    mid = mid or cout
    out = [(prefix + ".conv_block.0.weight", (mid, cin, 3, 3), "w")]
    out += _bn_shapes(prefix + ".conv_block.1", mid)
    out += [(prefix + ".conv_block.3.weight", (cout, mid, 3, 3), "w")]
    out += _bn_shapes(prefix + ".conv_block.4", cout)
    return out


def _bn_shapes(prefix, c):
    return [(prefix + ".weight", (c,), "gamma"), (prefix + ".bias", (c,), "beta"),
            (prefix + ".running_mean", (c,), "rmean"), (prefix + ".running_var", (c,), "rvar"),
            (prefix + ".num_batches_tracked", (), "count")]


def _attn_shapes(prefix, c):
    out = []
    for nm in ("query", "key", "value"):
        out += [(f"{prefix}.{nm}.weight", (c, c), "w"), (f"{prefix}.{nm}.bias", (c,), "b%d" % c)]
    out += [(prefix + ".norm.weight", (c,), "gamma"), (prefix + ".norm.bias", (c,), "beta")]
    return out


def _down_shapes(prefix, cin, cout, emb=256):
    out = _conv_block_shapes(prefix + ".maxpool_conv.1", cin, cin)
    out += _conv_block_shapes(prefix + ".maxpool_conv.2", cin, cout)
    out += _bn_shapes(prefix + ".maxpool_conv.3", cout)
    out += [(prefix + ".emb_layer.1.weight", (cout, emb), "w"), (prefix + ".emb_layer.1.bias", (cout,), "b%d" % emb)]
    return out


def _up_shapes(prefix, cin, cout, emb=256):
    out = _conv_block_shapes(prefix + ".conv.0", cin, cin)
    out += _conv_block_shapes(prefix + ".conv.1", cin, cout, cin // 2)
    out += _bn_shapes(prefix + ".conv.2", cout)
    out += [(prefix + ".emb_layer.1.weight", (cout, emb), "w"), (prefix + ".emb_layer.1.bias", (cout,), "b%d" % emb)]
    return out


def unet_state_shapes(c_in=3, c_out=3, three_head=False, embed_dim=16, hw=128):
    """Ordered (key, shape, kind) list == the reference UNet.state_dict() layout
    (ade_semantic.py:259-287; city_instance.py:217-252).  Checked against the real
    reference state_dict in tests/golden/make_golden.py."""
    s = _conv_block_shapes("initial_conv", c_in, 64)
    s += _down_shapes("downsample1", 64, 128) + _attn_shapes("self_attention1", 128)
    s += _down_shapes("downsample2", 128, 256) + _attn_shapes("self_attention2", 256)
    s += _down_shapes("downsample3", 256, 256) + _attn_shapes("self_attention3", 256)
    s += _conv_block_shapes("bottom1", 256, 512) + _conv_block_shapes("bottom2", 512, 512)
    s += _conv_block_shapes("bottom3", 512, 256)
    s += _up_shapes("upsample1", 512, 128) + _attn_shapes("self_attention4", 128)
    s += _up_shapes("upsample2", 256, 64) + _attn_shapes("self_attention5", 64)
    s += _up_shapes("upsample3", 128, 64) + _attn_shapes("self_attention6", 64)
    s += [("norm.weight", (64, hw, hw), "gamma"), ("norm.bias", (64, hw, hw), "beta")]
    s += [("final_layer.0.weight", (c_out, 64, 1, 1), "w"), ("final_layer.0.bias", (c_out,), "b64")]
    s += _bn_shapes("final_layer.1", c_out)
    if three_head:
        s += [("boundary_head.0.weight", (32, c_out, 3, 3), "w"), ("boundary_head.0.bias", (32,), "b%d" % (9 * c_out))]
        s += _bn_shapes("boundary_head.1", 32)
        s += [("boundary_head.3.weight", (1, 32, 1, 1), "w"), ("boundary_head.3.bias", (1,), "b32")]
        s += [("embedding_head.0.weight", (embed_dim, 64, 1, 1), "w"), ("embedding_head.0.bias", (embed_dim,), "b64")]
        s += _bn_shapes("embedding_head.1", embed_dim)
    return s


def make_tensor(rng, shape, kind):
    """One tensor of the recipe.  Distributions follow the torch default init ranges
     but BN/LN affine and running statistics are made non-trivial so a
    dropped gamma/beta/mean/var cannot pass a parity test."""
    import numpy as np
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        bound = 1.0 / math.sqrt(fan_in)
        a = rng.uniform(-bound, bound, size=shape)
    elif kind == "gamma":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif kind == "beta":
        a = rng.uniform(-0.2, 0.2, size=shape)
    elif kind[0] == "b" and kind[1:].isdigit():      # bias of a layer with that fan_in
        bound = 1.0 / math.sqrt(int(kind[1:]))
        a = rng.uniform(-bound, bound, size=shape)
    elif kind == "rmean":
        a = rng.normal(0.0, 0.1, size=shape)
    elif kind == "rvar":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif kind == "count":
        return torch.zeros((), dtype=torch.int64)
    else:
        raise KeyError(kind)
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def make_params(shapes, seed: int) -> Params:
    """Deterministic parameters from numpy default_rng(seed), drawn in list order."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return {k: make_tensor(rng, shp, kind) for k, shp, kind in shapes}


def make_keeps(seed: int, B: int, hw: int = 128) -> List[torch.Tensor]:
    """Six Bernoulli(0.5) keep-masks [B,N] for self_attention1..6."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ns = [(hw // 2) ** 2, (hw // 4) ** 2, (hw // 8) ** 2, (hw // 4) ** 2, (hw // 2) ** 2, hw ** 2]
    return [torch.from_numpy(rng.integers(0, 2, size=(B, n)).astype(np.uint8)) for n in ns]


def make_inputs(seed: int, B: int, c_out: int, hw: int = 128, ignore_frac: float = 0.0):
    """Synthetic images uniform [0,1) float32 (ToTensor range, ade_semantic.py:85) and int64 labels."""
    import numpy as np
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((B, 3, hw, hw), dtype=np.float32))
    y = rng.integers(0, c_out, size=(B, hw, hw))
    if ignore_frac > 0:
        y[rng.random((B, hw, hw)) < ignore_frac] = 255
    return x, torch.from_numpy(y.astype(np.int64))
