"""torch.autograd.Function ops over the C ABI (include/maskunet_hip.h).  All activations here are
NHWC tensors [B,H,W,C] (or token-major [B,N,C]) in the compute dtype (fp32 or fp16) whose channel
count is a multiple of 32; parameters are the fp32 torch parameters in reference layout (OIHW).

Each op cites the reference lines it replaces.  There is no eager/CPU fallback anywhere: every path
ends in `_lib.call`, which raises if the HIP library is missing or a kernel reports an error.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, call, dt, mdt, ptr, stream, workspace  # noqa: F401


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def _pad_vec(v, n, fill):
    """fp32 per-channel vector padded to n entries (pad value `fill`)."""
    v = v.detach().float()
    if v.numel() == n:
        return v.contiguous()
    return F.pad(v, (0, n - v.numel()), value=fill)


def _is_x(t_or_dtype) -> bool:
    """fp32 compute with the split-bf16 matrix products (set_float32_matmul_precision("high")): matrix operands travel chunk-encoded."""
    return mdt(t_or_dtype) == _lib.MU_F32X


def _enc(t):
    """Chunk-encoded copy of a matrix operand for the MU_F32X entry points (include/maskunet_hip.h); the tensor itself otherwise."""
    if not _is_x(t):
        return t
    t = t.contiguous()
    e = torch.empty_like(t)
    call("mu_split_encode", ptr(t), ptr(e), t.numel(), stream())
    return e


def _enc3(t):
    """fp32x: the 3x3-convolution operand encoding ([4 fp16 hi | 4 fp16 lo] per chunk, round 6) of an activation; the tensor itself otherwise."""
    if not _is_x(t):
        return t
    t = t.contiguous()
    e = torch.empty_like(t)
    call("mu_split_encode_h4", ptr(t), ptr(e), t.numel(), stream())
    return e


def _enc3x(t):
    """fp32x: (the 3x3 operand encoding of t, the fp16 rounding of t as plain rows) from one pass -- the first feeds the forward conv, the
    second is what its backward keeps for the one-term weight gradient (mu_conv_wgrad_h1)."""
    t = t.contiguous()
    e = torch.empty_like(t)
    t16 = torch.empty(t.shape, dtype=torch.float16, device=t.device)
    call("mu_split_encode_h4x", ptr(t), ptr(e), ptr(t16), t.numel(), stream())
    return e, t16


def _enc_for(t, taps):
    """The operand encoding a layer with `taps` taps reads: fp16 pairs for the 3x3 layers, bf16 pairs for 1x1 / Linear."""
    return _enc3(t) if taps == 9 else _enc(t)


def dy_encode_h(gy):
    """fp32x, 3x3 backward: a plain fp32 gradient as ONE power-of-two-scaled fp16 operand -> (tensor holding the halves, scale pair).  The
    halves live at the start of an fp32-typed buffer of gy's shape (the form the BatchNorm backward hands over through EncLink)."""
    gy = gy.contiguous()
    out = torch.empty_like(gy)
    scale = torch.empty(2, dtype=torch.float32, device=gy.device)
    ws = torch.empty(_lib.load().mu_dy_encode_h_workspace_bytes(), dtype=torch.uint8, device=gy.device)      # (own buffer: also called on the side stream)
    call("mu_dy_encode_h", ptr(gy), ptr(out), ptr(scale), gy.numel(), ptr(ws), ws.numel(), stream())
    return out, scale


def _enc_(t):
    """The same in place (buffers nobody reads as plain fp32 afterwards: prepared weights, qkv)."""
    if _is_x(t):
        call("mu_split_encode", ptr(t), ptr(t), t.numel(), stream())
    return t


def _enc_h_(t):
    """fp32x: the ATTENTION operand encoding ([8 fp16 hi | 8 fp16 lo] per eight fp32 values, include/maskunet_hip.h), in place -- qkv,
    which only the attention sweeps ever read."""
    if _is_x(t):
        call("mu_split_encode_h", ptr(t), ptr(t), t.numel(), stream())
    return t


class GradLink:
    """Side channel for ONE gradient tensor between two autograd nodes of one backward pass.

    Where a tensor has two consumers (a residual ConvBlock reads its input twice, ade_semantic.py:208; a skip connection feeds
    DownSample and UpSample, :292-309) autograd joins the two gradients with an elementwise kernel of its own (9 per step in the
    UNet: 0.35 ms).  The consumer whose backward runs FIRST puts its gradient here and returns None to autograd; the backward of the
    tensor's producer-side neighbour -- which data dependence orders later -- takes it and adds it inside its own kernel
    (mu_maxpool2_bwd_acc / mu_upcat_bwd_acc).  Links are made per forward call by the modules that own both ends (DownSample,
    UpSample, UNet); a link nobody fills is simply empty (take() -> None), and `armed` says a taker exists at all.

    A fill belongs to the backward pass (autograd graph task) that made it: a partial backward over a retained graph
    (`torch.autograd.grad(loss, late_params, retain_graph=True)`) can run the filler and prune the taker, and the gradient left
    behind must not be added to the NEXT pass's -- put() and take() drop a fill that carries another pass's id.
    Limitation (documented in INTEGRATION.md): the joined gradient bypasses autograd, so a tensor hook / retain_grad() on the linked
    tensors (the pooled tensor, the skip tensors x1..x3) sees only the part that still flows through autograd; MU_GRAD_LINKS=0
    restores autograd's own accumulation for such debugging."""
    __slots__ = ("t", "armed", "task")

    def __init__(self):
        self.t = None
        self.armed = False
        self.task = -1

    def put(self, t):
        task = _graph_task_id()
        if self.t is not None and self.task == task:
            self.t = self.t + t                              # two fills within one pass
        else:
            self.t, self.task = t, task                      # first fill of this pass (a stale fill of an earlier pass is dropped)

    def take(self):
        t, self.t = self.t, None
        return t if self.task == _graph_task_id() else None


_graph_task_id = getattr(torch._C, "_current_graph_task_id", lambda: -1)     # id of the running backward pass (-1 outside one)


class EncLink:
    """fp32x: tells a 3x3 conv's backward that the dy it is handed was WRITTEN as one scaled fp16 operand by the backward of the BatchNorm
    behind that conv (mu_bn_act_bwd_h) -- the BatchNorm's dx is the conv's dy and has no other consumer (ConvBlock wiring), so the separate
    encoding pass disappears.  Carries the device-side scale pair {S, 1 / S} of that tensor.  One link per (conv, BatchNorm) pair and
    forward call; valid only within the backward pass that marked it."""
    __slots__ = ("task", "scale")

    def __init__(self):
        self.task = None
        self.scale = None

    def mark(self, scale):
        self.task = _graph_task_id()
        self.scale = scale

    def take(self):
        """-> the scale pair if this backward pass marked the link, else None"""
        t, sc = self.task, self.scale
        self.task = self.scale = None
        return sc if (t is not None and t == _graph_task_id()) else None


def enc_link(x):
    """A fresh EncLink when x is an fp32 tensor in the fp32x mode and a backward can follow, else None (also None with the side-stream
    weight gradient, MU_WGRAD_SIDE=1, which reads dy plain)."""
    return EncLink() if (FUSED_ENCODE and not WGRAD_SIDE_STREAM and torch.is_grad_enabled() and _is_x(x)) else None


def _same_mode(ctx_is_x, t):
    """The fp32 matmul precision is process-wide state read in forward AND backward: saved operands were encoded (or not) under the
    forward's mode, so a backward under the other mode would mix encoded tensors with plain-fp32 kernels.  Refuse loudly."""
    if t.dtype == torch.float32 and bool(ctx_is_x) != _is_x(t):
        raise RuntimeError("maskunet_amd: set_float32_matmul_precision() changed between a forward pass and its backward")


FUSED_ENCODE = os.environ.get("MU_FUSED_ENCODE", "1") != "0"     # debug switch: 0 = every operand through mu_split_encode
GRAD_LINKS = os.environ.get("MU_GRAD_LINKS", "1") != "0"      # debug switch: 0 = leave every gradient join to autograd


def grad_link(*tensors):
    """A fresh link when a backward pass can follow (grad mode on and one of `tensors` requires grad), else None."""
    if not GRAD_LINKS or not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in tensors):
        return None
    return GradLink()


# ------------------------------------------------------------------------------------------------
# layout conversion at the module boundary
# ------------------------------------------------------------------------------------------------
class _ToNHWC(torch.autograd.Function):
    """NCHW fp32/fp16 [B,C,H,W] -> NHWC [B,H,W,pad32(C)] in the compute dtype (zero padded)."""

    @staticmethod
    def forward(ctx, x, dtype, cpad=None):
        B, C, H, W = x.shape
        x = x.contiguous()
        Cp = pad32(C) if cpad is None else int(cpad)
        y = torch.empty((B, H, W, Cp), dtype=dtype, device=x.device)
        call("mu_transpose_pad", ptr(x), dt(x), H * W, ptr(y), dt(y), Cp, B, C, H * W, Cp, stream())
        ctx.C, ctx.in_dtype = C, x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = g.contiguous()
        B, H, W, Cp = g.shape
        gx = torch.empty((B, ctx.C, H, W), dtype=ctx.in_dtype, device=g.device)
        call("mu_transpose", ptr(g), dt(g), Cp, ptr(gx), dt(gx), H * W, B, H * W, ctx.C, stream())
        return gx, None, None


class _ToNCHW(torch.autograd.Function):
    """NHWC [B,H,W,Cp] -> NCHW [B,C,H,W] (first C channels) in `out_dtype`."""

    @staticmethod
    def forward(ctx, x, C, out_dtype):
        x = x.contiguous()
        B, H, W, Cp = x.shape
        y = torch.empty((B, C, H, W), dtype=out_dtype, device=x.device)
        call("mu_transpose", ptr(x), dt(x), Cp, ptr(y), dt(y), H * W, B, H * W, C, stream())
        ctx.Cp, ctx.in_dtype = Cp, x.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = g.contiguous()
        B, C, H, W = g.shape
        gx = torch.empty((B, H, W, ctx.Cp), dtype=ctx.in_dtype, device=g.device)
        call("mu_transpose_pad", ptr(g), dt(g), H * W, ptr(gx), dt(gx), ctx.Cp, B, C, H * W, ctx.Cp, stream())
        return gx, None, None


def to_nhwc(x, dtype, cpad=None):
    """cpad: stored channel count (default pad32(C)); a wider zero padding for the ops that run on fixed widths (attn_width)."""
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    return _ToNHWC.apply(x, dtype, cpad)


def u8_hwc_to_nhwc(img_u8, dtype):
    """uint8 [B,H,W,C] image bytes (HWC, as decoded) -> [0,1] activations [B,H,W,pad32(C)] in the compute dtype
    (the ToTensor() of the reference datasets, ade_semantic.py:85, without an fp32 NCHW detour).  No gradient."""
    img_u8 = img_u8.contiguous()
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4:
        raise RuntimeError("u8_hwc_to_nhwc expects a uint8 [B,H,W,C] tensor")
    B, H, W, C = img_u8.shape
    y = torch.empty((B, H, W, pad32(C)), dtype=dtype, device=img_u8.device)
    call("mu_u8_to_nhwc", ptr(img_u8), ptr(y), B * H * W, C, pad32(C), dt(y), stream())
    return y


def resize_u8_to_nhwc(img_u8, size, dtype, bgr=False, return_u8=False):
    """Decoded image bytes uint8 [B,Hs,Ws,C] (HWC, any size) -> network input [B,Hd,Wd,pad32(C)] in the compute dtype: the dataset
    pipeline of the reference on the device -- cv2.cvtColor(BGR2RGB) when bgr=True (ade_semantic.py:65), cv2.resize(.., size,
    INTER_LINEAR) (:72, OpenCV's 8-bit fixed-point algorithm), ToTensor() (:85).  size = (width, height) like cv2.  No gradient."""
    img_u8 = img_u8.contiguous()
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] > 4:
        raise RuntimeError("resize_u8_to_nhwc expects a uint8 [B,H,W,C<=4] tensor")
    B, Hs, Ws, C = img_u8.shape
    Wd, Hd = int(size[0]), int(size[1])
    y = torch.empty((B, Hd, Wd, pad32(C)), dtype=dtype, device=img_u8.device)
    u8 = torch.empty((B, Hd, Wd, C), dtype=torch.uint8, device=img_u8.device) if return_u8 else None
    call("mu_resize_u8_nhwc", ptr(img_u8), B, Hs, Ws, C, int(bool(bgr)), ptr(y), ptr(u8), Hd, Wd, pad32(C), dt(y), stream())
    return (y, u8) if return_u8 else y


def resize_labels_u8(mask_u8, size):
    """Label map uint8 [B,Hs,Ws] -> int64 [B,Hd,Wd]: torch.from_numpy(cv2.resize(mask, size, interpolation=cv2.INTER_NEAREST)).long()
    (ade_semantic.py:73,78) on the device; size = (width, height)."""
    mask_u8 = mask_u8.contiguous()
    if mask_u8.dtype != torch.uint8 or mask_u8.dim() != 3:
        raise RuntimeError("resize_labels_u8 expects a uint8 [B,H,W] tensor")
    B, Hs, Ws = mask_u8.shape
    Wd, Hd = int(size[0]), int(size[1])
    out = torch.empty((B, Hd, Wd), dtype=torch.int64, device=mask_u8.device)
    call("mu_resize_nearest_u8", ptr(mask_u8), B, Hs, Ws, ptr(out), Hd, Wd, stream())
    return out


def to_nchw(x, C, out_dtype=torch.float32):
    """NHWC channel-padded -> NCHW [B,C,H,W] (the layout the reference's callers see).  The result remembers its NHWC source
    (`_mu_nhwc` = (source, C, version)): maskunet_amd.cross_entropy / mean_iou use the source when they are handed the untouched
    output, which is the same numbers in the layout the kernels prefer."""
    y = _ToNCHW.apply(x, C, out_dtype)
    y._mu_nhwc = (x, C, y._version)
    return y


# ------------------------------------------------------------------------------------------------
# convolution / linear
# ------------------------------------------------------------------------------------------------
# Prepared (compute-layout) weights are cached on the parameter for forwards under torch.no_grad() / inference_mode -- the validation
# loop of the reference scripts (`model.eval(); with torch.no_grad(): ...`, ade_semantic.py:421-430) converts the 45 weight tensors once
# instead of once per batch.  The entry is keyed on the parameter's version counter and storage address: optimizer
# steps, load_state_dict and .to() invalidate it (maskunet_amd.FusedAdamW writes through raw pointers and bumps the versions itself).
# Training forwards never use it -- the weights change every step anyway, and an in-place edit through `p.data` (which autograd's
# version counter does not see) must not be able to leave a training run on stale weights.  Never consulted during graph capture.
PREP_CACHE = os.environ.get("MU_PREP_CACHE", "1") != "0"


def _cache_ok():
    """Evaluated OUTSIDE the autograd.Function (inside Function.forward grad mode is always off)."""
    return PREP_CACHE and not torch.is_grad_enabled()


def _prep_cached(w, key, make, ok, extra_tag=()):
    if not ok or torch.cuda.is_current_stream_capturing():
        return make()
    cache = getattr(w, "_mu_prep", None)
    if cache is None:
        cache = {}
        try:
            w._mu_prep = cache
        except Exception:      # tensors that do not accept attributes
            return make()
    tag = (w._version, w.data_ptr()) + tuple(extra_tag)
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    out = make()
    cache[key] = (tag, out)
    return out


def _prep_weight(w, dtype, rows_pad, cols_pad, mode, cache_ok=False):
    """mode 0: forward layout; 1: data-gradient layout; 2: both from one launch -> (fwd, dgrad) views of one buffer."""
    if cache_ok and isinstance(w, torch.nn.Parameter):
        return _prep_cached(w, (dtype, mdt(dtype), rows_pad, cols_pad, mode), lambda: _prep_weight_raw(w, dtype, rows_pad, cols_pad, mode), True)
    return _prep_weight_raw(w, dtype, rows_pad, cols_pad, mode)


def _prep_weight_raw(w, dtype, rows_pad, cols_pad, mode):
    O, I = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
    n = taps * rows_pad * cols_pad
    dst = torch.empty((2 * n if mode == 2 else n,), dtype=dtype, device=w.device)
    wf = w.detach().float().contiguous()
    # fp32x (MU_F32X): the library writes the layouts in their operand encodings -- bf16 chunks for 1x1 layers, fp16 chunks (forward) and
    # HL rows (data gradient) for 3x3 layers (include/maskunet_hip.h)
    call("mu_prep_weight", ptr(wf), ptr(dst), mdt(dtype), O, I, taps, rows_pad, cols_pad, mode, stream())
    if mode == 2:
        return dst[:n].view(taps, rows_pad, cols_pad), dst[n:].view(taps, cols_pad, rows_pad)
    return dst.view(taps, rows_pad, cols_pad)


# One launch for the layouts of ALL conv weights of a model at the start of a training forward (33 launches of 5-25 us otherwise; the
# weights change every step, so a training forward cannot keep them).  Each weight gets a ONE-SHOT entry that the next _Conv.forward
# on it consumes -- nothing survives the forward it was made for, so the no-stale-weights rule of the cache above holds here too.
MULTI_PREP = os.environ.get("MU_MULTI_PREP", "1") != "0"
MULTI_PREP_MAX_JOBS = 128       # MU_PREP_MAX_JOBS of the kernel's shared job table


def prep_conv_weights(holder, weights, dtype, fwd_only=()):
    """weights: the conv Parameters (OIHW fp32) a forward is about to use, each once; fwd_only: those whose input needs no gradient
    (forward layout only).  holder: a dict owned by the caller that keeps the device-side job table between calls."""
    if not MULTI_PREP or not weights:
        return
    dev = weights[0].device
    if dev.type != "cuda" or any(w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4 for w in weights):
        return
    if len(weights) > MULTI_PREP_MAX_JOBS or any(w.shape[2] * w.shape[3] not in (1, 9) for w in weights):
        return                          # outside the kernel's job table / LDS tile (MU_PREP_MAX_JOBS, MAXT = 9): the per-layer path serves them
    only = {id(w) for w in fwd_only}
    key = (dtype, mdt(dtype), tuple((w.data_ptr(), tuple(w.shape)) for w in weights), tuple(id(w) in only for w in weights))
    # one device job table per distinct key, kept: a captured graph (GraphedStep) holds the raw address of the table its forward used,
    # and a later eager forward under another key (compute dtype, requires_grad of the input) must not free it under the graph.
    # Tables seen during a capture are pinned for the holder's lifetime; the others are bounded (oldest dropped beyond 8).
    plans = holder.setdefault("plans", {})
    plan = plans.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if plan is not None and capturing:
        holder.setdefault("pinned", {})[key] = plan
    if plan is None:
        if capturing:
            return                      # the table is an H2D copy: built by an eager forward (GraphedStep warms up eagerly)
        rows, metas, off, chunk = [], [], 0, 0
        for w in weights:
            O, I, taps = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            rp, cp = pad32(O), pad32(I)
            n = taps * rp * cp
            mode = 0 if id(w) in only else 2
            total = n if mode == 0 else 2 * n
            rows.append([w.data_ptr(), off, chunk, O, I, taps, rp, cp, mode, 0])
            metas.append((off, n, taps, rp, cp, mode))
            off += total
            chunk += (rp // 32) * (cp // 32)               # one block-iteration per 32 x 32 (out x in) tile
        plan = plans[key] = (key, torch.tensor(rows, dtype=torch.int64, device=dev), metas, off, chunk)
        pinned = holder.get("pinned", {})
        for old in [k for k in plans if k != key and k not in pinned][:max(0, len(plans) - len(pinned) - 8)]:
            del plans[old]
    _, table, metas, total, nchunks = plan
    dst = torch.empty(total, dtype=dtype, device=dev)
    call("mu_prep_weights_multi", ptr(table), len(weights), nchunks, ptr(dst), mdt(dtype), stream())      # (fp32x: encoded by the kernel)
    tag = (dtype, mdt(dtype))
    for w, (off, n, taps, rp, cp, mode) in zip(weights, metas):
        fwd = dst[off:off + n].view(taps, rp, cp)
        wd = dst[off + n:off + 2 * n].view(taps, cp, rp) if mode == 2 else None
        w._mu_step = (w._version, w.data_ptr(), tag, fwd, wd)


def _take_step_prep(w, dtype, taps, rows_pad, cols_pad, need_dgrad):
    """The one-shot layouts prep_conv_weights made for this forward: (fwd, dgrad-or-None), or None."""
    pre = getattr(w, "_mu_step", None)
    if pre is None:
        return None
    w._mu_step = None
    if pre[0] != w._version or pre[1] != w.data_ptr() or pre[2] != (dtype, mdt(dtype)) or tuple(pre[3].shape) != (taps, rows_pad, cols_pad):
        return None
    if need_dgrad and pre[4] is None:
        return None
    return pre[3], pre[4]


def _conv_raw(x, wprep, bias_p, Cout_p, taps, want_stats=False, x_encoded=False):
    """y = conv(x); with want_stats also the per-tile BatchNorm statistics rows of y ([rows, Cout_p, 2] floats) when the kernel
    serving this shape has a statistics epilogue (else None).  fp32x: x is chunk-encoded here unless the caller already did."""
    B, H, W, Cin_p = x.shape
    y = torch.empty((B, H, W, Cout_p), dtype=x.dtype, device=x.device)
    if not x_encoded:
        x = _enc_for(x, taps)
    if want_stats:
        rows = _lib.load().mu_conv_stats_rows(B, H, W, Cin_p, Cout_p, taps, mdt(x))
        if rows > 0:
            part = torch.empty((rows, Cout_p, 2), dtype=torch.float32, device=x.device)
            call("mu_conv_fwd_stats", ptr(x), ptr(wprep), ptr(bias_p), ptr(y), B, H, W, Cin_p, Cout_p, taps, Cin_p, Cout_p, mdt(x),
                 ptr(part), stream())
            return y, part
        call("mu_conv_fwd", ptr(x), ptr(wprep), ptr(bias_p), ptr(y), B, H, W, Cin_p, Cout_p, taps, Cin_p, Cout_p, mdt(x), stream())
        return y, None
    call("mu_conv_fwd", ptr(x), ptr(wprep), ptr(bias_p), ptr(y), B, H, W, Cin_p, Cout_p, taps, Cin_p, Cout_p, mdt(x), stream())
    return y


# Cut point of a segmented capture (maskunet_amd.GraphedStep over DataParallel): while CUT_HOOK is set, the tensor handed to cut_point()
# gets it as a gradient hook -- it fires in the backward pass once every node created AFTER the tensor in the forward has run (the
# engine orders nodes by creation), i.e. when the gradients of the bottleneck, the decoder and the heads are complete.
CUT_HOOK = None


def cut_point(t):
    if CUT_HOOK is not None and t.requires_grad:
        t.register_hook(CUT_HOOK)
    return t


# Gradient arena (maskunet_amd.DataParallel, round 5): id(parameter) -> that parameter's fp32 slice of its all-reduce bucket.  The
# kernels that produce the LARGE parameter gradients (conv / Linear weight gradients: 24.5 M of the model's 24.9 M parameters; the two
# affine tensors of LayerNorm([64,128,128])) write straight into the slice, autograd's AccumulateGrad adopts the tensor it is handed
# as p.grad (no copy while p.grad is None), and the bucket is all-reduced IN PLACE: no flattening torch.cat, no write-back.  Only while
# p.grad is None: an existing gradient (accumulation over micro-batches) is added to by autograd as always.
GRAD_ARENA = None
# Slices handed out since the current step was armed (DataParallel._arm / GraphedStep._body clear it), by address.  A parameter that
# feeds TWO autograd nodes of one backward pass (tied weights, a module applied twice) asks twice while p.grad is still None: the
# second kernel would overwrite the first one's result and the engine would then add two aliases of the same memory (2 x the last
# contribution instead of g1 + g2) -- the second request gets a fresh tensor and autograd sums the two as always.
# A slice is rewritten by the NEXT step's backward: a reference to p.grad held across steps sees the new values (like a graph's
# static gradient tensors); clone what must outlive the step.
GRAD_HANDED = set()


def grad_out(param, shape, device):
    """Destination for the gradient of `param`: its arena slice (viewed as `shape`) when one is registered, p.grad is None and the slice
    has not been handed out in this step yet, else a fresh fp32 tensor."""
    a = GRAD_ARENA
    if a is not None and param is not None and getattr(param, "grad", None) is None:
        v = a.get(id(param))
        if v is not None and v.device == device and v.numel() == math.prod(shape) and v.data_ptr() not in GRAD_HANDED:
            GRAD_HANDED.add(v.data_ptr())
            return v.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


WGRAD_X16 = os.environ.get("MU_WGRAD_X16", "1") != "0"         # debug switch: 0 = the two-term weight gradient on the encoded input (mu_conv_wgrad_h)


def _wgrad_raw(x, gy, w_shape, taps, st=None, ws=None, gy_encoded=False, x_encoded=False, param=None, gy_scale=None):
    """gy_scale (fp32x 3x3 layers): gy holds ONE scaled fp16 operand (dy_encode_h form) and this is its scale pair.  x may then be the
    fp16 rounding of the layer's input (a float16 tensor: the one-term weight gradient) or its chunk-encoded fp32x form (two terms)."""
    B, H, W, Cin_p = x.shape
    O, I = w_shape[0], w_shape[1]
    if x.dtype == torch.float16 and gy.dtype == torch.float32:      # (fp16-mode layers hand over an fp16 dy and take the ordinary path below)
        if gy_scale is None:
            gy, gy_scale = dy_encode_h(gy)
        Cout_p = gy.shape[-1]
        gw = grad_out(param, tuple(w_shape), x.device)
        if ws is None:
            ws = workspace(_lib.load().mu_conv_wgrad_workspace_bytes(B, H, W, Cin_p, Cout_p, 9), x.device)
        call("mu_conv_wgrad_h1", ptr(x), ptr(gy), ptr(gy_scale), ptr(gw), B, H, W, Cin_p, Cout_p, I, O, Cin_p, Cout_p, ptr(ws), ws.numel(),
             stream() if st is None else st)
        return gw
    code = mdt(x)
    if code == _lib.MU_F32X and taps == 9 and I <= 3 and not gy_encoded and not x_encoded:
        code = _lib.MU_F32               # the first layer's weight gradient is a plain-FMA kernel (no matrix cores): plain fp32 operands
    if code == _lib.MU_F32X and taps == 9:
        # two-term weight gradient (round 6): the saved input as fp16 pairs x dy as one scaled fp16 operand
        if not x_encoded:
            x = _enc3(x)
        if gy_scale is None:
            if gy_encoded:
                raise RuntimeError("conv weight gradient: a 3x3 layer takes dy as a scaled fp16 operand, not chunk-encoded")
            gy, gy_scale = dy_encode_h(gy)
        Cout_p = gy.shape[-1]
        gw = grad_out(param, tuple(w_shape), x.device)
        if ws is None:
            ws = workspace(_lib.load().mu_conv_wgrad_h_workspace_bytes(B, H, W, Cin_p, Cout_p), x.device)
        call("mu_conv_wgrad_h", ptr(x), ptr(gy), ptr(gy_scale), ptr(gw), B, H, W, Cin_p, Cout_p, I, O, Cin_p, Cout_p, ptr(ws), ws.numel(),
             stream() if st is None else st)
        return gw
    if code == _lib.MU_F32X:
        if not x_encoded:
            x = _enc(x)
        if not gy_encoded:
            gy = _enc(gy)
    Cout_p = gy.shape[-1]
    gw = grad_out(param, tuple(w_shape), x.device)
    if ws is None:
        ws = workspace(_lib.load().mu_conv_wgrad_workspace_bytes(B, H, W, Cin_p, Cout_p, taps), x.device)
    call("mu_conv_wgrad", ptr(x), ptr(gy), ptr(gw), B, H, W, Cin_p, Cout_p, taps, I, O, Cin_p, Cout_p, ptr(ws), ws.numel(),
         code, stream() if st is None else st)
    return gw


WGRAD_BIAS = os.environ.get("MU_WGRAD_BIAS", "1") != "0"        # debug switch: 0 = separate column-sum sweep for the bias gradients


def _wgrad_bias_raw(x, gy, w_shape, taps, param=None):
    """(dW, db) from one sweep where the library supports it (fp16 1x1 layers on the wide tiles), else None."""
    B, H, W, Cin_p = x.shape
    Cout_p = gy.shape[-1]
    lib = _lib.load()
    if not WGRAD_BIAS or not lib.mu_conv_wgrad_bias_supported(Cin_p, Cout_p, taps, dt(x)):
        return None
    O, I = w_shape[0], w_shape[1]
    gw = grad_out(param, tuple(w_shape), x.device)
    gb = torch.empty(O, dtype=torch.float32, device=x.device)
    ws = workspace(lib.mu_conv_wgrad_workspace_bytes(B, H, W, Cin_p, Cout_p, taps), x.device)
    call("mu_conv_wgrad_bias", ptr(x), ptr(gy), ptr(gw), ptr(gb), B, H, W, Cin_p, Cout_p, taps, I, O, Cin_p, Cout_p, ptr(ws), ws.numel(),
         dt(x), stream())
    return gw, gb


# Weight gradients on a side stream.  dW of a layer is needed by nobody until the backward pass is over, while the chain
# dgrad -> BatchNorm backward -> ... of the layers below is mostly HBM-bound: the MFMA-bound weight-gradient kernels run
# beside it (one 8-wave block per CU leaves room for the elementwise kernels' waves).  Ordering:
#   * the side stream waits for the main stream (x and gy are ready), the main stream re-joins in an end-of-backward callback
#     (queued on the autograd engine), so everything after loss.backward() sees finished gradients;
#   * x, gy and gw are recorded on the other stream for the caching allocator;
#   * only used when weight.grad is None (AccumulateGrad then adopts gw without launching a kernel; an existing .grad would be
#     added to on the main stream) -- gradient accumulation over micro-batches falls back to the in-stream path;
#   * maskunet_amd.DataParallel makes its bucket all-reduce wait for this stream as well (dp.py).
_SIDE = {}
_SIDE_WS = {}
_JOIN_QUEUED = set()
# Measured (B=64 bench, same box, two rounds): 36.20 / 36.25 ms per step in-stream vs 36.54 / 36.45 ms with the side stream
# (36.78 / 36.50 when launched ahead of the data gradient): the kernels do not overlap usefully, so this stays opt-in.
WGRAD_SIDE_STREAM = os.environ.get("MU_WGRAD_SIDE", "0") != "0"
# only layers whose feature map is at most this high (at 16^2 / 32^2 the data- and weight-gradient grids each fill half of the
# chip at B = 64, so the pair could run side by side; the big layers compete for HBM / LDS -- the round-3 finding)
WGRAD_SIDE_MAXHW = int(os.environ.get("MU_WGRAD_SIDE_MAXHW", "100000"))


def wgrad_stream(device):
    """The side stream weight gradients are computed on (None if it has not been used on this device)."""
    return _SIDE.get(torch.device(device).index if not isinstance(device, int) else device)


def _join_side(index):
    def cb():
        _JOIN_QUEUED.discard(index)
        torch.cuda.current_stream(index).wait_stream(_SIDE[index])
    return cb


def _wgrad_side(x, gy, w_shape, taps, x_encoded=False, param=None):
    dev = x.device
    side = _SIDE.get(dev.index)
    if side is None:
        side = _SIDE[dev.index] = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        B, H, W, Cin_p = x.shape
        nbytes = _lib.load().mu_conv_wgrad_workspace_bytes(B, H, W, Cin_p, gy.shape[-1], taps)
        if taps == 9 and _is_x(x):
            nbytes = max(nbytes, _lib.load().mu_conv_wgrad_h_workspace_bytes(B, H, W, Cin_p, gy.shape[-1]))
        ws = _SIDE_WS.get(dev.index)
        if ws is None or ws.numel() < nbytes:
            ws = _SIDE_WS[dev.index] = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        gw = _wgrad_raw(x, gy, w_shape, taps, st=side.cuda_stream, ws=ws, x_encoded=x_encoded, param=param)
    x.record_stream(side)
    gy.record_stream(side)
    gw.record_stream(main)
    if dev.index not in _JOIN_QUEUED:
        _JOIN_QUEUED.add(dev.index)
        torch.autograd.Variable._execution_engine.queue_callback(_join_side(dev.index))
    return gw


def _colsum(gy, n_valid, encoded=False):
    """Column sums (bias gradients); encoded = gy is a chunk-encoded fp32x operand."""
    C = gy.shape[-1]
    M = gy.numel() // C
    out = torch.empty(C, dtype=torch.float32, device=gy.device)
    ws = workspace(_lib.load().mu_colsum_workspace_bytes(C), gy.device)
    call("mu_colsum", ptr(gy), M, C, C, ptr(out), ptr(ws), ws.numel(), _lib.MU_F32X if encoded else dt(gy), stream())
    return out[:n_valid]


class _Conv(torch.autograd.Function):
    """nn.Conv2d k=3/pad=1 or k=1, NHWC (ade_semantic.py:199,202,284; city_instance.py:243-249)."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False, cache_ok=False, x_encoded=False, dy_link=None, x16=None):
        x = x.contiguous()
        ctx.dy_link = dy_link
        O, I = weight.shape[0], weight.shape[1]
        taps = weight.shape[2] * weight.shape[3]
        Cin_p, Cout_p = x.shape[-1], pad32(O)
        if pad32(I) != Cin_p:
            raise RuntimeError(f"conv: input has {Cin_p} (padded) channels, weight expects {I}")
        # the data-gradient layout is produced by the same launch when the backward will need it
        ctx.wd = None
        pre = _take_step_prep(weight, x.dtype, taps, Cout_p, Cin_p, ctx.needs_input_grad[0])
        if pre is not None:
            wprep, ctx.wd = pre[0], (pre[1] if ctx.needs_input_grad[0] else None)
        elif ctx.needs_input_grad[0]:
            wprep, ctx.wd = _prep_weight(weight, x.dtype, Cout_p, Cin_p, 2)
        else:
            wprep = _prep_weight(weight, x.dtype, Cout_p, Cin_p, 0, cache_ok)
        bias_p = _pad_vec(bias, Cout_p, 0.0) if bias is not None else None
        part = None
        # fp32x: the chunk-encoded input is what both the forward conv and the weight gradient read -- encode once, save THAT (the first
        # layer's weight gradient is a plain-FMA kernel and keeps the plain tensor)
        ctx.is_x = _is_x(x)
        ctx.x_enc = ctx.is_x and not (taps == 9 and I <= 3)
        if x_encoded and not ctx.x_enc:
            raise RuntimeError("conv: a pre-encoded input needs the fp32x mode and a matrix-core layer")
        # fp32x 3x3 layers (round 6): the backward keeps only the fp16 ROUNDING of the input (x16: half the bytes of the encoded form) for the
        # one-term weight gradient; the encoded form feeds the forward conv and is dropped.  x16 comes from the producer that wrote x
        # encoded (bn_act: `_mu_x16`), or from the same pass that encodes x here; without it the two-term form on the encoded input remains.
        keep16 = ctx.x_enc and taps == 9 and WGRAD_X16 and ctx.needs_input_grad[1]
        if ctx.x_enc and not x_encoded:
            if keep16:
                x, x16 = _enc3x(x)
            else:
                x = _enc_for(x, taps)
        if want_stats:
            y, part = _conv_raw(x, wprep, bias_p, Cout_p, taps, True, x_encoded=ctx.x_enc)
        else:
            y = _conv_raw(x, wprep, bias_p, Cout_p, taps, x_encoded=ctx.x_enc)
        ctx.save_for_backward(x16 if (keep16 and x16 is not None) else x, weight)
        ctx.wparam = weight                      # the Parameter itself: backward looks at its .grad
        ctx.has_bias, ctx.taps = bias is not None, taps
        if not want_stats:
            return y
        if part is None:
            part = torch.empty(0, dtype=torch.float32, device=x.device)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)         # no zero-filled "gradient" of the statistics rows in the backward
        return y, part

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gpart=None):
        if gy is None:
            return None, None, None, None, None, None, None, None
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        _same_mode(ctx.is_x, gy)
        gy_sc = ctx.dy_link.take() if ctx.dy_link is not None else None      # fp32x: dy arrived as ONE scaled fp16 operand from the BatchNorm's backward
        gy_pre = gy_sc is not None
        O, I = weight.shape[0], weight.shape[1]
        gx = gw = gb = None
        side = ctx.needs_input_grad[1] and WGRAD_SIDE_STREAM and ctx.wparam.grad is None and x.shape[1] <= WGRAD_SIDE_MAXHW
        if side and os.environ.get("MU_WGRAD_SIDE_FIRST"):
            gw = _wgrad_side(x, gy, tuple(weight.shape), ctx.taps, ctx.x_enc, ctx.wparam)
        if gy_pre and (side or not ctx.x_enc or ctx.taps != 9 or (ctx.has_bias and ctx.needs_input_grad[2])):
            raise RuntimeError("conv backward: an encoded dy reached a path that needs it plain")
        if ctx.is_x and ctx.taps == 9:
            # fp32x 3x3 layer (round 6): dy as ONE power-of-two-scaled fp16 operand, shared by the data gradient (against the fp16 pair of
            # the weights: two MFMAs per product) and the weight gradient (against the fp16 rounding of the saved input: one)
            wg_h = ctx.needs_input_grad[1] and not side and ctx.x_enc      # (a <= 3-channel layer's weight gradient: plain-FMA kernel, plain dy)
            gh = gy if gy_pre else None
            if gh is None and (ctx.needs_input_grad[0] or wg_h):
                gh, gy_sc = dy_encode_h(gy)
            if ctx.needs_input_grad[0]:
                wd = ctx.wd if ctx.wd is not None else _prep_weight(weight, gy.dtype, x.shape[-1], gy.shape[-1], 1)
                ctx.wd = None
                B, H, W, Cin_p = x.shape
                gx = torch.empty((B, H, W, Cin_p), dtype=torch.float32, device=x.device)
                call("mu_conv_dgrad_h", ptr(gh), ptr(wd), ptr(gy_sc), ptr(gx), B, H, W, gy.shape[-1], Cin_p, gy.shape[-1], Cin_p, stream())
            if side and gw is None:
                gw = _wgrad_side(x, gy, tuple(weight.shape), ctx.taps, ctx.x_enc, ctx.wparam)
            if wg_h:         # x: the fp16 rounding of the input (one term) or its encoded form (two terms) -- _wgrad_raw tells by dtype
                gw = _wgrad_raw(x, gh, tuple(weight.shape), 9, x_encoded=True, param=ctx.wparam, gy_scale=gy_sc)
            elif ctx.needs_input_grad[1] and not side:
                gw = _wgrad_raw(x, gy, tuple(weight.shape), 9, param=ctx.wparam)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                gb = _colsum(gy, O)
            return gx, gw, gb, None, None, None, None, None
        ge = _enc(gy) if (ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and not side)) else gy     # fp32x: one encoding of dy for both
        if ctx.needs_input_grad[0]:
            wd = ctx.wd if ctx.wd is not None else _prep_weight(weight, gy.dtype, x.shape[-1], gy.shape[-1], 1)
            ctx.wd = None
            gx = _conv_raw(ge, wd, None, x.shape[-1], ctx.taps, x_encoded=True)
        if side and gw is None:                  # behind the data gradient (both want every CU's LDS): it then runs beside the
            gw = _wgrad_side(x, gy, tuple(weight.shape), ctx.taps, ctx.x_enc, ctx.wparam)      # HBM-bound kernels that follow on the main stream
        if ctx.needs_input_grad[1] and not side:
            both = _wgrad_bias_raw(x, gy, tuple(weight.shape), ctx.taps, ctx.wparam) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            if both is not None:
                gw, gb = both
            else:
                first = not ctx.x_enc                                   # (the first layer's plain-FMA kernel: plain operands)
                gw = _wgrad_raw(x, gy if first else ge, tuple(weight.shape), ctx.taps, gy_encoded=_is_x(gy) and not first, x_encoded=ctx.x_enc,
                                param=ctx.wparam)
        if ctx.has_bias and ctx.needs_input_grad[2] and gb is None:
            gb = _colsum(gy, O)
        return gx, gw, gb, None, None, None, None, None


def conv(x, weight, bias=None):
    return _Conv.apply(x, weight, bias, False, _cache_ok(), False, None, None)


CONV_STATS = os.environ.get("MU_CONV_STATS", "1") != "0"      # debug switch: 0 = always the separate statistics sweep


def conv_stats(x, weight, bias=None, want=True, x_encoded=False, dy_link=None):
    """conv() that also returns the BatchNorm statistics rows of its output (an empty tensor when the kernel has none) --
    pass them to bn_act(..., stats=rows) to skip the separate statistics sweep.  fp32x: x_encoded = x was written chunk-encoded by its
    producer (bn_act(..., enc_out=True)); dy_link = EncLink shared with the BatchNorm behind this conv (see EncLink)."""
    x16 = getattr(x, "_mu_x16", None) if x_encoded else None      # the fp16 rounding its producer wrote beside the encoded form (bn_act)
    if not want or not CONV_STATS:
        return _Conv.apply(x, weight, bias, False, _cache_ok(), x_encoded, dy_link, x16), None
    return _Conv.apply(x, weight, bias, True, _cache_ok(), x_encoded, dy_link, x16)


# ------------------------------------------------------------------------------------------------
# BatchNorm2d (+ activation, + residual)
# ------------------------------------------------------------------------------------------------
class _BNAct(torch.autograd.Function):
    """act(res + BatchNorm2d(x)): ade_semantic.py:200-201 (BN,GELU), :204,208 (BN, +x, GELU), :219,240 (BN),
    :285-286 (BN, ReLU).  Training uses batch statistics and updates the running buffers in place."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, training, momentum, eps, act, nbt=None, stats=None, res_link=None,
                enc_out=False, dx_link=None, x16_box=None):
        x = x.contiguous()
        ctx.res_link = res_link if (res is not None and res_link is not None and res_link.armed) else None
        ctx.dx_link = dx_link if (dx_link is not None and _is_x(x)) else None
        enc_out = bool(enc_out) and _is_x(x)
        C = x.shape[-1]
        M = x.numel() // C
        cv = gamma.numel()
        g_p, b_p = _pad_vec(gamma, C, 1.0), _pad_vec(beta, C, 0.0)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = workspace(_lib.load().mu_bn_workspace_bytes(C), x.device)
        if training:
            rm = running_mean if running_mean is not None else None
            rv = running_var if running_var is not None else None
            if stats is not None and stats.numel() > 0:          # rows left by the producing conv's epilogue
                call("mu_bn_train_stats_rows", ptr(stats), stats.shape[0], M, C, ptr(mean), ptr(rstd), ptr(rm), ptr(rv), ptr(nbt), cv,
                     float(momentum), float(eps), ptr(ws), ws.numel(), stream())
            else:
                call("mu_bn_train_stats", ptr(x), M, C, C, ptr(mean), ptr(rstd), ptr(rm), ptr(rv), ptr(nbt), cv, float(momentum), float(eps),
                     ptr(ws), ws.numel(), dt(x), stream())
        else:
            call("mu_bn_eval_stats", ptr(running_mean), ptr(running_var), float(eps), ptr(mean), ptr(rstd), C, cv, stream())
        y = torch.empty_like(x)
        if res is not None:
            res = res.contiguous()
        if enc_out and x16_box is not None:
            # fp32x: y as the next 3x3 conv's encoded operand AND its fp16 rounding (what that conv's backward keeps: ops._Conv)
            y16 = torch.empty(x.shape, dtype=torch.float16, device=x.device)
            call("mu_bn_act_fwd_enc", ptr(x), ptr(res), ptr(y), ptr(y16), M, C, ptr(mean), ptr(rstd), ptr(g_p), ptr(b_p), act, stream())
            x16_box.append(y16)
        else:
            call("mu_bn_act_fwd", ptr(x), ptr(res), ptr(y), M, C, C, ptr(mean), ptr(rstd), ptr(g_p), ptr(b_p), act,
                 _lib.MU_F32X if enc_out else dt(x), stream())      # MU_F32X: y written as the next conv's chunk-encoded operand
        ctx.save_for_backward(x, res, mean, rstd, g_p, b_p)
        ctx.act, ctx.training, ctx.cv = act, bool(training), cv
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, res, mean, rstd, g_p, b_p = ctx.saved_tensors
        gy = gy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if res is not None else None
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        ws = workspace(_lib.load().mu_bn_workspace_bytes(C), x.device)
        enc = ctx.dx_link is not None            # fp32x: dx is the dy of the 3x3 conv in front -- written as ONE scaled fp16 operand, the conv is told (EncLink)
        if enc:
            sc = torch.empty(2, dtype=torch.float32, device=x.device)      # {S, 1 / S}, written on the device
            call("mu_bn_act_bwd_h", ptr(x), ptr(res), ptr(gy), ptr(dx), ptr(dres), M, C, ptr(mean), ptr(rstd), ptr(g_p), ptr(b_p),
                 ctx.act, int(ctx.training), ptr(dgamma), ptr(dbeta), ptr(sc), ptr(ws), ws.numel(), stream())
            ctx.dx_link.mark(sc)
        else:
            call("mu_bn_act_bwd", ptr(x), ptr(res), ptr(gy), ptr(dx), ptr(dres), M, C, C, ptr(mean), ptr(rstd), ptr(g_p), ptr(b_p),
                 ctx.act, int(ctx.training), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), dt(x), stream())
        if ctx.res_link is not None:             # the residual-branch gradient travels to the backward of x's producer (GradLink)
            ctx.res_link.put(dres)
            dres = None
        return dx, dres, dgamma[:ctx.cv], dbeta[:ctx.cv], None, None, None, None, None, None, None, None, None, None, None, None


class _BNPair(torch.autograd.Function):
    """BatchNorm2d(BatchNorm2d(x)) in training mode as ONE normalisation of x (ade_semantic.py:216-219, 237-240: the BatchNorm behind
    ConvBlock's last BatchNorm in DownSample / UpSample); see mu_bn_pair_compose in include/maskunet_hip.h for the algebra."""

    @staticmethod
    def forward(ctx, x, g1, b1, g2, b2, rm1, rv1, nbt1, mom1, eps1, rm2, rv2, nbt2, mom2, eps2, stats, dx_link=None):
        x = x.contiguous()
        ctx.dx_link = dx_link if (dx_link is not None and _is_x(x)) else None
        C = x.shape[-1]
        M = x.numel() // C
        cv = g1.numel()
        g1p, b1p, g2p, b2p = _pad_vec(g1, C, 1.0), _pad_vec(b1, C, 0.0), _pad_vec(g2, C, 1.0), _pad_vec(b2, C, 0.0)
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        coef = torch.empty(4, C, dtype=torch.float32, device=dev)      # gamma_eff, xhat_scale, dgamma2_coef, dgamma1_coef
        ws = workspace(_lib.load().mu_bn_workspace_bytes(C), dev)
        if stats is not None and stats.numel() > 0:
            call("mu_bn_train_stats_rows", ptr(stats), stats.shape[0], M, C, ptr(mean), ptr(rstd), ptr(rm1), ptr(rv1), ptr(nbt1), cv,
                 float(mom1), float(eps1), ptr(ws), ws.numel(), stream())
        else:
            call("mu_bn_train_stats", ptr(x), M, C, C, ptr(mean), ptr(rstd), ptr(rm1), ptr(rv1), ptr(nbt1), cv, float(mom1), float(eps1),
                 ptr(ws), ws.numel(), dt(x), stream())
        call("mu_bn_pair_compose", ptr(rstd), ptr(g1p), ptr(b1p), ptr(g2p), C, cv, M, float(eps1), float(eps2), float(mom2), ptr(rm2),
             ptr(rv2), ptr(nbt2), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), stream())
        y = torch.empty_like(x)
        call("mu_bn_act_fwd", ptr(x), None, ptr(y), M, C, C, ptr(mean), ptr(rstd), ptr(coef[0]), ptr(b2p), ACT_NONE, dt(x), stream())
        ctx.save_for_backward(x, mean, rstd, coef, b2p)
        ctx.cv = cv
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, mean, rstd, coef, b2p = ctx.saved_tensors
        gy = gy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(3, C, dtype=torch.float32, device=x.device)      # rows: dgamma2, dgamma1, dbeta1 (= 0), written by the finalize kernel
        dbeta2 = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace(_lib.load().mu_bn_workspace_bytes(C), x.device)
        enc = ctx.dx_link is not None
        if enc:
            sc = torch.empty(2, dtype=torch.float32, device=x.device)
            call("mu_bn_pair_bwd_h", ptr(x), ptr(gy), ptr(dx), M, C, ptr(mean), ptr(rstd), ptr(coef[0]), ptr(b2p), ptr(coef[1]), ptr(coef[2]),
                 ptr(coef[3]), ptr(dg), ptr(dbeta2), ptr(sc), ptr(ws), ws.numel(), stream())
            ctx.dx_link.mark(sc)
        else:
            call("mu_bn_pair_bwd", ptr(x), ptr(gy), ptr(dx), M, C, C, ptr(mean), ptr(rstd), ptr(coef[0]), ptr(b2p), ptr(coef[1]), ptr(coef[2]),
                 ptr(coef[3]), ptr(dg), ptr(dbeta2), ptr(ws), ws.numel(), dt(x), stream())
        cv = ctx.cv
        return (dx, dg[1, :cv], dg[2, :cv], dg[0, :cv], dbeta2[:cv]) + (None,) * 12


BN_PAIR = os.environ.get("MU_BN_PAIR", "1") != "0"          # debug switch: 0 = the two layers one after the other


def bn_pair(x, bn1, bn2, stats=None, dx_link=None):
    """bn2(bn1(x)) for two nn.BatchNorm2d containers applied back to back (no activation, no residual in between)."""
    both_train = (bn1.training and bn2.training and bn1.running_mean is not None and bn2.running_mean is not None
                  and bn1.weight is not None and bn2.weight is not None)
    ok = both_train and BN_PAIR and all(t is not None and t.device == x.device and t.dtype == torch.int64
                                        for t in (bn1.num_batches_tracked, bn2.num_batches_tracked))
    if not ok or bn1.momentum is None or bn2.momentum is None:
        return bn_act(bn_act(x, bn1, ACT_NONE, stats=stats, dx_link=dx_link), bn2, ACT_NONE)
    return _BNPair.apply(x, bn1.weight, bn1.bias, bn2.weight, bn2.bias, bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                         bn1.momentum, bn1.eps, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, bn2.momentum, bn2.eps, stats, dx_link)


def bn_act(x, bn, act=ACT_NONE, res=None, stats=None, res_link=None, enc_out=False, dx_link=None):
    """Apply the BatchNorm2d parameter container `bn` (an nn.BatchNorm2d used only for its
    parameters/buffers/flags) followed by `act`, optionally adding `res` before the activation.
    `stats`: statistics rows of x from conv_stats() (training mode only; ignored otherwise).
    `res_link`: GradLink that carries d(res) to the backward of res's producer (maxpool2 / upcat) instead of through autograd."""
    training = bn.training or bn.running_mean is None
    # the step counter is bumped by the statistics kernel (one tiny torch kernel per BatchNorm otherwise: 39 per step)
    nbt = bn.num_batches_tracked if (bn.training and bn.num_batches_tracked is not None) else None
    if nbt is not None and (nbt.device != x.device or nbt.dtype != torch.int64):
        nbt.add_(1)
        nbt = None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    box = [] if (enc_out and WGRAD_X16 and torch.is_grad_enabled() and _is_x(x)) else None
    y = _BNAct.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, act, nbt,
                     stats if training else None, res_link, enc_out, dx_link, box)
    if box:
        y._mu_x16 = box[0]
    return y


# ------------------------------------------------------------------------------------------------
# inference: Conv2d -> BatchNorm2d(eval) [-> BatchNorm2d(eval)] [-> + res] -> activation in ONE launch
# ------------------------------------------------------------------------------------------------
EVAL_FUSE = os.environ.get("MU_EVAL_FUSE", "1") != "0"      # debug switch: 0 = conv, then a BatchNorm-apply pass per layer


def eval_fusable(*bns):
    """True when the BatchNorm containers `bns` (None entries ignored) can be folded into the producing conv's epilogue: no gradient is
    being recorded (the validation loops of the reference run under torch.no_grad(), ade_semantic.py:443-447) and every layer
    normalises with its running statistics."""
    if not EVAL_FUSE or torch.is_grad_enabled():
        return False
    return all(bn is None or (not bn.training and bn.running_mean is not None and bn.running_var is not None) for bn in bns)


def conv_bn_act_eval(x, weight, conv_bias, bn, act=ACT_NONE, res=None, bn2=None):
    """act(res + bn2(bn(conv(x) + conv_bias))) with eval-mode BatchNorm layers, as one conv launch (mu_conv_fwd_fused): the layers'
    running statistics and affine parameters become a per-channel (scale, shift) of the conv epilogue (mu_bn_eval_fold), the
    residual add and GELU / ReLU follow in the same epilogue.  Removes one read + write pass per BatchNorm from the validation loop
    (ade_semantic.py:443-471).  Only under eval_fusable(bn, bn2); no autograd."""
    x = x.contiguous()
    B, H, W, Cin_p = x.shape
    O, I = weight.shape[0], weight.shape[1]
    taps = weight.shape[2] * weight.shape[3]
    Cout_p = pad32(O)
    if pad32(I) != Cin_p:
        raise RuntimeError(f"conv: input has {Cin_p} (padded) channels, weight expects {I}")
    wprep = _prep_weight(weight, x.dtype, Cout_p, Cin_p, 0, True)
    fold = torch.empty((2, Cout_p), dtype=torch.float32, device=x.device)
    f = lambda t: None if t is None else t.detach().float().contiguous()      # noqa: E731  (fp32 parameters: no copy)
    b2 = bn2 if bn2 is not None else None
    call("mu_bn_eval_fold", ptr(f(bn.running_mean)), ptr(f(bn.running_var)), ptr(f(bn.weight)), ptr(f(bn.bias)), float(bn.eps),
         ptr(f(b2.running_mean)) if b2 is not None else None, ptr(f(b2.running_var)) if b2 is not None else None,
         ptr(f(b2.weight)) if b2 is not None else None, ptr(f(b2.bias)) if b2 is not None else None, float(b2.eps) if b2 is not None else 0.0,
         ptr(f(conv_bias)), ptr(fold[0]), ptr(fold[1]), Cout_p, O, stream())
    y = torch.empty((B, H, W, Cout_p), dtype=x.dtype, device=x.device)
    if res is not None:
        res = res.contiguous()
    call("mu_conv_fwd_fused", ptr(_enc_for(x, taps)), ptr(wprep), ptr(fold[0]), ptr(fold[1]), ptr(res), act, ptr(y), B, H, W, Cin_p, Cout_p, taps, Cin_p,
         Cout_p, mdt(x), stream())
    return y


# ------------------------------------------------------------------------------------------------
# pooling / resampling / dropout
# ------------------------------------------------------------------------------------------------
class _MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2) (ade_semantic.py:216).  res_link: GradLink filled by the residual ConvBlock behind the pool with a second
    gradient of the pooled tensor; skip_link: GradLink filled by the UpSample that takes the pool's INPUT as its skip tensor with
    that tensor's other gradient.  Both are joined inside mu_maxpool2_bwd_acc."""

    @staticmethod
    def forward(ctx, x, res_link=None, skip_link=None):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        call("mu_maxpool2_fwd", ptr(x), ptr(y), B, H, W, C, dt(x), stream())
        ctx.save_for_backward(x)
        ctx.res_link, ctx.skip_link = res_link, skip_link
        if ctx.needs_input_grad[0]:              # a backward of this node will run: the fillers may route their gradients here
            if res_link is not None:
                res_link.armed = True
            if skip_link is not None:
                skip_link.armed = True
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        g2 = ctx.res_link.take() if ctx.res_link is not None else None
        ga = ctx.skip_link.take() if ctx.skip_link is not None else None
        if g2 is None and ga is None:
            call("mu_maxpool2_bwd", ptr(x), ptr(gy), ptr(dx), B, H, W, C, dt(x), stream())
        else:
            call("mu_maxpool2_bwd_acc", ptr(x), ptr(gy), ptr(g2), ptr(ga), ptr(dx), B, H, W, C, dt(x), stream())
        return dx, None, None


def maxpool2(x, res_link=None, skip_link=None):
    return _MaxPool2.apply(x, res_link, skip_link)


class _UpCat(torch.autograd.Function):
    """cat([skip, bilinear_x2(x, align_corners=True)], dim=C) (ade_semantic.py:235,250-253).  cx / cs: the true channel counts of
    x / skip; where they are not the stored (32-padded) counts the concat compacts [skip valid | up valid | zero pad]."""

    @staticmethod
    def forward(ctx, x, skip, cx, cs, res_link=None, skip_link=None):
        x, skip = x.contiguous(), skip.contiguous()
        B, h, w, Cx = x.shape
        Cs = skip.shape[-1]
        if skip.shape[:3] != (B, 2 * h, 2 * w):
            raise RuntimeError(f"upsample/concat: skip {tuple(skip.shape)} does not match 2x of {tuple(x.shape)}")
        ctx.compact = (cx != Cx or cs != Cs)
        Ct = pad32(cx + cs) if ctx.compact else Cs + Cx
        y = torch.empty((B, 2 * h, 2 * w, Ct), dtype=x.dtype, device=x.device)
        if ctx.compact:
            call("mu_upcat_compact_fwd", ptr(x), ptr(skip), ptr(y), B, h, w, Cx, cx, Cs, cs, Ct, dt(x), stream())
        else:
            call("mu_upcat_fwd", ptr(x), ptr(skip), ptr(y), B, h, w, Cx, Cs, dt(x), stream())
        ctx.dims = (B, h, w, Cx, Cs, cx, cs, Ct)
        # res_link (residual ConvBlock behind the concat -> this node): only the vector kernel joins a second gradient
        ctx.res_link = res_link if not ctx.compact else None
        if ctx.res_link is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            ctx.res_link.armed = True
        # skip_link (this node -> the maxpool that also consumes `skip`): used when that node armed it in its forward
        ctx.skip_link = skip_link if (skip_link is not None and skip_link.armed and ctx.needs_input_grad[1]) else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        B, h, w, Cx, Cs, cx, cs, Ct = ctx.dims
        gy = gy.contiguous()
        dx = torch.empty((B, h, w, Cx), dtype=gy.dtype, device=gy.device)
        dskip = torch.empty((B, 2 * h, 2 * w, Cs), dtype=gy.dtype, device=gy.device)
        g2 = ctx.res_link.take() if ctx.res_link is not None else None
        if ctx.compact:
            call("mu_upcat_compact_bwd", ptr(gy), ptr(dx), ptr(dskip), B, h, w, Cx, cx, Cs, cs, Ct, dt(gy), stream())
        elif g2 is not None:
            call("mu_upcat_bwd_acc", ptr(gy), ptr(g2), ptr(dx), ptr(dskip), B, h, w, Cx, Cs, dt(gy), stream())
        else:
            call("mu_upcat_bwd", ptr(gy), ptr(dx), ptr(dskip), B, h, w, Cx, Cs, dt(gy), stream())
        if ctx.skip_link is not None:            # d(skip) is added by the backward of the pool that shares the tensor (GradLink)
            ctx.skip_link.put(dskip)
            dskip = None
        return dx, dskip, None, None, None, None


def upcat(x, skip, cx=None, cs=None, res_link=None, skip_link=None):
    return _UpCat.apply(x, skip, x.shape[-1] if cx is None else int(cx), skip.shape[-1] if cs is None else int(cs), res_link, skip_link)


SEED_STEP = None        # int64 device tensor [1] or None; set by maskunet_amd.graph.GraphedStep during capture


class _Dropout(torch.autograd.Function):
    """nn.Dropout(p) in training (ade_semantic.py:273,304,307).  mask (uint8, same shape) overrides the generator."""

    @staticmethod
    def forward(ctx, x, p, seed, mask):
        x = x.contiguous()
        y = torch.empty_like(x)
        if mask is not None:
            mask = mask.to(device=x.device, dtype=torch.uint8).contiguous()
        step = SEED_STEP                                     # device step counter while a step is being captured (graph.py)
        call("mu_dropout_step", ptr(x), ptr(y), x.numel(), float(p), int(seed), ptr(step), ptr(mask), None, dt(x), stream())
        ctx.p, ctx.seed, ctx.step = p, seed, step
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (mask,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        call("mu_dropout_step", ptr(gy), ptr(gx), gy.numel(), float(ctx.p), int(ctx.seed), ptr(ctx.step), ptr(mask), None, dt(gy), stream())
        return gx, None, None, None


def dropout(x, p, training, mask=None):
    if not training or (p == 0.0 and mask is None):
        return x
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())   # CPU generator: honours torch.manual_seed
    return _Dropout.apply(x, p, seed, mask)


# ------------------------------------------------------------------------------------------------
# per-sample LayerNorm with full-shape affine
# ------------------------------------------------------------------------------------------------
class _LNSample(torch.autograd.Function):
    """nn.LayerNorm([64,H,W]) applied to the NCHW-flat memory [B, L] (ade_semantic.py:281,311)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        B = x.shape[0]
        L = x.numel() // B
        wf, bf = w.detach().float().contiguous().view(-1), b.detach().float().contiguous().view(-1)
        y = torch.empty_like(x)
        mean = torch.empty(B, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = workspace(_lib.load().mu_ln_sample_workspace_bytes(B), x.device)
        call("mu_ln_sample_fwd", ptr(x), ptr(wf), ptr(bf), ptr(y), ptr(mean), ptr(rstd), B, L, float(eps), ptr(ws), ws.numel(),
             dt(x), stream())
        ctx.save_for_backward(x, wf, mean, rstd)
        ctx.wshape = tuple(w.shape)
        ctx.wparam, ctx.bparam = w, b            # the leaves themselves: their gradients may have arena slices (grad_out)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, wf, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        B = x.shape[0]
        L = x.numel() // B
        dx = torch.empty_like(x)
        dw = grad_out(ctx.wparam, (L,), x.device)
        db = grad_out(ctx.bparam, (L,), x.device)
        ws = workspace(_lib.load().mu_ln_sample_workspace_bytes(B), x.device)
        call("mu_ln_sample_bwd", ptr(x), ptr(gy), ptr(wf), ptr(mean), ptr(rstd), ptr(dx), ptr(dw), ptr(db), B, L, ptr(ws),
             ws.numel(), dt(x), stream())
        return dx, dw.view(ctx.wshape), db.view(ctx.wshape), None


def ln_sample(x, w, b, eps):
    return _LNSample.apply(x, w, b, eps)


# ------------------------------------------------------------------------------------------------
# masked attention block
# ------------------------------------------------------------------------------------------------
def _transpose_tokens(src, R, C):
    """[B, R, C] -> [B, C, R] (same dtype)."""
    B = src.shape[0]
    dst = torch.empty((B, C, R), dtype=src.dtype, device=src.device)
    call("mu_transpose", ptr(src), dt(src), C, ptr(dst), dt(dst), R, B, R, C, stream())
    return dst


class _MaskAttention(torch.autograd.Function):
    """Mask2FormerAttention.forward (ade_semantic.py:163-190) on an NHWC activation.

    NHWC *is* the reference's token-major [B,N,C] view of x (:168), so the input needs no permute.  The
    reference then re-reads the [B,N,C] result as NCHW memory (:190); in NHWC terms that is a [C,N]->[N,C]
    transpose of each image's flat buffer, done here with mu_transpose (`scramble=True`), or skipped when the
    consumer wants the NCHW-flat memory itself (the final LayerNorm, :311)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, lnw, lnb, kidx, kcnt, eps, scramble, cache_ok=False, kidx_perm=False):
        x = x.contiguous()
        B, H, W, C = x.shape
        N = H * W
        cv = wq.shape[0]                          # true channel count; C = the width x is stored with (attn_width(cv))
        ctx.cv = cv
        if cv != C:
            # a channel count the kernels have no width for: every parameter zero-padded to C (slow path, torch pads; standalone
            # Mask2FormerAttention only -- the UNet's blocks are 64 / 128 / 256 wide)
            if scramble:
                raise RuntimeError("mask_attention: the re-viewed (scrambled) output needs an unpadded channel count")
            pw = lambda w: F.pad(w.detach().float(), (0, C - cv, 0, C - cv))      # noqa: E731
            pv = lambda v: F.pad(v.detach().float(), (0, C - cv))                # noqa: E731
            wq, wk, wv, bq, bk, bv, lnw, lnb = pw(wq), pw(wk), pw(wv), pv(bq), pv(bk), pv(bv), pv(lnw), pv(lnb)
            cache_ok = False

        def make_qkv():          # the three Linear layers as one [3C, C] 1x1 layer: forward + data-gradient layouts and the bias, one launch
            wbuf = torch.empty((6 * C * C,), dtype=x.dtype, device=x.device)
            bqkv_ = torch.empty((3 * C,), dtype=torch.float32, device=x.device)
            src = [t.detach() if t.dtype == torch.float32 and t.is_contiguous() else t.detach().float().contiguous()
                   for t in (wq, wk, wv, bq, bk, bv)]
            call("mu_prep_qkv", *[ptr(t) for t in src], ptr(wbuf), ptr(bqkv_), dt(x), C, stream())
            _enc_(wbuf)
            return bqkv_, (wbuf[:3 * C * C].view(1, 3 * C, C), wbuf[3 * C * C:].view(1, C, 3 * C))

        if cache_ok and isinstance(wq, torch.nn.Parameter):
            # cached on the query weight under ONE key per dtype; the tag carries all six parameters' versions, so a stale entry is
            # overwritten in place (keying on the versions grew the dict by one entry per train/validate cycle)
            vers = tuple((t._version, t.data_ptr()) for t in (wk, wv, bq, bk, bv))
            bqkv, (wprep, wd_) = _prep_cached(wq, ("qkv", x.dtype, mdt(x)), make_qkv, True, extra_tag=vers)
        else:
            bqkv, (wprep, wd_) = make_qkv()
        ctx.wd = wd_ if ctx.needs_input_grad[0] else None
        xe = _enc(x)                                                   # fp32x: the projection's operand form, kept for its weight gradient (x itself otherwise)
        if _is_x(x) and QKV_ENCODED and (3 * C) % 64 == 0:
            # [B,H,W,3C] == [B,N,3C], written by the projection's epilogue in the attention operand encoding (fp16 pairs): only the
            # attention sweeps, forward and backward, ever read it
            qkv = torch.empty((B, H, W, 3 * C), dtype=x.dtype, device=x.device)
            call("mu_conv1x1_fwd_enc_h", ptr(xe), ptr(wprep), ptr(bqkv), ptr(qkv), B * N, C, 3 * C, C, 3 * C, stream())
        else:
            qkv = _enc_h_(_conv_raw(xe, wprep, bqkv, 3 * C, 1, x_encoded=True))
        out = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
        oattn = torch.empty_like(out)
        lse2 = torch.empty((B, N), dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty_like(lse2), torch.empty_like(lse2)
        g, b_ = lnw.detach().float().contiguous(), lnb.detach().float().contiguous()
        if cv == C:
            call("mu_attn_fwd", ptr(qkv), ptr(x), ptr(kidx), ptr(kcnt), ptr(g), ptr(b_), ptr(out), ptr(oattn), ptr(lse2), ptr(mean),
                 ptr(rstd), B, N, C, kidx.shape[1], float(eps), mdt(x), stream())
        else:
            call("mu_attn_fwd_padded", ptr(qkv), ptr(x), ptr(kidx), ptr(kcnt), ptr(g), ptr(b_), ptr(out), ptr(oattn), ptr(lse2), ptr(mean),
                 ptr(rstd), B, N, C, cv, kidx.shape[1], float(eps), mdt(x), stream())
        ctx.save_for_backward(x, qkv, oattn, lse2, mean, rstd, g, kidx, kcnt, xe)
        ctx.is_x = _is_x(x)
        ctx.scramble, ctx.dims, ctx.kidx_perm = scramble, (B, H, W, C), bool(kidx_perm) and kidx.shape[1] == N
        if scramble:
            return _transpose_tokens(out.view(B, C, N), C, N).view(B, H, W, C)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, qkv, oattn, lse2, mean, rstd, g, kidx, kcnt, xe = ctx.saved_tensors
        B, H, W, C = ctx.dims
        N = H * W
        gout = gout.contiguous()
        _same_mode(ctx.is_x, gout)
        if ctx.scramble:
            gout = _transpose_tokens(gout.view(B, N, C), N, C)      # back to token-major flat [B, (N,C)]
        dY = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
        dqkv = torch.empty((B, N, 3 * C), dtype=x.dtype, device=x.device)
        delta = torch.empty((B, N), dtype=torch.float32, device=x.device)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        ws = workspace(_lib.load().mu_attn_bwd_workspace_bytes(B, N, C), x.device)
        # MU_ATTN_KIDX_PERMUTATION (8) is a promise the CALLER of mask_attention makes (kidx_perm=True: every kidx row is a whole
        # permutation with the masked keys last, as mu_compact_keys / a stable descending argsort give): the dK/dV sweep then zeroes
        # the masked rows itself.  Without the promise phase 1 memsets dqkv (kidx from outside may be padded behind kcnt).
        perm = 8 if ctx.kidx_perm else 0
        # fp32x: the sweeps write dqkv in the chunk encoding the projection's gradient kernels read (MU_ATTN_DQKV_ENCODED) -- no separate
        # encoding pass over the block's largest tensor; its column sums (the bias gradients) are taken from the encoded form
        enc_dqkv = ctx.is_x and DQKV_ENCODED
        if enc_dqkv:
            perm |= 16
        cv = ctx.cv
        for phase in (1, 2, 4):      # LayerNorm-backward prepass, dQ sweep, dK/dV sweep (separate calls: each can be timed)
            if cv == C:
                call("mu_attn_bwd_phases", ptr(qkv), ptr(x), ptr(oattn), ptr(gout), ptr(kidx), ptr(kcnt), ptr(lse2), ptr(mean), ptr(rstd),
                     ptr(g), ptr(dY), ptr(delta), ptr(dqkv), ptr(dg), ptr(db), B, N, C, kidx.shape[1], ptr(ws), ws.numel(), mdt(x),
                     phase | perm, stream())
            else:
                call("mu_attn_bwd_phases_padded", ptr(qkv), ptr(x), ptr(oattn), ptr(gout), ptr(kidx), ptr(kcnt), ptr(lse2), ptr(mean),
                     ptr(rstd), ptr(g), ptr(dY), ptr(delta), ptr(dqkv), ptr(dg), ptr(db), B, N, C, cv, kidx.shape[1], ptr(ws), ws.numel(),
                     mdt(x), phase | perm, stream())
        dqkv4 = dqkv.view(B, H, W, 3 * C)
        dqkv_e = dqkv4 if enc_dqkv else _enc(dqkv4)      # fp32x: dqkv as a matrix operand (projection data- and weight-gradient); dqkv4 otherwise
        gx = None
        if ctx.needs_input_grad[0]:
            wd = ctx.wd                          # kept on ctx (views of one small buffer): a second backward over a retained graph needs it again
            if ATTN_FUSED_ADD and _lib.load().mu_conv1x1_add_supported(3 * C, C, mdt(x)):
                # gx = dqkv @ Wqkv + dY: the residual branch (:187) joins the projection's data-gradient in its epilogue
                gx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
                call("mu_conv1x1_fwd_add", ptr(dqkv_e), ptr(wd), ptr(dY), ptr(gx), B * N, 3 * C, C, 3 * C, C, mdt(x), stream())
            else:
                gx = _conv_raw(dqkv_e, wd, None, C, 1, x_encoded=True)
                call("mu_add", ptr(gx), ptr(dY), ptr(gx), gx.numel(), dt(gx), stream())
        both = None if enc_dqkv else _wgrad_bias_raw(x, dqkv4, (3 * C, C, 1, 1), 1)       # projection weight and bias gradients from one sweep over dqkv
        if both is not None:
            gw, gb = both[0].view(3 * C, C), both[1]
        else:
            gw = _wgrad_raw(xe, dqkv_e, (3 * C, C, 1, 1), 1, gy_encoded=True, x_encoded=ctx.is_x).view(3 * C, C)
            gb = _colsum(dqkv4, 3 * C, encoded=enc_dqkv)
        if cv != C:                  # gradients of the real (unpadded) parameters
            return (gx, gw[:cv, :cv], gb[:cv], gw[C:C + cv, :cv], gb[C:C + cv], gw[2 * C:2 * C + cv, :cv], gb[2 * C:2 * C + cv],
                    dg[:cv], db[:cv], None, None, None, None, None, None)
        return (gx, gw[:C], gb[:C], gw[C:2 * C], gb[C:2 * C], gw[2 * C:], gb[2 * C:], dg, db, None, None, None, None, None, None)


ATTN_FUSED_ADD = os.environ.get("MU_ATTN_FUSED_ADD", "1") != "0"      # debug switch: 0 = projection data-gradient + mu_add
QKV_ENCODED = os.environ.get("MU_QKV_ENCODED", "1") != "0"            # debug switch: 0 = plain projection output + a separate mu_split_encode_h pass
DQKV_ENCODED = os.environ.get("MU_DQKV_ENCODED", "1") != "0"          # debug switch: 0 = plain dqkv + a separate mu_split_encode pass


ATTN_WIDTHS = (32, 64, 128, 256)


def attn_width(channels: int) -> int:
    """Stored (zero-padded) width an attention block of `channels` channels runs at: one of the flash-style sweeps' widths up to 256,
    the next multiple of 32 above (the generic GEMM path, _WideMaskAttention)."""
    for w in ATTN_WIDTHS:
        if channels <= w:
            return w
    return pad32(channels)


def _gemm_nt(a, a_ld, M, K, w, n_out, out, out_ld, code, bias=None):
    """out[M, n_out] (rows out_ld apart) = a[M, K] (rows a_ld apart) @ w[n_out, K]^T (+ bias): the 1x1-conv entry point as a plain GEMM."""
    call("mu_conv_fwd", ptr(a), ptr(w), ptr(bias) if bias is not None else None, ptr(out), 1, 1, M, K, n_out, 1, a_ld, out_ld, code, stream())


def _gemm_tn(x, x_ld, dy, dy_ld, M, K, n_out, code):
    """fp32 [n_out, K] = dy[M, n_out]^T @ x[M, K]: the 1x1 weight-gradient entry point as a plain GEMM."""
    out = torch.empty((n_out, K), dtype=torch.float32, device=x.device)
    ws = workspace(_lib.load().mu_conv_wgrad_workspace_bytes(1, 1, M, K, n_out, 1), x.device)
    call("mu_conv_wgrad", ptr(x), ptr(dy), ptr(out), 1, 1, M, K, n_out, 1, K, n_out, x_ld, dy_ld, ptr(ws), ws.numel(), code, stream())
    return out


def _colsum_wide(t2d):
    """Column sums of a [M, C] tensor of any width (mu_colsum sweeps up to 1024 columns per call: column blocks of 512 with ld = C)."""
    M, C = t2d.shape
    out = torch.empty(C, dtype=torch.float32, device=t2d.device)
    ws = workspace(_lib.load().mu_colsum_workspace_bytes(512), t2d.device)
    for c0 in range(0, C, 512):
        cb = min(512, C - c0)
        call("mu_colsum", ptr(t2d[:, c0:]), M, cb, C, ptr(out[c0:]), ptr(ws), ws.numel(), dt(t2d), stream())
    return out


class _WideMaskAttention(torch.autograd.Function):
    """Mask2FormerAttention.forward (ade_semantic.py:163-190) for channel counts above the flash-style sweeps' widths (C > 256): the
    GENERIC path.  Per image the products run as GEMMs on the 1x1-conv entry points over the kept key rows, with the row kernels of
    csrc/attn_wide.hip in between; one N x N score tile per image exists at a time (the backward recomputes it).  fp16 or exact fp32
    (the fp32x mode runs this block in exact fp32).  Off the model's path: generality, not speed."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, lnw, lnb, kidx, kcnt, eps, scramble):
        x = x.contiguous()
        B, H, W, C = x.shape
        N = H * W
        cv = wq.shape[0]
        if scramble and cv != C:
            raise RuntimeError("mask_attention: the re-viewed (scrambled) output needs an unpadded channel count")
        code = _lib.MU_F16 if x.dtype == torch.float16 else _lib.MU_F32
        pw = lambda w: F.pad(w.detach().float(), (0, C - cv, 0, C - cv))      # noqa: E731
        pv = lambda v: F.pad(v.detach().float(), (0, C - cv))                # noqa: E731
        wqkv = torch.cat([pw(wq), pw(wk), pw(wv)], 0).to(x.dtype).contiguous()          # [3C, C]
        bqkv = torch.cat([pv(bq), pv(bk), pv(bv)]).contiguous()
        g, b_ = pv(lnw).contiguous(), pv(lnb).contiguous()
        qkv = torch.empty((B, N, 3 * C), dtype=x.dtype, device=x.device)
        _gemm_nt(x, C, B * N, C, wqkv, 3 * C, qkv, 3 * C, code, bias=bqkv)
        nk = pad32(kidx.shape[1])
        oattn = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
        S = torch.empty((N, nk), dtype=x.dtype, device=x.device)
        scale = 1.0 / math.sqrt(cv)
        for b in range(B):
            Kg, Vg = _WideMaskAttention._kept(qkv[b], kidx[b], kcnt[b:b + 1], nk, C, code)
            _gemm_nt(qkv[b], 3 * C, N, C, Kg, nk, S, nk, code)                                       # S = Q Kg^T
            call("mu_softmax_rows", ptr(S), N, nk, ptr(kcnt[b:b + 1]), scale, code, stream())
            VgT = _transpose_tokens(Vg.view(1, nk, C), nk, C).view(C, nk)
            _gemm_nt(S, nk, N, nk, VgT, C, oattn[b], C, code)                                        # O = P Vg
        out = torch.empty_like(oattn)
        mean = torch.empty((B, N), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        call("mu_ln_rows_fwd", ptr(oattn), ptr(x), ptr(g), ptr(b_), ptr(out), ptr(mean), ptr(rstd), B * N, C, cv, float(eps), code, stream())
        ctx.save_for_backward(x, qkv, oattn, mean, rstd, g, kidx, kcnt, wqkv)
        ctx.dims, ctx.cv, ctx.scramble, ctx.code, ctx.nk = (B, H, W, C), cv, scramble, code, nk
        if scramble:
            return _transpose_tokens(out.view(B, C, N), C, N).view(B, H, W, C)
        return out

    @staticmethod
    def _kept(qkv_b, kidx_b, kcnt_b, nk, C, code):
        """The kept key / value rows of one image, [nk, C] each, zero rows behind the kept ones."""
        Kg = torch.empty((nk, C), dtype=qkv_b.dtype, device=qkv_b.device)
        Vg = torch.empty_like(Kg)
        call("mu_gather_rows", ptr(qkv_b[:, C:]), 3 * C, ptr(kidx_b), ptr(kcnt_b), ptr(Kg), nk, C, code, stream())
        call("mu_gather_rows", ptr(qkv_b[:, 2 * C:]), 3 * C, ptr(kidx_b), ptr(kcnt_b), ptr(Vg), nk, C, code, stream())
        return Kg, Vg

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, qkv, oattn, mean, rstd, g, kidx, kcnt, wqkv = ctx.saved_tensors
        B, H, W, C = ctx.dims
        N, cv, code, nk = H * W, ctx.cv, ctx.code, ctx.nk
        gout = gout.contiguous()
        if ctx.scramble:
            gout = _transpose_tokens(gout.view(B, N, C), N, C)
        dY = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
        gxh = torch.empty_like(dY)
        call("mu_ln_rows_bwd", ptr(gout), ptr(oattn), ptr(x), ptr(mean), ptr(rstd), ptr(g), ptr(dY), ptr(gxh), B * N, C, cv, code, stream())
        dg, db = _colsum_wide(gxh.view(B * N, C)), _colsum_wide(gout.view(B * N, C))
        dqkv = torch.zeros((B, N, 3 * C), dtype=x.dtype, device=x.device)              # masked keys: zero dK / dV rows
        S = torch.empty((N, nk), dtype=x.dtype, device=x.device)
        dP = torch.empty_like(S)
        scale = 1.0 / math.sqrt(cv)
        for b in range(B):
            cnt = kcnt[b:b + 1]
            Kg, Vg = _WideMaskAttention._kept(qkv[b], kidx[b], cnt, nk, C, code)
            _gemm_nt(qkv[b], 3 * C, N, C, Kg, nk, S, nk, code)
            call("mu_softmax_rows", ptr(S), N, nk, ptr(cnt), scale, code, stream())                   # P again
            _gemm_nt(dY[b], C, N, C, Vg, nk, dP, nk, code)                                           # dP = dO Vg^T
            dVg = _gemm_tn(dY[b], C, S, nk, N, C, nk, code)                                          # dVg = P^T dO
            call("mu_attn_wide_ds", ptr(S), ptr(dP), N, nk, ptr(cnt), scale, code, stream())          # dP <- dS
            KgT = _transpose_tokens(Kg.view(1, nk, C), nk, C).view(C, nk)
            _gemm_nt(dP, nk, N, nk, KgT, C, dqkv[b], 3 * C, code)                                    # dQ = dS Kg
            dKg = _gemm_tn(qkv[b], 3 * C, dP, nk, N, C, nk, code)                                    # dKg = dS^T Q
            call("mu_scatter_rows", ptr(dKg), ptr(kidx[b]), ptr(cnt), ptr(dqkv[b][:, C:]), 3 * C, N, C, code, stream())
            call("mu_scatter_rows", ptr(dVg), ptr(kidx[b]), ptr(cnt), ptr(dqkv[b][:, 2 * C:]), 3 * C, N, C, code, stream())
        gx = None
        if ctx.needs_input_grad[0]:
            wT = _transpose_tokens(wqkv.view(1, 3 * C, C), 3 * C, C).view(C, 3 * C)
            gx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
            _gemm_nt(dqkv, 3 * C, B * N, 3 * C, wT, C, gx, C, code)
            call("mu_add", ptr(gx), ptr(dY), ptr(gx), gx.numel(), dt(gx), stream())
        gw = _gemm_tn(x, C, dqkv, 3 * C, B * N, C, 3 * C, code)                                      # [3C, C]
        gb = _colsum_wide(dqkv.view(B * N, 3 * C))
        return (gx, gw[:cv, :cv], gb[:cv], gw[C:C + cv, :cv], gb[C:C + cv], gw[2 * C:2 * C + cv, :cv], gb[2 * C:2 * C + cv],
                dg[:cv], db[:cv], None, None, None, None)


def mask_attention(x, q, k, v, norm, kidx, kcnt, scramble=True, kidx_perm=False):
    """q,k,v: nn.Linear containers; norm: nn.LayerNorm([C]) container.  kidx [B, nkmax] int32 lists the visible keys of image b in
    kidx[b, :kcnt[b]].  kidx_perm=True promises that nkmax == N and every row is a whole permutation of 0..N-1 with the masked keys
    behind the visible ones (compact_keys() output); leave it False for index lists from anywhere else."""
    if x.shape[-1] > ATTN_WIDTHS[-1]:            # wider than the flash-style sweeps: the generic GEMM path
        return _WideMaskAttention.apply(x, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias, norm.weight, norm.bias, kidx, kcnt,
                                        norm.eps, scramble)
    return _MaskAttention.apply(x, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias, norm.weight, norm.bias, kidx, kcnt,
                                norm.eps, scramble, _cache_ok(), kidx_perm)


def compact_keys(keep):
    """keep [B, N] uint8 or int64 (non-zero = key visible) -> (kidx int32 [B, N], kcnt int32 [B], keep8 uint8 [B, N]): the visible
    keys of every image in ascending order followed by the masked ones (== torch.argsort(keep, 1, descending=True, stable=True)),
    on the current stream, no host sync."""
    if keep.dim() != 2 or keep.dtype not in (torch.uint8, torch.int64, torch.bool):
        raise RuntimeError("compact_keys expects a [B, N] uint8 / bool / int64 tensor")
    keep = keep.contiguous()
    if keep.dtype == torch.bool:
        keep = keep.view(torch.uint8)
    B, N = keep.shape
    kidx = torch.empty((B, N), dtype=torch.int32, device=keep.device)
    kcnt = torch.empty((B,), dtype=torch.int32, device=keep.device)
    keep8 = torch.empty((B, N), dtype=torch.uint8, device=keep.device)
    call("mu_compact_keys", ptr(keep), keep.element_size(), B, N, ptr(kidx), ptr(kcnt), ptr(keep8), stream())
    return kidx, kcnt, keep8
