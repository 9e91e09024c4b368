"""nn.Module surface of the reference (code/ade20k/ade_semantic.py:152-314 and
code/cityscapes/city_instance.py:216-276), backed by the HIP kernels.

Same class names, constructor signatures, forward signatures, error behaviour and ``state_dict``
keys/shapes as the reference, so its training scripts and checkpoints work unchanged; the
``north_star`` aliases (DoubleConv, Down, Up, MaskAttention, OutConv) are exported too.

The torch sub-modules (nn.Conv2d, nn.BatchNorm2d, nn.Linear, nn.LayerNorm ...) are used purely as
PARAMETER CONTAINERS -- they give identical parameter names, shapes and default initialisation
 -- their ``forward`` is never called.  Every forward here runs HIP kernels through
``maskunet_amd.ops``; CPU tensors raise (there is no fallback).

Public ``forward`` takes and returns NCHW tensors like the reference.  Inside a UNet the blocks
talk NHWC (``forward_nhwc``) in the compute dtype so no layout conversion happens between them.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU

_DEFAULT_DTYPE = torch.float32


def set_default_compute_dtype(dtype):
    """Compute/storage dtype of activations for modules created afterwards: torch.float32 (parity
    path, exact-fp32 MFMA) or torch.float16 (fp16 storage, fp32 accumulate)."""
    global _DEFAULT_DTYPE
    if dtype not in (torch.float32, torch.float16):
        raise TypeError("compute dtype must be torch.float32 or torch.float16")
    _DEFAULT_DTYPE = dtype


class _HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.compute_dtype = _DEFAULT_DTYPE

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.float16):
            raise TypeError("compute dtype must be torch.float32 or torch.float16")
        for m in self.modules():
            if isinstance(m, _HipModule):
                m.compute_dtype = dtype
        return self

    def _check_device(self, x):
        if not x.is_cuda:
            raise RuntimeError("maskunet_amd modules run on the GPU only (HIP kernels, no CPU fallback); "
                               "move the model and inputs to cuda")


class Mask2FormerAttention(_HipModule):
    """ade_semantic.py:152-190.  ``mask`` keeps the reference's attribute semantics: ``None`` until the first
    forward, then an additive {0,-inf} key mask cached while H*W is unchanged; it can be injected from outside
    in the reference's own form ([B,N,N] expand view, or anything indexable as mask[:,0,:]) or via
    ``set_keep_mask`` ([B,N], 1 = key visible).  ``mask_mode='resample'`` redraws it on every forward (what the
    reference does under multi-GPU nn.DataParallel)."""

    def __init__(self, channels, size):
        super().__init__()
        self.channels = channels
        self.size = size
        self.query = nn.Linear(channels, channels)
        self.key = nn.Linear(channels, channels)
        self.value = nn.Linear(channels, channels)
        self.norm = nn.LayerNorm([channels])
        self.mask_mode = "fixed"
        self._keep = None
        self._kidx = None
        self._kcnt = None

    # -- mask state -----------------------------------------------------------------------------
    @property
    def mask(self):
        if self._keep is None:
            return None
        keep = self._keep
        add = torch.where(keep > 0, torch.zeros((), device=keep.device), torch.full((), -float("inf"), device=keep.device))
        return add.unsqueeze(1).expand(-1, keep.shape[1], -1)

    @mask.setter
    def mask(self, value):
        if value is None:
            self._keep = self._kidx = self._kcnt = None
            return
        if value.dim() == 3:
            value = value[:, 0, :]
        self.set_keep_mask(value == 0 if value.is_floating_point() else value)

    def set_keep_mask(self, keep):
        """keep: [B,N], non-zero = key visible.  The compacted kept-key index list is built on the device at the next forward."""
        keep = (keep != 0).to(torch.uint8)
        self._keep = keep
        self._kidx = self._kcnt = None

    def _compact(self, device):
        if self._kidx is None or self._kidx.device != device:
            # visible keys first, ascending, then the masked ones (== a stable descending argsort): one HIP launch, no host sync
            self._kidx, self._kcnt, self._keep = ops.compact_keys(self._keep.to(device))
        return self._kidx, self._kcnt

    def _mask_for(self, x_nhwc):
        B, H, W, _ = x_nhwc.shape
        N = H * W
        if self._keep is None or self._keep.shape[-1] != N or self.mask_mode == "resample":
            # same draw as the reference (ade_semantic.py:178): randint(0,2,(B,H,W)) on the input's device; the int64 draw goes
            # straight into the compaction kernel (no uint8 copy, no torch.argsort: the per-step cost of mask_mode="resample")
            binary = torch.randint(0, 2, (B, H, W), device=x_nhwc.device)
            self._kidx, self._kcnt, self._keep = ops.compact_keys(binary.view(B, -1))
        kidx, kcnt = self._compact(x_nhwc.device)
        if kidx.shape[0] != B:
            if kidx.shape[0] == 1:      # reference: a cached batch-1 mask broadcasts
                kidx, kcnt = kidx.expand(B, -1).contiguous(), kcnt.expand(B).contiguous()
            else:
                raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({kidx.shape[0]}) at "
                                   "non-singleton dimension 0")
        return kidx, kcnt

    # -- forward --------------------------------------------------------------------------------
    def forward_nhwc(self, x, scramble=True):
        if x.shape[-1] != ops.attn_width(self.channels):
            raise ValueError("Input channel size does not match initialized channel size.")
        kidx, kcnt = self._mask_for(x)
        # kidx comes from ops.compact_keys: whole permutations, masked keys last -> the backward may skip the memset of dqkv
        return ops.mask_attention(x, self.query, self.key, self.value, self.norm, kidx, kcnt, scramble, kidx_perm=True)

    def forward(self, x):
        batch_size, channels, height, width = x.size()
        if channels != self.channels:
            raise ValueError("Input channel size does not match initialized channel size.")
        self._check_device(x)
        # any channel count (the reference takes any, :153-161): widths the flash-style kernels are not built for run zero-padded to
        # the next of 32 / 64 / 128 / 256 (scores scaled by the true 1/sqrt(C), LayerNorm over the true channels); above 256 the
        # generic GEMM path at the next multiple of 32 (ops._WideMaskAttention)
        cw = ops.attn_width(channels)
        y = self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype, cw), scramble=False)    # [B,N,cw] token-major
        if cw != channels:
            y = y.view(batch_size, height * width, cw)[:, :, :channels].contiguous()
        # the reference returns this buffer re-viewed as [B,C,H,W] (ade_semantic.py:190)
        return y.view(batch_size, channels, height, width).to(x.dtype)


class ConvBlock(_HipModule):
    """ade_semantic.py:192-210."""

    def __init__(self, in_channels, out_channels, mid_channels=None, residual=False):
        super().__init__()
        self.residual = residual
        if not mid_channels:
            mid_channels = out_channels
        self.conv_block = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.GELU(),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
        )

    def forward_nhwc(self, x, tail_bn=None, res_link=None, x_encoded=False, enc_out=False):
        """tail_bn: the nn.BatchNorm2d DownSample / UpSample apply right behind this block (:219,240); in training mode it is
        folded into the block's last BatchNorm (ops.bn_pair).  res_link: ops.GradLink shared with the op that produced x (residual
        blocks only): the gradient of the `x +` branch (:208) is joined inside that op's backward kernel.
        fp32x mode only (ignored otherwise): x_encoded = x arrives as a chunk-encoded matrix operand (non-residual blocks: x feeds
        nothing but the first conv); enc_out = the caller promises that the output feeds nothing but a conv, so the last
        BatchNorm-apply pass writes it chunk-encoded.  Inside the block the BatchNorm outputs / gradients that only feed a conv are
        written encoded by the BatchNorm kernels themselves (ops.EncLink)."""
        cb = self.conv_block
        if x_encoded and self.residual:
            raise RuntimeError("ConvBlock: a residual block adds its input and needs it plain")
        if ops.eval_fusable(cb[1], cb[4], tail_bn):
            # inference: every BatchNorm (+ residual add, + GELU) lives in the epilogue of the conv in front of it
            y = ops.conv_bn_act_eval(x, cb[0].weight, None, cb[1], ACT_GELU)
            if self.residual:
                y = ops.conv_bn_act_eval(y, cb[3].weight, None, cb[4], ACT_GELU, res=x)
                return y if tail_bn is None else ops.bn_act(y, tail_bn, ACT_NONE)
            return ops.conv_bn_act_eval(y, cb[3].weight, None, cb[4], ACT_NONE, bn2=tail_bn)
        # fp32x: the dy of a conv is the dx of the BatchNorm behind it (one link per pair); not for the 3-channel stem, whose weight
        # gradient is a plain-FMA kernel.  mid: bn1's output feeds conv2 only.
        l0 = ops.enc_link(x) if cb[0].weight.shape[1] > 3 else None
        l1 = ops.enc_link(x) if cb[3].weight.shape[1] > 3 else None      # (a second conv with <= 3 input channels is a plain-FMA layer too)
        fx = ops.enc_link(x) is not None                                 # fp32x mode with a backward to come
        fm = l1 is not None                                              # ... and the mid activation feeds a matrix-core conv: written encoded
        y, st = ops.conv_stats(x, cb[0].weight, want=cb[1].training, x_encoded=x_encoded and ops._is_x(x), dy_link=l0)   # BatchNorm statistics from the conv epilogue where it has one
        y = ops.bn_act(y, cb[1], ACT_GELU, stats=st, enc_out=fm, dx_link=l0)
        y, st = ops.conv_stats(y, cb[3].weight, want=cb[4].training, x_encoded=fm, dy_link=l1)
        if self.residual:
            y = ops.bn_act(y, cb[4], ACT_GELU, res=x, stats=st, res_link=res_link, dx_link=l1,
                           enc_out=enc_out and tail_bn is None and fx)            # gelu(x + BN(conv(...)))  (:208)
            return y if tail_bn is None else ops.bn_act(y, tail_bn, ACT_NONE)
        if tail_bn is not None:
            return ops.bn_pair(y, cb[4], tail_bn, stats=st, dx_link=l1)
        return ops.bn_act(y, cb[4], ACT_NONE, stats=st, dx_link=l1, enc_out=enc_out and fx)

    def forward(self, x):
        self._check_device(x)
        cout = self.conv_block[3].out_channels
        return ops.to_nchw(self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype)), cout, x.dtype)


class DownSample(_HipModule):
    """ade_semantic.py:212-229.  emb_layer is dead weight kept for checkpoint compatibility (:222-225)."""

    def __init__(self, in_channels, out_channels, emb_dim=256):
        super().__init__()
        self.maxpool_conv = nn.Sequential(
            nn.MaxPool2d(2),
            ConvBlock(in_channels, in_channels, residual=True),
            ConvBlock(in_channels, out_channels),
            nn.BatchNorm2d(out_channels),
        )
        self.emb_layer = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, out_channels))
        self.out_channels = out_channels

    def forward_nhwc(self, x, skip_link=None):
        """skip_link: ops.GradLink shared with the UpSample that takes x as its skip tensor (UNet wiring, :292-309)."""
        mc = self.maxpool_conv
        link = ops.grad_link(x)                           # residual branch of mc[1] -> the pool's backward
        x = ops.maxpool2(x, link, skip_link)
        fx = ops.enc_link(x) is not None                  # fp32x mode with a backward to come: mc[1]'s output goes to mc[2]'s first conv only
        x = mc[1].forward_nhwc(x, res_link=link, enc_out=fx)
        return mc[2].forward_nhwc(x, tail_bn=mc[3], x_encoded=fx)       # ConvBlock + the BatchNorm behind it (:218-219)

    def forward(self, x):
        self._check_device(x)
        return ops.to_nchw(self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype)), self.out_channels, x.dtype)


class UpSample(_HipModule):
    """ade_semantic.py:231-256."""

    def __init__(self, in_channels, out_channels, emb_dim=256):
        super().__init__()
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.out_channels = out_channels
        self.conv = nn.Sequential(
            ConvBlock(in_channels, in_channels, residual=True),
            ConvBlock(in_channels, out_channels, in_channels // 2),
            nn.BatchNorm2d(out_channels),
        )
        self.emb_layer = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, out_channels))

    def forward_nhwc(self, x, skip_x, cx=None, cs=None, skip_link=None):
        """cx / cs: true channel counts of x / skip_x when they are not multiples of the 32-channel storage padding.
        skip_link: ops.GradLink shared with the DownSample that also consumes skip_x (UNet wiring)."""
        link = ops.grad_link(x, skip_x)          # residual branch of conv[0] -> the concat's backward
        x = ops.upcat(x, skip_x, cx, cs, link, skip_link)         # cat([skip_x, up(x)], dim=1)  (:250-253)
        fx = ops.enc_link(x) is not None                  # (as in DownSample)
        x = self.conv[0].forward_nhwc(x, res_link=link, enc_out=fx)
        return self.conv[1].forward_nhwc(x, tail_bn=self.conv[2], x_encoded=fx)      # ConvBlock + the BatchNorm behind it (:239-240)

    def forward(self, x, skip_x):
        self._check_device(x)
        y = self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype), ops.to_nhwc(skip_x, self.compute_dtype), x.shape[1], skip_x.shape[1])
        return ops.to_nchw(y, self.out_channels, x.dtype)


class _Head(nn.Sequential):
    """Conv2d 1x1 (bias) -> BatchNorm2d -> ReLU container with the reference's Sequential indices."""


def _head_1x1_bn_relu(seq, x):
    if ops.eval_fusable(seq[1]):
        return ops.conv_bn_act_eval(x, seq[0].weight, seq[0].bias, seq[1], ACT_RELU)
    y = ops.conv(x, seq[0].weight, seq[0].bias)
    return ops.bn_act(y, seq[1], ACT_RELU)


class UNet(_HipModule):
    """ade_semantic.py:258-314 (1-head) and city_instance.py:216-276 (3-head, when ``embed_dim`` is given).

    ``UNet(c_in, c_out)`` -> forward returns [B,c_out,H,W]; ``UNet(c_in, c_out, embed_dim)`` -> forward returns
    (semantic, boundary, embeddings).  ``hw`` (keyword, default 128) sizes ``norm = LayerNorm([64,hw,hw])``; the
    reference hard-codes 128 (:281)."""

    def __init__(self, c_in=3, c_out=3, embed_dim=None, *, hw=128):
        super().__init__()
        self.c_out = c_out
        self.initial_conv = ConvBlock(c_in, 64)
        self.downsample1 = DownSample(64, 128)
        self.self_attention1 = Mask2FormerAttention(128, 128)
        self.downsample2 = DownSample(128, 256)
        self.self_attention2 = Mask2FormerAttention(256, 256)
        self.downsample3 = DownSample(256, 256)
        self.self_attention3 = Mask2FormerAttention(256, 256)

        self.bottom1 = ConvBlock(256, 512)
        self.bottom2 = ConvBlock(512, 512)
        self.bottom3 = ConvBlock(512, 256)

        self.dropout = nn.Dropout(0.3)

        self.upsample1 = UpSample(512, 128)
        self.self_attention4 = Mask2FormerAttention(128, 128)
        self.upsample2 = UpSample(256, 64)
        self.self_attention5 = Mask2FormerAttention(64, 64)
        self.upsample3 = UpSample(128, 64)
        self.self_attention6 = Mask2FormerAttention(64, 64)
        self.norm = nn.LayerNorm([64, hw, hw])
        self.final_layer = nn.Sequential(nn.Conv2d(64, c_out, kernel_size=1), nn.BatchNorm2d(c_out), nn.ReLU())
        self.three_head = embed_dim is not None
        if self.three_head:
            self.boundary_head = nn.Sequential(
                nn.Conv2d(c_out, 32, kernel_size=3, padding=1), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 1, kernel_size=1))
            self.embedding_head = nn.Sequential(nn.Conv2d(64, embed_dim, kernel_size=1), nn.BatchNorm2d(embed_dim), nn.ReLU())
            self.embed_dim = embed_dim
        # test hook: explicit dropout keep-masks (NHWC uint8) for the two dropout sites
        self.dropout_masks = None

    def attention_blocks(self):
        return [getattr(self, f"self_attention{i}") for i in range(1, 7)]

    def set_keep_masks(self, keeps):
        """Inject the six key keep-masks ([B,N_k] each) -- parity tests and reproducible benches."""
        for blk, k in zip(self.attention_blocks(), keeps):
            blk.set_keep_mask(k)

    def set_mask_mode(self, mode):
        if mode not in ("fixed", "resample"):
            raise ValueError("mask_mode must be 'fixed' or 'resample'")
        for blk in self.attention_blocks():
            blk.mask_mode = mode

    def _drop(self, x, i):
        m = None if self.dropout_masks is None else self.dropout_masks[i]
        return ops.dropout(x, self.dropout.p, self.training, m)

    def forward_nhwc(self, x):
        """x: NHWC [B,H,W,32] (3 real channels) -> tuple of NHWC head outputs (padded channels)."""
        # x1, x2, x3 each feed a DownSample and, as skip tensors, an UpSample (:292-309): their two gradients are joined inside
        # the pool's backward kernel (ops.GradLink) instead of by three autograd accumulation kernels
        if not (torch.is_grad_enabled() and self.training and ops.MULTI_PREP):
            return self._forward_nhwc(x)
        # training: every conv weight's compute layouts from ONE launch; each conv below consumes its one-shot entry, and whatever an
        # exception leaves behind is dropped so that no later forward can pick up layouts of older weights
        ws = self.__dict__.get("_conv_ws")
        if ws is None:
            ws = self.__dict__["_conv_ws"] = ([m for m in self.modules() if isinstance(m, nn.Conv2d)], {})
        # the tensors the convs below will see NOW: under torch.func.functional_call (GraphedStep) `m.weight` is the call's fresh leaf,
        # not the Parameter -- the one-shot entries must sit on those objects or every conv falls back to its own prep launch
        weights = [m.weight for m in ws[0]]
        first = self.initial_conv.conv_block[0].weight
        ops.prep_conv_weights(ws[1], weights, x.dtype, fwd_only=() if x.requires_grad else (first,))
        try:
            return self._forward_nhwc(x)
        finally:
            for w in weights:
                w._mu_step = None

    def _forward_nhwc(self, x):
        x1 = self.initial_conv.forward_nhwc(x)
        l1 = ops.grad_link(x1)
        x2 = self.downsample1.forward_nhwc(x1, skip_link=l1)
        x2 = self.self_attention1.forward_nhwc(x2)
        l2 = ops.grad_link(x2)
        x3 = self.downsample2.forward_nhwc(x2, skip_link=l2)
        x3 = self.self_attention2.forward_nhwc(x3)
        l3 = ops.grad_link(x3)
        x4 = self.downsample3.forward_nhwc(x3, skip_link=l3)
        x4 = ops.cut_point(self.self_attention3.forward_nhwc(x4))     # encoder | bottleneck + decoder: where GraphedStep may split its capture

        fx = ops.enc_link(x4) is not None                # fp32x mode with a backward to come: the bottleneck blocks hand their outputs on chunk-encoded
        x4 = self.bottom1.forward_nhwc(x4, enc_out=fx)
        x4 = self.bottom2.forward_nhwc(x4, x_encoded=fx, enc_out=fx)
        x4 = self.bottom3.forward_nhwc(x4, x_encoded=fx)

        y = self.upsample1.forward_nhwc(x4, x3, skip_link=l3)
        y = self._drop(y, 0)
        y = self.self_attention4.forward_nhwc(y)
        y = self.upsample2.forward_nhwc(y, x2, skip_link=l2)
        y = self._drop(y, 1)
        y = self.self_attention5.forward_nhwc(y)
        y = self.upsample3.forward_nhwc(y, x1, skip_link=l1)
        B, H, W, _ = y.shape
        # attention 6 feeds the per-sample LayerNorm, which works on the NCHW-flat memory == the
        # token-major buffer itself, so the scramble transpose is skipped here (:310-311)
        y = self.self_attention6.forward_nhwc(y, scramble=False)             # [B,N,64] == NCHW-flat [B,64,H,W]
        if tuple(self.norm.normalized_shape) != (64, H, W):
            raise RuntimeError(f"Given normalized_shape={list(self.norm.normalized_shape)}, expected input with shape "
                               f"[*, 64, {H}, {W}]")
        y = ops.ln_sample(y.view(B, -1), self.norm.weight, self.norm.bias, self.norm.eps)
        y = ops.to_nhwc(y.view(B, 64, H, W), self.compute_dtype)             # NCHW-flat -> NHWC for the 1x1 heads
        sem = _head_1x1_bn_relu(self.final_layer, y)
        if not self.three_head:
            return (sem,)
        emb = _head_1x1_bn_relu(self.embedding_head, y)
        bh = self.boundary_head
        if ops.eval_fusable(bh[1]):
            b = ops.conv_bn_act_eval(sem, bh[0].weight, bh[0].bias, bh[1], ACT_RELU)
        else:
            b = ops.conv(sem, bh[0].weight, bh[0].bias)
            b = ops.bn_act(b, bh[1], ACT_RELU)
        b = ops.conv(b, bh[3].weight, bh[3].bias)
        return sem, b, emb

    def logits_nhwc(self, x):
        """Semantic head output as the internal NHWC channel-padded tensor [B,H,W,pad32(c_out)] (compute dtype) -- feed it to
        ``maskunet_amd.pixel_cross_entropy_nhwc`` / ``mean_iou`` to skip the NCHW fp32 materialisation in a training step."""
        self._check_device(x)
        return self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype))[0]

    def forward_u8(self, images_u8_hwc, bgr=False):
        """Forward from decoded image bytes: uint8 [B,H,W,3] (HWC) of ANY size, standing in for the reference's pipeline
        ``forward(ToTensor(cv2.resize(img, (hw, hw), interpolation=cv2.INTER_LINEAR)))`` (ade_semantic.py:72-76,85); ``bgr=True`` also
        applies ``cv2.cvtColor(img, cv2.COLOR_BGR2RGB)`` (:65), i.e. takes what cv2.imread returns.  The resize is the RESTATED OpenCV
        4.10 8-bit algorithm (11-bit fixed-point coefficients, 2x2 area shortcut; its CPU restatement lives with the test
        infrastructure) -- exact against that restatement, NOT pinned to bytes produced by a real cv2 build (cv2 is not installed here and not vendored by the reference:
        SURVEY 8-f4 stays "partial" for this half).  ToTensor's division by 255 is exact.  Images that already have the network's size
        skip the resize (cv2.resize to the same size is the identity)."""
        self._check_device(images_u8_hwc)
        H, W = int(self.norm.normalized_shape[1]), int(self.norm.normalized_shape[2])
        if tuple(images_u8_hwc.shape[1:3]) == (H, W) and not bgr:
            x = ops.u8_hwc_to_nhwc(images_u8_hwc, self.compute_dtype)
        else:
            x = ops.resize_u8_to_nhwc(images_u8_hwc, (W, H), self.compute_dtype, bgr=bgr)
        return self._finish(self.forward_nhwc(x))

    def forward(self, x):
        self._check_device(x)
        return self._finish(self.forward_nhwc(ops.to_nhwc(x, self.compute_dtype)))

    def _finish(self, outs):
        sem = ops.to_nchw(outs[0], self.c_out, torch.float32)
        if not self.three_head:
            return sem
        return sem, ops.to_nchw(outs[1], 1, torch.float32), ops.to_nchw(outs[2], self.embed_dim, torch.float32)


class InstanceUNet(UNet):
    """The 3-head UNet of code/cityscapes/city_instance.py:216-276 with its own constructor signature
    ``UNet(c_in=3, c_out=3, embed_dim=16)``."""

    def __init__(self, c_in=3, c_out=3, embed_dim=16, *, hw=128):
        super().__init__(c_in, c_out, embed_dim, hw=hw)


# north_star aliases -------------------------------------------------------------------------------
DoubleConv = ConvBlock
Down = DownSample
Up = UpSample
MaskAttention = Mask2FormerAttention


class OutConv(_HipModule):
    """Alias-style wrapper for the reference's ``final_layer`` (Conv1x1 -> BN -> ReLU, ade_semantic.py:283-287)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.final_layer = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1), nn.BatchNorm2d(out_channels), nn.ReLU())
        self.out_channels = out_channels

    def forward(self, x):
        self._check_device(x)
        y = _head_1x1_bn_relu(self.final_layer, ops.to_nhwc(x, self.compute_dtype))
        return ops.to_nchw(y, self.out_channels, x.dtype)
