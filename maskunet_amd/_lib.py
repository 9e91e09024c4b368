"""ctypes binding of libmaskunet_hip.so (C ABI declared in include/maskunet_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the product
path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C maskunet_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_ulonglong, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MU_LIB_PATH") or os.path.join(_HERE, "libmaskunet_hip.so")     # MU_LIB_PATH: debug builds (tools/build_attn_variant.sh)

MU_F32, MU_F16, MU_F32X = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
_ERR = {-1: "MU_ERR_ARG", -2: "MU_ERR_SHAPE", -3: "MU_ERR_LAUNCH", -4: "MU_ERR_WORKSPACE"}

# name -> (restype, argtypes); must list EVERY symbol of include/maskunet_hip.h (tests/test_abi.py checks)
P, I, L, F = c_void_p, c_int, c_long, c_float
SIGNATURES = {
    "mu_version_host": (c_char_p, []),
    "mu_transpose": (I, [P, I, L, P, I, L, I, I, I, P]),
    "mu_transpose_pad": (I, [P, I, L, P, I, L, I, I, I, I, P]),
    "mu_cast": (I, [P, I, P, I, L, P]),
    "mu_split_encode": (I, [P, P, L, P]),
    "mu_split_encode_h": (I, [P, P, L, P]),
    "mu_split_encode_h4": (I, [P, P, L, P]),
    "mu_split_encode_h4x": (I, [P, P, P, L, P]),
    "mu_conv_wgrad_h1": (I, [P, P, P, P, I, I, I, I, I, I, I, L, L, P, L, P]),
    "mu_bn_act_fwd_enc": (I, [P, P, P, P, L, I, P, P, P, P, I, P]),
    "mu_dy_encode_h_workspace_bytes": (L, []),
    "mu_dy_encode_h": (I, [P, P, P, L, P, L, P]),
    "mu_conv_dgrad_h": (I, [P, P, P, P, I, I, I, I, I, L, L, P]),
    "mu_conv_wgrad_h_workspace_bytes": (L, [I, I, I, I, I]),
    "mu_conv_wgrad_h": (I, [P, P, P, P, I, I, I, I, I, I, I, L, L, P, L, P]),
    "mu_bn_act_bwd_h": (I, [P, P, P, P, P, L, I, P, P, P, P, I, I, P, P, P, P, L, P]),
    "mu_bn_pair_bwd_h": (I, [P, P, P, L, I, P, P, P, P, P, P, P, P, P, P, P, L, P]),
    "mu_conv1x1_fwd_enc_h": (I, [P, P, P, P, L, I, I, L, L, P]),
    "mu_prep_weight": (I, [P, P, I, I, I, I, I, I, I, P]),
    "mu_conv_fwd": (I, [P, P, P, P, I, I, I, I, I, I, L, L, I, P]),
    "mu_conv_stats_rows": (I, [I, I, I, I, I, I, I]),
    "mu_conv_fwd_stats": (I, [P, P, P, P, I, I, I, I, I, I, L, L, I, P, P]),
    "mu_conv_wgrad_workspace_bytes": (L, [I, I, I, I, I, I]),
    "mu_conv_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, L, L, P, L, I, P]),
    "mu_conv_wgrad_bias_supported": (I, [I, I, I, I]),
    "mu_conv_wgrad_bias": (I, [P, P, P, P, I, I, I, I, I, I, I, I, L, L, P, L, I, P]),
    "mu_colsum_workspace_bytes": (L, [I]),
    "mu_colsum": (I, [P, L, I, L, P, P, L, I, P]),
    "mu_bn_workspace_bytes": (L, [I]),
    "mu_bn_train_stats": (I, [P, L, I, L, P, P, P, P, P, I, F, F, P, L, I, P]),
    "mu_bn_train_stats_rows": (I, [P, I, L, I, P, P, P, P, P, I, F, F, P, L, P]),
    "mu_bn_eval_stats": (I, [P, P, F, P, P, I, I, P]),
    "mu_bn_act_fwd": (I, [P, P, P, L, I, L, P, P, P, P, I, I, P]),
    "mu_bn_act_bwd": (I, [P, P, P, P, P, L, I, L, P, P, P, P, I, I, P, P, P, L, I, P]),
    "mu_bn_pair_compose": (I, [P, P, P, P, I, I, L, F, F, F, P, P, P, P, P, P, P, P]),
    "mu_bn_act_bwd_scaled": (I, [P, P, P, P, P, L, I, L, P, P, P, P, I, I, P, P, P, P, L, I, P]),
    "mu_bn_pair_bwd": (I, [P, P, P, L, I, L, P, P, P, P, P, P, P, P, P, P, L, I, P]),
    "mu_ln_sample_workspace_bytes": (L, [I]),
    "mu_ln_sample_fwd": (I, [P, P, P, P, P, P, I, L, F, P, L, I, P]),
    "mu_ln_sample_bwd": (I, [P, P, P, P, P, P, P, P, I, L, P, L, I, P]),
    "mu_maxpool2_fwd": (I, [P, P, I, I, I, I, I, P]),
    "mu_maxpool2_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "mu_upcat_fwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "mu_upcat_bwd": (I, [P, P, P, I, I, I, I, I, I, P]),
    "mu_upcat_compact_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "mu_upcat_compact_bwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "mu_dropout": (I, [P, P, L, F, c_ulonglong, P, P, I, P]),
    "mu_dropout_step": (I, [P, P, L, F, c_ulonglong, P, P, P, I, P]),
    "mu_add": (I, [P, P, P, L, I, P]),
    "mu_attn_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, I, P]),
    "mu_attn_bwd_workspace_bytes": (L, [I, I, I]),
    "mu_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, L, I, P]),
    "mu_gather_rows": (I, [P, L, P, P, P, I, I, I, P]),
    "mu_scatter_rows": (I, [P, P, P, P, L, I, I, I, P]),
    "mu_softmax_rows": (I, [P, I, I, P, F, I, P]),
    "mu_attn_wide_ds": (I, [P, P, I, I, P, F, I, P]),
    "mu_ln_rows_fwd": (I, [P, P, P, P, P, P, P, L, I, I, F, I, P]),
    "mu_ln_rows_bwd": (I, [P, P, P, P, P, P, P, P, L, I, I, I, P]),
    "mu_ce_workspace_bytes": (L, []),
    "mu_ce_fwd": (I, [P, P, L, I, I, L, P, P, P, P, L, I, P]),
    "mu_ce_bwd": (I, [P, P, P, P, P, F, L, I, I, L, P, I, P]),
    "mu_ce_nchw_fwd": (I, [P, P, I, I, L, L, P, P, P, P, L, I, P]),
    "mu_ce_nchw_bwd": (I, [P, P, P, P, P, F, I, I, L, L, P, I, P]),
    "mu_mean_iou": (I, [P, P, L, I, L, L, L, L, F, P, P, I, P]),
    "mu_inst_triplet_workspace_bytes": (L, [I, I]),
    "mu_inst_triplet_fwd": (I, [P, P, I, I, I, I, I, F, P, I, I, P, L, P, P]),
    "mu_inst_triplet_bwd": (I, [P, I, I, I, I, P, I, I, P, P, P]),
    "mu_u8_to_nhwc": (I, [P, P, L, I, I, I, P]),
    "mu_adamw_chunk": (I, []),
    "mu_adamw_multi": (I, [P, P, P, I, I, F, F, F, F, F, F, P, P, I, P, P]),
    "mu_prep_qkv": (I, [P, P, P, P, P, P, P, P, I, I, P]),
    "mu_prep_weights_multi": (I, [P, I, L, P, I, P]),
    "mu_conv1x1_add_supported": (I, [I, I, I]),
    "mu_conv1x1_fwd_add": (I, [P, P, P, P, L, I, I, L, L, I, P]),
    "mu_maxpool2_bwd_acc": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "mu_upcat_bwd_acc": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "mu_compact_keys": (I, [P, I, I, I, P, P, P, P]),
    "mu_resize_u8_nhwc": (I, [P, I, I, I, I, I, P, P, I, I, I, I, P]),
    "mu_resize_nearest_u8": (I, [P, I, I, I, P, I, I, P]),
    "mu_conv_fwd_fused": (I, [P, P, P, P, P, I, P, I, I, I, I, I, I, L, L, I, P]),
    "mu_bn_eval_fold": (I, [P, P, P, P, F, P, P, P, P, F, P, P, P, I, I, P]),
    "mu_attn_bwd_phases": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, L, I, I, P]),
    "mu_clock_probe": (I, [P, P, I, I, P]),
    "mu_attn_fwd_padded": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, I, P]),
    "mu_attn_bwd_phases_padded": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, L, I, I, P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"maskunet_amd: HIP library not found at {LIB_PATH}. It must be built (hipcc --offload-arch=gfx950): "
            "run `make -C maskunet_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("MU_LIB_PATH") and not hasattr(lib, name):
            continue                     # debug builds of an older source tree (A/B profiling): entry points added since are simply absent
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return MU_F32
    if d == torch.float16:
        return MU_F16
    raise TypeError(f"maskunet_amd supports float32 and float16 compute, got {d}")


# fp32 compute: how the MATRIX products run (conv / Linear / attention; everything else is plain fp32 either way).
#   "highest": exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, 157 TF/s peak) -- bit-level fp32 FMA chains, the parity path;
#   "high":    MU_F32X -- fp32 storage, matrix products on the 16-bit matrix cores with split operands, ~3x faster:
#              * conv / Linear forward: every operand as bf16 hi + lo, three bf16 MFMAs per product, fp32 accumulate (~1e-5 relative
#                per product, unbiased) -- the scheme torch.set_float32_matmul_precision("high") names;
#              * attention: q / k / v / dY as fp16 hi + lo pairs, the softmax probabilities P and dS as ONE fp16 operand straight from
#                the accumulators (2^-12 relative per element) under exact power-of-two range scales: two MFMAs per P / dS product;
#              * 3x3 conv backward (round 6): dy as ONE power-of-two-scaled fp16 operand against an fp16 hi + lo partner (weights for
#                the data gradient, the saved input for the weight gradient): two MFMAs per product.
#              Outputs stay within north_star's 1e-3 of the reference, gradients within the fp32 gates of tests/_gpu_checks.py.
F32_MATMUL_PRECISION = os.environ.get("MU_F32_MATMUL", "highest")
if F32_MATMUL_PRECISION not in ("highest", "high"):
    raise ValueError(f'MU_F32_MATMUL must be "highest" or "high", got {F32_MATMUL_PRECISION!r}')


def set_float32_matmul_precision(precision: str):
    """Process-wide, like torch.set_float32_matmul_precision: "highest" (default: exact-fp32 MFMA) or "high" (MU_F32X: fp32 storage, split
    16-bit operands on the matrix cores -- three bf16 terms per forward product, single-term fp16 P / dS / dy with fp16-pair partners in
    attention and the 3x3 backward; see the comment above F32_MATMUL_PRECISION).  fp16 compute is unaffected."""
    global F32_MATMUL_PRECISION
    if precision not in ("highest", "high"):
        raise ValueError('precision must be "highest" or "high"')
    F32_MATMUL_PRECISION = precision


def get_float32_matmul_precision() -> str:
    return F32_MATMUL_PRECISION


def mdt(t_or_dtype) -> int:
    """dtype code for the matrix entry points (mu_conv_fwd*, mu_conv1x1_fwd_add, mu_conv_wgrad*, mu_attn_*)."""
    d = dt(t_or_dtype)
    return MU_F32X if (d == MU_F32 and F32_MATMUL_PRECISION == "high") else d


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("maskunet_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (every entry point takes it as its last argument).
    torch.cuda.current_stream() builds a Python Stream object per call (~8 us, several hundred calls per training step); the raw query is the
    same value without the object."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# optional HIP-event probe (bench.py): {"pred": f(name, args) -> bool | str, "events": []} brackets the matching
# entry points with events on the launch stream; None = off (no overhead).
PROBE = None


_fns = {}


def _fn(name):
    f = _fns.get(name)
    if f is None:
        f = _fns[name] = getattr(load(), name)
    return f


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    probe = PROBE
    tag = probe["pred"](name, args) if probe is not None else None
    if tag:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.ExternalStream(args[-1]) if isinstance(args[-1], int) and args[-1] else torch.cuda.current_stream()
        e0.record(st)                # on the stream the kernel is launched on (the last argument of every entry point)
        rc = _fn(name)(*args)
        e1.record(st)
        probe["events"].append((name if tag is True else tag, e0, e1))      # pred may name the launch (a str) instead of True
    else:
        rc = _fn(name)(*args)
    if rc != 0:
        raise RuntimeError(f"maskunet_amd: {name} failed with {_ERR.get(rc, rc)}")


_workspaces = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, current stream).  Kernels that use it are stream-ordered, and every consumer finishes
    with it before the next launch on the same stream touches it; work on another stream (side-stream weight gradients, a
    graph-capture stream) gets a buffer of its own."""
    key = (device.type, device.index, stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
