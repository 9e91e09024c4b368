"""GPU parity tests, kernel level: every HIP op (called through maskunet_amd.ops -> ctypes -> the C ABI of
libmaskunet_hip.so) against stock torch / the CPU oracle on the same seeded inputs.  fp32 compute must meet the
north_star's 1e-3; fp16 compute (fp16 storage, fp32 accumulate) is held to 3e-2."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16]


def _assert_all(results):
    bad = [(n, e, t) for n, e, t in results if not (e <= t)]
    assert not bad, "parity failures (name, err, tol): " + "; ".join(f"{n}: {e:.3e} > {t:.1e}" for n, e, t in bad)


def test_library_loaded_is_in_tree():
    from maskunet_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.mu_version_host()
    assert _lib.LIB_PATH.endswith("maskunet_amd/libmaskunet_hip.so")


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_transpose(dtype) + G.check_layout_roundtrip(dtype) + G.check_prep_weight(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fwd_dgrad_wgrad(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_conv(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_act(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_bn_act(dtype) + G.check_conv_stats(dtype) + G.check_bn_pair(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_upsample_concat(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_pool_up(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_ln_sample(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_ln_sample(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_dropout(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_fwd_bwd(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_attention(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_any_channel_count(dtype):
    """Mask2FormerAttention(channels, size) takes any channel count in the reference (ade_semantic.py:153-161); widths the kernels are
    not built for run zero-padded (scores scaled by the true 1/sqrt(C), LayerNorm over the true channels) -- VERDICT r3 missing #6."""
    from tests import _gpu_checks as G
    _assert_all(G.check_attention(dtype, cases=[(2, 8, 8, 24), (2, 8, 12, 48), (1, 12, 12, 96), (1, 8, 8, 200), (1, 16, 4, 130)]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_attention_above_256_channels_generic_path(dtype):
    """Mask2FormerAttention(channels > 256): the generic GEMM path (ops._WideMaskAttention, csrc/attn_wide.hip) against the CPU oracle,
    forward + every gradient -- 288 and 512 channels, a channel count that is no multiple of 32, a batch-1 image, odd token counts."""
    from tests import _gpu_checks as G
    _assert_all(G.check_attention(dtype, cases=[(2, 8, 8, 288), (1, 6, 10, 300), (2, 4, 4, 512), (1, 16, 8, 320)]))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_forward_overflow_redo_path(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_overflow_redo(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_bwd_masked_rows_zeroed_by_sweep_or_memset(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_bwd_masked_rows(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_image_without_visible_keys_is_nan_like_the_reference(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_no_visible_key(dtype))


def test_attention_mask_semantics():
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_mask_semantics())


def test_compact_keys_matches_stable_argsort():
    from tests import _gpu_checks as G
    _assert_all(G.check_compact_keys())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gradient_joins_and_qkv_prep(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_grad_joins(dtype) + G.check_prep_qkv(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_grad_links_whole_model(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_grad_links_model(dtype))


def test_error_codes_not_exceptions():
    """Bad arguments come back as negative status codes and surface as RuntimeError on the Python side."""
    from maskunet_amd import _lib
    lib = _lib.load()
    assert lib.mu_conv_fwd(None, None, None, None, 1, 1, 1, 32, 32, 9, 32, 32, 0, None) == -1
    x = torch.zeros(1, 4, 4, 48, device="cuda")
    assert lib.mu_conv_fwd(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 1, 4, 4, 48, 32, 9, 48, 32, 0, None) == -2
    with pytest.raises(RuntimeError, match="MU_ERR"):
        _lib.call("mu_maxpool2_fwd", x.data_ptr(), x.data_ptr(), 1, 1, 4, 48, 0, None)      # no 2 x 2 window fits a 1-row image


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_persistent_tile_walk(dtype, monkeypatch):
    """The persistent 3x3 kernel (blocks walk several spatial tiles with the DMA pipeline running across tile borders) is
    normally selected only for big grids; force it on small ones (3 blocks per channel slice, uneven tile counts)."""
    from tests import _gpu_checks as G
    monkeypatch.setenv("MU_CONV_PERSIST_BLOCKS", "3")
    _assert_all(G.check_conv(dtype, cases=[(2, 32, 32, 64, 128, 3), (1, 64, 48, 128, 128, 3), (3, 16, 16, 256, 256, 3), (2, 40, 16, 128, 256, 3)]))


@pytest.fixture
def fp32x():
    import maskunet_amd
    maskunet_amd.set_float32_matmul_precision("high")
    yield
    maskunet_amd.set_float32_matmul_precision("highest")


def test_conv_fp32x(fp32x):
    """Every conv / weight-gradient shape of the kernel suite in the split-bf16 mode, at the fp32 gate (1e-3)."""
    from tests import _gpu_checks as G
    _assert_all(G.check_conv(torch.float32))


def test_attention_fp32x(fp32x):
    from tests import _gpu_checks as G
    _assert_all(G.check_attention(torch.float32))
    _assert_all(G.check_attention(torch.float32, cases=[(2, 8, 12, 48), (1, 8, 8, 200)]))
    _assert_all(G.check_attention_bwd_masked_rows(torch.float32))


def test_split_encode_is_a_per_chunk_hi_lo_rewrite():
    """mu_split_encode through the C ABI: every aligned 16-byte chunk of four fp32 values becomes [4 bf16 hi | 4 bf16 lo] with
    hi = bf16_rne(x), lo = bf16_rne(x - hi); hi + lo reproduces x to 2^-16 relative (2^-17 bound + bf16 subnormal floor), in place too."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4096 + 8, device="cuda", generator=g) * torch.logspace(-6, 4, 4104, device="cuda")
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 65504.0], device="cuda")
    e = torch.empty_like(x)
    _lib.call("mu_split_encode", x.data_ptr(), e.data_ptr(), x.numel(), _lib.stream())
    w = e.view(torch.int32).view(-1, 4)                                  # dwords: hi(x0,x1) hi(x2,x3) lo(x0,x1) lo(x2,x3)
    def halves(d):                                                       # a dword of two bf16 -> the two floats
        return torch.stack([(d << 16).view(torch.float32), (d & -65536).view(torch.float32)], -1)
    hi = torch.cat([halves(w[:, 0]), halves(w[:, 1])], -1).reshape(-1)
    lo = torch.cat([halves(w[:, 2]), halves(w[:, 3])], -1).reshape(-1)
    assert torch.equal(hi, x.bfloat16().float())                         # hi is the round-to-nearest-even bf16 of x
    assert torch.equal(lo, (x - hi).bfloat16().float())
    rel = ((hi.double() + lo.double() - x.double()).abs() / x.double().abs().clamp(min=1e-30)).max().item()
    assert rel <= 2.0 ** -16, rel
    y = x.clone()
    _lib.call("mu_split_encode", y.data_ptr(), y.data_ptr(), y.numel(), _lib.stream())     # in place
    assert torch.equal(y.view(torch.int32), e.view(torch.int32))
    with pytest.raises(RuntimeError, match="MU_ERR_ARG"):
        _lib.call("mu_split_encode", x.data_ptr(), e.data_ptr(), 6, _lib.stream())         # not whole chunks


def test_split_encode_h_is_a_per_group_fp16_hi_lo_rewrite():
    """mu_split_encode_h through the C ABI (the fp32x ATTENTION operand encoding): every aligned 32-byte group of eight fp32 values becomes
    [8 fp16 hi | 8 fp16 lo] with hi = fp16_rne(x), lo = fp16_rne(x - hi); hi + lo reproduces x to 2^-21 relative or 2^-25 absolute (fp16's
    subnormal floor), in place too."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(8192, device="cuda", generator=g) * torch.logspace(-4, 3, 8192, device="cuda")
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 1e-9, 2.0 ** -24], device="cuda")
    e = torch.empty_like(x)
    _lib.call("mu_split_encode_h", x.data_ptr(), e.data_ptr(), x.numel(), _lib.stream())
    h = e.view(torch.float16).view(-1, 16)
    hi, lo = h[:, :8].reshape(-1).float(), h[:, 8:].reshape(-1).float()
    assert torch.equal(hi, x.half().float())
    assert torch.equal(lo, (x - hi).half().float())
    err = (hi.double() + lo.double() - x.double()).abs()
    assert bool((err <= torch.maximum(x.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -25, device="cuda", dtype=torch.float64))).all())
    y = x.clone()
    _lib.call("mu_split_encode_h", y.data_ptr(), y.data_ptr(), y.numel(), _lib.stream())     # in place
    assert torch.equal(y.view(torch.int32), e.view(torch.int32))
    with pytest.raises(RuntimeError, match="MU_ERR_ARG"):
        _lib.call("mu_split_encode_h", x.data_ptr(), e.data_ptr(), 12, _lib.stream())         # not whole groups


def test_attention_overflow_redo_fp32x():
    """fp32x: P is an fp16 operand like in the fp16 kernels, so the forward runs the optimistic sweep too -- force its redo."""
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_overflow_redo(torch.float32, fp32x=True))


@pytest.mark.parametrize("gscale", [0.0, 1e-9, 1.0, 3e5])
def test_attention_fp32x_backward_is_invariant_to_the_gradient_magnitude(gscale):
    """fp32x backward: dY travels as fp16 pairs scaled by a power of two chosen from max|dY| on the device, dS as ONE fp16 operand scaled
    by 2^pshift -- the relative error of dqkv must not depend on how large the incoming gradient is (1e-9: the magnitude of a mean-reduced
    loss over 10^6 pixels; without the scale everything underflows fp16; 0.0: an all-zero gradient gives exact zeros, not 0 * inf)."""
    from tests import _gpu_checks as G
    _assert_all(G.check_attention_overflow_redo(torch.float32, fp32x=True, hot=False, gout_scale=gscale))


def test_colsum_of_a_chunk_encoded_tensor():
    """mu_colsum(MU_F32X): column sums (bias gradients) straight from the chunk-encoded operand form -- what the fp32x attention backward
    leaves in dqkv (MU_ATTN_DQKV_ENCODED) -- equal the sums of hi + lo of every element."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    for M, C in [(4096 + 37, 192), (1000, 64), (257, 768)]:
        x = torch.randn(M, C, device="cuda", generator=g) * 3.0
        e = torch.empty_like(x)
        _lib.call("mu_split_encode", x.data_ptr(), e.data_ptr(), x.numel(), _lib.stream())
        out = torch.empty(C, device="cuda")
        ws = _lib.workspace(_lib.load().mu_colsum_workspace_bytes(C), torch.device("cuda"))
        _lib.call("mu_colsum", e.data_ptr(), M, C, C, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.MU_F32X, _lib.stream())
        w = e.view(torch.int32).view(-1, 4)
        def halves(d):
            return torch.stack([(d << 16).view(torch.float32), (d & -65536).view(torch.float32)], -1)
        dec = (torch.cat([halves(w[:, 0]), halves(w[:, 1])], -1) + torch.cat([halves(w[:, 2]), halves(w[:, 3])], -1)).reshape(M, C)
        ref = dec.double().sum(0)
        assert float((out.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max() + dec.abs().double().sum(0).max())



# ------------------------------------------------------------------------------------------------
# round 6: the fp32x 3x3 convolutions -- fp16-pair operands, dy as ONE scaled fp16 operand, two-term backward -- through the C ABI
# ------------------------------------------------------------------------------------------------
def test_split_encode_h4_is_a_per_chunk_fp16_hi_lo_rewrite():
    """mu_split_encode_h4 (the fp32x 3x3-CONVOLUTION operand encoding): every aligned 16-byte chunk of four fp32 values becomes
    [4 fp16 hi | 4 fp16 lo] with hi = fp16_rne(x), lo = fp16_rne(x - hi); hi + lo reproduces x to 2^-21 relative or 2^-25 absolute, in place too."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(8192, device="cuda", generator=g) * torch.logspace(-5, 3, 8192, device="cuda")
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, -65504.0, 1e-9, 2.0 ** -24], device="cuda")
    e = torch.empty_like(x)
    _lib.call("mu_split_encode_h4", x.data_ptr(), e.data_ptr(), x.numel(), _lib.stream())
    h = e.view(torch.float16).view(-1, 8)
    hi, lo = h[:, :4].reshape(-1).float(), h[:, 4:].reshape(-1).float()
    assert torch.equal(hi, x.half().float())
    assert torch.equal(lo, (x - hi).half().float())
    err = (hi.double() + lo.double() - x.double()).abs()
    assert bool((err <= torch.maximum(x.double().abs() * 2.0 ** -21, torch.tensor(2.0 ** -25, device="cuda", dtype=torch.float64))).all())
    y = x.clone()
    _lib.call("mu_split_encode_h4", y.data_ptr(), y.data_ptr(), y.numel(), _lib.stream())
    assert torch.equal(y.view(torch.int32), e.view(torch.int32))
    with pytest.raises(RuntimeError, match="MU_ERR_ARG"):
        _lib.call("mu_split_encode_h4", x.data_ptr(), e.data_ptr(), 6, _lib.stream())


@pytest.mark.parametrize("mag", [1e-9, 3e-5, 1.0, 7e3])
def test_dy_encode_h_scale_is_an_exact_power_of_two(mag):
    """mu_dy_encode_h: dy_h = fp16(S dy) with S a power of two chosen on the device so that S max|dy| lands in [2^13, 2^14); the scale
    pair is {S, 1 / S}; zeros give S = 1; a NaN stays a NaN."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(11)
    dy = torch.randn(3 * 1024 * 40, device="cuda", generator=g) * mag
    out = torch.empty(dy.numel(), dtype=torch.float16, device="cuda")
    sc = torch.empty(2, device="cuda")
    ws = torch.empty(_lib.load().mu_dy_encode_h_workspace_bytes(), dtype=torch.uint8, device="cuda")
    _lib.call("mu_dy_encode_h", dy.data_ptr(), out.data_ptr(), sc.data_ptr(), dy.numel(), ws.data_ptr(), ws.numel(), _lib.stream())
    S, inv = float(sc[0]), float(sc[1])
    assert S > 0 and S * inv == 1.0 and math.log2(S) == int(math.log2(S))
    top = S * float(dy.abs().max())
    assert 2.0 ** 13 <= top < 2.0 ** 14, top
    assert torch.equal(out, (dy * S).half())
    z = torch.zeros(64, device="cuda")
    _lib.call("mu_dy_encode_h", z.data_ptr(), out.data_ptr(), sc.data_ptr(), 64, ws.data_ptr(), ws.numel(), _lib.stream())
    assert float(sc[0]) == 1.0 and float(out[:64].abs().max()) == 0.0
    dy[5] = float("nan")
    _lib.call("mu_dy_encode_h", dy.data_ptr(), out.data_ptr(), sc.data_ptr(), dy.numel(), ws.data_ptr(), ws.numel(), _lib.stream())
    assert bool(torch.isnan(out[5])) and float(sc[0]) == S          # (fmax ignores the NaN: the scale follows the finite values)
    with pytest.raises(RuntimeError, match="MU_ERR_ARG"):
        _lib.call("mu_dy_encode_h", dy.data_ptr(), dy.data_ptr(), sc.data_ptr(), dy.numel(), ws.data_ptr(), ws.numel(), _lib.stream())


# (B, H, W, Cin, Cout): the ping-pong kernel (Cin of the layer % 128, H % 16), the halo-tile kernel at 128 / 64 output channels, the
# generic register-staged kernel (odd sizes, 32-channel operands)
H2_SHAPES = [(2, 32, 32, 128, 128), (1, 16, 16, 256, 64), (2, 8, 16, 64, 128), (1, 24, 16, 64, 64), (2, 13, 9, 32, 64), (1, 7, 20, 96, 32),
             (1, 16, 16, 512, 512), (1, 8, 16, 128, 64), (1, 24, 32, 256, 128)]      # (the last two: the 128-channel halo-tile form, H % 16 != 0)


@pytest.mark.parametrize("shape", H2_SHAPES)
@pytest.mark.parametrize("gmag", [1e-7, 1.0])
def test_conv3x3_two_term_backward_through_the_abi(shape, gmag):
    """mu_conv_dgrad_h / mu_conv_wgrad_h against fp64 torch: dy as ONE scaled fp16 operand (mu_dy_encode_h), weights as HL rows
    (mu_prep_weight, MU_F32X, mode 1), the input chunk-encoded fp16 pairs (mu_split_encode_h4).  Gate 1e-3 of the gradient's maximum,
    independent of the gradient's magnitude (the scale is an exact power of two)."""
    from maskunet_amd import _lib
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / math.sqrt(9 * Cin))
    dy = torch.randn(B, Cout, H, W, generator=g) * gmag
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr, padding=1).backward(dy.double())
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.cuda().contiguous()
    st = _lib.stream()
    whl = torch.empty(9 * Cin * Cout, device="cuda")
    _lib.call("mu_prep_weight", wd.data_ptr(), whl.data_ptr(), _lib.MU_F32X, Cout, Cin, 9, Cin, Cout, 1, st)
    dyh = torch.empty(dyd.numel(), dtype=torch.float16, device="cuda")
    sc = torch.empty(2, device="cuda")
    ws0 = torch.empty(_lib.load().mu_dy_encode_h_workspace_bytes(), dtype=torch.uint8, device="cuda")
    _lib.call("mu_dy_encode_h", dyd.data_ptr(), dyh.data_ptr(), sc.data_ptr(), dyd.numel(), ws0.data_ptr(), ws0.numel(), st)
    dx = torch.full((B, H, W, Cin), float("nan"), device="cuda")
    _lib.call("mu_conv_dgrad_h", dyh.data_ptr(), whl.data_ptr(), sc.data_ptr(), dx.data_ptr(), B, H, W, Cout, Cin, Cout, Cin, st)
    xe = torch.empty_like(xd)
    _lib.call("mu_split_encode_h4", xd.data_ptr(), xe.data_ptr(), xd.numel(), st)
    nws = _lib.load().mu_conv_wgrad_h_workspace_bytes(B, H, W, Cin, Cout)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    gw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    _lib.call("mu_conv_wgrad_h", xe.data_ptr(), dyh.data_ptr(), sc.data_ptr(), gw.data_ptr(), B, H, W, Cin, Cout, Cin, Cout, Cin, Cout,
              ws.data_ptr(), ws.numel(), st)
    ex = float((dx.permute(0, 3, 1, 2).double().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
    ew = float((gw.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())
    assert ex <= 1e-3 and ew <= 1e-3, (shape, gmag, ex, ew)
    # the ONE-term weight gradient on the fp16 rounding of the input (mu_split_encode_h4x's second output; mu_conv_wgrad_h1): what the model
    # path runs -- dW sums over every pixel, so the 2^-12 roundings of x average out
    xe2, x16 = torch.empty_like(xd), torch.empty(xd.shape, dtype=torch.float16, device="cuda")
    _lib.call("mu_split_encode_h4x", xd.data_ptr(), xe2.data_ptr(), x16.data_ptr(), xd.numel(), st)
    assert torch.equal(xe2.view(torch.int32), xe.view(torch.int32)) and torch.equal(x16, xd.half())
    ws1 = torch.empty(_lib.load().mu_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 9), dtype=torch.uint8, device="cuda")
    gw1 = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    _lib.call("mu_conv_wgrad_h1", x16.data_ptr(), dyh.data_ptr(), sc.data_ptr(), gw1.data_ptr(), B, H, W, Cin, Cout, Cin, Cout, Cin, Cout,
              ws1.data_ptr(), ws1.numel(), st)
    ew1 = float((gw1.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())
    assert ew1 <= 1e-3, (shape, gmag, ew1)


@pytest.mark.parametrize("gmag", [1e-8, 1.0, 3e4])
def test_bn_backward_writes_dx_as_one_scaled_fp16_operand(gmag):
    """mu_bn_act_bwd_h against mu_bn_act_bwd on the same fp32 operands: dgamma / dbeta / dres bit-identical, dx_h / S = dx to fp16
    rounding (2^-11 of |S dx| per element), S a power of two with S max|dx| in [2^11, 2^14) (the device-side per-channel bound is tight
    to ~2 bits), whatever the gradient's magnitude; zero-padded channels (gamma = 1, rstd = 1 / sqrt(eps), dy = 0) do not move the scale."""
    from maskunet_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(21)
    for (M, C, cv, act, use_res) in [(2048, 64, 64, _lib.ACT_GELU, False), (1536, 160, 150, _lib.ACT_NONE, False), (4096, 32, 19, _lib.ACT_GELU, True)]:
        x = torch.randn(M, C, device="cuda", generator=g) * 2.0 + 0.5
        x[:, cv:] = 0.0
        dy = torch.randn(M, C, device="cuda", generator=g) * gmag
        dy[:, cv:] = 0.0
        res = (torch.randn(M, C, device="cuda", generator=g) if use_res else None)
        gamma = torch.rand(C, device="cuda", generator=g) + 0.5
        beta = torch.randn(C, device="cuda", generator=g) * 0.1
        gamma[cv:], beta[cv:] = 1.0, 0.0
        mean = x.mean(0)
        rstd = torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
        ws = torch.empty(lib.mu_bn_workspace_bytes(C), dtype=torch.uint8, device="cuda")
        outs = []
        for h_ in (False, True):
            dx = torch.full((M, C), float("nan"), device="cuda")
            dres = torch.empty_like(x) if use_res else None
            dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
            sc = torch.zeros(2, device="cuda")
            rp, drp = (res.data_ptr() if use_res else None), (dres.data_ptr() if use_res else None)
            if h_:
                _lib.call("mu_bn_act_bwd_h", x.data_ptr(), rp, dy.data_ptr(), dx.data_ptr(), drp, M, C, mean.data_ptr(), rstd.data_ptr(),
                          gamma.data_ptr(), beta.data_ptr(), act, 1, dg.data_ptr(), db.data_ptr(), sc.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
            else:
                _lib.call("mu_bn_act_bwd", x.data_ptr(), rp, dy.data_ptr(), dx.data_ptr(), drp, M, C, C, mean.data_ptr(), rstd.data_ptr(),
                          gamma.data_ptr(), beta.data_ptr(), act, 1, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), _lib.MU_F32, _lib.stream())
            outs.append((dx, dres, dg, db, sc))
        (dx0, dr0, dg0, db0, _), (dxh, dr1, dg1, db1, sc) = outs
        assert torch.equal(dg0, dg1) and torch.equal(db0, db1) and (not use_res or torch.equal(dr0, dr1))
        S = float(sc[0])
        assert S > 0 and S * float(sc[1]) == 1.0 and math.log2(S) == int(math.log2(S))
        top = S * float(dx0.abs().max())
        assert 2.0 ** 11 <= top < 2.0 ** 14, (M, C, gmag, top)
        got = dxh.view(torch.float16).view(-1)[: M * C].view(M, C).float() / S      # the halves sit at the start of the buffer
        assert float((got - dx0).abs().max()) <= 2.0 ** -11 * float(dx0.abs().max()) * 1.01


def test_bn_act_fwd_enc_writes_the_encoded_output_and_its_fp16_rounding():
    """mu_bn_act_fwd_enc: y in the 3x3 operand encoding (bit-identical to mu_bn_act_fwd with MU_F32X) plus y16 = fp16(y) as plain rows."""
    from maskunet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(31)
    M, C = 1536, 96
    x = torch.randn(M, C, device="cuda", generator=g)
    res = torch.randn(M, C, device="cuda", generator=g)
    mean, rstd = x.mean(0), torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    for act, r in ((_lib.ACT_GELU, None), (_lib.ACT_NONE, None), (_lib.ACT_GELU, res)):
        rp = r.data_ptr() if r is not None else None
        y_plain, y_enc, y_enc2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        y16 = torch.empty(M, C, dtype=torch.float16, device="cuda")
        _lib.call("mu_bn_act_fwd", x.data_ptr(), rp, y_plain.data_ptr(), M, C, C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, _lib.MU_F32, _lib.stream())
        _lib.call("mu_bn_act_fwd", x.data_ptr(), rp, y_enc.data_ptr(), M, C, C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, _lib.MU_F32X, _lib.stream())
        _lib.call("mu_bn_act_fwd_enc", x.data_ptr(), rp, y_enc2.data_ptr(), y16.data_ptr(), M, C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, _lib.stream())
        assert torch.equal(y_enc.view(torch.int32), y_enc2.view(torch.int32))
        assert torch.equal(y16, y_plain.half())
        h = y_enc.view(torch.float16).view(-1, 8)
        assert torch.equal(h[:, :4].reshape(M, C), y16)                     # the hi halves of the encoding ARE the fp16 rounding
