"""GPU parity tests, kernel level: every HIP op (called through maskunet_amd.ops -> ctypes -> the C ABI of
libmaskunet_hip.so) against stock torch / the CPU oracle on the same seeded inputs.  fp32 compute must meet the
north_star's 1e-3; fp16 compute (fp16 storage, fp32 accumulate) is held to 3e-2."""
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

