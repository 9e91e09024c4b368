"""GPU parity tests, module/model level: the reference-named modules against (a) the committed golden vectors
generated from the REAL reference (tests/golden/make_golden.py) and (b) the CPU oracle run live."""
import glob
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16]
_G = os.path.join(os.path.dirname(__file__), "golden")
MODULE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(_G, "*.npz"))
                      if not os.path.basename(p).startswith(("unet", "instloss", "miou", "resize")))


def _assert_all(results):
    bad = [(n, e, t) for n, e, t in results if not (e <= t)]
    assert not bad, "parity failures (name, err, tol): " + "; ".join(f"{n}: {e:.3e} > {t:.1e}" for n, e, t in bad)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", MODULE_CASES)
def test_golden_module(name, dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_golden_module(name, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["unet1_c150_b2_eval", "unet1_c150_b2_train", "unet3_c19_b2_train", "unet1_c133_b2_train"])
def test_unet_golden(name, dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_golden(name, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_unet_vs_oracle_with_dropout(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_vs_oracle(dtype, B=2, c_out=150))


def test_unet_fp32_gradients_within_3x_of_the_references_own_fp32_noise(capsys):
    """fp32 HIP path vs the fp64 oracle, gated per parameter at 3x the fp32 oracle's own error against fp64: the
    gate follows the measured noise floor of the reference arithmetic instead of a constant."""
    from tests import _gpu_checks as G
    res = G.check_unet_vs_oracle(torch.float32, B=2, c_out=150, with_dropout=False, seed=310, noise_floor=True)
    with capsys.disabled():
        for n, e, t in res:
            print(f"  {n}: {e:.3e} (gate {t:.1e})")
    _assert_all(res)


def test_unet_b8_train_mode_vs_oracle():
    """Training mode at B = 8 against the live CPU oracle (fp32): >= 8 images per XCD in the attention block order, the persistent
    conv kernel, multi-round split-K plans -- the batch-dependent dispatch under a VALUE check, not a property."""
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_vs_oracle(torch.float32, B=8, c_out=150, seed=320))


def test_kernels_are_bitwise_deterministic_at_bench_shapes():
    """tools/stress_determinism.py (race screen of the hand-placed vmcnt / barrier schedules) with 20 launches per kernel and shape;
    profiles/ holds the log of a 300-launch run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_determinism.py"), "20"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "all launches bit-identical" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_eval_mode_batchnorm_folded_into_conv_epilogue(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_eval_fused(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_one_launch_weight_prep_is_bit_identical(dtype):
    from tests import _gpu_checks as G
    _assert_all(G.check_prep_weights_multi(dtype))


def test_unet_3head_vs_oracle():
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_vs_oracle(torch.float32, B=2, c_out=19, three_head=True, seed=400))


def test_unet_wrong_size_raises():
    import maskunet_amd
    m = maskunet_amd.UNet(3, 5).cuda()
    with pytest.raises(RuntimeError, match="normalized_shape"):
        m(torch.zeros(1, 3, 64, 64, device="cuda"))


def test_state_dict_roundtrip_and_dp_prefix():
    import maskunet_amd
    m = maskunet_amd.UNet(3, 7).cuda()
    sd = {"module." + k: v for k, v in m.state_dict().items()}          # what a DataParallel checkpoint looks like
    m2 = maskunet_amd.UNet(3, 7).cuda()
    m2.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})   # ade_panoptic.py:434
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_unet_other_resolution_vs_oracle():
    """The path is not tied to 128x128 (configs[4] runs 256x256): a 64x64 model (hw=64) against the oracle, fp32."""
    import numpy as np
    import torch.nn.functional as F
    import maskunet_amd
    from oracle import maskunet_oracle as O
    hw, c_out, B, seed = 64, 21, 2, 700
    shapes = O.unet_state_shapes(3, c_out, False, hw=hw)
    params = O.make_params(shapes, seed)
    model = maskunet_amd.UNet(3, c_out, hw=hw)
    model.load_state_dict(params)
    model.cuda().train()
    model.dropout.p = 0.0
    keeps = O.make_keeps(seed + 1, B, hw)
    model.set_keep_masks(keeps)
    x, labels = O.make_inputs(seed + 2, B, c_out, hw)
    p = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    ref = O.unet_forward(p, x, keeps, training=True)
    O.pixel_cross_entropy(ref, labels).backward()
    out = model(x.cuda())
    F.cross_entropy(out, labels.cuda()).backward()
    assert float((out.cpu() - ref).abs().max()) <= 1e-3
    worst = 0.0
    gmax = max(float(v.grad.abs().max()) for v in p.values() if v.requires_grad and v.grad is not None)
    for k, v in model.named_parameters():
        r = p[k].grad
        if r is None or float(r.abs().max()) < 1e-3 * gmax:
            continue
        g = v.grad.float().cpu()
        worst = max(worst, 1.0 - float((g.double() * r.double()).sum() / (g.double().norm() * r.double().norm())))
    assert worst <= 1e-3, worst      # 8x8 bottleneck with B=2: 128 samples per BN channel amplify fp32 reassociation noise


def test_training_step_is_bitwise_reproducible():
    """No float atomics, fixed reduction orders, and -- for the LDS-DMA ring kernels whose ordering is hand-counted -- no
    races: three fp16 training steps of the full-resolution model from identical state give bit-identical gradients."""
    import torch.nn.functional as F
    import maskunet_amd
    import bench
    torch.manual_seed(3)
    B = 12
    model = maskunet_amd.UNet(3, 150).cuda()
    model.set_compute_dtype(torch.float16).train()
    x, labels, keeps = bench.synth(B, 150, 128, 9, torch.device("cuda"))
    model.set_keep_masks(keeps)
    g = torch.Generator().manual_seed(5)
    model.dropout_masks = [torch.randint(0, 2, (B, 32, 32, 128), generator=g, dtype=torch.uint8).cuda(),      # NHWC keep-masks of the
                           torch.randint(0, 2, (B, 64, 64, 64), generator=g, dtype=torch.uint8).cuda()]       # two dropout sites
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(3):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        out = model(x)
        loss = F.cross_entropy(out, labels)          # (torch's own loss reduction uses atomics: its VALUE is not compared)
        (loss * 1024.0).backward()
        runs.append((out.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    for out, grads in runs[1:]:
        assert torch.equal(out, runs[0][0]), "forward output differs between identical runs"
        for n, gref in runs[0][1].items():
            assert torch.equal(grads[n], gref), f"gradient of {n} differs between identical runs"


@pytest.mark.parametrize("precision", ["highest", "high"])
def test_training_trajectory_matches_oracle(precision):
    """Four optimisation steps (fp32 compute -- exact-fp32 MFMA and the fp32x split-bf16 mode --, FusedAdamW, maskunet_amd.CrossEntropyLoss)
    follow the CPU oracle trained with torch.optim.AdamW on the same batch: the loss after every step agrees, and it moves by far more
    than the tolerance."""
    import maskunet_amd
    maskunet_amd.set_float32_matmul_precision(precision)
    try:
        _trajectory()
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")


def _trajectory():
    import maskunet_amd
    from oracle import maskunet_oracle as O
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(19, False, 610, torch.float32, True, 2)
    p = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    lr, wd = 1e-3, 1e-2
    ref_opt = torch.optim.AdamW([v for v in p.values() if v.requires_grad], lr=lr, weight_decay=wd)
    ref_losses = []
    for _ in range(4):
        loss = O.pixel_cross_entropy(O.unet_forward(p, x, keeps, training=True), labels)
        ref_opt.zero_grad(set_to_none=True)
        loss.backward()
        ref_opt.step()
        ref_losses.append(float(loss.detach()))
    opt = maskunet_amd.FusedAdamW(model.parameters(), lr=lr, weight_decay=wd)
    crit = maskunet_amd.CrossEntropyLoss()
    xd, yd = x.cuda(), labels.cuda()
    got = []
    for _ in range(4):
        loss = crit(model(xd), yd)
        model.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    assert abs(ref_losses[-1] - ref_losses[0]) > 2e-2, ref_losses           # the steps do change the loss
    for a, b in zip(got, ref_losses):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (got, ref_losses)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_backward_twice_over_a_retained_graph(dtype):
    """the projection's data-gradient layout stays on the autograd node, so a second backward over the same graph
    (retain_graph=True, or two losses backpropagated separately) gives the same gradients instead of MU_ERR_ARG."""
    import maskunet_amd
    torch.manual_seed(5)
    m = maskunet_amd.Mask2FormerAttention(64, 64).cuda().set_compute_dtype(dtype)
    x = torch.randn(2, 64, 16, 16, device="cuda", requires_grad=True)
    m.set_keep_mask(torch.randint(0, 2, (2, 256), dtype=torch.uint8))
    y = m(x)
    gy = torch.randn_like(y)
    y.backward(gy, retain_graph=True)
    g1 = [x.grad.clone()] + [p.grad.clone() for p in m.parameters()]
    x.grad = None
    m.zero_grad(set_to_none=True)
    y.backward(gy)
    g2 = [x.grad] + [p.grad for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


def test_partial_backward_does_not_leak_a_linked_gradient_into_the_next_pass():
    """torch.autograd.grad towards LATE parameters over a retained graph runs the GradLink fillers (residual ConvBlock
    / UpSample backward) and prunes the takers (max-pool backward); the gradient parked in the link belongs to THAT pass and must
    not be added to the full backward that follows."""
    import torch.nn.functional as F
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(150, False, 330, torch.float32, True, 2)
    xd, ld = x.to("cuda"), labels.to("cuda")
    F.cross_entropy(model(xd), ld).backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(xd), ld)
    late = [model.downsample1.maxpool_conv[1].conv_block[0].weight, model.upsample3.conv[0].conv_block[0].weight]
    part = torch.autograd.grad(loss, late, retain_graph=True)          # fills links whose takers are pruned from this pass
    assert torch.equal(part[0], ref["downsample1.maxpool_conv.1.conv_block.0.weight"])
    loss.backward()
    for n, p in model.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), n


# ------------------------------------------------------------------------------------------------
# fp32x: fp32 storage, matrix products as three bf16 MFMAs on (hi, lo) operand splits
# (maskunet_amd.set_float32_matmul_precision("high")) -- held to the SAME gates as the exact-fp32 path: north_star's 1e-3 on outputs,
# the fp32 gradient gates of the whole-model goldens, per-parameter 1 - cos <= 1e-4 against the live oracle.
# ------------------------------------------------------------------------------------------------
@pytest.fixture
def fp32x():
    import maskunet_amd
    maskunet_amd.set_float32_matmul_precision("high")
    yield
    maskunet_amd.set_float32_matmul_precision("highest")


@pytest.mark.parametrize("name", ["unet1_c150_b2_eval", "unet1_c150_b2_train", "unet3_c19_b2_train", "unet1_c133_b2_train"])
def test_unet_golden_fp32x(name, fp32x, capsys):
    from tests import _gpu_checks as G
    res = G.check_unet_golden(name, torch.float32)
    with capsys.disabled():
        print("  fp32x", name, "; ".join(f"{n.split(name)[-1].strip()}: {e:.2e}" for n, e, t in res[:4]))
    _assert_all(res)


@pytest.mark.parametrize("name", MODULE_CASES)
def test_golden_module_fp32x(name, fp32x):
    from tests import _gpu_checks as G
    _assert_all(G.check_golden_module(name, torch.float32))


def test_unet_vs_oracle_fp32x(fp32x):
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_vs_oracle(torch.float32, B=2, c_out=150))


def test_fp32x_is_really_the_split_path_and_fp32_is_untouched():
    """The switch changes the matrix products (different bits from the exact-fp32 MFMA, error ~1e-5 against it) and switching back
    restores the exact path bit for bit."""
    import maskunet_amd
    from maskunet_amd import ops
    torch.manual_seed(0)
    x = torch.randn(2, 16, 16, 64, device="cuda")
    w = torch.randn(128, 64, 3, 3, device="cuda") * 0.05
    exact = ops.conv(x, w)
    maskunet_amd.set_float32_matmul_precision("high")
    try:
        assert maskunet_amd.get_float32_matmul_precision() == "high"
        split = ops.conv(x, w)
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")
    again = ops.conv(x, w)
    assert torch.equal(exact, again)
    err = float((split - exact).abs().max()) / float(exact.abs().max())
    assert 0 < err < 1e-4, err


def test_eval_fused_and_one_launch_weight_prep_fp32x(fp32x):
    """The inference epilogue (eval-mode BatchNorm folded into the conv) and the one-launch weight layouts in the split-bf16 mode: the
    prepared weights are chunk-encoded behind both, keyed apart from the exact-fp32 layouts."""
    from tests import _gpu_checks as G
    _assert_all(G.check_eval_fused(torch.float32))
    _assert_all(G.check_prep_weights_multi(torch.float32))


def test_precision_modes_do_not_share_cached_weight_layouts():
    """An eval forward under torch.no_grad() caches the prepared (compute-layout) weights on the parameters; switching the fp32 matmul
    precision must not hand the exact-fp32 kernels chunk-encoded weights or vice versa."""
    import maskunet_amd
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(150, False, 340, torch.float32, False, 2)
    xd = x.cuda()
    with torch.no_grad():
        a = model(xd)
        maskunet_amd.set_float32_matmul_precision("high")
        try:
            b = model(xd)
        finally:
            maskunet_amd.set_float32_matmul_precision("highest")
        c = model(xd)
    assert torch.equal(a, c)
    err = float((a - b).abs().max()) / max(1.0, float(a.abs().max()))
    assert 0 < err < 1e-3, err


@pytest.mark.parametrize("cin,cout,residual", [(16, 3, False), (3, 3, True)])
def test_convblock_with_a_three_channel_second_conv_fp32x(cin, cout, residual, fp32x):
    """a ConvBlock whose SECOND conv has <= 3 input channels (a plain-FMA layer like the stem) must run in the fp32x training
    mode too -- its mid activation / gradient stay plain instead of chunk-encoded."""
    import maskunet_amd
    from oracle import maskunet_oracle as O
    shapes = O._conv_block_shapes("m", cin, cout)
    p = O.make_params(shapes, 77)
    m = maskunet_amd.ConvBlock(cin, cout, residual=residual)
    m.load_state_dict({k[2:]: v for k, v in p.items()})
    m.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, cin, 16, 16, generator=g)
    go = torch.randn(2, cout, 16, 16, generator=g)
    for v in p.values():
        if v.dtype.is_floating_point and v.dim() > 0:
            v.requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O.conv_block(xr, {k: v for k, v in p.items()}, "m", residual, True, {})
    yr.backward(go)
    xd = x.cuda().requires_grad_(True)
    y = m(xd)
    y.backward(go.cuda())
    assert float((y.cpu() - yr).abs().max()) <= 1e-3 * max(1.0, float(yr.abs().max()))
    assert float((xd.grad.cpu() - xr.grad).abs().max()) <= 1e-3 * float(xr.grad.abs().max())
    gw = m.conv_block[3].weight.grad.cpu()
    gr = p["m.conv_block.3.weight"].grad
    assert float((gw - gr).abs().max()) <= 1e-3 * float(gr.abs().max())


def test_precision_switch_between_forward_and_backward_is_refused():
    """the fp32 matmul precision is process-wide and read in forward and backward; a backward under the other mode would mix
    encoded saved operands with plain-fp32 kernels -- it raises instead."""
    import maskunet_amd
    from maskunet_amd import ops
    x = torch.randn(1, 8, 8, 64, device="cuda", requires_grad=True)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    y = ops.conv(x, w)
    maskunet_amd.set_float32_matmul_precision("high")
    try:
        with pytest.raises(RuntimeError, match="changed between a forward pass and its backward"):
            y.sum().backward()
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")

