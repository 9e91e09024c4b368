"""GPU tests for the "next" rows: fused pixel cross-entropy, on-device mean IoU, fused AdamW."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3)])
@pytest.mark.parametrize("C,ignore", [(150, -100), (19, 255), (133, 255)])
def test_pixel_cross_entropy_matches_torch(dtype, tol, C, ignore):
    import maskunet_amd
    from maskunet_amd import ops
    g = np.random.default_rng(C)
    B, H, W = 2, 12, 10
    logits = torch.from_numpy(g.standard_normal((B, C, H, W)).astype(np.float32) * 3)
    labels = torch.from_numpy(g.integers(0, C, (B, H, W)))
    if ignore == 255:
        labels[torch.from_numpy(g.random((B, H, W)) < 0.2)] = 255
    lr = logits.to(dtype).float().clone().requires_grad_(True)
    ref = F.cross_entropy(lr, labels, ignore_index=ignore)
    ref.backward()
    ld = logits.cuda().requires_grad_(True)
    nhwc = ops.to_nhwc(ld, dtype)
    loss = maskunet_amd.pixel_cross_entropy_nhwc(nhwc, labels.cuda(), C, ignore)
    (loss * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item()))
    err = (ld.grad.cpu() / 3.0 - lr.grad).abs().max().item() / lr.grad.abs().max().item()
    assert err <= 20 * tol, err


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3)])
@pytest.mark.parametrize("shape,ignore", [((3, 150, 16, 12), -100), ((2, 19, 9, 7), 255), ((4, 133, 32, 32), 255), ((1, 1, 4, 4), -100)])
def test_cross_entropy_nchw_matches_torch(dtype, tol, shape, ignore):
    """maskunet_amd.CrossEntropyLoss on the module's NCHW output == nn.CrossEntropyLoss (ade_semantic.py:377,399)."""
    import maskunet_amd
    g = torch.Generator().manual_seed(sum(shape))
    B, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 3).to(dtype)
    labels = torch.randint(0, C, (B, H, W), generator=g)
    if ignore == 255:
        labels[torch.rand(B, H, W, generator=g) < 0.2] = 255
    from oracle import maskunet_oracle as O
    xr = x.double().requires_grad_(True)
    ref = O.pixel_cross_entropy(xr, labels, ignore_index=ignore)       # the oracle's restatement of nn.CrossEntropyLoss, in fp64
    ref.backward()
    xg = x.cuda().requires_grad_(True)
    crit = maskunet_amd.CrossEntropyLoss(ignore_index=ignore)
    loss = crit(xg, labels.cuda())
    loss.backward()
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item()))
    gtol = 2e-7 if dtype == torch.float32 else 2e-3 * float(xr.grad.abs().max())
    assert float((xg.grad.double().cpu() - xr.grad).abs().max()) <= gtol
    # grad_scale multiplies the backward only
    xs = x.cuda().requires_grad_(True)
    l2 = maskunet_amd.cross_entropy(xs, labels.cuda(), ignore, grad_scale=64.0)
    l2.backward()
    assert l2.item() == loss.item()
    assert torch.allclose(xs.grad.float() / 64.0, xg.grad.float(), rtol=2e-3, atol=2e-7)


def test_cross_entropy_on_module_output_uses_nhwc_source():
    """criterion(model(x), labels): the loss on the untouched module output (read through its NHWC source) equals the loss on a copy
    of that output (read as NCHW), value and parameter gradients; a modified output falls back to the NCHW kernels."""
    import maskunet_amd
    from maskunet_amd import losses
    from tests import _gpu_checks as G
    if not losses.NHWC_SOURCE:
        pytest.skip("MU_CE_NHWC_SOURCE=0")
    model, params, keeps, x, labels = G.build_unet(19, False, 501, torch.float16, True, 2)
    x, labels = x.cuda(), labels.cuda()
    labels[:, :3] = 255
    crit = maskunet_amd.CrossEntropyLoss(ignore_index=255)

    def run(copy):
        for bn in model.modules():
            if isinstance(bn, torch.nn.BatchNorm2d):
                bn.reset_running_stats()
        model.zero_grad(set_to_none=True)
        out = model(x)
        assert losses._nhwc_source(out) is not None
        if copy:
            out = out * 1.0                                   # same values, no NHWC source
            assert losses._nhwc_source(out) is None
        loss = crit(out, labels)
        (loss * 256.0).backward()
        return loss.item(), {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l1))
    assert g0.keys() == g1.keys()
    worst = 0.0
    for k in g0:
        if float(g1[k].abs().max()) > 1e-3 * 256:             # parameters with analytically zero gradients carry rounding noise only
            c = 1.0 - float((g0[k].double() * g1[k].double()).sum() / (g0[k].double().norm() * g1[k].double().norm()))
            worst = max(worst, c)
    assert worst <= 1e-3, worst
    out = model(x)
    out.add_(0.0)                                             # in-place use invalidates the remembered source
    assert losses._nhwc_source(out) is None


def test_graphed_step_equals_eager_step():
    """maskunet_amd.GraphedStep (forward + criterion + backward as one HIP graph) reproduces the eager step bit for bit when dropout is
    off, and draws a new dropout mask per replay when it is on (device step counter mixed into the captured seed)."""
    import maskunet_amd
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(19, False, 502, torch.float16, True, 2)
    x, labels = x.cuda(), labels.cuda()
    crit = maskunet_amd.CrossEntropyLoss()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.dropout.p = 0.0
    loss_e = crit(model(x), labels)
    (loss_e * 128.0).backward()
    ge = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    rm_e = model.downsample1.maxpool_conv[3].running_mean.clone()
    model.zero_grad(set_to_none=True)
    model.load_state_dict(state)
    step = maskunet_amd.GraphedStep(model, crit, x, labels, loss_scale=128.0, warmup=1)
    model.load_state_dict(state)                              # warm-up and capture advanced the BatchNorm buffers
    loss_g = step(x, labels)
    assert loss_g.item() == loss_e.item()
    for k, v in model.named_parameters():
        if k in ge:
            assert torch.equal(v.grad, ge[k]), k
    assert torch.equal(model.downsample1.maxpool_conv[3].running_mean, rm_e)
    # other inputs through the same graph
    x2 = torch.rand_like(x)
    l2 = step(x2, labels).item()
    # optimizer.zero_grad() (set_to_none=True is torch's default) between replays must not lose the gradients, and a returned
    # loss must not alias the next replay's
    model.zero_grad(set_to_none=True)
    l3 = step(x, labels)
    assert all(v.grad is not None and torch.equal(v.grad, ge[k]) for k, v in model.named_parameters() if k in ge)
    l4 = step(x2, labels)
    assert l3.item() == loss_e.item() and l4.item() == l2 and l3.data_ptr() != l4.data_ptr()
    model.zero_grad(set_to_none=True)
    # dropout on: a fresh capture; two replays on the same inputs see different masks
    model.dropout.p = 0.3
    step = maskunet_amd.GraphedStep(model, crit, x, labels, loss_scale=128.0, warmup=1)
    a = step(x, labels).item()
    b = step(x, labels).item()
    assert a != b and a == a and b == b and l2 == l2


def test_cross_entropy_nchw_rejects_bad_labels():
    import maskunet_amd
    x = torch.randn(2, 5, 4, 4, device="cuda")
    with pytest.raises(RuntimeError):
        maskunet_amd.cross_entropy(x, torch.zeros(2, 4, 3, dtype=torch.int64, device="cuda"))
    with pytest.raises(RuntimeError):
        maskunet_amd.cross_entropy(x, torch.zeros(2, 4, 4, dtype=torch.int32, device="cuda"))


def test_pixel_cross_entropy_grad_scale_and_padding():
    import maskunet_amd
    C, Cp = 19, 32
    x = torch.randn(2, 4, 4, Cp, device="cuda", requires_grad=True)
    lab = torch.randint(0, C, (2, 4, 4), device="cuda")
    l1 = maskunet_amd.pixel_cross_entropy_nhwc(x, lab, C)
    l1.backward()
    g1 = x.grad.clone()
    x.grad = None
    l2 = maskunet_amd.pixel_cross_entropy_nhwc(x, lab, C, grad_scale=128.0)
    l2.backward()
    assert torch.allclose(l1, l2) and torch.allclose(x.grad, g1 * 128.0, rtol=1e-5, atol=1e-7)
    assert float(x.grad[..., C:].abs().max()) == 0.0        # padded channels get exact zeros


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_mean_iou_matches_reference_definition(layout):
    import maskunet_amd
    from maskunet_amd import ops
    from oracle import maskunet_oracle as O
    g = np.random.default_rng(3)
    B, C, H, W = 2, 21, 16, 16
    y = torch.from_numpy(g.standard_normal((B, C, H, W)).astype(np.float32))
    t = torch.from_numpy(g.integers(0, C - 3, (B, H, W)))           # some classes absent from the labels
    y[:, C - 1] = -10.0                                              # and one never predicted either -> union == 0, skipped
    ref = O.mean_iou(y, t, C)
    pred = y.cuda() if layout == "nchw" else ops.to_nhwc(y.cuda(), torch.float32)
    got = maskunet_amd.mean_iou(pred, t.cuda(), C)
    assert abs(got.item() - ref.item()) <= 1e-6


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("name", ["miou_dense", "miou_absent_classes", "miou_ties_150"])
def test_mean_iou_matches_reference_goldens(name, layout):
    """mu_mean_iou against fixtures produced by the reference's own mean_iou (tests/golden/make_golden_losses.py)."""
    import os
    import maskunet_amd
    from maskunet_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    y, t, C = torch.from_numpy(g["y"]).cuda(), torch.from_numpy(g["t"]).cuda(), int(g["num_classes"])
    pred = y if layout == "nchw" else ops.to_nhwc(y, torch.float32)
    got = maskunet_amd.mean_iou(pred, t, C)
    assert abs(got.item() - float(g["miou"])) <= 1e-6


def test_fused_adamw_matches_torch():
    import maskunet_amd
    torch.manual_seed(0)
    shapes = [(64, 3, 3, 3), (64,), (150, 64, 1, 1), (5000,), (7,)]
    ref_p = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=5e-3, weight_decay=1e-1)
    our = maskunet_amd.FusedAdamW(our_p, lr=5e-3, weight_decay=1e-1)
    scale = 1024.0
    for it in range(4):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i == 4 and it % 2 == 0:            # a parameter that sometimes has no gradient
                a.grad = b.grad = None
                continue
            gr = torch.randn_like(a)
            a.grad = gr.clone()
            b.grad = gr * scale                    # loss-scaled gradients, un-scaled inside the fused step
        ref.step()
        our.step(grad_scale=scale)
    for a, b in zip(ref_p, our_p):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), (a - b).abs().max()


def test_training_step_with_fused_loss_and_optimizer():
    """One fp16 training step through the fused-loss path equals the public-API path (module output + torch CE)."""
    import maskunet_amd
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(19, False, 500, torch.float16, True, 2)
    x, labels = x.cuda(), labels.cuda()
    out = model(x)
    l_ref = F.cross_entropy(out, labels)
    (l_ref * 512.0).backward()
    gref = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    model.zero_grad(set_to_none=True)
    for bn in model.modules():
        if isinstance(bn, torch.nn.BatchNorm2d):
            bn.reset_running_stats()
    loss = maskunet_amd.pixel_cross_entropy_nhwc(model.logits_nhwc(x), labels, 19, grad_scale=512.0)
    loss.backward()
    assert abs(loss.item() - l_ref.item()) <= 2e-3
    worst = 0.0
    for k, v in model.named_parameters():
        if k in gref and float(gref[k].abs().max()) > 1e-3 * 512:
            c = 1.0 - float((v.grad.double() * gref[k].double()).sum() / (v.grad.double().norm() * gref[k].double().norm()))
            worst = max(worst, c)
    assert worst <= 2e-2, worst
    opt = maskunet_amd.FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)
    before = model.final_layer[0].weight.detach().clone()
    opt.step(grad_scale=512.0)
    assert not torch.equal(before, model.final_layer[0].weight)


def test_u8_input_pipeline_matches_totensor():
    """8-f4: forward_u8(uint8 HWC) == forward(ToTensor(image)) (ade_semantic.py:72-76,85)."""
    import maskunet_amd
    from maskunet_amd import ops
    img = torch.randint(0, 256, (2, 128, 128, 3), dtype=torch.uint8, device="cuda")
    y = ops.u8_hwc_to_nhwc(img, torch.float32)
    ref = (img.cpu().float() / 255.0).cuda()         # divided on the HOST like ToTensor (torch's GPU division by a scalar multiplies by the reciprocal)
    assert torch.equal(y[..., :3], ref) and float(y[..., 3:].abs().max()) == 0.0
    m = maskunet_amd.UNet(3, 5).cuda().eval()
    m.set_keep_masks([torch.ones(2, n, dtype=torch.uint8) for n in (4096, 1024, 256, 1024, 4096, 16384)])
    a = m.forward_u8(img)
    b = m(ref.permute(0, 3, 1, 2).contiguous())
    assert torch.allclose(a, b, atol=1e-5)


def test_resize_pipeline_matches_cv2_restatement():
    """8-f4, resize half: cv2.resize INTER_LINEAR (image, + BGR2RGB + ToTensor) and INTER_NEAREST (label map) on the device, bit for bit
    against the numpy restatement of OpenCV 4.10's 8-bit algorithm and its committed fixture (ade_semantic.py:65,72-73,78,85)."""
    from tests import _gpu_checks as G
    bad = [(n, e, t) for n, e, t in G.check_resize_u8() if not (e <= t)]
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["instloss_ade_small", "instloss_ade_sparse", "instloss_city_ignore", "instloss_none"])
def test_instance_contrastive_loss_matches_reference_goldens(name):
    """Device InstanceContrastiveLoss vs the fixtures generated from the reference class (tests/golden/make_golden_losses.py)."""
    import os
    import numpy as np
    import maskunet_amd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    ign = int(g["ignore"])
    feat = torch.from_numpy(g["feat"]).cuda().requires_grad_(True)
    crit = maskunet_amd.InstanceContrastiveLoss(margin=1.0, ignore_index=None if ign < 0 else ign, id_cap=32768, max_instances=64)
    loss = crit(feat, torch.from_numpy(g["mask"]).cuda(), torch.from_numpy(g["u"]).cuda())
    (loss * 3.0).backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5
    assert float((feat.grad.cpu() / 3.0 - torch.from_numpy(g["dfeat"])).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_instance_contrastive_loss_matches_oracle_at_model_shape():
    """B=4, 19 classes, 128x128, Cityscapes-style ids with an ignore label: device loss/gradient vs the oracle restatement."""
    import numpy as np
    import maskunet_amd
    from oracle import maskunet_oracle as O
    rng = np.random.default_rng(77)
    B, C, H, W = 4, 19, 128, 128
    feat = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    mask = torch.zeros((B, H, W), dtype=torch.int64)
    ids = [24000, 24001, 26000, 26001, 26002, 33000]
    for i, v in enumerate(ids):                              # rectangles of instance ids, some overlapping, plus an ignore band
        b = i % B
        mask[b, 10 * i:10 * i + 30, 5 * i:5 * i + 40] = v
    mask[:, 120:, :] = 255
    mask[1, 3, 3] = 31000                                    # a one-pixel instance is skipped
    u = torch.from_numpy(rng.random(64).astype(np.float32))
    fr = feat.clone().requires_grad_(True)
    lr = O.instance_contrastive_loss(fr, mask, u, 0.5, 255)
    lr.backward()
    fd = feat.cuda().requires_grad_(True)
    ld = maskunet_amd.InstanceContrastiveLoss(margin=0.5, ignore_index=255, max_instances=64)(fd, mask.cuda(), u.cuda())
    ld.backward()
    assert abs(float(ld) - float(lr)) <= 1e-5 * max(1.0, abs(float(lr)))
    assert float((fd.grad.cpu() - fr.grad).abs().max()) <= 1e-5


def test_weight_layout_cache_only_without_grad_and_invalidated_by_updates():
    """Forwards under torch.no_grad() re-use the prepared weight layouts (validation loop); training forwards never do; an optimizer
    step (FusedAdamW writes through raw pointers) or load_state_dict invalidates the entries."""
    import maskunet_amd
    from maskunet_amd import _lib, ops
    torch.manual_seed(0)
    m = maskunet_amd.ConvBlock(32, 64).cuda().eval()
    x = torch.randn(2, 32, 16, 16, device="cuda")
    calls = []
    _lib.PROBE = {"pred": lambda name, a: (calls.append(name) or False) if name == "mu_prep_weight" else False, "events": []}
    try:
        with torch.no_grad():
            y0 = m(x)
            n0 = len(calls)
            y1 = m(x)
            assert len(calls) == n0 and n0 == 2 and torch.equal(y0, y1)          # second forward: no conversions
        m.train()
        xr = x.clone().requires_grad_(True)
        m(xr).sum().backward()
        m(xr).sum().backward()
        assert len(calls) == n0 + 4                                               # training forwards always convert
        opt = maskunet_amd.FusedAdamW(m.parameters(), lr=0.1)
        opt.step()
        m.eval()
        with torch.no_grad():
            y2 = m(x)
            assert len(calls) == n0 + 6 and not torch.equal(y2, y0)              # updated weights: converted again, new result
            sd = {k: v.clone() for k, v in m.state_dict().items()}
            for v in sd.values():
                if v.dtype.is_floating_point:
                    v.mul_(0.5)
            m.load_state_dict(sd)
            y3 = m(x)
            assert len(calls) == n0 + 8 and not torch.equal(y3, y2)
            ref = maskunet_amd.ConvBlock(32, 64).cuda().eval()
            ref.load_state_dict(sd)
            ops_cache, ops.PREP_CACHE = ops.PREP_CACHE, False
            try:
                assert torch.equal(ref(x), y3)
            finally:
                ops.PREP_CACHE = ops_cache
    finally:
        _lib.PROBE = None


def test_fused_adamw_skips_an_overflowed_step_entirely():
    """one inf / NaN gradient must not reach the fp32 master weights.  A step whose gradients hold a non-finite
    value changes NOTHING (parameters, moments, effective step count) -- decided on the device, no host sync -- and the trajectory
    afterwards equals torch.optim.AdamW's over the finite steps only."""
    import maskunet_amd
    torch.manual_seed(1)
    shapes = [(64, 3, 3, 3), (150, 64, 1, 1), (5000,)]
    ref_p = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    our_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, lr=5e-3, weight_decay=1e-1)
    our = maskunet_amd.FusedAdamW(our_p, lr=5e-3, weight_decay=1e-1)
    scale = 1024.0
    for it in range(5):
        grads = [torch.randn_like(a) for a in ref_p]
        overflow = it in (1, 3)
        for a, b, g in zip(ref_p, our_p, grads):
            a.grad = g.clone()
            b.grad = g * scale
        if overflow:
            our_p[1].grad.view(-1)[77] = float("inf") if it == 1 else float("nan")
            before = [p.detach().clone() for p in our_p]
            moments = [our.state[p]["exp_avg"].clone() for p in our_p]
        else:
            ref.step()
        our.step(grad_scale=scale)
        assert our.last_step_skipped() == overflow
        if overflow:
            assert all(torch.equal(a, b) for a, b in zip(before, our_p))
            assert all(torch.equal(m, our.state[p]["exp_avg"]) for m, p in zip(moments, our_p))
    for a, b in zip(ref_p, our_p):
        assert torch.isfinite(b).all() and torch.allclose(a, b, rtol=2e-5, atol=2e-6), (a - b).abs().max()
    assert our.effective_steps(our_p[0]) == 3
    assert our.state_dict()["state"][0]["step"] == 3          # folded into the checkpointed counters


def test_fused_adamw_under_torch_gradscaler():
    """torch.cuda.amp.GradScaler.step(FusedAdamW): the scaler's device-side scale / found_inf drive the kernel (no .item() sync in
    GradScaler for optimisers with _step_supports_amp_scaling); an overflowed step is skipped and the scale backs off."""
    import maskunet_amd
    torch.manual_seed(2)
    w = torch.randn(4096, device="cuda").requires_grad_(True)
    w_ref = w.detach().clone().requires_grad_(True)
    our = maskunet_amd.FusedAdamW([w], lr=1e-2, weight_decay=0.0)
    ref = torch.optim.AdamW([w_ref], lr=1e-2, weight_decay=0.0)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=1000)
    x = torch.randn(4096, device="cuda")
    for it in range(4):
        our.zero_grad(set_to_none=True)
        loss = (w * x).square().mean()
        scaler.scale(loss).backward()
        if it == 2:
            w.grad[5] = float("inf")                     # the overflow a too-large scale produces
        else:
            ref.zero_grad(set_to_none=True)
            (w_ref * x).square().mean().backward()
            ref.step()
        s0 = scaler.get_scale()
        scaler.step(our)
        scaler.update()
        assert (scaler.get_scale() < s0) == (it == 2)
    assert torch.isfinite(w).all() and torch.allclose(w, w_ref, rtol=2e-5, atol=2e-6)


def test_fused_adamw_skip_counters_survive_rollback_reallocation_and_an_unchecked_step():
    """the device-side skip counters and the host `step` must stay consistent through (1) load_state_dict() in a
    process that skipped steps since its last state_dict(), (2) a re-allocated parameter list (the plan is keyed on data_ptr), (3) a plain
    step() without a finite check after scaled steps -- the bias-correction exponent t = step - skipped must never reach 0 or jump."""
    import maskunet_amd
    torch.manual_seed(3)
    w = torch.randn(3000, device="cuda").requires_grad_(True)
    w_ref = w.detach().clone().requires_grad_(True)
    our = maskunet_amd.FusedAdamW([w], lr=1e-2, weight_decay=1e-2)
    ref = torch.optim.AdamW([w_ref], lr=1e-2, weight_decay=1e-2)

    def both(g, skip=False, scale=1.0, **kw):
        w.grad = g * scale
        if skip:
            w.grad[5] = float("inf")
        else:
            w_ref.grad = g.clone()
            ref.step()
        our.step(grad_scale=scale, **kw)

    both(torch.randn_like(w), scale=64.0)
    snap = our.state_dict()                                      # step = 1 (nothing skipped yet)
    snap = {"state": {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in snap["state"].items()},
            "param_groups": snap["param_groups"]}
    w_snap, ref_snap, ref_w_snap = w.detach().clone(), ref.state_dict(), w_ref.detach().clone()
    ref_snap = {"state": {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in ref_snap["state"].items()},
                "param_groups": ref_snap["param_groups"]}
    both(torch.randn_like(w), skip=True, scale=64.0)             # skipped on the device: skipped[0] = 1, host step = 2
    both(torch.randn_like(w), skip=True, scale=64.0)             # skipped[0] = 2, host step = 3
    assert our.effective_steps(w) == 1
    # (1) roll back to the snapshot: the loaded step (1) must not have the two stale skips subtracted (t would be -1 -> inf / NaN weights)
    our.load_state_dict(snap)
    ref.load_state_dict(ref_snap)
    with torch.no_grad():
        w.copy_(w_snap)
        w_ref.copy_(ref_w_snap)
    assert our.effective_steps(w) == 1
    both(torch.randn_like(w), scale=64.0)
    assert torch.isfinite(w).all() and torch.allclose(w, w_ref, rtol=2e-5, atol=2e-6)
    # (3) a skipped step, then a plain unchecked step: t continues from the applied count (2 -> 3), it does not jump by the skips
    both(torch.randn_like(w), skip=True, scale=64.0)
    both(torch.randn_like(w))                                    # grad_scale 1 -> no finite check, skipped[] still subtracted
    assert our.effective_steps(w) == 3
    assert torch.allclose(w, w_ref, rtol=2e-5, atol=2e-6)
    # (2) re-allocated storage (what model.to() / .float() do): the old plan's skip is folded into `step`, not dropped
    with torch.no_grad():
        w.data = w.data.clone()
    both(torch.randn_like(w), scale=64.0)
    assert our.effective_steps(w) == 4 and our.state_dict()["state"][0]["step"] == 4
    assert torch.allclose(w, w_ref, rtol=2e-5, atol=2e-6)



@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp16", "fp32x"])
def test_training_forward_takes_every_conv_layout_from_the_one_launch(precision, monkeypatch):
    """A training forward makes the compute layouts of ALL conv weights in one launch (ops.prep_conv_weights); no conv may fall back to
    its own mu_prep_weight launch -- neither in an eager step nor inside GraphedStep, whose forward runs under
    torch.func.functional_call on fresh leaves (round 6: the one-shot entries used to be parked on the Parameters there, and all 31
    convs of a captured step ran their own prep launch beside the unused combined one)."""
    import maskunet_amd
    from maskunet_amd import ops
    from tests import _gpu_checks as G
    dtype = torch.float16 if precision == "fp16" else torch.float32
    maskunet_amd.set_float32_matmul_precision("high" if precision == "fp32x" else "highest")
    try:
        model, params, keeps, x, labels = G.build_unet(19, False, 733, dtype, True, 2)
        x, labels = x.cuda(), labels.cuda()
        crit = maskunet_amd.CrossEntropyLoss()
        calls = []
        raw = ops._prep_weight_raw
        monkeypatch.setattr(ops, "_prep_weight_raw", lambda w, *a: (calls.append(tuple(w.shape)), raw(w, *a))[1])
        crit(model(x), labels).backward()
        assert calls == [], f"eager step: {len(calls)} per-layer prep launches {calls[:4]}"
        model.zero_grad(set_to_none=True)
        step = maskunet_amd.GraphedStep(model, crit, x, labels, warmup=1)
        step(x, labels)
        assert calls == [], f"captured step: {len(calls)} per-layer prep launches {calls[:4]}"
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")


@pytest.mark.gpu
def test_graphed_step_redraws_the_key_masks_per_replay():
    """GraphedStep with mask_mode = "resample" -- the reference's multi-GPU semantics (a fresh randint(0, 2, (B, H, W)) per replica
    forward, ade_semantic.py:177-181 under nn.DataParallel :373; VERDICT r5 #5a): the draws and the key compactions are captured with
    the step, every replay sees new masks, and the gradients of a replay equal an EAGER step run with the masks that replay drew."""
    import maskunet_amd
    from tests import _gpu_checks as G
    model, params, keeps, x, labels = G.build_unet(19, False, 611, torch.float16, True, 2)
    x, labels = x.cuda(), labels.cuda()
    crit = maskunet_amd.CrossEntropyLoss()
    model.dropout.p = 0.0
    model.set_mask_mode("resample")
    state = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(1234)
    step = maskunet_amd.GraphedStep(model, crit, x, labels, loss_scale=128.0, warmup=1)
    drawn, grads, losses_ = [], [], []
    for it in range(3):
        model.load_state_dict(state)                          # (the BatchNorm buffers advance with every step)
        losses_.append(step(x, labels).item())
        drawn.append([blk._keep.clone() for blk in model.attention_blocks()])
        grads.append({k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None})
    for a, b in ((0, 1), (1, 2), (0, 2)):                     # six masks of 256 ... 16384 fair coin flips each: no two replays agree
        assert all(not torch.equal(ka, kb) for ka, kb in zip(drawn[a], drawn[b])), (a, b)
    assert all(0.4 < float(k.float().mean()) < 0.6 for ks in drawn for k in ks)
    assert len(set(losses_)) == 3
    # an eager step with replay 1's masks, fixed
    model.zero_grad(set_to_none=True)
    model.load_state_dict(state)
    model.set_mask_mode("fixed")
    model.set_keep_masks([k.clone() for k in drawn[1]])
    loss_e = crit(model(x), labels)
    (loss_e * 128.0).backward()
    assert loss_e.item() == losses_[1]
    for k, v in model.named_parameters():
        if v.grad is not None:
            assert torch.equal(v.grad, grads[1][k]), k
