"""GPU parity tests for the BASELINE.json configurations the B=2 goldens do not reach:

  configs[2]  COCO panoptic 128x128, c_out = 133 (coco_panoptic.py:279,472; head padded 133 -> 160 channels)
  configs[4]  COCO semantic 256x256 (hw=256: N = 65 536 tokens in self_attention6; the reference cannot run it, ade_semantic.py:281)
  configs[1]  B = 64: batch-dependent kernel dispatch (persistent conv tiles, one-round split-K plans, >= 8 images per XCD
              attention order) checked through batch invariance in eval mode.

Where a CPU autograd graph is not affordable (256x256: 17 GB) the backward is held to size-independent properties: the fp16
path against the HIP fp32 path (per-parameter cosine), bit-reproducibility, and exact zeros for masked keys' dK/dV.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16]


def _assert_all(results):
    bad = [(n, e, t) for n, e, t in results if not (e <= t)]
    assert not bad, "parity failures (name, err, tol): " + "; ".join(f"{n}: {e:.3e} > {t:.1e}" for n, e, t in bad)


@pytest.mark.parametrize("dtype", DTYPES)
def test_unet_c133_vs_oracle(dtype):
    """configs[2] shape at B=2, training mode with dropout masks, against the oracle run live (outputs, loss, every parameter
    gradient, BatchNorm running statistics)."""
    from tests import _gpu_checks as G
    _assert_all(G.check_unet_vs_oracle(dtype, B=2, c_out=133, seed=1330))


def _build(c_out, hw, seed, dtype, training, B):
    import maskunet_amd
    from oracle import maskunet_oracle as O
    params = O.make_params(O.unet_state_shapes(3, c_out, False, hw=hw), seed)
    model = maskunet_amd.UNet(3, c_out, hw=hw)
    model.load_state_dict(params)
    model.cuda().set_compute_dtype(dtype).train(training)
    model.dropout.p = 0.0
    keeps = O.make_keeps(seed + 1, B, hw)
    model.set_keep_masks(keeps)
    x, labels = O.make_inputs(seed + 2, B, c_out, hw)
    return model, params, keeps, x, labels


@pytest.mark.parametrize("training", [False, True])
def test_unet_256_forward_vs_blockwise_oracle(training):
    """configs[4]: UNet(3,133,hw=256), B=1, forward in fp32 and fp16 against the oracle with query-blocked attention
    (O(q_block*N) memory, same math; tests/test_oracle_golden.py::test_attention_blockwise_equals_materialised)."""
    from oracle import maskunet_oracle as O
    from tests import _gpu_checks as G
    torch.set_num_threads(min(torch.get_num_threads(), 32))
    model, params, keeps, x, labels = _build(133, 256, 2560, torch.float32, training, 1)
    ns = {}
    with torch.no_grad():
        ref = O.unet_forward(params, x, keeps, training=training, new_stats=ns, q_block=2048)
    res = []
    for dtype in DTYPES:
        model.load_state_dict(params)
        model.set_compute_dtype(dtype)
        with torch.no_grad():
            out = model(x.cuda())
        res.append((f"unet256 {'train' if training else 'eval'} out {dtype}", G._err(out, ref), G.TOL[dtype]))
        if training:
            worst = max(G._err(v, ns[k]) for k, v in model.state_dict().items() if k in ns)
            res.append((f"unet256 running stats {dtype}", worst, G.TOL[dtype]))
    _assert_all(res)


def _grads(model, x, labels, scale):
    model.zero_grad(set_to_none=True)
    out = model(x)
    (F.cross_entropy(out, labels) * scale).backward()
    return out.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _worst_cos(ga, gb, sa, sb):
    gmax = max(float(v.abs().max()) / sb for v in gb.values())
    worst = (0.0, "")
    for n, r in gb.items():
        r = r.double() / sb
        if float(r.abs().max()) < 1e-3 * gmax:      # analytically-zero gradients (gamma/beta in front of a second BatchNorm, key.bias)
            continue
        g = ga[n].double() / sa
        c = 1.0 - float((g * r).sum() / (g.norm() * r.norm() + 1e-300))
        if c > worst[0]:
            worst = (c, n)
    return worst


@pytest.mark.parametrize("hw,c_out,B", [(256, 133, 1), (128, 150, 2)])
def test_fp16_backward_tracks_fp32_hip_backward(hw, c_out, B):
    """The fp16 path (fp16 storage, fp32 accumulate, loss scale 1024) against the HIP fp32 path on the same inputs: per-parameter
    1 - cosine of every live gradient.  Both runs share kernels' structure, so what this isolates is fp16 rounding; at 256x256 it is
    the backward check (a CPU autograd graph of that size is not affordable).  Also: both runs are bit-reproducible."""
    model, params, keeps, x, labels = _build(c_out, hw, 2570 + hw, torch.float32, True, B)
    xd, yd = x.cuda(), labels.cuda()
    o32, g32 = _grads(model, xd, yd, 1.0)
    model.load_state_dict(params)
    o32b, g32b = _grads(model, xd, yd, 1.0)
    assert torch.equal(o32, o32b) and all(torch.equal(g32[n], g32b[n]) for n in g32), "fp32 step is not bit-reproducible"
    model.load_state_dict(params)
    model.set_compute_dtype(torch.float16)
    o16, g16 = _grads(model, xd, yd, 1024.0)
    model.load_state_dict(params)
    o16b, g16b = _grads(model, xd, yd, 1024.0)
    assert torch.equal(o16, o16b) and all(torch.equal(g16[n], g16b[n]) for n in g16), "fp16 step is not bit-reproducible"
    assert set(g16) == set(g32)
    assert all(torch.isfinite(v).all() for v in g16.values())
    err_out = float((o16 - o32).abs().max()) / max(1.0, float(o32.abs().max()))
    worst = _worst_cos(g16, g32, 1024.0, 1.0)
    print(f"fp16 vs fp32 HIP at {hw}x{hw}: out err {err_out:.3e}, worst 1-cos {worst[0]:.3e} [{worst[1]}]")
    assert err_out <= 3e-2, err_out
    assert worst[0] <= 2e-2, worst


def test_attention_65536_tokens_masked_keys_get_exact_zero_gradients():
    """self_attention6 at 256x256 (N = 65 536, C = 64) through the C ABI: dK and dV rows of masked keys are exact zeros, dQ rows
    are not; forward rows agree with a torch recomputation on a sample of queries."""
    from maskunet_amd import _lib
    B, N, C = 1, 256 * 256, 64
    for dtype, tol in ((torch.float32, 1e-3), (torch.float16, 3e-2)):
        g_ = torch.Generator(device="cuda").manual_seed(5)
        qkv = torch.randn(B, N, 3 * C, device="cuda", generator=g_).to(dtype)
        x = torch.randn(B, N, C, device="cuda", generator=g_).to(dtype)
        keep = torch.randint(0, 2, (B, N), device="cuda", generator=g_, dtype=torch.uint8)
        kidx = torch.argsort(keep, dim=1, descending=True, stable=True).to(torch.int32).contiguous()
        kcnt = keep.sum(1, dtype=torch.int32).contiguous()
        g = torch.ones(C, device="cuda")
        b_ = torch.zeros(C, device="cuda")
        out, oattn = torch.empty_like(x), torch.empty_like(x)
        lse = torch.empty(B, N, device="cuda")
        mean, rstd, delta = torch.empty_like(lse), torch.empty_like(lse), torch.empty_like(lse)
        dY, dqkv = torch.empty_like(x), torch.full_like(qkv, 7.0)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        gout = torch.randn(B, N, C, device="cuda", generator=g_).to(dtype)
        ws = _lib.workspace(_lib.load().mu_attn_bwd_workspace_bytes(B, N, C), torch.device("cuda"))
        st = _lib.stream()
        _lib.call("mu_attn_fwd", qkv.data_ptr(), x.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), g.data_ptr(), b_.data_ptr(),
                  out.data_ptr(), oattn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, C, N, 1e-5, _lib.dt(x), st)
        _lib.call("mu_attn_bwd", qkv.data_ptr(), x.data_ptr(), oattn.data_ptr(), gout.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(),
                  lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dY.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                  dg.data_ptr(), db.data_ptr(), B, N, C, N, ws.data_ptr(), ws.numel(), _lib.dt(x), st)
        masked = keep[0] == 0
        dk, dv, dq = dqkv[0, :, C:2 * C], dqkv[0, :, 2 * C:], dqkv[0, :, :C]
        assert float(dk[masked].abs().max()) == 0.0 and float(dv[masked].abs().max()) == 0.0
        assert float(dk[~masked].abs().max()) > 0 and float(dq.abs().max()) > 0 and torch.isfinite(dqkv).all()
        # forward sample: 256 queries against all kept keys, fp64 on the device
        qs = torch.arange(0, N, N // 256, device="cuda")
        q, k, v = (qkv[0, :, i * C:(i + 1) * C].double() for i in range(3))
        s = (q[qs] @ k[~masked].T) / (C ** 0.5)
        o = torch.softmax(s, dim=-1) @ v[~masked] + x[0, qs].double()
        ref = F.layer_norm(o, (C,), eps=1e-5)
        err = float((out[0, qs].double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err <= tol, (dtype, err)


def test_batch64_eval_is_batch_invariant():
    """configs[1] batch size: in eval mode (running-statistics BatchNorm, no dropout) every image is independent, so images 0..1
    of a B=64 fp16 batch must reproduce the B=2 run -- bit for bit where the kernels' per-image arithmetic does not depend on the
    batch-dependent dispatch (persistent conv tile walk, attention XCD order), which is the design."""
    from oracle import maskunet_oracle as O
    model, params, keeps, x, _ = _build(150, 128, 640, torch.float16, False, 64)
    with torch.no_grad():
        big = model(x.cuda())
    model.set_keep_masks([k[:2] for k in keeps])
    with torch.no_grad():
        small = model(x[:2].cuda())
    assert torch.isfinite(big).all()
    diff = float((big[:2] - small).abs().max())
    assert diff == 0.0, f"B=64 and B=2 disagree on the same images: max abs diff {diff:.3e}"
    # and the B=2 run is the one the reference goldens pin (same code path as tests/test_gpu_modules.py::test_unet_golden)
    with torch.no_grad():
        ref = O.unet_forward(params, x[:2], [k[:2] for k in keeps], training=False)
    assert float((small.cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max())) <= 3e-2


def test_batch64_training_step_matches_split_batches_statistics_free_parts():
    """B = 64 fp16 training step: finite loss and gradients, bit-reproducible, and the step's loss equals the eval-free recomputation
    of the criterion on its own output (guards the B=64-only dispatch paths of the backward: one-round split-K plans and the
    persistent data-gradient kernel)."""
    import maskunet_amd
    model, params, keeps, x, labels = _build(150, 128, 641, torch.float16, True, 64)
    xd, yd = x.cuda(), labels.cuda()
    crit = maskunet_amd.CrossEntropyLoss()
    runs = []
    for _ in range(2):
        model.load_state_dict(params)
        model.zero_grad(set_to_none=True)
        out = model(xd)
        loss = crit(out, yd)
        (loss * 1024.0).backward()
        runs.append((out.detach().clone(), loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert runs[0][1] == runs[1][1] and torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(runs[0][2][n], runs[1][2][n]) for n in runs[0][2])
    assert all(torch.isfinite(v).all() for v in runs[0][2].values())
    ref_loss = F.cross_entropy(runs[0][0].float(), yd).item()
    assert abs(ref_loss - runs[0][1]) <= 1e-4 * max(1.0, abs(ref_loss))
    # weight gradients of a B=64 batch vs the sum over its two halves is NOT an identity in training mode (batch statistics),
    # so the cross-check of the split-K plans is made where it is one: the first conv's weight gradient from (x, dy)
    from maskunet_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    for (cin, cout, hw) in [(128, 128, 128), (64, 64, 128), (256, 256, 32)]:
        xa = torch.randn(64, hw, hw, cin, device="cuda", generator=g).half()
        dy = torch.randn(64, hw, hw, cout, device="cuda", generator=g).half()
        full = ops._wgrad_raw(xa, dy, (cout, cin, 3, 3), 9)
        halves = ops._wgrad_raw(xa[:32].contiguous(), dy[:32].contiguous(), (cout, cin, 3, 3), 9) + \
            ops._wgrad_raw(xa[32:].contiguous(), dy[32:].contiguous(), (cout, cin, 3, 3), 9)
        rel = float((full - halves).abs().max()) / float(halves.abs().max())
        assert rel <= 1e-3, ((cin, cout, hw), rel)


def test_three_head_batch64_training_step():
    """configs[3] per-GPU shape: the 3-head Cityscapes model (c_out = 19, embed 16) at B = 64, fp16, with the reference's criterion
    (CrossEntropyLoss(ignore_index=255) + 0.1 * InstanceContrastiveLoss on the embeddings, city_instance.py:372-377): finite,
    bit-reproducible, every head's parameters get gradients except the boundary head (unused by that loss, as in the reference)."""
    import maskunet_amd
    import bench
    torch.manual_seed(5)
    B = 64
    model = maskunet_amd.InstanceUNet(3, 19, 16).cuda()
    model.set_compute_dtype(torch.float16).train()
    model.dropout.p = 0.0
    x, labels, keeps = bench.synth(B, 19, 128, 11, torch.device("cuda"), ignore_frac=0.1)
    model.set_keep_masks(keeps)
    g = torch.Generator().manual_seed(3)
    blocks = torch.randint(0, 21, (B, 8, 8), generator=g)
    inst = blocks.repeat_interleave(16, 1).repeat_interleave(16, 2).cuda()
    u = torch.rand(1024, generator=g).cuda()
    crit = maskunet_amd.CrossEntropyLoss(ignore_index=255)
    iloss = maskunet_amd.InstanceContrastiveLoss(margin=1.0, ignore_index=255)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(2):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        sem, bnd, emb = model(x)
        loss = crit(sem, labels) + 0.1 * iloss(emb, inst, u)
        (loss * 1024.0).backward()
        runs.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0] and runs[0][0] == runs[0][0]
    assert all(torch.equal(runs[0][1][n], runs[1][1][n]) and torch.isfinite(runs[0][1][n]).all() for n in runs[0][1])
    names = set(runs[0][1])
    assert any(n.startswith("embedding_head") for n in names) and any(n.startswith("final_layer") for n in names)
    assert not any(n.startswith("boundary_head") for n in names) and not any("emb_layer" in n for n in names)
    assert tuple(sem.shape) == (B, 19, 128, 128) and tuple(bnd.shape) == (B, 1, 128, 128) and tuple(emb.shape) == (B, 16, 128, 128)


@pytest.mark.parametrize("c_out,hw,B", [(133, 128, 128), (133, 256, 32)])
def test_full_size_configs_training_step_and_batch_invariance(c_out, hw, B):
    """configs[2] at its full batch (COCO panoptic shape, c_out 133, B = 128) and configs[4] at its per-GPU batch (256x256, B = 32):
    the batch-dependent dispatch (persistent tile walk, split-K plans, XCD image order, 64-bit offsets -- the 256x256 tensors are
    671 MB) under the same property set as the B = 64 tests: a fp16 training step is finite and bit-reproducible, its loss equals a
    recomputation of the criterion on its own output, and in eval mode images 0..1 of the big batch equal the B = 2 run bit for bit
    (the B = 2 run is what the reference goldens / the live oracle pin at these shapes: test_unet_golden[unet1_c133_b2_train],
    test_unet_c133_vs_oracle, test_unet_256_forward_vs_blockwise_oracle)."""
    import maskunet_amd
    model, params, keeps, x, labels = _build(c_out, hw, 7000 + hw + B, torch.float16, True, B)
    xd, yd = x.cuda(), labels.cuda()
    crit = maskunet_amd.CrossEntropyLoss()
    runs = []
    for _ in range(2):
        model.load_state_dict(params)
        model.zero_grad(set_to_none=True)
        out = model(xd)
        loss = crit(out, yd)
        (loss * 1024.0).backward()
        runs.append((out.detach().clone(), loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
        del out, loss
    assert runs[0][1] == runs[1][1] and torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(runs[0][2][n], runs[1][2][n]) for n in runs[0][2])
    assert all(torch.isfinite(v).all() for v in runs[0][2].values()) and len(runs[0][2]) >= 160
    ref_loss = F.cross_entropy(runs[0][0].float(), yd).item()
    assert abs(ref_loss - runs[0][1]) <= 1e-4 * max(1.0, abs(ref_loss))
    # the gradient of the last image's logits is not zero: every image of the batch took part (a dropped tail of the batch would
    # leave finite, reproducible, but wrong results)
    gin = torch.autograd.grad(crit(model(xd.requires_grad_(True)), yd), xd)[0]
    assert float(gin[-1].abs().max()) > 0 and float(gin[B // 2].abs().max()) > 0 and torch.isfinite(gin).all()
    del runs, gin
    model.zero_grad(set_to_none=True)
    model.load_state_dict(params)
    model.eval()
    with torch.no_grad():
        big = model(xd.detach())
    model.set_keep_masks([k[:2] for k in keeps])
    with torch.no_grad():
        small = model(xd.detach()[:2])
    assert torch.isfinite(big).all()
    diff = float((big[:2] - small).abs().max())
    assert diff == 0.0, f"B={B} and B=2 disagree on the same images: max abs diff {diff:.3e}"


def test_attention_65536_tokens_batch32_offsets():
    """self_attention6 of configs[4] at its per-GPU batch through the C ABI (fp16): qkv is 805 MB, so image 31 sits behind 32-bit
    element offsets; its forward rows agree with a torch recomputation and its masked keys get exact-zero dK / dV."""
    from maskunet_amd import _lib
    B, N, C = 32, 256 * 256, 64
    g_ = torch.Generator(device="cuda").manual_seed(9)
    qkv = torch.randn(B, N, 3 * C, device="cuda", generator=g_).half()
    x = torch.randn(B, N, C, device="cuda", generator=g_).half()
    keep = torch.randint(0, 2, (B, N), device="cuda", generator=g_, dtype=torch.uint8)
    kidx = torch.argsort(keep, dim=1, descending=True, stable=True).to(torch.int32).contiguous()
    kcnt = keep.sum(1, dtype=torch.int32).contiguous()
    g, b_ = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    out, oattn = torch.empty_like(x), torch.empty_like(x)
    lse = torch.empty(B, N, device="cuda")
    mean, rstd, delta = torch.empty_like(lse), torch.empty_like(lse), torch.empty_like(lse)
    dY, dqkv = torch.empty_like(x), torch.full_like(qkv, 7.0)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    gout = torch.randn(B, N, C, device="cuda", generator=g_).half()
    ws = _lib.workspace(_lib.load().mu_attn_bwd_workspace_bytes(B, N, C), torch.device("cuda"))
    st = _lib.stream()
    _lib.call("mu_attn_fwd", qkv.data_ptr(), x.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), g.data_ptr(), b_.data_ptr(),
              out.data_ptr(), oattn.data_ptr(), lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, C, N, 1e-5, _lib.dt(x), st)
    _lib.call("mu_attn_bwd", qkv.data_ptr(), x.data_ptr(), oattn.data_ptr(), gout.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(),
              lse.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dY.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
              dg.data_ptr(), db.data_ptr(), B, N, C, N, ws.data_ptr(), ws.numel(), _lib.dt(x), st)
    assert torch.isfinite(dqkv).all()
    for b in (0, 17, 31):
        masked = keep[b] == 0
        dk, dv = dqkv[b, :, C:2 * C], dqkv[b, :, 2 * C:]
        assert float(dk[masked].abs().max()) == 0.0 and float(dv[masked].abs().max()) == 0.0 and float(dk[~masked].abs().max()) > 0
        qs = torch.arange(0, N, N // 128, device="cuda")
        q, k, v = (qkv[b, :, i * C:(i + 1) * C].double() for i in range(3))
        s = (q[qs] @ k[~masked].T) / (C ** 0.5)
        o = torch.softmax(s, dim=-1) @ v[~masked] + x[b, qs].double()
        ref = F.layer_norm(o, (C,), eps=1e-5)
        err = float((out[b, qs].double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err <= 3e-2, (b, err)


def test_fp32x_batch64_training_step_is_reproducible_and_tracks_fp32():
    """The fp32x mode (fp32 storage, split-bf16 matrix products on chunk-encoded operands) at the bench shape: a B = 64 training step is
    finite and bit-reproducible (the xf32 instantiations of the ring / LDS-DMA kernels under the batch-dependent dispatch), and its
    outputs / gradients agree with the exact-fp32 HIP path of the same step far inside the fp32 gates."""
    import maskunet_amd
    model, params, keeps, x, labels = _build(150, 128, 7640, torch.float32, True, 64)
    xd, yd = x.cuda(), labels.cuda()

    def run():
        model.load_state_dict(params)
        return _grads(model, xd, yd, 1.0)
    o32, g32 = run()
    maskunet_amd.set_float32_matmul_precision("high")
    try:
        ox, gx = run()
        ox2, gx2 = run()
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")
    assert torch.equal(ox, ox2) and all(torch.equal(gx[n], gx2[n]) for n in gx), "fp32x step is not bit-reproducible"
    assert set(gx) == set(g32) and all(torch.isfinite(v).all() for v in gx.values())
    err_out = float((ox - o32).abs().max()) / max(1.0, float(o32.abs().max()))
    worst = _worst_cos(gx, g32, 1.0, 1.0)
    print(f"fp32x vs fp32 HIP at B=64: out err {err_out:.3e}, worst 1-cos {worst[0]:.3e} [{worst[1]}]")
    assert err_out <= 1e-3, err_out
    assert worst[0] <= 1e-4, worst
