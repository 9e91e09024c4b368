"""CPU: pin the oracle (oracle/maskunet_oracle.py) against golden vectors generated from the
real reference by tests/golden/make_golden.py.  fp32 oracle vs fp32 reference: tolerance 2e-5
absolute on O(1) values (the fp32-vs-fp64 noise floor of the path is 7e-6..1.9e-5)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import maskunet_oracle as O

TOL = 2e-5


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def _params(rec, prefix_to, requires_grad=True):
    p = {}
    for k, v in rec.items():
        if k.startswith("param/"):
            t = torch.from_numpy(v.copy())
            if requires_grad and t.dtype.is_floating_point and "running" not in k:
                t.requires_grad_(True)
            p[prefix_to + "." + k[len("param/"):]] = t
    return p


def _check(a, b, tol=TOL, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.2f})"


MODULE_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(
    os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if not os.path.basename(p).startswith(("unet", "instloss", "miou", "resize")))


def test_have_module_cases():
    assert len(MODULE_CASES) >= 13


@pytest.mark.parametrize("name", MODULE_CASES)
def test_module_case(golden_dir, name):
    rec = _load(golden_dir, name)
    training = bool(rec["training"])
    p = _params(rec, "m")
    ins = [torch.from_numpy(rec[f"in/{i}"].copy()).requires_grad_(True) for i in range(2) if f"in/{i}" in rec]
    ns = {}
    if name.startswith("convblock"):
        out = O.conv_block(ins[0], p, "m", "res" in name, training, ns)
    elif name.startswith("down"):
        out = O.downsample(ins[0], p, "m", training, ns)
    elif name.startswith("up"):
        out = O.upsample(ins[0], ins[1], p, "m", training, ns)
    elif name.startswith("attn"):
        out = O.mask_attention(ins[0], p, "m", torch.from_numpy(rec["keep"]))
    else:
        raise AssertionError(name)
    _check(out, rec["out"], what="out")
    out.backward(torch.from_numpy(rec["gout"]))
    for i, t in enumerate(ins):
        _check(t.grad, rec[f"gin/{i}"], what=f"gin{i}")
    for k, v in rec.items():
        if k.startswith("gparam/"):
            _check(p["m." + k[len("gparam/"):]].grad, v, tol=5e-5, what=k)
        if k.startswith("newstat/") and training and not name.startswith("attn"):
            _check(ns["m." + k[len("newstat/"):]], v, what=k)


def test_attention_blockwise_equals_materialised(golden_dir):
    rec = _load(golden_dir, "attn_64_16x16")
    p = _params(rec, "m", requires_grad=False)
    x = torch.from_numpy(rec["in/0"])
    keep = torch.from_numpy(rec["keep"])
    a = O.mask_attention(x, p, "m", keep)
    b = O.mask_attention(x, p, "m", keep, q_block=48)
    _check(b, a.numpy(), tol=1e-6)


def test_attention_channel_mismatch_raises(golden_dir):
    rec = _load(golden_dir, "attn_32_8x8")
    p = _params(rec, "m", requires_grad=False)
    with pytest.raises(ValueError, match="Input channel size does not match"):
        O.mask_attention(torch.zeros(1, 16, 8, 8), p, "m", torch.ones(1, 64, dtype=torch.uint8))


def test_upsample_matches_torch():
    x = torch.randn(2, 3, 5, 7)
    ref = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    _check(O.upsample_bilinear2x(x), ref.numpy(), tol=1e-6)


def _unet_case(golden_dir, name, three_head):
    rec = _load(golden_dir, name)
    B, c_out, seed = int(rec["B"]), int(rec["c_out"]), int(rec["seed"])
    training = bool(rec["training"])
    shapes = O.unet_state_shapes(3, c_out, three_head)
    p = O.make_params(shapes, seed)
    for k, v in p.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(training)
    keeps = O.make_keeps(seed + 1, B)
    x, labels = O.make_inputs(seed + 2, B, c_out, ignore_frac=0.1 if three_head else 0.0)
    ns = {}
    out = O.unet_forward(p, x, keeps, training=training, new_stats=ns, three_head=three_head)
    outs = out if three_head else (out,)
    for i, o in enumerate(outs):
        _check(o[:, :, ::16, ::16], rec[f"out{i}_slice"], tol=5e-5, what=f"out{i}")
        assert abs(float(o.detach().double().sum()) - float(rec[f"out{i}_sum"])) <= 2e-5 * float(rec[f"out{i}_abssum"]) + 1e-3
    loss = O.pixel_cross_entropy(outs[0], labels, 255 if three_head else -100)
    if three_head:
        loss = loss + 0.5 * out[2].square().mean() + 0.25 * out[1].square().mean()
    assert abs(loss.item() - float(rec["loss"])) <= 2e-5 * max(1.0, abs(float(rec["loss"])))
    if not training:
        return
    loss.backward()
    # gamma/beta of a BN that feeds straight into another BN (ade_semantic.py:218-219) have an
    # analytically ~zero gradient: pure rounding noise, so compare with an absolute floor.
    floor = 1e-6 * max(float(v) for k, v in rec.items() if k.startswith("gnorm/"))
    for k, v in rec.items():
        if k.startswith("gnorm/"):
            key = k[len("gnorm/"):]
            has = bool(rec["ghas/" + key])
            g = p[key].grad
            assert (g is not None) == has, key
            if has:
                assert abs(float(g.double().norm()) - float(v)) <= 2e-4 * float(v) + floor, key
        elif k.startswith("g/"):
            err = float(np.abs(p[k[2:]].grad.numpy() - v).max())
            assert err <= 1e-4 * float(np.abs(v).max()) + floor, (k, err)
        elif k.startswith("newstat/"):
            _check(ns[k[len("newstat/"):]], v, what=k)
    _check(p["norm.weight"].grad[:, ::16, ::16], rec["g_slice/norm.weight"], tol=1e-4)
    _check(p["initial_conv.conv_block.0.weight"].grad, rec["g_slice/initial_conv.conv_block.0.weight"], tol=1e-4)


def test_unet1_eval(golden_dir):
    _unet_case(golden_dir, "unet1_c150_b2_eval", False)


def test_unet1_train(golden_dir):
    _unet_case(golden_dir, "unet1_c150_b2_train", False)


def test_unet3_train(golden_dir):
    _unet_case(golden_dir, "unet3_c19_b2_train", True)


def test_unet1_c133_train(golden_dir):
    """BASELINE configs[2] (COCO panoptic, c_out = 133): vectors from coco_panoptic.py's own UNet class (:279,472)."""
    _unet_case(golden_dir, "unet1_c133_b2_train", False)


@pytest.mark.parametrize("name", ["miou_dense", "miou_absent_classes", "miou_ties_150"])
def test_oracle_mean_iou(golden_dir, name):
    """oracle.mean_iou == the reference's mean_iou (ade_semantic.py:128-146) on the fixtures written by
    tests/golden/make_golden_losses.py (absent classes, union == 0 skip, argmax ties)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    got = O.mean_iou(torch.from_numpy(g["y"]), torch.from_numpy(g["t"]), int(g["num_classes"]))
    assert abs(float(got) - float(g["miou"])) <= 1e-7


@pytest.mark.parametrize("name", ["instloss_ade_small", "instloss_ade_sparse", "instloss_city_ignore", "instloss_none"])
def test_oracle_instance_contrastive_loss(golden_dir, name):
    """oracle.instance_contrastive_loss == the reference's InstanceContrastiveLoss (loss and feature gradient) on the fixtures
    written by tests/golden/make_golden_losses.py."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    feat = torch.from_numpy(g["feat"]).requires_grad_(True)
    ign = int(g["ignore"])
    loss = O.instance_contrastive_loss(feat, torch.from_numpy(g["mask"]), torch.from_numpy(g["u"]), 1.0, None if ign < 0 else ign)
    if loss.requires_grad:
        loss.backward()
    grad = feat.grad if feat.grad is not None else torch.zeros_like(feat)
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    assert float((grad - torch.from_numpy(g["dfeat"])).abs().max()) <= 1e-6


# ------------------------------------------------------------------------------------------------
# SURVEY 8-f4, resize half: oracle/cv2_resize_oracle.py restates OpenCV 4.10's 8-bit cv2.resize (cv2 is not installed here and not
# vendored by the reference).  Pinned by values computed by hand from the published fixed-point formulas.
# ------------------------------------------------------------------------------------------------
def test_cv2_resize_linear_hand_computed():
    from oracle import cv2_resize_oracle as R
    src = np.array([[0, 100], [200, 255]], np.uint8)
    # x: dx=0 -> sx=-1 -> (0, weight 2048); dx=1 -> fx=.25 -> (1536, 512); dx=2 -> fx=.75 -> (512, 1536); dx=3 -> right tap outside -> (1, 2048)
    # y: dy=0 -> sy=-1, fy=.75 but both rows clamp to row 0; dy=1 -> (1536, 512); ...
    # (1,1): D0 = 0*1536 + 100*512 = 51200, D1 = 200*1536 + 255*512 = 437760
    #        ((1536*(51200>>4))>>16) + ((512*(437760>>4))>>16) + 2 >> 2 = (75 + 213 + 2) >> 2 = 72
    # (1,2): D0 = 153600, D1 = 494080 -> (225 + 241 + 2) >> 2 = 117
    want = np.array([[0, 25, 75, 100], [50, 72, 117, 139], [150, 167, 200, 216], [200, 214, 241, 255]], np.uint8)
    assert np.array_equal(R.resize_linear_u8(src, (4, 4)), want)
    # exact 2x downscale: cv::resize switches INTER_LINEAR to the INTER_AREA fast path, (a + b + c + d + 2) >> 2
    a = (np.arange(16, dtype=np.uint8).reshape(4, 4) * 10)
    assert np.array_equal(R.resize_linear_u8(a, (2, 2)), np.array([[25, 45], [105, 125]], np.uint8))
    assert np.array_equal(R.resize_linear_u8(np.array([[1, 2], [2, 2]], np.uint8), (1, 1)), np.array([[2]], np.uint8))     # (7 + 2) >> 2
    # same size: every fx = fy = 0 -> identity; channels are independent
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_linear_u8(img, (5, 7)), img)
    assert np.array_equal(R.resize_linear_u8(img, (11, 13))[:, :, 1], R.resize_linear_u8(np.ascontiguousarray(img[:, :, 1]), (11, 13)))
    # a constant image stays constant for every scale (a0 + a1 == 2048 wherever both taps are used ... up to the >>4 / >>16 floors:
    # 255 -> D = 522240, (b0*(32640))>>16 + (b1*32640)>>16 + 2 >> 2 = 255 when b0 + b1 = 2048 and the two floors lose < 2 together)
    assert int(R.resize_linear_u8(np.full((9, 14), 255, np.uint8), (31, 23)).min()) == 255


def test_cv2_resize_nearest_hand_computed():
    from oracle import cv2_resize_oracle as R
    src = np.arange(12, dtype=np.uint8).reshape(3, 4)
    # columns floor(dx * 4/6) = 0,0,1,2,2,3; rows floor(dy * 3/5) = 0,0,1,1,2
    want = np.array([[0, 0, 1, 2, 2, 3], [0, 0, 1, 2, 2, 3], [4, 4, 5, 6, 6, 7], [4, 4, 5, 6, 6, 7], [8, 8, 9, 10, 10, 11]], np.uint8)
    assert np.array_equal(R.resize_nearest(src, (6, 5)), want)
    assert np.array_equal(R.resize_nearest(src, (2, 1)), np.array([[0, 2]], np.uint8))           # floor(dx * 2), row 0


def test_cv2_resize_fixture_is_reproducible(golden_dir):
    """tests/golden/resize_cases.npz is what tests/golden/make_golden_resize.py writes from the oracle (the GPU tests hold the HIP
    kernels to it bit for bit)."""
    from oracle import cv2_resize_oracle as R
    z = np.load(os.path.join(golden_dir, "resize_cases.npz"))
    n = len([k for k in z.files if k.endswith("_img")])
    assert n >= 6
    for i in range(n):
        dw, dh = (int(v) for v in z[f"c{i}_dsize"])
        assert np.array_equal(R.resize_linear_u8(z[f"c{i}_img"], (dw, dh)), z[f"c{i}_lin"])
        assert np.array_equal(R.resize_nearest(z[f"c{i}_lab"], (dw, dh)), z[f"c{i}_near"])
    img, lab = R.prepare_sample(z["c0_img"], z["c0_lab"], (128, 128))
    assert img.shape == (3, 128, 128) and img.dtype == np.float32 and lab.dtype == np.int64 and float(img.max()) <= 1.0
    assert np.array_equal((img * 255).round().astype(np.uint8).transpose(1, 2, 0), R.resize_linear_u8(np.ascontiguousarray(z["c0_img"][:, :, ::-1]), (128, 128)))


def test_cv2_resize_restatement_against_an_independent_bilinear():
    """Not a cv2 pin (cv2 is not in the image: SURVEY 8-f4 stays "partial"), but an INDEPENDENT cross-check of the restatement: OpenCV's
    INTER_LINEAR samples at half-pixel centres with clamped borders -- the definition torch's float `interpolate(mode="bilinear",
    align_corners=False)` implements -- in 11-bit fixed point, so the restated bytes must sit within 1 LSB of the rounded float result at
    every pixel (ragged up- and down-scales, the exact 2x case where cv2 takes its area shortcut included); INTER_NEAREST is
    floor(dst * scale), torch's legacy "nearest": exact."""
    from oracle import cv2_resize_oracle as R
    rng = np.random.default_rng(11)
    for (h, w), (dh, dw) in [((37, 53), (128, 128)), ((300, 200), (128, 128)), ((256, 256), (128, 128)), ((64, 48), (96, 80)), ((131, 517), (128, 128))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = R.resize_linear_u8(img, (dw, dh)).astype(np.int32)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        err = np.abs(got - ref)
        assert float(err.max()) <= 1.0 + 1e-9, ((h, w), (dh, dw), float(err.max()))
        assert float(err.mean()) <= 0.30, float(err.mean())                  # rounding noise, not a systematic offset
        lab = rng.integers(0, 151, (h, w), dtype=np.uint8)
        near = R.resize_nearest(lab, (dw, dh))
        tn = torch.nn.functional.interpolate(torch.from_numpy(lab)[None, None].float(), size=(dh, dw), mode="nearest")[0, 0].numpy().astype(np.uint8)
        assert np.array_equal(near, tn)

