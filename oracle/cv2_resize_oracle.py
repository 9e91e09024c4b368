"""CPU restatement of the image / label-map preparation in front of the path.  TEST INFRASTRUCTURE ONLY.

The reference datasets (code/ade20k/ade_semantic.py:60-78, same in the other eight scripts) do, per sample:
    image_rgb = cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB)                                   (:62-65)
    image_rgb = cv2.resize(image_rgb, (128, 128), interpolation=cv2.INTER_LINEAR)                   (:72)
    mask      = cv2.resize(mask,      (128, 128), interpolation=cv2.INTER_NEAREST)                  (:73)
    image     = ToTensor()(image_rgb)      # uint8 HWC -> float32 CHW / 255                          (:75-76, 85)
    mask      = torch.from_numpy(mask).long()                                                       (:78)

cv2 is a third-party dependency that is NOT vendored under /root/reference and NOT installed in this image
(requirement.txt:168 pins opencv-python-headless==4.10.0.84), so these functions restate the published algorithm of that
version -- modules/imgproc/src/resize.cpp -- for 8-bit images:
  * INTER_NEAREST (resizeNN): sx = min(floor(dx * (1 / inv_scale_x)), W - 1), likewise in y; no half-pixel offset;
  * INTER_LINEAR on CV_8U: fixed point with INTER_RESIZE_COEF_BITS = 11.  Per destination column
    fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; columns left of the image take (sx, fx) = (0, 0), columns whose
    right tap falls outside take (W - 1, 0); alpha = saturate_cast<short>(cvRound((1 - fx, fx) * 2048)) (round half to even).
    Rows: the same fy / beta, but the two source rows are clamped to [0, H - 1] and fy is kept.  Horizontal pass (HResizeLinear)
    D = S[sx] * a0 + S[sx + 1] * a1 as int (S[sx] * 2048 past xmax); vertical pass (VResizeLinear<uchar, int, short>)
    dst = uchar((((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2);
  * the 2x2 shortcut of cv::resize: INTER_LINEAR with scale_x == scale_y == 2 exactly runs the INTER_AREA fast path,
    dst = (s00 + s01 + s10 + s11 + 2) >> 2.
Parity status of THIS file: pinned to hand-computed cases only (tests/test_oracle_golden.py::test_cv2_resize_*): cv2 itself cannot be
run here, so "parity unpinned against cv2" -- the restatement follows the cited source, nothing more is claimed.
Only tests/ may import this module (the HIP kernels mu_resize_u8_nhwc / mu_resize_nearest_u8 are checked against it bit for bit).
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _coeffs(dn: int, sn: int, clamp_taps: bool):
    """Per destination index: first source index and the two short coefficients (resize.cpp, INTER_LINEAR branch of the offset /
    coefficient loop).  clamp_taps=True is the x direction (out-of-range taps collapse to an edge pixel with weight 2048)."""
    inv_scale = float(dn) / float(sn)
    scale = 1.0 / inv_scale
    d = np.arange(dn, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_taps:
        left = s < 0
        f[left], s[left] = 0.0, 0
        right = s + 1 >= sn                      # sx + ksize2 >= width  ->  sx >= width - 1  ->  fx = 0, sx = width - 1
        f[right], s[right] = 0.0, sn - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int64)      # cvRound: half to even, as np.rint
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)
    c0, c1 = np.clip(c0, -32768, 32767), np.clip(c1, -32768, 32767)
    return s, c0, c1


def resize_linear_u8(src: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(src, (width, height), interpolation=cv2.INTER_LINEAR) for uint8 [H, W] or [H, W, C]."""
    assert src.dtype == np.uint8
    dw, dh = int(dsize[0]), int(dsize[1])
    img = src if src.ndim == 3 else src[:, :, None]
    sh, sw, _ = img.shape
    if sw == 2 * dw and sh == 2 * dh:            # cv::resize: INTER_LINEAR -> INTER_AREA when both scales are exactly 2
        a = img.astype(np.int64)
        out = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
        out = out.astype(np.uint8)
        return out if src.ndim == 3 else out[:, :, 0]
    sx, a0, a1 = _coeffs(dw, sw, True)
    sy, b0, b1 = _coeffs(dh, sh, False)
    s = img.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)             # never read where a1 == 0 at the right edge; clamped for the gather only
    hor = s[:, sx, :] * a0[None, :, None] + s[:, sx1, :] * a1[None, :, None]       # [sh, dw, C] ints (HResizeLinear)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    d0, d1 = hor[r0], hor[r1]                    # [dh, dw, C]
    out = (((b0[:, None, None] * (d0 >> 4)) >> 16) + ((b1[:, None, None] * (d1 >> 4)) >> 16) + 2) >> 2
    out = (out & 0xFF).astype(np.uint8)          # uchar(...) of the reference code: a plain narrowing cast
    return out if src.ndim == 3 else out[:, :, 0]


def resize_nearest(src: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(src, (width, height), interpolation=cv2.INTER_NEAREST) (resizeNN)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]
    ifx, ify = 1.0 / (float(dw) / float(sw)), 1.0 / (float(dh) / float(sh))
    sx = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * ifx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * ify).astype(np.int64), sh - 1)
    return src[sy][:, sx]


def prepare_sample(image_bgr: np.ndarray, mask: np.ndarray, size=(128, 128)):
    """ADE20KSegmentationDataset.__getitem__ after the two cv2.imread calls (ade_semantic.py:65-78 with transforms=ToTensor()):
    -> (float32 CHW image in [0, 1], int64 label map)."""
    rgb = image_bgr[:, :, ::-1]                                          # cv2.COLOR_BGR2RGB
    rgb = resize_linear_u8(np.ascontiguousarray(rgb), size)
    m = resize_nearest(mask, size)
    img = rgb.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)  # torchvision ToTensor(): HWC uint8 -> CHW float / 255
    return img, m.astype(np.int64)
