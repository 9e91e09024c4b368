#!/usr/bin/env python3
"""Fixtures for the resize half of SURVEY 8-f4 (cv2.resize INTER_LINEAR / INTER_NEAREST on 8-bit images, ade_semantic.py:72-73).

cv2 (opencv-python-headless==4.10.0.84, requirement.txt:168) is neither vendored under /root/reference nor installed in the build
image, so these vectors come from oracle/cv2_resize_oracle.py -- the numpy restatement of OpenCV's published 8-bit algorithm -- and
NOT from cv2 itself; the hand-computed cases in tests/test_oracle_golden.py are what pins that restatement.  Run from the repo root:
    python tests/golden/make_golden_resize.py        -> tests/golden/resize_cases.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cv2_resize_oracle as R  # noqa: E402

# (source height, source width) -> 128 x 128 unless noted: ragged downscale, ADE20K-like aspect, exact 2x (area shortcut), upscale,
# integer 4x (stays linear), one-pixel-wide source
CASES = [((37, 53), (128, 128)), ((171, 128), (128, 128)), ((64, 64), (32, 32)), ((50, 70), (128, 128)), ((64, 64), (16, 16)),
         ((9, 1), (16, 8)), ((150, 100), (64, 96)), ((32, 32), (32, 32))]


def main():
    rng = np.random.default_rng(20240)
    out = {}
    for i, ((sh, sw), (dw, dh)) in enumerate(CASES):
        img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        lab = rng.integers(0, 151, (sh, sw), dtype=np.uint8)
        if i == 1:                                  # a smooth image too: gradients + a saturated corner
            yy, xx = np.mgrid[0:sh, 0:sw]
            img = np.stack([(yy * 255 // (sh - 1)), (xx * 255 // (sw - 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
            img[:8, :8] = 255
        out[f"c{i}_img"] = img
        out[f"c{i}_lab"] = lab
        out[f"c{i}_dsize"] = np.array([dw, dh])
        out[f"c{i}_lin"] = R.resize_linear_u8(img, (dw, dh))
        out[f"c{i}_near"] = R.resize_nearest(lab, (dw, dh))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resize_cases.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
