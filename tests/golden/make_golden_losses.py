#!/usr/bin/env python3
"""Golden vectors for the reference's InstanceContrastiveLoss (build container only; needs /root/reference, CPU).

The class is AST-extracted from ade_panoptic.py (no ignore label) and city_instance.py (ignore label 255) and run on small
inputs.  torch.randint -- the reference's draw of the negative pixel -- is patched for the duration of the call to return
floor(u[k] * n) for its k-th call, so the fixtures are reproducible from the stored `u`.  Stored: inputs, u, loss, d loss / d features.
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_class(path, name):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name]
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def run(cls, feat, mask, u):
    calls = {"k": 0}
    real = torch.randint

    def fake(low, high, size, **kw):
        k = calls["k"]
        calls["k"] += 1
        return torch.tensor([min(int(float(u[k]) * high), high - 1)])

    torch.randint = fake
    try:
        f = feat.clone().requires_grad_(True)
        loss = cls(margin=1.0)(f, mask)
        if loss.requires_grad:
            loss.backward()
        g = f.grad if f.grad is not None else torch.zeros_like(f)
    finally:
        torch.randint = real
    return loss.detach(), g, calls["k"]


def case(name, cls, B, C, H, W, ids, ignore, seed):
    rng = np.random.default_rng(seed)
    feat = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    mask = torch.from_numpy(rng.choice(np.array(ids, dtype=np.int64), size=(B, H, W)))
    u = torch.from_numpy(rng.random(64).astype(np.float32))
    loss, g, draws = run(cls, feat, mask, u)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), feat=feat.numpy(), mask=mask.numpy(), u=u.numpy(), loss=loss.numpy(),
                        dfeat=g.numpy(), draws=np.array(draws), ignore=np.array(-1 if ignore is None else ignore))
    print("wrote", name, "loss", float(loss), "draws", draws)


def load_function(path, name):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def miou_case(name, fn, B, C, H, W, seed, label_hi, never_predicted=(), tie=False):
    """mean_iou (ade_semantic.py:128-146) on random logits.  label_hi < C leaves classes absent from the labels;
    `never_predicted` classes get -10 logits (absent from both sides when also >= label_hi: union == 0, skipped);
    tie: two classes share the maximal logit on some pixels (argmax must pick the first, as torch.argmax does)."""
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((B, C, H, W)).astype(np.float32)
    for c in never_predicted:
        y[:, c] = -10.0
    if tie:
        y[:, 3, ::2] = y.max(axis=1)[:, ::2] + 1.0
        y[:, 5, ::2] = y[:, 3, ::2]
    t = rng.integers(0, label_hi, (B, H, W)).astype(np.int64)
    out = fn(torch.from_numpy(y), torch.from_numpy(t), C)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), y=y, t=t, num_classes=np.array(C), miou=np.array(float(out), dtype=np.float64))
    print("wrote", name, "mean_iou", float(out))


def main():
    miou = load_function(os.path.join(REF, "ade20k/ade_semantic.py"), "mean_iou")
    miou_case("miou_dense", miou, 2, 7, 12, 12, 601, 7)
    miou_case("miou_absent_classes", miou, 2, 21, 16, 16, 602, 18, never_predicted=(20,))     # class 20: union == 0 -> skipped
    miou_case("miou_ties_150", miou, 1, 150, 24, 20, 603, 150, never_predicted=(7, 149), tie=True)
    ade = load_class(os.path.join(REF, "ade20k/ade_panoptic.py"), "InstanceContrastiveLoss")
    city = load_class(os.path.join(REF, "cityscapes/city_instance.py"), "InstanceContrastiveLoss")
    case("instloss_ade_small", ade, 2, 5, 8, 8, [0, 1, 2, 3, 7], None, 501)
    case("instloss_ade_sparse", ade, 3, 4, 10, 12, [0, 0, 0, 5, 9, 11, 4000], None, 512)      # one-pixel instances are likely
    case("instloss_city_ignore", city, 2, 6, 8, 16, [0, 255, 255, 26001, 26002, 24000], 255, 507)
    case("instloss_none", ade, 2, 3, 4, 4, [0], None, 504)                                      # no instance at all -> 0


if __name__ == "__main__":
    main()
