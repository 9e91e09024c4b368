"""CPU: host-side logic of the drop-in boundary -- module names/signatures, state_dict contract,
mask attribute semantics (8-b5), error behaviour, and that the product path refuses CPU tensors."""
import pytest
import torch

import maskunet_amd
from oracle import maskunet_oracle as O


def test_state_dict_contract_1head():
    m = maskunet_amd.UNet(3, 150)
    shapes = O.unet_state_shapes(3, 150)            # verified against the real reference in make_golden.py
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in shapes]
    for k, shp, _ in shapes:
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in m.parameters()) == 24920962


def test_state_dict_contract_3head():
    m = maskunet_amd.InstanceUNet(3, 19)
    shapes = O.unet_state_shapes(3, 19, True)
    assert list(m.state_dict().keys()) == [k for k, _, _ in shapes]
    assert sum(p.numel() for p in m.parameters()) == 24918858


def test_load_reference_style_checkpoint_with_module_prefix():
    params = O.make_params(O.unet_state_shapes(3, 7), 5)
    ckpt = {"module." + k: v for k, v in params.items()}             # nn.DataParallel checkpoint (ade_semantic.py:412)
    m = maskunet_amd.UNet(3, 7)
    m.load_state_dict({k.replace("module.", ""): v for k, v in ckpt.items()})     # ade_panoptic.py:434
    assert torch.equal(m.state_dict()["bottom2.conv_block.3.weight"], params["bottom2.conv_block.3.weight"])


def test_aliases_and_signatures():
    assert maskunet_amd.DoubleConv is maskunet_amd.ConvBlock and maskunet_amd.Down is maskunet_amd.DownSample
    assert maskunet_amd.Up is maskunet_amd.UpSample and maskunet_amd.MaskAttention is maskunet_amd.Mask2FormerAttention
    cb = maskunet_amd.ConvBlock(8, 16, mid_channels=4, residual=True)
    assert cb.conv_block[0].weight.shape == (4, 8, 3, 3) and cb.conv_block[3].weight.shape == (16, 4, 3, 3)
    assert maskunet_amd.ConvBlock(8, 16, 0).conv_block[0].out_channels == 16      # `if not mid_channels` (:196)
    d = maskunet_amd.DownSample(16, 32)
    assert d.emb_layer[1].weight.shape == (32, 256)
    u = maskunet_amd.UpSample(64, 16)
    assert u.conv[1].conv_block[0].weight.shape == (32, 64, 3, 3)                  # mid = in // 2 (:239)
    a = maskunet_amd.Mask2FormerAttention(64, 99)
    assert a.size == 99 and a.channels == 64
    assert maskunet_amd.OutConv(64, 5).final_layer[0].weight.shape == (5, 64, 1, 1)


def test_mask_attribute_semantics_on_host():
    a = maskunet_amd.Mask2FormerAttention(32, 32)
    assert a.mask is None                                            # ade_semantic.py:160
    keep = torch.tensor([[1, 0, 1, 1], [0, 0, 1, 0]], dtype=torch.uint8)
    add = torch.where(keep > 0, torch.zeros(()), torch.full((), -float("inf"))).unsqueeze(1).expand(-1, 4, -1)
    a.mask = add                                                     # the reference's own form (:180-181)
    assert tuple(a.mask.shape) == (2, 4, 4) and torch.equal(a.mask, add)
    assert a.mask.stride(1) == 0                                     # expand view: key-only mask
    assert a._keep.tolist() == keep.tolist() and a._kidx is None     # the index list is built lazily, on the device ...
    with pytest.raises(RuntimeError, match="GPU"):
        a._compact(torch.device("cpu"))                              # ... by mu_compact_keys: there is no host fallback (GPU test: check_compact_keys)
    a.mask = None
    assert a.mask is None and a._kidx is None


def test_cpu_tensors_are_refused_no_fallback():
    m = maskunet_amd.UNet(3, 5)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 128, 128))
    with pytest.raises(RuntimeError, match="GPU only"):
        maskunet_amd.ConvBlock(32, 32)(torch.zeros(1, 32, 8, 8))


def test_channel_mismatch_error_matches_reference():
    a = maskunet_amd.Mask2FormerAttention(32, 32)
    with pytest.raises(ValueError, match="Input channel size does not match initialized channel size."):
        a(torch.zeros(1, 16, 4, 4))


def test_product_never_imports_oracle():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "maskunet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} references the oracle"


def test_compute_dtype_switch():
    m = maskunet_amd.UNet(3, 5)
    assert m.compute_dtype == torch.float32
    m.set_compute_dtype(torch.float16)
    assert m.downsample1.maxpool_conv[1].compute_dtype == torch.float16
    with pytest.raises(TypeError):
        m.set_compute_dtype(torch.bfloat16)


def test_float32_matmul_precision_switch_maps_to_the_abi_dtype():
    from maskunet_amd import _lib
    assert maskunet_amd.get_float32_matmul_precision() == "highest"
    assert _lib.mdt(torch.float32) == _lib.MU_F32 and _lib.mdt(torch.float16) == _lib.MU_F16
    maskunet_amd.set_float32_matmul_precision("high")
    try:
        assert _lib.mdt(torch.float32) == _lib.MU_F32X          # the matrix entry points take the split-bf16 mode ...
        assert _lib.dt(torch.float32) == _lib.MU_F32            # ... every other kernel sees plain fp32 storage
        assert _lib.mdt(torch.float16) == _lib.MU_F16           # fp16 compute is unaffected
    finally:
        maskunet_amd.set_float32_matmul_precision("highest")
    with pytest.raises(ValueError):
        maskunet_amd.set_float32_matmul_precision("medium")


def test_attention_width_table():
    from maskunet_amd import ops
    assert [ops.attn_width(c) for c in (1, 24, 32, 33, 64, 96, 128, 200, 256)] == [32, 32, 32, 64, 64, 128, 128, 256, 256]
    # above the flash-style sweeps' widths: the generic path stores the next multiple of 32
    assert [ops.attn_width(c) for c in (257, 288, 300, 512, 1000)] == [288, 288, 320, 512, 1024]


def test_grad_link_drops_a_fill_from_another_backward_pass(monkeypatch):
    """ops.GradLink: a gradient parked by one backward pass (autograd graph task) is never handed to, or added into, another's."""
    from maskunet_amd import ops
    task = {"id": 7}
    monkeypatch.setattr(ops, "_graph_task_id", lambda: task["id"])
    link = ops.GradLink()
    a, b = torch.ones(3), torch.full((3,), 2.0)
    link.put(a)
    link.put(b)                                  # two fills within one pass add up
    assert torch.equal(link.take(), a + b) and link.take() is None
    link.put(a)                                  # filled by pass 7, whose taker was pruned ...
    task["id"] = 8
    link.put(b)                                  # ... pass 8 starts from its own fill, not a + b
    assert torch.equal(link.take(), b)
    link.put(a)
    task["id"] = 9
    assert link.take() is None                   # a taker of pass 9 never sees pass 8's leftover


def test_bench_spawns_a_child_launcher_for_bare_gpus(monkeypatch):
    """bench.py --gpus N with WORLD_SIZE unset: the ranks are started as a CHILD `python -m torch.distributed.run` (never an exec of a
    process that may have touched the GPU), with the same arguments, on 127.0.0.1; rank 0's JSON line and the exit code are relayed."""
    import subprocess
    import sys
    import bench
    seen = {}

    class R:
        returncode = 0
        stdout = 'noise\n{"metric": "128x128 images/sec (fwd+bwd)", "n_gpus": 4}\n'

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("MU_DIST_BACKEND", raising=False)
    # N ranks over RCCL need N visible devices: refused with a message BEFORE anything is spawned (device_count() does not touch the GPU)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4)
    assert "needs 4 visible GPUs" in str(e.value) and "cmd" not in seen
    monkeypatch.setenv("MU_DIST_BACKEND", "gloo")               # debug backend: ranks may share a device
    bench.need_devices(4)
    monkeypatch.delenv("MU_DIST_BACKEND")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    bench.spawn_ranks(4)
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    R.returncode = 3
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4)
    assert e.value.code == 3


def test_enc_link_is_valid_only_within_the_marking_backward_pass(monkeypatch):
    """ops.EncLink (fp32x): "the dy you are handed is one scaled fp16 operand, and this is its scale" holds only for the backward pass in
    which the BatchNorm behind the conv wrote it so; a leftover mark of another pass (partial backward) must read as plain."""
    from maskunet_amd import ops
    task = {"id": 3}
    monkeypatch.setattr(ops, "_graph_task_id", lambda: task["id"])
    link = ops.EncLink()
    assert link.take() is None
    sc = torch.tensor([8.0, 0.125])
    link.mark(sc)
    assert link.take() is sc and link.take() is None              # consumed once, hands the scale pair over
    link.mark(sc)
    task["id"] = 4
    assert link.take() is None                                    # another pass
    assert ops.enc_link(torch.zeros(4)) is None                   # exact-fp32 mode: no links at all


def test_bench_sweep_roofline_arithmetic():
    """bench.sweep_rooflines: executed FLOPs = 2 * products * N * Nk * C per image (Nk = kept keys), fp32x counts the 16-bit MFMA FLOPs
    ISSUED (two terms for S and P V, one for dP and the gradient products of the backward sweeps: round 6) against the dense 16-bit peak and
    carries the fp32-grade rate next to it; `kernels` holds all three sweeps."""
    import bench

    class Ev:
        def __init__(self, t=0.0):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t                        # milliseconds

    def events(ms):
        return [(tag, Ev(0.0), Ev(v)) for tag, v in ms.items() for _ in range(3)]

    B, hw, kept = 64, 128, 0.5
    N = hw * hw
    r = bench.sweep_rooflines("fp16", events({"fwd": 2.0, "dq": 2.5, "dkv": 3.5}), hw, B, kept, {"clock_mhz": 2000.0})
    exec_dkv = 8.0 * N * N * 64 * B * kept
    assert abs(r["flops_per_launch"] - exec_dkv) < 1 and r["launches_timed"] == 3 and r["ms_per_launch"] == 3.5
    assert abs(r["achieved"] - exec_dkv / 3.5e-3 / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-4
    assert abs(r["frac_at_measured_clock"] - r["achieved"] / (2500.0 * 2000.0 / 2400.0)) < 1e-4
    assert set(r["kernels"]) == {"fwd", "dq", "dkv"} and abs(r["kernels"]["fwd"]["flops_per_launch"] - exec_dkv / 2) < 1
    assert abs(r["algorithmic_achieved"] - 2 * r["achieved"]) < 0.02           # full key set = twice the kept half
    x = bench.sweep_rooflines("fp32x", events({"fwd": 4.0, "dq": 6.0, "dkv": 8.0}), hw, B, kept, None)
    assert abs(x["flops_per_launch"] - exec_dkv * 5 / 4) < 1                    # (2 + 1 + 1 + 1) terms over 4 products
    assert abs(x["kernels"]["fwd"]["flops_per_launch"] - (exec_dkv / 2) * 4 / 2) < 1 and abs(x["kernels"]["dq"]["flops_per_launch"] - (exec_dkv * 3 / 4) * 4 / 3) < 1
    assert abs(x["useful_tflops"] - exec_dkv / 8e-3 / 1e12) < 0.01 and x["peak"] == 2500.0 and x["frac_at_measured_clock"] is None
    assert bench.sweep_rooflines("fp16", events({"fwd": 2.0}), hw, B, kept, None) is None     # no dominant-kernel launch timed



def test_conv_stats_rows_mirror_the_kernel_dispatch():
    """mu_conv_stats_rows is host logic: it names the number of BatchNorm-statistics rows the 3x3 forward kernel of a shape writes
    (0 = that kernel has no statistics epilogue and the caller runs the separate sweep).  One case per kernel of the dispatch."""
    from maskunet_amd import _lib
    rows = _lib.load().mu_conv_stats_rows
    F32, F16, F32X = _lib.MU_F32, _lib.MU_F16, _lib.MU_F32X
    # fp16 ping-pong kernel (Cout % 128, H % 16, Cin % 64): four wave rows per 16 x 16 tile
    assert rows(64, 128, 128, 128, 128, 9, F16) == 64 * 8 * 8 * 4
    # fp16 weights-resident kernel (64 -> 64, >= 512 tiles): one row per wave of a group
    assert rows(64, 128, 128, 64, 64, 9, F16) == 64 * 8 * 8 * 4
    assert rows(64, 64, 64, 64, 64, 9, F16) == 64 * 4 * 4 * 4
    # ... below 512 tiles the halo-tile kernel serves the shape: 8 x 16 tiles, four wave rows (Cout = 64) / two (Cout % 128, H % 16 != 0)
    assert rows(2, 128, 128, 64, 64, 9, F16) == 2 * 16 * 8 * 4
    assert rows(3, 8, 48, 64, 128, 9, F16) == 3 * 1 * 3 * 2
    # fp32x: ping-pong kernel from 192 blocks on, the halo-tile kernel below
    assert rows(64, 32, 32, 128, 128, 9, F32X) == 64 * 2 * 2 * 4
    assert rows(2, 16, 16, 128, 128, 9, F32X) == 2 * 2 * 1 * 2
    assert rows(64, 128, 128, 64, 64, 9, F32X) == 64 * 16 * 8 * 4
    # no epilogue: exact fp32, 1x1 layers, widths that are not a multiple of 16, channel counts the tile kernels do not take
    assert rows(64, 128, 128, 64, 64, 9, F32) == 0
    assert rows(64, 128, 128, 64, 192, 1, F16) == 0
    assert rows(2, 16, 24, 64, 64, 9, F16) == 0
    assert rows(2, 16, 16, 32, 64, 9, F16) == 0


def test_data_parallel_buckets_end_with_a_small_tail_bucket():
    """The last bucket of maskunet_amd.DataParallel (the first layers' gradients: its all-reduce starts when the backward has ended and is
    exposed in full) is a small one; every trainable parameter sits in exactly one bucket, in reverse registration order."""
    import maskunet_amd
    m = maskunet_amd.UNet(3, 150)
    dp = maskunet_amd.DataParallel(m)
    flat = [p for b in dp.buckets for p in b.params]
    want = [p for p in reversed(list(m.parameters())) if p.requires_grad]
    assert [id(p) for p in flat] == [id(p) for p in want]
    assert len(dp.buckets) >= 3
    assert dp.buckets[-1].numel * 4 <= 2 * (1 << 20) < dp.buckets[-2].numel * 4
    assert any(p is m.initial_conv.conv_block[0].weight for p in dp.buckets[-1].params)
    one = maskunet_amd.DataParallel(maskunet_amd.UNet(3, 150), tail_mb=0.0)       # tail_mb = 0: the plain size-capped split
    assert len(one.buckets) == len(dp.buckets) - 1


def test_wgrad_workspace_covers_the_row_band_kernels():
    """mu_conv_wgrad_workspace_bytes (host-only) must reserve the slabs of the nine-taps-per-block kernels: up to 256 bands per channel-tile
    grid (256 / ((Cin/64)(Cout/64)) bands of [9][Cout][Cin] floats), for the fp16 layers those kernels serve."""
    from maskunet_amd import _lib
    lib = _lib.load()
    B = 64
    for H, Cin, Cout in [(128, 64, 64), (128, 128, 64), (64, 64, 64), (64, 64, 128), (64, 128, 64), (32, 128, 128),
                         (16, 256, 256), (16, 512, 512), (16, 256, 512)]:
        tiles = (Cin // 64) * (Cout // 64)
        rows = B * H
        want = max(1, min(256 // tiles, rows))
        rpb = -(-rows // want)
        if H == 16:
            rpb = (rpb + 1) // 2 * 2
        nb = -(-rows // rpb)
        need = nb * 9 * Cout * Cin * 4
        got = lib.mu_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 9)
        assert got >= need, (H, Cin, Cout, got, need)
