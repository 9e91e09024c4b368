#!/usr/bin/env python3
"""Benchmark of the MaskAttn-UNet forward+backward hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): 128x128 images/sec, forward + backward, whole job (all N GPUs).
Workload at N=1: configs[1] = "ADE20K semantic 128x128, batch=64, 1xMI355X fp16": UNet(3,150), B=64 per GPU,
fp16 storage / fp32 accumulate, synthetic images/labels/key-masks, random-init weights, train mode (batch-stat
BatchNorm, dropout 0.3), loss = mean pixel cross-entropy on the module's NCHW fp32 output (maskunet_amd.CrossEntropyLoss,
the reference's criterion as HIP kernels; --torch-loss = torch's F.cross_entropy) with a static loss scale
for the fp16 backward.  N>1: one process per GPU,
the same per-GPU batch (weak scaling), one bucketed RCCL all-reduce of the gradients per step overlapped with the
backward (maskunet_amd/dp.py), and -- like the reference under multi-GPU nn.DataParallel -- fresh attention key masks every step
(`config.mask_mode` = "resample": randint + mu_compact_keys on the device).

One JSON line on rank 0.
  * `roofline`: the dominant kernel (the dK/dV sweep of self_attention6), duration measured live with HIP events on the launch
    stream during the timed steps.  `achieved`/`frac` count the matrix products the kernel EXECUTES (masked keys are skipped
    exactly, so they are not work); the SURVEY 8-d4 full-key-set convention is carried as `algorithmic_*`.
  * `step_roofline`: the whole step and the forward alone against both rooflines (SURVEY 8-d3/d4 per-image figures: 136.2 GF /
    200.7 MB forward, ~450 GF / 602 MB forward+backward at 128x128, c_out=150; computed analytically for other shapes).
  * `cpu_baseline`: the CPU oracle (oracle/maskunet_oracle.py == restatement of the reference's PyTorch CPU path) timed on this
    host's cores on configs[0] (B=4, fp32, train mode) -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense, MI355X_MICROARCH.md.  fp32x = fp32 storage, every matrix product evaluated as several 16-bit MFMAs on (hi, lo) operand splits:
# three per product in the forward convolutions, two wherever one operand is a single fp16 term (attention: P, dS, the key in Q K^T, the value
# in dO V^T; 3x3 backward: the scaled dy -- round 6).  The fp32x
# `peak` is the dense 16-bit peak; `achieved` counts the MFMA FLOPs the kernel ISSUES (useful FLOPs x terms, TERMS below), so `frac` is
# a matrix-pipe utilisation like the fp16 line's; the useful rate is carried next to it.
PEAK_MFMA_TFLOPS = {"fp16": 2500.0, "fp32": 157.3, "fp32x": 2500.0}
PEAK_HBM_GBS = 8000.0
DOMINANT_KERNEL = "attn_bwd_dkv3_kernel"
# the three sweeps of one attention block: products per (query, key) pair [each 2*C FLOP] and, for fp32x, 16-bit MFMA terms per product
# (round 6: S = Q K^T with the key as ONE fp16 term against the two-term query in all three sweeps, P V two terms in the forward; in the
#  backward sweeps dP and the gradient products run as single fp16 MFMAs on exactly scaled operands)
SWEEPS = {"fwd": {"kernel": "attn_fwd2_kernel", "products": 2, "terms_fp32x": (2, 2)},                  # S = Q K^T, O = P V
          "dq": {"kernel": "attn_bwd_dq2_kernel", "products": 3, "terms_fp32x": (2, 1, 1)},             # S, dP = dO V^T, dQ = dS K
          "dkv": {"kernel": "attn_bwd_dkv3_kernel", "products": 4, "terms_fp32x": (2, 1, 1, 1)}}        # S, dP, dV = P^T dO, dK = dS^T Q
REF_CLOCK_MHZ = 1900.0       # convention for `clock.ms_per_step_at_ref_clock` (about what the pool's boxes hold in the MFMA probe)
# share of the step spent in MFMA-bound kernels (attention, conv, weight-grad: 23.5 of 29.5 ms, profiles/r04_kernel_time_split.txt) --
# measured on the configs[1] fp16 line ONLY, so `ms_per_step_at_ref_clock` is emitted for that configuration only
MFMA_SHARE = 0.8
# PMC traffic of the dominant kernel per workload shape: profiles/r0N_dkv_traffic_b{batch}_c{c_out}_hw{hw}_{dtype}[_3head].json (load_traffic)
DATASET_BY_COUT = {150: "ADE20K-semantic", 151: "ADE20K-semantic", 133: "COCO-panoptic", 19: "Cityscapes", 81: "COCO-instance"}


def synth(B, c_out, hw, seed, device, ignore_frac=0.0):
    """SURVEY 8-d2 synthetic batch from numpy default_rng(seed) (42 = the reference's nominal seed, ade_semantic.py:24; + rank for
    the other ranks): images uniform [0,1) like ToTensor(), int64 labels (a fraction set to 255 for the Cityscapes shape), one
    Bernoulli(0.5) key keep-mask per attention block."""
    import numpy as np
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.random((B, 3, hw, hw), dtype=np.float32))
    lab = rng.integers(0, c_out, (B, hw, hw))
    if ignore_frac > 0:
        lab[rng.random((B, hw, hw)) < ignore_frac] = 255
    y = torch.from_numpy(lab.astype(np.int64))
    ns = [(hw // 2) ** 2, (hw // 4) ** 2, (hw // 8) ** 2, (hw // 4) ** 2, (hw // 2) ** 2, hw ** 2]
    keeps = [torch.from_numpy(rng.integers(0, 2, (B, n)).astype(np.uint8)) for n in ns]
    return x.to(device), y.to(device), [k.to(device) for k in keeps]


def algorithmic_counts(c_out, hw, three_head=False, embed_dim=16):
    """Per-image algorithmic FLOPs and HBM elements of the forward (SURVEY 8-d4 convention: every conv reads its input and writes
    its output once, every BatchNorm-apply reads + writes once, attention moves 8*N*C elements and NO N x N tensor, the final
    LayerNorm reads + writes + two affine tensors).  Reproduces the survey's 136.2 GF and 100.4 M elements at 128x128, c_out=150."""
    convs, bns = [], []

    def block(cin, cout, h, mid=None):
        mid = mid or cout
        convs.extend([(cin, mid, h, 9), (mid, cout, h, 9)])
        bns.extend([(mid, h), (cout, h)])

    block(3, 64, hw)
    for cin, cout, h in ((64, 128, hw // 2), (128, 256, hw // 4), (256, 256, hw // 8)):
        block(cin, cin, h)
        block(cin, cout, h)
        bns.append((cout, h))
    for cin, cout in ((256, 512), (512, 512), (512, 256)):
        block(cin, cout, hw // 8)
    for cin, cout, h in ((512, 128, hw // 4), (256, 64, hw // 2), (128, 64, hw)):
        block(cin, cin, h)
        block(cin, cout, h, cin // 2)
        bns.append((cout, h))
    heads = [(64, c_out, 1)] + ([(64, embed_dim, 1), (c_out, 32, 9), (32, 1, 1)] if three_head else [])
    attn = [(128, hw // 2), (256, hw // 4), (256, hw // 8), (128, hw // 4), (64, hw // 2), (64, hw)]
    conv_fl = sum(2.0 * ci * co * h * h * t for ci, co, h, t in convs) + sum(2.0 * ci * co * t * hw * hw for ci, co, t in heads)
    proj_fl = sum(6.0 * c * c * h * h for c, h in attn)
    attn_fl = sum(4.0 * float(h * h) ** 2 * c for c, h in attn)
    elems = (sum((ci + co) * h * h for ci, co, h, _ in convs) + sum((ci + co) * hw * hw for ci, co, _ in heads)
             + sum(2 * c * h * h for c, h in bns) + 2 * sum(co for _, co, _ in heads[:3 if three_head else 1]) * hw * hw
             + sum(8 * c * h * h for c, h in attn) + 4 * 64 * hw * hw)
    return {"conv_proj_flops": conv_fl + proj_fl, "attn_flops": attn_fl, "fwd_flops": conv_fl + proj_fl + attn_fl, "fwd_elems": elems,
            # backward = 2x the conv/projection products (data + weight gradient) and 2.5x attention (recompute), traffic ~2x forward
            "step_flops": 3.0 * (conv_fl + proj_fl) + 3.5 * attn_fl, "step_elems": 3 * elems}


def cpu_baseline(c_out, hw, B=4, iters=3, budget_s=150.0):
    """Oracle (kind 'port') fwd+bwd on the host cores: configs[0] = "ADE20K semantic 128x128, batch=4, PyTorch CPU reference"
    (fp32, train mode, B=4, warm-up 1 + median of >= 3 iterations, inputs from default_rng(42))."""
    from oracle import maskunet_oracle as O
    # torch's intra-op scaling collapses with hundreds of threads on this path (256 threads on the GPU box's
    # host: 232 s/iteration); 32 threads is the fastest setting found, `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    shapes = O.unet_state_shapes(3, c_out, False, hw=hw)
    p = O.make_params(shapes, 1)
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    x, labels, keeps = synth(B, c_out, hw, 42, "cpu")
    times = []
    t_all = time.time()
    for it in range(iters + 1):
        if it >= 2 and time.time() - t_all + times[-1] > budget_s:      # bounded sample: stop early on a slow host (>= 1 timed iteration)
            break
        t0 = time.time()
        out = O.unet_forward(p, x, keeps, training=True)
        loss = O.pixel_cross_entropy(out, labels)
        loss.backward()
        for v in p.values():
            v.grad = None
        times.append(time.time() - t0)
    steady = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(B / steady, 4), "unit": "images/sec", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"configs[0]: oracle fwd+bwd, fp32, train mode, B={B}, c_out={c_out}, {hw}x{hw}, default_rng(42) inputs, "
                      f"median of {len(times) - 1} after 1 warm-up ({steady:.2f} s/iteration)"}


class SmiSampler:
    """Board power rails of the benched GPU from sysfs hwmon, sampled by a helper thread over the timed region -- context only: the
    clock figure of the line is `clock_mhz` from in-kernel stamps (MI355X_MICROARCH.md: sysfs sclk reads up to ~10 % off an MFMA-dense
    loop's clock, and on this pool 95-2300 MHz for one load).  The sysfs node is found through the benched device's PCI address
    (never by list position: `card10` sorts before `card2`, and render-only nodes shift the index); EVERY power*_average / power*_input
    rail is reported with its label, because a single rail is not the board (round 4 reported one rail: 254 W for a saturated part).
    Rank 0 only.  All fields None / empty where the box does not expose the files to an ordinary user."""

    def __init__(self, dev, period=0.05):
        import glob
        import threading
        self.dev = self.pci = None
        self.rails = []                                           # (label, path)
        try:
            pr = torch.cuda.get_device_properties(dev)
            dom, bus, dv = getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
            if bus is not None and dv is not None:
                self.pci = f"{dom or 0:04x}:{bus:02x}:{dv:02x}.0"
        except Exception:
            pass
        if self.pci:
            for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
                try:
                    if os.path.basename(os.path.realpath(d)).lower() == self.pci:
                        self.dev = d
                        break
                except OSError:
                    pass
        if self.dev:
            for f in sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*", "power*_average")) +
                            glob.glob(os.path.join(self.dev, "hwmon", "hwmon*", "power*_input"))):
                lab = f.rsplit("_", 1)[0] + "_label"
                try:
                    name = open(lab).read().strip()
                except OSError:
                    name = ""
                self.rails.append((f"{os.path.basename(f)}{' (' + name + ')' if name else ''}", f))
        self.period = period
        self.samples = {lab: [] for lab, _ in self.rails}
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            for lab, f in self.rails:
                try:
                    self.samples[lab].append(float(open(f).read()) * 1e-6)
                except (OSError, ValueError):
                    pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.rails:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.rails:
            self._th.join(timeout=1.0)

    def summary(self):
        return {"pci": self.pci, "power_rails_w": {lab: (round(sum(v) / len(v), 1) if v else None) for lab, v in self.samples.items()},
                "samples": max([len(v) for v in self.samples.values()] or [0]),
                "note": "hwmon rails of the benched device, mean over the timed region; context only -- not used for any normalisation"}


def clock_probe(dev, seconds=0.5, iters=16384):
    """Clock the chip holds under a dense fp16 MFMA load, right after the timed region (the chip is warm): mu_clock_probe launched
    back to back for `seconds`, median over the CUs of d(s_memtime) / d(s_memrealtime) of the last launches (the guide's check 6).
    Also the matrix rate of that register-only loop -- the most this box gives a 16x16x32 fp16 MFMA stream on random operands."""
    from maskunet_amd import _lib
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    stamps = torch.zeros(ncu, 2, dtype=torch.int64, device=dev)
    sink = torch.empty(ncu * 256, dtype=torch.float32, device=dev)
    keep = []
    t_end = time.perf_counter() + seconds
    n = 0
    while time.perf_counter() < t_end or n < 8:
        _lib.call("mu_clock_probe", _lib.ptr(stamps), _lib.ptr(sink), ncu, iters, _lib.stream())
        n += 1
        if n % 4 == 0:
            keep.append(stamps.clone())
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    st = torch.stack(keep[-4:]).double().cpu()              # [launch][cu][2]
    clk = (st[..., 0] / st[..., 1].clamp(min=1) * 100.0).flatten().median().item()
    t_loop = st[..., 1].flatten().median().item() * 1e-8    # seconds per launch (100 MHz ticks)
    flops = ncu * 4 * iters * 16 * 16384.0                  # 4 waves per block, 16 MFMAs per iteration, 2*16*16*32 FLOP each
    return {"clock_mhz": round(clk, 1), "probe_tflops": round(flops / t_loop / 1e12, 1), "launches": n,
            "method": "mu_clock_probe: register-only v_mfma_f32_16x16x32_f16 loop on random operands, one 4-wave block per CU, "
                      "100 MHz * d(s_memtime)/d(s_memrealtime), median over CUs of the last 4 sampled launches"}


def parity_gate(args, dtype, dtype_name=None):
    """The B = 2 whole-model golden (generated from the reference's own classes, tests/golden/make_golden.py) run once through the
    timed dtype BEFORE the timed region: which parity gate the benched path passes, in the bench line itself.  Checker only."""
    name = "unet3_c19_b2_train" if args.three_head else {150: "unet1_c150_b2_train", 133: "unet1_c133_b2_train"}.get(args.c_out)
    if name is None or args.hw != 128 or (args.three_head and args.c_out != 19):
        return {"fixture": None, "note": "no reference-generated whole-model golden for this shape (tests/golden/ holds c_out 150 / 133 / 3-head 19 at 128x128)"}
    try:
        from tests import _gpu_checks as G
        res = G.check_unet_golden(name, dtype)
    except Exception as e:                                   # the bench line must not depend on the checker
        return {"fixture": f"tests/golden/{name}.npz", "error": repr(e)[:200]}
    obs = {"out": max(e for n, e, t in res if " out" in n and "slice" in n),
           "loss": max(e for n, e, t in res if n.endswith(" loss")),
           "grad_maxnorm_worst_param": max(e for n, e, t in res if "worst grad" in n),
           "running_stats": max([e for n, e, t in res if "running_" in n] or [0.0])}
    gates = {"out": G.TOL[dtype], "loss": G.TOL[dtype], "grad_maxnorm_worst_param": max(t for n, e, t in res if "worst grad" in n),
             "grad_1_minus_cos_per_param (tests/test_gpu_modules.py, vs the live oracle)": 1e-4 if dtype == torch.float32 else 2e-2}
    return {"fixture": f"tests/golden/{name}.npz (outputs / loss / gradients of the reference's own UNet, B=2, train mode)",
            "dtype": {"fp16": "f16", "fp32": "f32", "fp32x": "f32 storage, split 16-bit products"}[dtype_name or args.dtype], "gate": gates,
            "observed": {k: float(f"{v:.3g}") for k, v in obs.items()}, "passed": all(e <= t for _, e, t in res),
            "north_star_gate": "1e-3 (fp32 outputs): met by --dtype fp32 (observed ~1e-5) and --dtype fp32x, not by the fp16-storage path timed here"
                               if dtype == torch.float16 else "1e-3 (fp32 outputs)"}


def attention_probe(hw):
    """_lib.PROBE predicate: HIP events around the three sweeps of self_attention6 (N = hw * hw, C = 64), tagged fwd / dq / dkv.
    mu_attn_fwd(qkv, x, kidx, kcnt, gamma, beta, out, oattn, lse2, mean, rstd, B, N, C, ...): args[12] = N;
    mu_attn_bwd_phases(..., B, N, C, nkmax, ws, ws_bytes, dtype, phases, stream): args[16] = N, args[22] = phases."""
    N6 = hw * hw

    def pred(name, a):
        if name == "mu_attn_fwd" and a[12] == N6:
            return "fwd"
        if name == "mu_attn_bwd_phases" and a[16] == N6:
            return {2: "dq", 4: "dkv"}.get(a[22] & 7, False)
        return False
    return pred


def sweep_rooflines(dtype_name, events, hw, batch, kept, clk, traffic=None, traffic_source=None):
    """`roofline` (the dominant kernel: the dK/dV sweep of self_attention6) and `roofline.kernels` (all three sweeps of that block, each
    with its own executed-FLOP fraction, so that the WEAKEST sweep is visible, not only the longest) from the HIP-event durations."""
    N6, C6 = hw * hw, 64
    peak = PEAK_MFMA_TFLOPS[dtype_name]
    esz = 2 if dtype_name == "fp16" else 4
    out = {}
    for tag, sw in SWEEPS.items():
        durs = [e0.elapsed_time(e1) * 1e-3 for t, e0, e1 in events if t == tag]
        if not durs:
            continue
        tk = sum(durs) / len(durs)
        useful_full = 2.0 * sw["products"] * N6 * N6 * C6 * batch            # the full key set
        useful = useful_full * kept                                          # executed: masked keys are skipped exactly
        issued = useful * (sum(sw["terms_fp32x"]) / sw["products"] if dtype_name == "fp32x" else 1.0)
        ach = issued / tk / 1e12
        out[tag] = {"kernel": sw["kernel"], "ms_per_launch": round(tk * 1e3, 3), "launches_timed": len(durs),
                    "achieved": round(ach, 2), "frac": round(ach / peak, 4),
                    "frac_at_measured_clock": round(ach / (peak * clk["clock_mhz"] / 2400.0), 4) if clk else None,
                    "useful_tflops": round(useful / tk / 1e12, 2), "flops_per_launch": issued,
                    "algorithmic_achieved": round(useful_full / tk / 1e12, 2), "algorithmic_frac": round(useful_full / tk / 1e12 / peak, 4)}
    d = out.get("dkv")
    if d is None:
        return None
    note = ("achieved/frac = matrix FLOPs the kernel EXECUTES (8*N*Nk*C per image, Nk = kept keys; masked keys are skipped exactly) / "
            "HIP-event time; algorithmic_* FLOPs = SURVEY 8-d4's full-key-set count 8*N*N*C; algorithmic_bytes = Q, dO, K, V read + dK, dV "
            "written once (6*N*C elements per image)")
    if dtype_name == "fp32x":
        note += ("; fp32x: achieved/frac count the 16-bit MFMA FLOPs ISSUED (useful x terms: round 6 -- two for S and P V, one for dP and the gradient products of the backward sweeps) against the "
                 "dense 16-bit peak, `useful_tflops` the fp32-grade products delivered")
    return {"bound": "mfma", "achieved": d["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": d["frac"],
            # the same FLOPs against the matrix peak at the clock this chip holds under a dense MFMA load
            # (peak is quoted at 2400 MHz; `clock`): what is left is the kernel's own issue efficiency
            "frac_at_measured_clock": d["frac_at_measured_clock"], "traffic": traffic, "traffic_source": traffic_source,
            "kernel": f"{DOMINANT_KERNEL} (self_attention6 dK/dV sweep, N={N6}, C=64)",
            "ms_per_launch": d["ms_per_launch"], "launches_timed": d["launches_timed"], "kept_keys": round(kept, 4),
            "flops_per_launch": d["flops_per_launch"], "useful_tflops": d["useful_tflops"],
            "algorithmic_achieved": d["algorithmic_achieved"], "algorithmic_frac": d["algorithmic_frac"],
            # the dK/dV sweep's own operands, each once: reads Q, dO, K, V; writes dK, dV ([N, C] each; the 8*N*C of
            # SURVEY 8-d4 is the whole attention block) -- the unit `traffic` is measured in
            "algorithmic_bytes_per_launch": 6.0 * N6 * C6 * esz * batch,
            "kernels": out, "note": note}


def load_traffic(args, dtype_name):
    """HBM-side bytes of the dominant launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md): counters cannot be
    read inside this process, so the figure comes from the committed rocprofv3 --pmc run of the same command -- one file per workload
    shape, named in `traffic_source`; null where no counter pass was taken for the shape."""
    tkey = f"b{args.batch}_c{args.c_out}_hw{args.hw}_{dtype_name}" + ("_3head" if args.three_head else "")
    for rnd in ("r06", "r05", "r04"):
        tfile = f"{rnd}_dkv_traffic_{tkey}.json"
        tpath = os.path.join(ROOT, "profiles", tfile)
        if os.path.exists(tpath):
            rec = json.load(open(tpath))
            return rec.get("hbm_bytes_per_launch"), (f"profiles/{tfile} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command on another "
                                                     f"box, not this run" + (f"; session {rec['session']}" if rec.get("session") else "") + ")")
    return None, None


def parity_grade_line(args, dev, steps, warmup, clk):
    """The SAME workload in the fp32x mode (fp32 storage, split 16-bit matrix products: the fastest path under north_star's 1e-3 gate) as
    a first-class measurement of its own: own model, own parity-gate run, `warmup` untimed + `steps` timed steps bracketed by
    synchronize(), own `roofline` block from HIP events on the launch stream -- behind the timed region of the line's own dtype.
    It is never `value`: the line benches the dtype named in `dtype` (BASELINE.json configs[1] names fp16)."""
    import maskunet_amd
    from maskunet_amd import _lib
    maskunet_amd.set_float32_matmul_precision("high")
    try:
        gate = parity_gate(args, torch.float32, "fp32x")
        torch.manual_seed(1234)
        model = (maskunet_amd.InstanceUNet(3, args.c_out, 16, hw=args.hw) if args.three_head else maskunet_amd.UNet(3, args.c_out, hw=args.hw)).to(dev)
        model.set_compute_dtype(torch.float32).train()
        x, labels, keeps = synth(args.batch, args.c_out, args.hw, 42, dev, ignore_frac=0.1 if args.three_head else 0.0)
        model.set_keep_masks(keeps)
        crit = maskunet_amd.CrossEntropyLoss(ignore_index=255) if args.three_head else maskunet_amd.CrossEntropyLoss()

        def one():
            out = model(x)
            loss = crit(out[0] if args.three_head else out, labels)
            loss.backward()
            model.zero_grad(set_to_none=True)
            return loss
        for _ in range(warmup):
            one()
        _lib.PROBE = {"pred": attention_probe(args.hw), "events": []}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / steps
        events = _lib.PROBE["events"]
        _lib.PROBE = None
        kept = float(model.self_attention6._keep.float().mean().item())
        traffic, src = load_traffic(args, "fp32x")
        return {"dtype": "f32 storage, split 16-bit matrix products, f32 accumulate (maskunet_amd.set_float32_matmul_precision('high'))",
                "value": round(args.batch / dt_, 1), "unit": "images/sec", "ms_per_step": round(1e3 * dt_, 3), "steps": steps, "warmup": warmup,
                "finite": bool(torch.isfinite(loss.detach()).all()),
                "roofline": sweep_rooflines("fp32x", events, args.hw, args.batch, kept, clk, traffic, src),
                "parity_gate": {"gate_out": 1e-3, "observed": gate.get("observed"), "passed": gate.get("passed"), "fixture": gate.get("fixture")},
                "note": "the same workload and batch in the parity-grade precision mode, timed behind the line's own timed region; the line's "
                        "`value` is the dtype named in `dtype`"}
    except Exception as e:                                     # never break the bench line
        return {"error": repr(e)[:300]}
    finally:
        _lib.PROBE = None
        maskunet_amd.set_float32_matmul_precision("highest")


def need_devices(n):
    """One process per GPU over RCCL needs n visible devices: refuse with a message instead of n - ndev ranks dying in hipSetDevice /
    ncclCommInitRank (torch.cuda.device_count() does not initialise the GPU).  MU_DIST_BACKEND=gloo (debug) lets ranks share a device."""
    ndev = torch.cuda.device_count()
    if os.environ.get("MU_DIST_BACKEND", "nccl") == "nccl" and n > ndev:
        raise SystemExit(f"bench.py: --gpus {n} over the nccl (RCCL) backend needs {n} visible GPUs, this node shows {ndev} "
                         f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); MU_DIST_BACKEND=gloo shares one device between ranks for debugging")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process and hand back its output and exit code."""
    import socket
    import subprocess
    need_devices(n)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this image)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln                                            # rank 0's JSON line
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32", "fp32x"],
                    help="fp16: fp16 storage / fp32 accumulate (the configs[1] line); fp32: exact-fp32 MFMA (the 1e-3 parity path); fp32x: fp32 "
                         "storage, matrix products as two / three 16-bit MFMAs on (hi, lo) splits (maskunet_amd.set_float32_matmul_precision('high'))")
    ap.add_argument("--c-out", type=int, default=150)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--loss-scale", type=float, default=1024.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the B=2 golden check that fills `parity_gate`")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the MFMA-loop clock probe after the timed region")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip `host_inclusive` (the step with the batch copied from pinned host "
                    "memory every iteration, timed behind the timed region; N = 1 only)")
    ap.add_argument("--no-fp32x-line", action="store_true", help="skip `parity_grade_path` (the same workload timed in the fp32x "
                    "parity-grade mode behind the timed region; N = 1 and --dtype fp16 only)")
    ap.add_argument("--pg-steps", type=int, default=0, help="timed steps of `parity_grade_path` (default: max(10, --steps // 2); 2 warm-up steps)")
    ap.add_argument("--three-head", action="store_true")
    ap.add_argument("--fused-loss", action="store_true",
                    help="SURVEY 8-f1 path: fused NHWC cross-entropy on the internal logits instead of module output + torch CE")
    ap.add_argument("--torch-loss", action="store_true", help="torch's F.cross_entropy on the module output instead of maskunet_amd.CrossEntropyLoss")
    ap.add_argument("--graph", action="store_true", help="replay the step (forward + criterion + backward) as one HIP graph "
                    "(maskunet_amd.GraphedStep; single GPU)")
    ap.add_argument("--optimizer", action="store_true", help="also run the fused AdamW step (8-f2) inside the timed step")
    ap.add_argument("--mask-mode", default="auto", choices=["auto", "fixed", "resample"],
                    help="attention key masks: 'fixed' = drawn once and cached (the reference on ONE GPU, ade_semantic.py:177), 'resample' = "
                         "redrawn + re-compacted on the device every step (what the reference does under multi-GPU nn.DataParallel, whose "
                         "replicas are discarded each step, :373); auto = fixed at N = 1, resample at N > 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (torch.distributed.run as a CHILD process, before anything in this
        # process has touched the GPU -- never an exec of a process that has) and relay rank 0's JSON line and the exit code
        return spawn_ranks(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MU_DIST_BACKEND", "nccl")      # "gloo" lets two ranks share one GPU in a debug run
    # MU_BENCH_FORCE_DP=1: run the multi-rank code path (process group, DataParallel hooks and buckets, barrier, max-over-ranks) in a
    # group of ONE rank -- what a single-GPU box can exercise of the N > 1 invocation over RCCL (tests/test_gpu_dp.py)
    multi = world > 1 or os.environ.get("MU_BENCH_FORCE_DP") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            need_devices(world)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            local = local % max(ndev, 1)
    if world != args.gpus:
        # a launcher that set WORLD_SIZE to something else than --gpus: refuse instead of benching a different job than asked for
        # (WORLD_SIZE unset and --gpus N > 1 never gets here: spawn_ranks above)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: they must agree "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ..., or no launcher at all)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import maskunet_amd
    from maskunet_amd import _lib
    dtype = torch.float16 if args.dtype == "fp16" else torch.float32
    maskunet_amd.set_float32_matmul_precision("high" if args.dtype == "fp32x" else "highest")
    torch.manual_seed(1234)
    model = (maskunet_amd.InstanceUNet(3, args.c_out, 16, hw=args.hw) if args.three_head
             else maskunet_amd.UNet(3, args.c_out, hw=args.hw)).to(dev)
    model.set_compute_dtype(dtype).train()
    x, labels, keeps = synth(args.batch, args.c_out, args.hw, 42 + rank, dev, ignore_frac=0.1 if args.three_head else 0.0)
    model.set_keep_masks(keeps)
    # N > 1 times the step the reference runs under nn.DataParallel: every replica forward draws a fresh mask (SURVEY 0 trap #2,
    # 8-e1), i.e. per step six randint draws + six key compactions (mu_compact_keys) on the device stream, no host sync
    mask_mode = args.mask_mode if args.mask_mode != "auto" else ("resample" if multi else "fixed")
    model.set_mask_mode(mask_mode)
    net = maskunet_amd.DataParallel(model, force_sync=True) if multi else model
    scale = args.loss_scale if dtype == torch.float16 else 1.0

    inst_loss = inst_labels = None
    if args.three_head:              # synthetic instance ids: 16x16 blocks of ids 0..20 (0 = background), device-side loss (row f1b)
        g = torch.Generator().manual_seed(7 + rank)
        blocks = torch.randint(0, 21, (args.batch, args.hw // 16, args.hw // 16), generator=g)
        inst_labels = blocks.repeat_interleave(16, 1).repeat_interleave(16, 2).to(dev)
        inst_loss = maskunet_amd.InstanceContrastiveLoss(margin=1.0, ignore_index=255)
    criterion = maskunet_amd.CrossEntropyLoss(ignore_index=255) if args.three_head else maskunet_amd.CrossEntropyLoss()
    # ^ the reference's nn.CrossEntropyLoss() on the module output (ade_semantic.py:377,399)
    opt = maskunet_amd.FusedAdamW(model.parameters(), lr=5e-5, weight_decay=1e-1) if args.optimizer else None

    graphed = None
    if args.graph:
        if args.fused_loss or args.three_head or args.torch_loss:
            raise SystemExit("--graph: maskunet_amd.CrossEntropyLoss, 1-head model only")
        # mask_mode "resample" (the reference's N > 1 semantics): the six randint draws + key compactions are captured too -- torch's CUDA
        # generator hands a captured graph a fresh Philox offset on every replay, mu_compact_keys is stream-ordered (round 6)
        # N > 1: each replica replays its own graph and the bucketed all-reduce follows the replay (a replay runs no autograd hooks,
        # so the exchange does not overlap the backward) -- opt-in, for small per-GPU batches where the host enqueue is the bound
        graphed = maskunet_amd.GraphedStep(net, criterion, x, labels, loss_scale=scale)

    def step():
        nonlocal graphed, fwd_events
        if graphed is not None:
            loss = graphed(x, labels)
            if opt is not None:
                opt.step(grad_scale=scale)
            return loss
        if args.fused_loss and not args.three_head:
            if multi:
                net._arm()
            loss = maskunet_amd.pixel_cross_entropy_nhwc(model.logits_nhwc(x), labels, args.c_out, grad_scale=scale)
            loss.backward()
        else:
            if fwd_events is not None:
                ef0, ef1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ef0.record()
            out = net(x)
            if fwd_events is not None:
                ef1.record()
                fwd_events.append((ef0, ef1))
            sem = out[0] if args.three_head else out
            loss = F.cross_entropy(sem, labels) if args.torch_loss else criterion(sem, labels)
            if args.three_head:          # city_instance.py:372-377: seg_loss + LAMBDA_IE * InstanceContrastiveLoss(embeddings, inst_labels)
                loss = loss + 0.1 * inst_loss(out[2], inst_labels)
            (loss * scale).backward()
        if multi:
            net.finish_gradient_sync()
        if opt is not None:
            opt.step(grad_scale=scale)
        model.zero_grad(set_to_none=True)
        return loss

    # which parity gate the timed dtype passes: the reference-generated B = 2 golden, once, outside the timed region
    gate = None
    if rank == 0 and not args.no_parity_gate:
        gate = parity_gate(args, dtype)
        torch.cuda.synchronize()

    fwd_events = None
    for _ in range(args.warmup):
        step()
    fwd_events = []          # forward-only time: events around net(x) on the current stream (every launch of the path is on it)
    # HIP-event probe on the three sweeps of self_attention6 (N = hw*hw, C = 64): the dK/dV sweep (attn_bwd_dkv3_kernel) is the
    # dominant kernel of the step, the forward and dQ sweeps are reported next to it (`roofline.kernels`)
    N6 = args.hw * args.hw
    _lib.PROBE = {"pred": attention_probe(args.hw), "events": []}
    sampler = SmiSampler(dev) if rank == 0 else None            # hwmon power rails over the timed region (rank 0's GPU only)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__()
    probe = _lib.PROBE
    _lib.PROBE = None
    clk = None
    if rank == 0 and not args.no_clock_probe:
        clk = clock_probe(dev)                                   # right behind the timed steps: the chip is still warm
    _lib.PROBE = probe
    graphed_mode = graphed is not None
    if graphed is not None:                     # a replayed graph makes no Python calls to probe: time the kernel in two eager steps
        graphed = None                          # after the timed region (on every rank: eager steps all-reduce)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    _lib.PROBE = None
    # the same step with the batch handed over as HOST buffers (what the reference's loop does: `inputs.to(device)`, `labels.to(device)`
    # per iteration, ade_semantic.py:394-395): pinned fp32 NCHW images + int64 labels copied on the launch stream inside every step.
    # A side figure (`host_inclusive`), never `value`.
    host_incl = None
    if world == 1 and not args.no_host_inclusive and graphed_mode is False and not args.fused_loss:
        xh, lh = x.detach().cpu().pin_memory(), labels.detach().cpu().pin_memory()
        for _ in range(2):
            x.copy_(xh, non_blocking=True); labels.copy_(lh, non_blocking=True)
            step()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(args.steps):
            x.copy_(xh, non_blocking=True); labels.copy_(lh, non_blocking=True)
            step()
        torch.cuda.synchronize()
        th = (time.perf_counter() - th) / args.steps
        host_incl = {"value": round(args.batch / th, 1), "unit": "images/sec", "ms_per_step": round(1e3 * th, 3), "steps": args.steps,
                     "bytes_per_step": int(xh.numel() * xh.element_size() + lh.numel() * lh.element_size()),
                     "note": "the timed step preceded by the copy of the batch from pinned host memory (fp32 NCHW images + int64 labels, "
                             "as the reference's loop hands them over) on the launch stream; `value` of the line has the inputs resident in HBM"}
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not torch.isfinite(loss.detach()).all():
        raise RuntimeError("non-finite loss in the timed region")

    if rank == 0:
        imgs = args.batch * world * args.steps
        # fraction of keys the kernel really multiplies (the rest are skipped, not computed): of the mask the last step ran with
        kept = float(model.self_attention6._keep.float().mean().item())
        peak = PEAK_MFMA_TFLOPS[args.dtype]
        traffic, traffic_source = load_traffic(args, args.dtype)
        roof = sweep_rooflines(args.dtype, probe["events"], args.hw, args.batch, kept, clk, traffic, traffic_source)
        # whole step and forward alone against both rooflines (per-image algorithmic figures)
        cnt = algorithmic_counts(args.c_out, args.hw, args.three_head)
        esz = 2 if dtype == torch.float16 else 4
        t_step = elapsed / args.steps
        headline_cfg = (args.dtype == "fp16" and args.batch == 64 and args.c_out == 150 and args.hw == 128 and not args.three_head
                        and not args.optimizer and not args.graph and not args.fused_loss and not args.torch_loss and world == 1)
        nimg = args.batch                      # per GPU: the step time is per GPU too (weak scaling)
        attn_exec = cnt["attn_flops"] * 0.5    # Bernoulli(0.5) key masks: half the keys are skipped
        step = {"flops_per_img": round(cnt["step_flops"] / 1e9, 1), "flops_per_img_unit": "GFLOP (3x conv/proj + 3.5x attention, full key set)",
                "bytes_per_img": round(cnt["step_elems"] * esz / 1e6, 1), "bytes_per_img_unit": "MB",
                "mfma_frac": round(cnt["step_flops"] * nimg / t_step / (peak * 1e12), 4),
                "mfma_frac_executed": round((3.0 * cnt["conv_proj_flops"] + 3.5 * attn_exec) * nimg / t_step / (peak * 1e12), 4),
                "hbm_frac": round(cnt["step_elems"] * esz * nimg / t_step / (PEAK_HBM_GBS * 1e9), 4),
                "bound": "mfma (ridge ~300 FLOP/B, the step has ~750 FLOP/B)"}
        if fwd_events:
            t_fwd = sum(a.elapsed_time(b) for a, b in fwd_events) * 1e-3 / len(fwd_events)
            step.update({"fwd_ms": round(t_fwd * 1e3, 3), "fwd_flops_per_img": round(cnt["fwd_flops"] / 1e9, 1),
                         "fwd_bytes_per_img": round(cnt["fwd_elems"] * esz / 1e6, 1),
                         "fwd_hbm_frac": round(cnt["fwd_elems"] * esz * nimg / t_fwd / (PEAK_HBM_GBS * 1e9), 4),
                         "fwd_mfma_frac": round(cnt["fwd_flops"] * nimg / t_fwd / (peak * 1e12), 4),
                         "fwd_mfma_frac_executed": round((cnt["conv_proj_flops"] + attn_exec) * nimg / t_fwd / (peak * 1e12), 4),
                         "fwd_note": "north_star target '>= 60 % of the fp16 HBM roofline for the fused forward': the fused forward is "
                                     "MFMA-bound (680 FLOP/B), so its HBM fraction is set by its matrix time"})
        rec = {
            "metric": "128x128 images/sec (fwd+bwd)", "value": round(imgs / elapsed, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp16": "f16", "fp32": "f32", "fp32x": "f32 storage, split 16-bit matrix products (f32 accumulate)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": f"{DATASET_BY_COUT.get(args.c_out, 'custom')} shape {args.hw}x{args.hw}, c_out={args.c_out}, batch={args.batch}/GPU, "
                                   f"{'3-head' if args.three_head else '1-head'} MaskAttn-UNet fwd+bwd, train mode",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "loss_scale": scale, "mask_mode": mask_mode,
                       "loss": "fused NHWC CE kernel" if args.fused_loss else ("torch CE on module output" if args.torch_loss else
                                                                                "maskunet_amd.CrossEntropyLoss on module output"), "optimizer_in_step": bool(opt), "hip_graph": bool(args.graph)},
            "roofline": roof,
            "step_roofline": step,
            "clock": dict(clk or {}, smi=sampler.summary()),
            "parity_gate": gate,
        }
        if host_incl is not None:
            rec["host_inclusive"] = host_incl
        if clk and headline_cfg:
            # box-to-box comparison: ~0.8 of the configs[1] fp16 step is MFMA kernels whose time follows this clock, the rest HBM streams
            # that do not (profiles/r04_kernel_time_split.txt); REF_CLOCK_MHZ is a convention.  Only for the configuration the share was
            # measured on (other dtypes / shapes / --optimizer have another split)
            rec["clock"].update({"ms_per_step_at_ref_clock": round(1e3 * t_step * (MFMA_SHARE * clk["clock_mhz"] / REF_CLOCK_MHZ + 1.0 - MFMA_SHARE), 3),
                                 "ref_clock_mhz": REF_CLOCK_MHZ, "mfma_share_of_step": MFMA_SHARE})
        if world == 1 and args.dtype == "fp16" and not args.no_fp32x_line and not args.graph:
            del model, net                                      # (the timed model's activations are gone; its parameters go with it)
            torch.cuda.empty_cache()
            rec["parity_grade_path"] = parity_grade_line(args, dev, args.pg_steps or max(10, args.steps // 2), 2, clk)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args.c_out, args.hw)
        print(json.dumps(rec), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
