#!/usr/bin/env python3
"""Benchmark of the correlation hot path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over this rank's batch of synthetic feature maps:
build the pyramid once, then `iters` lookups (what one RAFTStereo.forward does,
core/raft_stereo.py:100,108-110).  Pairs are independent, so ranks share nothing on the data
path (weak scaling: every GPU gets the same per-GPU batch); the only collectives are the
barrier and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "corr build TFLOP/s + lookup GB/s; stereo pairs/sec at 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_GBS = 6290.0          # the same guide: what a float4 copy kernel measures (79 % of spec)
MFMA16_PEAK_TF = 2500.0        # dense fp16/bf16
MFMA32_PEAK_TF = 157.3         # f32-input MFMA

# name -> (per-GPU batch, H, W, dtype, iters, impl, train)   feature depth is 256, 4 levels, radius 4
WORKLOADS = {
    # BASELINE.json configs[3]: KITTI 375x1242 -> 384x1248 padded, /4, 32 iters, 64 pairs over 8 GPUs
    "kitti": dict(batch=8, H=96, W=312, dtype="f16", iters=32, impl="reg_hip", train=False),
    # configs[1]: realtime 480x640, n_downsample 3, 7 iters, fp16
    "realtime": dict(batch=1, H=60, W=80, dtype="f16", iters=7, impl="reg_hip", train=False),
    # configs[0]: the reference's CPU-runnable case, 384x512 fp32, 32 iters
    "cpu_case": dict(batch=1, H=96, W=128, dtype="f32", iters=32, impl="reg_hip", train=False),
    # configs[2]: SceneFlow training crop 320x720, batch 8, 22 iters, bf16, forward + backward
    "sceneflow_train": dict(batch=8, H=80, W=180, dtype="bf16", iters=22, impl="reg_hip", train=True),
    # configs[4]: Middlebury-F 2016x3008 /4, alt (volume never built); row slabs of 63 rows per GPU
    "middlebury_alt": dict(batch=1, H=63, W=752, dtype="f16", iters=32, impl="alt_hip", train=False),
    # the whole Middlebury-F frame on one GPU
    "middlebury_alt_full": dict(batch=1, H=504, W=752, dtype="f16", iters=32, impl="alt_hip", train=False),
}
DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
D, LEVELS, RADIUS = 256, 4, 4


def note(msg):
    """Progress line on stderr (RC_BENCH_VERBOSE=1): the JSON line on stdout stays the only stdout output."""
    if os.environ.get("RC_BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()
if os.environ.get("RC_BENCH_DUMP_AFTER"):
    # a hang names itself: every thread's Python stack on stderr after N seconds, then exit
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["RC_BENCH_DUMP_AFTER"]), exit=True)


def synthetic_coords(B, H, W, seed=1234, adversarial=False):
    """SURVEY 8(d) lookup coordinates: x = w - disp with disp piecewise smooth in [0, W/2] (a bilinearly
    upsampled 8x8 grid of U(0, W/3) plus U(-0.5, 0.5) noise), or i.i.d. U(0, W) for the adversarial case;
    y = the row index.  [B, 2, H, W] float32 (numpy)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    xs = np.arange(W, dtype=f32)[None, None, :].repeat(B, 0).repeat(H, 1)
    ys = np.arange(H, dtype=f32)[None, :, None].repeat(B, 0).repeat(W, 2)
    if adversarial:
        disp = rng.uniform(0, W, size=(B, H, W)).astype(f32)
    else:
        coarse = rng.uniform(0, W / 3, size=(B, 8, 8)).astype(f32)
        gy, gx = np.linspace(0, 7, H, dtype=f32), np.linspace(0, 7, W, dtype=f32)
        y0, x0 = np.clip(np.floor(gy).astype(int), 0, 6), np.clip(np.floor(gx).astype(int), 0, 6)
        fy, fx = (gy - y0)[None, :, None], (gx - x0)[None, None, :]
        rows0, rows1 = coarse[:, y0], coarse[:, y0 + 1]
        smooth = (rows0[:, :, x0] * (1 - fy) * (1 - fx) + rows0[:, :, x0 + 1] * (1 - fy) * fx
                  + rows1[:, :, x0] * fy * (1 - fx) + rows1[:, :, x0 + 1] * fy * fx)
        disp = np.clip(smooth + rng.uniform(-0.5, 0.5, size=(B, H, W)), 0, W / 2).astype(f32)
    return np.ascontiguousarray(np.stack([xs - disp, ys], axis=1).astype(f32))


def synthetic_inputs(cfg, device, seed):
    """N(0,1) feature maps, piecewise-smooth disparity; all on the GPU before timing."""
    g = torch.Generator(device=device).manual_seed(seed)
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    f1 = torch.randn((B, D, H, W), generator=g, device=device).to(DTYPES[cfg["dtype"]])
    f2 = torch.randn((B, D, H, W), generator=g, device=device).to(DTYPES[cfg["dtype"]])
    coords = torch.from_numpy(synthetic_coords(B, H, W, seed=seed)).to(device)
    return f1, f2, coords


def moving_coords(coords, iters, seed):
    """One coordinate tensor per lookup: the GRU moves coords1 a little every iteration
    (core/raft_stereo.py:120-123), so no two lookups of a pass read exactly the same windows.  x gets a
    seeded U(-1, 1) px offset per pixel and lookup (y, the row index, stays); made before the timed region."""
    g = torch.Generator(device=coords.device).manual_seed(seed + 4242)
    out = []
    for _ in range(iters):
        c = coords.clone()
        c[:, 0] += torch.rand(c[:, 0].shape, generator=g, device=coords.device) * 2 - 1
        out.append(c)
    return out


def measured_traffic(workload, batch, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this workload
    (profiles/<round>/<workload>_b<batch>_pmc.json, written by tools/pmc_summarize.py): 2 x FETCH_SIZE +
    WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md.  Returns (bytes, source file) of the newest
    round that has one, (None, None) when no profile of this exact workload is committed.  This is a figure
    READ BACK from a committed profile of the same command, not a counter collected in this run (counters need
    rocprofv3 attached: tools/profile_gpu.sh)."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{workload}_b{batch}_pmc.json"))):
        try:
            with open(path) as f:
                v = json.load(f).get(kernel, {}).get("traffic_bytes_corrected")
        except (OSError, ValueError):
            v = None
        if v is not None:
            best, src = v, os.path.relpath(path, ROOT)
    return (None, None) if best is None else (int(best), src)


def algorithmic_counts(cfg):
    """Per-step algorithmic work: build FLOPs/bytes, lookup bytes per launch.

    Bytes are what the SHIPPED data layout has to move: the build reads both feature maps
    once and writes the stored (even) levels of the pyramid, 1 + 1/4 (+ 1/16 ...) of level 0; the alt lookup
    reads the fragment-order copy of f1 and the even levels of the pooled f2 copy, s*D*(1 + 1.25) per pixel.
    SURVEY 8(d) counted a pyramid with every level stored (1.875 x level 0; f2 copies 1.875 x): those older,
    larger counts are kept as *_survey_count so that a fraction priced on them can still be read."""
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    s = 4 if cfg["dtype"] == "f32" else 2
    nt = 2 * RADIUS + 1
    px = B * H * W
    stored = sum(0.5 ** i for i in range(0, LEVELS, 2))       # even levels: 1.25 for 4 levels
    every = sum(0.5 ** i for i in range(LEVELS))              # 1.875
    build_flops = 2.0 * B * H * W * W * D
    build_in = B * H * 2 * W * D * s
    build_bytes = build_in + B * H * W * W * s * stored
    build_bytes_survey = build_in + B * H * W * W * s * every
    lookup_bytes = px * (LEVELS * (2 * RADIUS + 2) * s + 4 + LEVELS * nt * 4)      # 228 B (16-bit) / 308 B (f32)
    alt_flops = px * 2.0 * D * LEVELS * (2 * RADIUS + 2)
    alt_bytes = px * (s * D * (1 + stored) + LEVELS * nt * 4 + 4)
    alt_bytes_survey = px * (s * D * (1 + every) + LEVELS * nt * 4 + 4)
    return dict(pixels=px, build_flops=build_flops, build_bytes=build_bytes, build_bytes_survey_count=build_bytes_survey,
                lookup_bytes=lookup_bytes, alt_flops=alt_flops, alt_bytes=alt_bytes,
                alt_bytes_survey_count=alt_bytes_survey)


EVENT_EVERY = 4     # timed steps between two that carry HIP events (see main)


def roof_entry(kernel, alg_bytes, us, traffic=(None, None), **more):
    """One `roofline` object: algorithmic bytes per launch / average launch time against the 8 TB/s spec peak,
    with the same rate against what a copy kernel reaches on this part beside it."""
    ach = alg_bytes / (us * 1e-6) / 1e9
    d = dict(bound="hbm", kernel=kernel, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
             frac=round(ach / HBM_PEAK_GBS, 4), frac_of_measured_copy=round(ach / HBM_COPY_GBS, 4),
             traffic=traffic[0], traffic_source=traffic[1],
             algorithmic_bytes_per_launch=int(alg_bytes), avg_launch_us=round(us, 2))
    d.update(more)
    return d


class Step:
    """One pass of the hot path; records HIP events around the build and around the lookups."""

    def __init__(self, cfg, device, seed):
        import raft_stereo_amd as R
        self.R, self.cfg = R, cfg
        self.f1, self.f2, self.coords = synthetic_inputs(cfg, device, seed)
        self.coords_list = moving_coords(self.coords, cfg["iters"], seed)
        self.train = cfg["train"]
        if self.train:
            self.f1.requires_grad_(True)
            self.f2.requires_grad_(True)
            # one gradient tensor PER LOOKUP, as training has (each iteration's loss term reaches its own lookup):
            # rounds 1-2 and the first r03 profiles passed one tensor to all of them, which then sat in the
            # caches while the fused adjoint read it 22 times (68 us); with 22 tensors (365 MB at config 3,
            # more than the Infinity Cache) the same kernel takes 108 us
            g = torch.Generator(device=device).manual_seed(seed + 1)
            self.gouts = [torch.randn((cfg["batch"], LEVELS * (2 * RADIUS + 1), cfg["H"], cfg["W"]), generator=g,
                                      device=device) for _ in range(cfg["iters"])]
            if os.environ.get("RC_BENCH_SHARED_GOUT"):      # lab: the old behaviour, to compare with old profiles
                self.gouts = [self.gouts[0]] * cfg["iters"]
        self.cls = R.CORR_IMPLEMENTATIONS[cfg["impl"]][0]
        self.events = []          # one set of 4 HIP events per recorded step (torch's current stream
        self.build_ms, self.lookup_ms, self.bwd_ms = [], [], []   # is the stream the kernels run on)

    def run(self, record):
        cfg = self.cfg
        if record:
            self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            self.events.append(self.ev)
            self.ev[0].record()
        blk = self.cls(self.f1, self.f2, num_levels=LEVELS, radius=RADIUS)
        if record:
            self.ev[1].record()
        outs = []
        for k in range(cfg["iters"]):
            out = blk(self.coords_list[k])
            if self.train:
                outs.append(out)
        if record:
            self.ev[2].record()
        if self.train:
            self.f1.grad = self.f2.grad = None
            torch.autograd.backward(outs, self.gouts)
        if record:
            self.ev[3].record()
        return out

    def collect(self):
        """Call after a device synchronize: event times of every recorded step."""
        for ev in self.events:
            self.build_ms.append(ev[0].elapsed_time(ev[1]))
            self.lookup_ms.append(ev[1].elapsed_time(ev[2]))
            self.bwd_ms.append(ev[2].elapsed_time(ev[3]))
        self.events = []


def fused_convc1(step, cfg, device, n=50):
    """SURVEY 8(f) rank 1, measured beside the contract's metric: relu(convc1(lookup)) in one kernel
    (rc_lookup_convc1_fwd) against the library's lookup followed by torch's 1x1 convolution + ReLU in
    the pyramid's dtype (what autocast runs for core/update.py:78).  HIP events on the current stream."""
    g = torch.Generator(device=device).manual_seed(77)
    dt = step.f1.dtype
    wgt = torch.randn((64, LEVELS * (2 * RADIUS + 1), 1, 1), generator=g, device=device) * 0.2
    bias = torch.randn((64,), generator=g, device=device)
    wq, bq = wgt.to(dt), bias.to(dt)
    with torch.no_grad():
        blk = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)

        def timed(fn):
            for _ in range(5):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            b.synchronize()
            return a.elapsed_time(b) / n * 1e3

        fused = timed(lambda: blk.lookup_convc1(step.coords, wgt, bias))
        unfused = timed(lambda: torch.relu_(torch.nn.functional.conv2d(blk(step.coords).to(dt), wq, bq)))
        vt = None
        if cfg["dtype"] != "f32" and step.cls is step.R.CorrBlock1DHip:
            # the lookup with its taps in the volume's 16-bit type (rc_lookup_fwd_vt: what the reference's
            # CorrBlockFast1D returns under mixed precision), coordinates moved every call like the timed region
            blk16 = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS, out_dtype=dt)
            it = iter(range(10 ** 9))
            vt = timed(lambda: blk16(step.coords_list[next(it) % len(step.coords_list)]))
    px = cfg["batch"] * cfg["H"] * cfg["W"]
    s = 4 if cfg["dtype"] == "f32" else 2
    alg = px * (LEVELS * (2 * RADIUS + 2) * s + 4 + 64 * s)
    res = dict(fused_lookup_convc1_us=round(fused, 2), unfused_lookup_then_convc1_us=round(unfused, 2),
               fused_lookup_convc1_gbps=round(alg / (fused * 1e-6) / 1e9, 1))
    if vt is not None:
        alg16 = px * (LEVELS * (2 * RADIUS + 2) * s + 4 + LEVELS * (2 * RADIUS + 1) * s)
        res.update(lookup_16bit_out_us=round(vt, 2), lookup_16bit_out_gbps=round(alg16 / (vt * 1e-6) / 1e9, 1))
    return res


def sampler_twin(cfg, device, n=50):
    """SURVEY 8(a) rows a6/a7 as the reference ships them: the `corr_sampler` extension's forward and backward on
    the caller's own row-major level-0 volume [B,H,W1,W2] (sampler/sampler_kernel.cu:19-166), measured beside
    the contract's metric.  Algorithmic bytes: forward reads 2r+2 volume entries and one coordinate per pixel
    and writes 2r+1 values; backward writes the whole zero-filled gradient volume (the reference allocates
    zeros_like and adds 2r+2 entries per pixel, :148) and reads 2r+1 gradients and one coordinate per pixel."""
    from raft_stereo_amd import sampler
    dt = DTYPES[cfg["dtype"]]
    s = 4 if cfg["dtype"] == "f32" else 2
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    g = torch.Generator(device=device).manual_seed(123)
    vol = torch.randn((B, H, W, W), generator=g, device=device).to(dt)
    coords = torch.from_numpy(synthetic_coords(B, H, W, seed=11)).to(device)
    nt = 2 * RADIUS + 1
    gcorr = torch.randn((B, nt, H, W), generator=g, device=device).to(dt)

    def timed(fn):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / n * 1e3

    fwd = timed(lambda: sampler.forward(vol, coords, RADIUS))
    bwd = timed(lambda: sampler.backward(vol, coords, gcorr, RADIUS))
    px = B * H * W
    fwd_bytes = px * ((nt + 1) * s + 4 + nt * s)
    bwd_bytes = px * (W * s + nt * s + 4)
    return dict(sampler_twin_fwd_us=round(fwd, 2), sampler_twin_fwd_gbps=round(fwd_bytes / (fwd * 1e-6) / 1e9, 1),
                sampler_twin_bwd_us=round(bwd, 2), sampler_twin_bwd_gbps=round(bwd_bytes / (bwd * 1e-6) / 1e9, 1))


def update_ops(cfg, device, n=50):
    """SURVEY 8(f) ranks 3 and 4, measured beside the contract's metric: the ConvGRU gate arithmetic on the
    1/f-resolution hidden state ([B,128,H,W], core/update.py:27-31) as two library kernels against the same
    chain of torch ops, and the convex upsampling (core/raft_stereo.py:55-67) against its torch ops.  HIP
    events on the current stream; algorithmic bytes = every operand once."""
    import raft_stereo_amd as R
    dt = DTYPES[cfg["dtype"]]
    s = 4 if cfg["dtype"] == "f32" else 2
    B, H, W, C = cfg["batch"], cfg["H"], cfg["W"], 128
    nd = 3 if W * 8 <= 640 else 2          # realtime runs at 1/8 resolution, everything else at 1/4
    f = 2 ** nd
    g = torch.Generator(device=device).manual_seed(99)
    ctx = torch.randn((B, 3 * C, H, W), generator=g, device=device).to(dt)
    cz, cr, cq = ctx.split(split_size=C, dim=1)
    zc, rc, qc = (torch.randn((B, C, H, W), generator=g, device=device).to(dt) for _ in range(3))
    h = torch.tanh(torch.randn((B, C, H, W), generator=g, device=device)).to(dt)
    flow = torch.randn((B, 2, H, W), generator=g, device=device) * 4
    mask = torch.randn((B, 9 * f * f, H, W), generator=g, device=device).to(dt)

    def timed(fn):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / n * 1e3

    def torch_gru():
        z = torch.sigmoid(zc + cz)
        r = torch.sigmoid(rc + cr)
        rh = r * h
        q = torch.tanh(qc + cq)
        return rh, (1 - z) * h + z * q

    def torch_up():
        m = torch.softmax(mask.view(B, 1, 9, f, f, H, W), dim=2)
        u = torch.nn.functional.unfold(f * flow, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
        return torch.sum(m * u, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, f * H, f * W)[:, :1]

    # the whole gru08 of the default model (hidden 128, inputs 128 + 128): ConvGRUHip against the reference's
    # forward spelled with torch ops on the same parameters (convolutions are torch's in both)
    gru = R.ConvGRUHip(C, 2 * C).to(device)
    xm = torch.randn((B, C, H, W), generator=g, device=device)            # motion features are fp32 (update.py:85)
    xu = torch.randn((B, C, H, W), generator=g, device=device).to(dt)

    def torch_convgru():
        x = torch.cat([xm, xu], dim=1)
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(gru.convz(hx) + cz)
        r = torch.sigmoid(gru.convr(hx) + cr)
        q = torch.tanh(gru.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q

    with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
        t_gru_mod = timed(lambda: gru(h, cz, cr, cq, xm, xu), )
        t_gru_mod_torch = timed(torch_convgru)
    with torch.no_grad():
        t_upd = timed(lambda: R.gru_update(zc, cz, qc, cq, h))
        t_gate = timed(lambda: R.gru_gate_r(rc, cr, h))
        t_torch = timed(torch_gru)
        t_up = timed(lambda: R.upsample_flow_hip(flow, mask, nd, channels=1))
        t_up_torch = timed(torch_up)
    # the adjoints (training): autograd through the library's functions, both flow channels
    fl, mk = flow.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    up = R.upsample_flow_hip(fl, mk, nd)
    gup = torch.randn_like(up)
    t_up_bwd = timed(lambda: torch.autograd.grad(up, (fl, mk), gup, retain_graph=True))
    leaves = [t.clone().requires_grad_(True) for t in (zc, qc, h)]
    hn = R.gru_update(leaves[0], cz, leaves[1], cq, leaves[2])
    ghn = torch.randn_like(hn)
    t_upd_bwd = timed(lambda: torch.autograd.grad(hn, leaves, ghn, retain_graph=True))
    el = B * C * H * W
    up_bytes = B * H * W * (9 * f * f * s + 4 + f * f * 4)
    loss = {}
    if cfg["train"]:
        # sequence_loss over the iterations' full-resolution predictions (train_stereo.py:35-70), forward + backward
        n = cfg["iters"]
        gt = -torch.rand((B, 1, f * H, f * W), generator=g, device=device) * 100
        valid = (torch.rand((B, f * H, f * W), generator=g, device=device) > 0.1).float()
        preds = [(gt + torch.randn((B, 1, f * H, f * W), generator=g, device=device)).requires_grad_(True)
                 for _ in range(n)]

        def ours():
            l, _ = R.sequence_loss_hip(preds, gt, valid)
            l.backward()

        def torch_loss():      # the reference's ops, masks included
            vm = ((valid >= 0.5) & (torch.sum(gt ** 2, dim=1).sqrt() < 700)).unsqueeze(1)
            tot = 0.0
            for i in range(n):
                tot = tot + 0.9 ** (15 / (n - 1) * (n - i - 1)) * (preds[i] - gt).abs()[vm.bool()].mean()
            epe = torch.sum((preds[-1] - gt) ** 2, dim=1).sqrt().view(-1)[vm.view(-1)]
            _ = (epe.mean().item(), (epe < 1).float().mean().item(), (epe < 3).float().mean().item(),
                 (epe < 5).float().mean().item())
            tot.backward()

        def wall(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        # SURVEY 8(f) rank 1 in training: all iterations' lookup -> convc1 -> relu, forward + backward, fused
        # (lookup recomputed in the backward, 64-channel activations kept) against lookup + torch's convolution
        wgt = (torch.randn((64, LEVELS * (2 * RADIUS + 1), 1, 1), generator=g, device=device) * 0.2).requires_grad_(True)
        bias = torch.randn((64,), generator=g, device=device).requires_grad_(True)
        f1 = torch.randn((B, D, H, W), generator=g, device=device).to(dt).requires_grad_(True)
        f2 = torch.randn((B, D, H, W), generator=g, device=device).to(dt).requires_grad_(True)
        crd = torch.from_numpy(synthetic_coords(B, H, W, seed=5)).to(device)
        gact = torch.randn((B, 64, H, W), generator=g, device=device).to(dt)

        def c1_step(fused):
            blk = R.CorrBlock1DHip(f1, f2, num_levels=LEVELS, radius=RADIUS)
            if fused:
                outs = [blk.lookup_convc1(crd, wgt, bias) for _ in range(n)]
            else:
                outs = [torch.relu(torch.nn.functional.conv2d(blk(crd).to(dt), wgt.to(dt), bias.to(dt)))
                        for _ in range(n)]
            torch.autograd.backward(outs, [gact] * n)
        loss = dict(train_fused_convc1_ms=round(wall(lambda: c1_step(True)), 3),
                    train_unfused_convc1_ms=round(wall(lambda: c1_step(False)), 3),
                    sequence_loss_fwd_bwd_ms=round(wall(ours), 3), sequence_loss_torch_ms=round(wall(torch_loss), 3),
                    sequence_loss_predictions=n)
    return dict(**loss, gru_update_us=round(t_upd, 2), gru_update_gbps=round(6 * el * s / (t_upd * 1e-6) / 1e9, 1),
                gru_gate_r_us=round(t_gate, 2), gru_gate_r_gbps=round(4 * el * s / (t_gate * 1e-6) / 1e9, 1),
                gru_pointwise_torch_us=round(t_torch, 2),
                convgru_module_us=round(t_gru_mod, 2), convgru_module_torch_us=round(t_gru_mod_torch, 2),
                upsample_us=round(t_up, 2), upsample_gbps=round(up_bytes / (t_up * 1e-6) / 1e9, 1),
                upsample_torch_us=round(t_up_torch, 2), upsample_bwd_us=round(t_up_bwd, 2),
                gru_update_bwd_us=round(t_upd_bwd, 2))


def _torch_ops_refine(blk, corr_fn, net_list, inp_list, args, iters, test_mode=True):
    """The same loop spelled with torch ops on the same parameters, the way the reference's update block and
    forward pass are written (core/update.py:23-32, 77-85, 115-138; core/raft_stereo.py:55-67, 108-139): plain
    lookups from the same correlation block, then torch for everything else.  Baseline for `refine_loop` only."""
    import torch.nn.functional as F
    amp = args.mixed_precision
    layers = args.n_gru_layers
    f = 2 ** args.n_downsample

    def gru(m, h, cz, cr, cq, *xs):
        x = torch.cat(xs, dim=1)
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(m.convz(hx) + cz)
        r = torch.sigmoid(m.convr(hx) + cr)
        q = torch.tanh(m.convq(torch.cat([r * h, x], dim=1)) + cq)
        return (1 - z) * h + z * q

    def pool2x(x):
        return F.avg_pool2d(x, 3, stride=2, padding=1)

    def interp(x, dest):
        return F.interpolate(x, dest.shape[2:], mode="bilinear", align_corners=True)

    def block(net, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True):
        if iter32:
            net[2] = gru(blk.gru32, net[2], *inp_list[2], pool2x(net[1]))
        if iter16:
            net[1] = gru(blk.gru16, net[1], *inp_list[1], pool2x(net[0]), *([interp(net[2], net[1])] if layers > 2 else []))
        if iter08:
            e = blk.encoder
            cor = F.relu(e.convc2(F.relu(e.convc1(corr))))
            flo = F.relu(e.convf2(F.relu(e.convf1(flow))))
            motion = torch.cat([F.relu(e.conv(torch.cat([cor, flo], dim=1))), flow], dim=1)
            net[0] = gru(blk.gru08, net[0], *inp_list[0], motion, *([interp(net[1], net[0])] if layers > 1 else []))
        if not update:
            return net
        return net, .25 * blk.mask(net[0]), blk.flow_head(net[0])

    n, _, h, w = net_list[0].shape
    ys, xs = torch.meshgrid(torch.arange(h, device=net_list[0].device), torch.arange(w, device=net_list[0].device),
                            indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(n, 1, 1, 1)
    coords1 = coords0.clone()
    net, preds = list(net_list), []
    for itr in range(iters):
        coords1 = coords1.detach()
        corr = corr_fn(coords1)
        flow = coords1 - coords0
        with torch.autocast("cuda", enabled=amp):
            if layers == 3 and args.slow_fast_gru:
                net = block(net, iter32=True, iter16=False, iter08=False, update=False)
            if layers >= 2 and args.slow_fast_gru:
                net = block(net, iter32=layers == 3, iter16=True, iter08=False, update=False)
            net, up_mask, delta_flow = block(net, corr, flow, iter32=layers == 3, iter16=layers >= 2)
        delta_flow[:, 1] = 0.0
        coords1 = coords1 + delta_flow
        if test_mode and itr < iters - 1:
            continue
        fl = coords1 - coords0
        mask = torch.softmax(up_mask.view(n, 1, 9, f, f, h, w), dim=2)
        up = F.unfold(f * fl, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
        up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2, f * h, f * w)
        preds.append(up[:, :1])
    return (fl, preds[-1]) if test_mode else preds


def refine_loop(step, cfg, device, reps=3, capture=True, baseline=True):
    """Everything RAFTStereo.forward does after its encoders (core/raft_stereo.py:100-141) on this package's
    operators, beside the contract's metric: correlation block + default-width update block (hidden 128, 3 GRU
    layers; its dense convolutions are torch's) + convex upsampling, test mode, autocast for 16-bit feature maps.
    Random-init weights, synthetic context tensors.  Wall clock, synchronised."""
    from argparse import Namespace
    import raft_stereo_amd as R
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    nd = 3 if W * 8 <= 640 else 2
    dt = DTYPES[cfg["dtype"]]
    amp = dt != torch.float32
    realtime = nd == 3        # BASELINE config 2: n_downsample 3, 2 GRU layers, slow_fast_gru; else the defaults
    args = Namespace(hidden_dims=[128] * 3, corr_levels=LEVELS, corr_radius=RADIUS, n_downsample=nd,
                     slow_fast_gru=realtime, n_gru_layers=2 if realtime else 3, mixed_precision=amp)
    g = torch.Generator(device=device).manual_seed(5)
    blk = R.BasicMultiUpdateBlockHip(args, hidden_dims=args.hidden_dims).to(device).eval()
    sizes = [(H, W), ((H + 1) // 2, (W + 1) // 2), ((H + 3) // 4, (W + 3) // 4)]
    net = [torch.tanh(torch.randn((B, 128, h, w), generator=g, device=device)).to(dt) for h, w in sizes]
    inp = [list(torch.randn((B, 384, h, w), generator=g, device=device).to(dt).split(128, dim=1)) for h, w in sizes]
    iters = cfg["iters"]

    def once():
        corr_fn = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        return R.refine_disparity(blk, corr_fn, net, inp, args, iters=iters, test_mode=True)

    with torch.no_grad():
        once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / reps
    if not baseline:      # --loop-only: a kernel trace of the library's loop alone
        return dict(refine_loop_ms=round(dt_s * 1e3, 2))

    def once_torch():
        corr_fn = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        return _torch_ops_refine(blk, corr_fn, net, inp, args, iters)

    with torch.no_grad():
        ref_up = once_torch()[1]
        agree = float((once()[1] - ref_up).abs().mean())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            once_torch()
        torch.cuda.synchronize()
        dt_torch = (time.perf_counter() - t0) / reps
    out = dict(refine_loop_torch_ops_ms=round(dt_torch * 1e3, 2),
               refine_loop_mean_abs_diff_px_vs_torch_ops=round(agree, 5))
    out.update(refine_loop_config=f"{args.n_gru_layers} GRU layers, slow_fast_gru={args.slow_fast_gru}, hidden 128, "
                                  f"{iters} iterations, 1/{2 ** nd} resolution, autocast={amp}",
               refine_loop_ms=round(dt_s * 1e3, 2), refine_loop_pairs_per_s=round(B / dt_s, 2),
               refine_loop_ms_per_iteration=round(dt_s * 1e3 / iters, 3))
    # the same loop captured once in a HIP graph and replayed (nothing in it allocates outside torch's graph
    # pool or synchronises): what a latency-bound caller (the realtime configuration) pays per pair
    if not capture:       # multi-rank runs: no graph capture beside RCCL
        return out
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                once()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                once()
            graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            torch.cuda.synchronize()
            out["refine_loop_hip_graph_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
    except Exception as e:      # a side metric must not take the contract's line down
        out["refine_loop_hip_graph_ms"] = None
        out["refine_loop_hip_graph_error"] = str(e)[:120]
    return out


def refine_train_step(step, cfg, device, reps=2, dist=None):
    """A training step of everything after the encoders (SceneFlow shape): refinement loop in training mode
    (a prediction per iteration), sequence loss, backward down to the feature maps and the update block's
    parameters -- this package's operators against the same step in torch ops (reference-style loss included).
    Wall clock, synchronised; random-init weights, synthetic context tensors and ground truth.

    With more than one rank (BASELINE config 3: RCCL gradient all-reduce at 1/2/4/8 GPUs) every rank runs the step
    on its own batch shard with the update block wrapped in DistributedDataParallel (one process per GPU, backend
    "nccl" = RCCL, instead of the reference's nn.DataParallel, train_stereo.py:137), and the loss is normalised
    over the GLOBAL batch's valid pixels (train_stereo.py:58) through two scalar all-reduces
    (raft_stereo_amd.global_batch_loss).  The torch-ops leg is single-rank only."""
    from argparse import Namespace
    import raft_stereo_amd as R
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    dt = DTYPES[cfg["dtype"]]
    amp = dt != torch.float32
    world = dist.get_world_size() if dist is not None else 1
    args = Namespace(hidden_dims=[128] * 3, corr_levels=LEVELS, corr_radius=RADIUS, n_downsample=2,
                     slow_fast_gru=False, n_gru_layers=3, mixed_precision=amp)
    g = torch.Generator(device=device).manual_seed(6)          # same weights on every rank
    blk = R.BasicMultiUpdateBlockHip(args, hidden_dims=args.hidden_dims).to(device)
    for p in blk.parameters():                                  # (nn init uses the global RNG: make it rank-independent)
        p.data.copy_(torch.randn(p.shape, generator=g, device=device) * 0.05)
    g = torch.Generator(device=device).manual_seed(60 + (dist.get_rank() if dist is not None else 0))
    sizes = [(H, W), ((H + 1) // 2, (W + 1) // 2), ((H + 3) // 4, (W + 3) // 4)]
    net = [torch.tanh(torch.randn((B, 128, h, w), generator=g, device=device)).to(dt) for h, w in sizes]
    inp = [list(torch.randn((B, 384, h, w), generator=g, device=device).to(dt).split(128, dim=1)) for h, w in sizes]
    gt = -torch.rand((B, 1, 4 * H, 4 * W), generator=g, device=device) * 50
    valid = (torch.rand((B, 4 * H, 4 * W), generator=g, device=device) > 0.1).float()
    iters = cfg["iters"]
    class Refiner(torch.nn.Module):
        """The trainable module of the step: the whole refinement loop as ONE forward, so that
        DistributedDataParallel sees one forward and one backward per step."""

        def __init__(self, block):
            super().__init__()
            self.block = block

        def forward(self, corr_fn):
            return R.refine_disparity(self.block, corr_fn, net, inp, args, iters=iters, test_mode=False)

    model = Refiner(blk)
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel
        # the default 25 MB buckets, all-reduced while the backward pass is still running.  (main() refuses this
        # leg when several ranks share one device: tools/lab_ddp_gloo_control.py.  CPU tensors over gloo are
        # covered by tests/test_cpu_host.py; RCCL needs more than one GPU.)
        model = DistributedDataParallel(model, device_ids=[device.index], gradient_as_bucket_view=False)
        note("update block wrapped in DistributedDataParallel")
        if os.environ.get("RC_BENCH_VERBOSE"):
            import faulthandler
            faulthandler.dump_traceback_later(100, exit=False)
    count_local = int(((valid >= 0.5) & (gt[:, 0].abs() < 700)).sum().item())

    def ours():
        blk.zero_grad(set_to_none=True)
        step.f1.grad = step.f2.grad = None
        corr_fn = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        preds = model(corr_fn)
        loss, _ = R.sequence_loss_hip(preds, gt, valid)
        note("forward done")
        loss_bwd, loss_global = R.global_batch_loss(loss, count_local)     # identity with one rank
        loss_bwd.backward()
        note("backward done")
        return loss_global

    def torch_ops():
        blk.zero_grad(set_to_none=True)
        step.f1.grad = step.f2.grad = None
        corr_fn = step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        preds = _torch_ops_refine(blk, corr_fn, net, inp, args, iters, test_mode=False)
        vm = ((valid >= 0.5) & (torch.sum(gt ** 2, dim=1).sqrt() < 700)).unsqueeze(1)
        loss = 0.0
        for i, p in enumerate(preds):
            loss = loss + 0.9 ** (15 / (iters - 1) * (iters - i - 1)) * (p - gt).abs()[vm.bool()].mean()
        epe = torch.sum((preds[-1] - gt) ** 2, dim=1).sqrt().view(-1)[vm.view(-1)]
        _ = (epe.mean().item(), (epe < 1).float().mean().item())
        loss.backward()
        return loss

    def wall(fn):
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            last = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, float(last)

    torch.cuda.reset_peak_memory_stats(device)
    t_ours, l_ours = wall(ours)
    peak_ours = torch.cuda.max_memory_allocated(device) / 2 ** 30
    out = dict(refine_train_step_ms=round(t_ours, 1), refine_train_loss=round(l_ours, 4),
               refine_train_peak_gib=round(peak_ours, 2))
    if world > 1:
        t = torch.tensor([t_ours], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)
        # every rank holds the same averaged gradients after the step: compare a checksum across ranks
        chk = torch.stack([p.grad.double().sum() for p in blk.parameters() if p.grad is not None]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        grad_bytes = sum(p.numel() * p.element_size() for p in blk.parameters() if p.requires_grad)
        out.update(refine_train_step_ms=round(float(t.item()), 1), n_ranks_seen=int(seen.item()),
                   allreduce_bytes=int(grad_bytes + 2 * 8),
                   allreduce_what="DistributedDataParallel gradient buckets of the update block (fp32) + 2 scalars "
                                  "(valid-pixel count, L1 numerator) per step",
                   refine_train_global_batch=B * world,
                   refine_train_grads_identical_across_ranks=bool(abs(float(hi.item()) - float(lo.item())) <=
                                                                  1e-9 * max(1.0, abs(float(hi.item())))))
        return out
    torch.cuda.reset_peak_memory_stats(device)
    t_torch, l_torch = wall(torch_ops)
    peak_torch = torch.cuda.max_memory_allocated(device) / 2 ** 30
    out.update(refine_train_step_torch_ops_ms=round(t_torch, 1), refine_train_loss_torch_ops=round(l_torch, 4),
               refine_train_peak_gib_torch_ops=round(peak_torch, 2))
    return out


def reg_hip_beside_alt(step, cfg, device, n=20):
    """Beside an alt workload: the same lookups from a MATERIALISED pyramid (reg_hip) of the same feature maps.
    The reference offers alt to save memory on large frames (README: 'significantly decrease memory consumption ...
    slower'); on a 288 GB part the Middlebury-F pyramid is 0.7 GB, so a caller should see what not materialising
    costs per lookup here, and what the build costs once per pair."""
    import raft_stereo_amd as R
    with torch.no_grad():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blk = R.CorrBlock1DHip(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        torch.cuda.synchronize()
        a.record()
        blk = R.CorrBlock1DHip(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)
        b.record()
        b.synchronize()
        build_us = a.elapsed_time(b) * 1e3
        for c in step.coords_list[:3]:
            blk(c)
        a.record()
        for k in range(n):
            blk(step.coords_list[k % len(step.coords_list)])
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        agree = float((blk(step.coords_list[0]) - step.cls(step.f1, step.f2, num_levels=LEVELS, radius=RADIUS)(
            step.coords_list[0])).abs().max())
    return dict(reg_hip_at_this_size_us=round(us, 2), reg_hip_build_at_this_size_us=round(build_us, 1),
                reg_hip_pyramid_gb=round(blk.volume.numel() * blk.volume.element_size() / 1e9, 3),
                reg_hip_vs_alt_max_abs_diff=round(agree, 5))


def hbm_resident_lookup(cfg, device, seed, batch=64, passes=3):
    """The lookup's roofline once more at a batch whose touched pyramid lines and outputs no longer fit
    the 256 MiB Infinity Cache (64 KITTI pairs: what one pass reads and writes is ~0.5 GB), beside the
    per-GPU headline batch: build once, then `iters` lookups with moving coordinates, HIP events on
    the current stream, average per launch over `passes` passes."""
    big = dict(cfg, batch=batch)
    st = Step(big, device, seed)
    with torch.no_grad():
        blk = st.cls(st.f1, st.f2, num_levels=LEVELS, radius=RADIUS)
        for c in st.coords_list[:3]:
            blk(c)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(passes):
            for c in st.coords_list:
                blk(c)
        b.record()
        b.synchronize()
    us = a.elapsed_time(b) / (passes * len(st.coords_list)) * 1e3
    alg = algorithmic_counts(big)["lookup_bytes"]
    return roof_entry("lookup_fwd_kernel", alg, us, measured_traffic("kitti", batch, "lookup_fwd_kernel"), batch=batch)


def hip_graph_step(cfg, device, seed, n=50):
    """The same step (build/prepare + lookups, and for a training workload the backward to the feature maps)
    captured once in a HIP graph and replayed: what a caller that captures its loop pays per pair batch once
    launch and Python overhead are gone.  Reported beside `ms_per_step` (which stays the eager number); the
    library needs nothing special for this: it neither allocates nor synchronises.  The step gets fresh input
    tensors whose first passes run on the capture's side stream (torch wants the leaves' gradient accumulators
    made there: with accumulators from earlier passes on the default stream the capture of the backward
    crashed on the host).  Training: the replayed gradients are compared bit for bit with the eager ones."""
    step = Step(cfg, device, seed)
    ctx = torch.enable_grad() if cfg["train"] else torch.no_grad()
    with ctx:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step.run(record=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eager_grads = None
        if cfg["train"]:
            eager_grads = (step.f1.grad.clone(), step.f2.grad.clone())
            step.f1.grad = step.f2.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step.run(record=False)
        for _ in range(3):
            graph.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            graph.replay()
        b.record()
        b.synchronize()
    out = dict(hip_graph_ms_per_step=round(a.elapsed_time(b) / n, 4))
    if eager_grads is not None:
        out["hip_graph_grads_equal_eager"] = bool(torch.equal(step.f1.grad, eager_grads[0])
                                                  and torch.equal(step.f2.grad, eager_grads[1]))
    return out


def cpu_baseline(cfg, budget_s=15.0):
    """The reference's reg path (torch ops port, oracle/reg_torch.py) on this host's cores, on a
    bounded sample of the same workload: ONE pair of this shape in fp32, repeated for ~budget_s."""
    from oracle.reg_torch import time_reg_path
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    if cfg["impl"] == "alt_hip":
        sample_h = min(cfg["H"], 63)
    else:
        sample_h = cfg["H"]
    n, tb, tl = 0, 0.0, 0.0
    t0 = time.perf_counter()
    while True:
        b, l, used = time_reg_path(1, D, sample_h, cfg["W"], cfg["iters"], LEVELS, RADIUS, threads=threads)
        n += 1
        tb += b
        tl += l
        if time.perf_counter() - t0 > budget_s or n >= 500:
            break
    per_pair = (tb + tl) / n * (cfg["H"] / sample_h)
    return dict(value=round(1.0 / per_pair, 4), unit="pairs/s", cores=used, kind="port",
                sample=f"1 pair of the {cfg['H']}x{cfg['W']} workload in fp32 (reg path: einsum + avg_pool2d + "
                       f"grid_sample, {cfg['iters']} lookups), {n} passes; build {tb / n * 1e3:.1f} ms, "
                       f"lookups {tl / n * 1e3:.1f} ms per pass")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps after 10 (a step is 0.4 ms, 3.6 ms at the largest workload; 20 steps were 8 ms and
    # noisy); under --profile 20 after 5, the command the committed profiles were collected with
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="kitti", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--dtype", choices=["f32", "f16", "bf16"], default=None, help="override the feature dtype")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-refine-loop", action="store_true",
                    help="skip the refinement-loop side metric (torch convolutions, HIP graph capture): for profiler runs")
    ap.add_argument("--miopen-find", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: MIOpen searches its solvers for the loop's convolutions")
    ap.add_argument("--loop-only", action="store_true",
                    help="after the timed region run only the library's refine loop (no torch-ops leg, no side metrics)")
    ap.add_argument("--adversarial", action="store_true", help="i.i.d. uniform disparity (worst-case gather)")
    ap.add_argument("--profile", action="store_true",
                    help="profiler runs: no CPU baseline, no refinement-loop legs, no 64-pair leg (one instantiation "
                         "of each kernel at one size per run)")
    ap.add_argument("--debug-leg", choices=["loop_eager", "graph_step", "loop_capture"], default=None,
                    help="after the timed region run exactly one side leg (to bisect a fault under a profiler)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.profile else 100
    if args.warmup is None:
        args.warmup = 5 if args.profile else 10
    if args.profile:
        args.no_cpu_baseline = args.no_refine_loop = True
    if args.miopen_find and WORKLOADS[args.workload]["train"]:
        # MIOpen's exhaustive search over backward-data / backward-weight solvers of every convolution shape ran
        # past three silent minutes on the first training step (round 1): refuse instead of looking hung
        raise SystemExit("--miopen-find is for the inference workloads; refusing it with --workload sceneflow_train")
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    # HIP graph capture is kept out of profiler runs by construction: under rocprofv3 counter collection the
    # capture of the refinement loop aborted the process ("profiler fault")
    profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or \
        "rocprofiler" in os.environ.get("LD_PRELOAD", "")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > max(torch.cuda.device_count(), 1) and WORKLOADS[args.workload]["train"] and not args.no_refine_loop \
            and not args.profile and not args.loop_only:
        # several ranks on one device + DistributedDataParallel: torch's DDP over gloo stalls in a backward pass in
        # that arrangement, with or without a kernel of this library (profiles/r02/lab_ddp_gloo_control_one_gpu.txt)
        raise SystemExit(f"refusing the DDP training leg with {world} ranks on {torch.cuda.device_count()} GPU(s): "
                         "DistributedDataParallel with several ranks sharing one device stalls in backward "
                         "(profiles/r02/lab_ddp_gloo_control_one_gpu.txt); use one GPU per rank, or --no-refine-loop "
                         "for the forward/backward correlation step alone")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)     # (an inference rehearsal may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; RC_BENCH_BACKEND=gloo only to rehearse the multi-rank path on a one-GPU box
        backend = os.environ.get("RC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = dict(WORKLOADS[args.workload])
    if args.batch:
        cfg["batch"] = args.batch
    if args.dtype:
        cfg["dtype"] = args.dtype
    step = Step(cfg, device, seed=1234 + rank)
    if args.adversarial:
        step.coords = torch.from_numpy(synthetic_coords(cfg["batch"], cfg["H"], cfg["W"], seed=7, adversarial=True)).to(device)
        step.coords_list = moving_coords(step.coords, cfg["iters"], 7)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step.run(record=False)
    # the timed steps are eager and their launches run barely ahead of the GPU (8 us of host time beside a 9 us
    # kernel): Python's cyclic collector is kept out of the timed region, as timeit keeps it out of its own
    import gc
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP events bracket the build and the lookups of every EVENT_EVERY-th timed step: an event record drains
        # the queue (3.5 us each, 14 us per step = 3.5 % of this step when every step carried its four), so the
        # kernels' durations are sampled inside the timed region instead of taxing all of it
        step.run(record=(i % EVENT_EVERY == 0))
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    step.collect()                  # event times are read after the timed region

    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    note("timed region done")
    cnt = algorithmic_counts(cfg)
    pairs_total = cfg["batch"] * world * args.steps
    value = pairs_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    build_ms = float(np.median(step.build_ms))
    lookup_ms = float(np.median(step.lookup_ms)) / cfg["iters"]        # average per launch
    is_alt = cfg["impl"] == "alt_hip"
    stock = not args.dtype and not args.adversarial     # the committed profiles are of the stock workloads
    if is_alt:
        roof = roof_entry("alt_fwd_kernel", cnt["alt_bytes"], lookup_ms * 1e3,
                          measured_traffic(args.workload, cfg["batch"], "alt_fwd_kernel") if stock else (None, None),
                          mfma_tflops=round(cnt["alt_flops"] / (lookup_ms * 1e-3) / 1e12, 2),
                          frac_survey_count=round(cnt["alt_bytes_survey_count"] / (lookup_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    else:
        roof = roof_entry("lookup_fwd_kernel", cnt["lookup_bytes"], lookup_ms * 1e3,
                          measured_traffic(args.workload, cfg["batch"], "lookup_fwd_kernel") if stock else (None, None))
    if not is_alt:
        # where the lookup's reads come from: a pyramid of up to 200 MB is left in the 256 MiB Infinity Cache by the
        # build (rc_build_pyramid's size rule), so only the writes of such a lookup reach HBM and its rate can pass
        # that of a copy; the HBM-resident figure is `roofline_hbm_resident` (64 pairs)
        es = 4 if cfg["dtype"] == "f32" else 2
        pyramid_bytes = cfg["batch"] * cfg["H"] * cfg["W"] * cfg["W"] * 1.25 * es
        roof["pyramid_reads_from"] = "infinity_cache" if pyramid_bytes <= (200 << 20) else "hbm"
    ach = roof["achieved"]
    extra = dict(lookup_gbps=round(ach, 1), lookup_us=round(lookup_ms * 1e3, 2))
    if not is_alt:
        peak = MFMA32_PEAK_TF if cfg["dtype"] == "f32" else MFMA16_PEAK_TF
        tf = cnt["build_flops"] / (build_ms * 1e-3) / 1e12
        hbm_cap_tf = cnt["build_flops"] / cnt["build_bytes"] * HBM_PEAK_GBS / 1e3
        hbm_cap_survey = cnt["build_flops"] / cnt["build_bytes_survey_count"] * HBM_PEAK_GBS / 1e3
        build_gbps = cnt["build_bytes"] / (build_ms * 1e-3) / 1e9
        build_tr = (None, None)
        for kname in (["build_kernel"] if cfg["dtype"] == "f32" else ["build16g_kernel", "build16_kernel"]):
            if stock and build_tr[0] is None:        # (the LDS-DMA kernel where the shape is eligible, else round 2's)
                build_tr = measured_traffic(args.workload, cfg["batch"], kname)
        extra.update(build_tflops=round(tf, 2), build_ms=round(build_ms, 4), build_frac_mfma=round(tf / peak, 4),
                     build_gbps=round(build_gbps, 1), build_algorithmic_bytes=int(cnt["build_bytes"]),
                     build_frac_of_binding_roof=round(tf / min(peak, hbm_cap_tf), 4),
                     build_frac_of_measured_copy=round(build_gbps / HBM_COPY_GBS, 4),
                     build_frac_of_binding_roof_survey_count=round(tf / min(peak, hbm_cap_survey), 4),
                     build_binding_roof="hbm" if hbm_cap_tf < peak else "mfma",
                     build_traffic=build_tr[0], build_traffic_source=build_tr[1])
    else:
        extra.update(prepare_ms=round(build_ms, 4))
        if world == 1 and not args.loop_only:
            extra.update(reg_hip_beside_alt(step, cfg, device))       # side leg, after the timed region
    if cfg["train"]:
        extra.update(backward_ms=round(float(np.median(step.bwd_ms)), 4))

    if args.debug_leg:
        if args.debug_leg == "loop_eager":
            extra.update(refine_loop(step, cfg, device, capture=False, baseline=False))
        elif args.debug_leg == "graph_step":
            extra.update(hip_graph_step(cfg, device, 1234 + rank))
        else:
            extra.update(refine_loop(step, cfg, device, capture=True, baseline=True))
        args.loop_only = True      # nothing else
    elif args.loop_only:
        if rank == 0 and not cfg["train"]:
            extra.update(refine_loop(step, cfg, device, capture=False, baseline=False))
    elif not is_alt and not cfg["train"] and world == 1:       # side legs: single-GPU runs only (with more ranks
        extra.update(fused_convc1(step, cfg, device))      # after the timed region: not part of `value`
        if cfg["H"] * cfg["W"] * cfg["W"] * cfg["batch"] <= (1 << 28):
            extra.update(sampler_twin(cfg, device))
    if not args.loop_only and world == 1 and cfg["H"] * cfg["W"] * cfg["batch"] <= 300000:   # the others would idle in a barrier)
        extra.update(update_ops(cfg, device))              # side metrics, after the timed region
    if (not args.loop_only and not cfg["train"] and world == 1 and cfg["H"] * cfg["W"] * cfg["batch"] <= 300000
            and not args.no_refine_loop):
        extra.update(refine_loop(step, cfg, device, capture=world == 1 and not profiler))
    if cfg["train"] and not args.no_refine_loop and not args.loop_only and (rank == 0 or world > 1):
        # config 3's gradient sync: with several ranks every rank takes part (DDP over RCCL)
        note("side legs done; training step")
        extra.update(refine_train_step(step, cfg, device, dist=dist))
        note("training step done")
    if world == 1 and not args.loop_only and not profiler:
        # side metric; kept out of multi-rank runs (no capture beside RCCL)
        extra.update(hip_graph_step(cfg, device, 1234 + rank))
    if profiler:
        extra["profiler_attached"] = True
    if (args.workload == "kitti" and not args.batch and not args.dtype and world == 1 and not args.loop_only
            and not args.profile and not args.adversarial):
        # the same kernel with nothing left in the Infinity Cache between lookups (after the timed region)
        extra["roofline_hbm_resident"] = hbm_resident_lookup(cfg, device, 1234 + rank)
        extra["lookup_frac_hbm_resident"] = extra["roofline_hbm_resident"]["frac"]

    line = {
        "metric": METRIC, "value": round(value, 3), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "event_timed_steps": len(step.build_ms),
        "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.workload}: correlation hot path only ({cfg['impl']}: "
                               f"{'prepare' if is_alt else 'build+pyramid'} + {cfg['iters']} lookups"
                               f"{' + backward' if cfg['train'] else ''}), fmaps [{cfg['batch']},{D},{cfg['H']},{cfg['W']}] "
                               f"per GPU, {LEVELS} levels, radius {RADIUS}",
                   "per_gpu_batch": cfg["batch"], "global_batch": cfg["batch"] * world, "iters": cfg["iters"],
                   "partition": "image pairs per rank, no data-path collective",
                   "coords": ("adversarial (i.i.d. uniform disparity)" if args.adversarial else
                              "piecewise-smooth disparity (SURVEY 8d)") +
                             ", x moved by a seeded U(-1,1) px per pixel before every lookup"},
        "roofline": roof,
    }
    line.update(extra)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
