"""CPU-side checks: the C-ABI library loads and exports every symbol include/raftcorr.h
declares, the layout arithmetic is what the oracle-side un-skew expects, and the host-side
dispatch / sharding logic (including a 2-rank gloo run) is right.  No compute calls."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from raft_stereo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "raftcorr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib.rc_version() == _lib.RC_VERSION == int(re.search(r"#define RC_VERSION (\d+)", hdr).group(1))


def test_layout_matches_documented_formula():
    from raft_stereo_amd._lib import lib, Layout, RC_F16, RC_F32
    lay = Layout()
    assert lib.rc_pyramid_layout(8, 96, 312, 312, 4, RC_F16, lay) == 0
    assert lay.pitch == 320 and lay.rb == 8 and lay.dtype == RC_F16
    assert list(lay.width) == [312, 156, 78, 39] and list(lay.widthp) == [312, 160, 80, 40]
    # a forward pyramid keeps its even levels (the lookup derives levels 1 and 3 in registers) ...
    off = 0
    assert list(lay.stored) == [1, 0, 1, 0] and lay.full == 0
    for i in (0, 2):
        assert lay.offset[i] == off
        off += 8 * 96 * lay.widthp[i] * lay.pitch
    assert lay.total == off and lay.offset[1] == 0 and lay.offset[3] == 0
    # ... the float gradient pyramid of rc_lookup_bwd every level
    glay = Layout()
    assert lib.rc_grad_pyramid_layout(8, 96, 312, 312, 4, glay) == 0
    assert list(glay.stored) == [1, 1, 1, 1] and glay.full == 1 and glay.rb == 4 and glay.dtype == RC_F32
    off = 0
    for i in range(4):
        assert glay.offset[i] == off
        off += 8 * 96 * glay.widthp[i] * glay.pitch
    assert glay.total == off
    assert lib.rc_lookup_bwd(16, 2, 16, 16, lay, 4, 0, None) == -1          # needs the gradient layout
    assert lib.rc_lookup_fwd(16, glay, 16, 2, 16, 4, 0, None) == -1         # needs the forward layout
    assert lib.rc_pyramid_layout(8, 96, 312, 312, 4, RC_F32, lay) == 0 and lay.rb == 4
    assert lib.rc_pyramid_layout(1, 1, 16, 4, 4, RC_F32, lay) != 0          # W2 too small for 4 levels
    assert b"too small" in lib.rc_last_error_string()
    assert lib.rc_pyramid_layout(1, 1, 16, 64, 5, RC_F32, lay) != 0         # more levels than compiled
    assert lib.rc_pyramid_layout(0, 1, 16, 64, 4, RC_F32, lay) != 0
    assert lib.rc_pyramid_layout(1, 1, 16, 64, 4, 7, lay) != 0              # unknown dtype


def test_null_and_bad_arguments_are_rejected_without_a_gpu():
    from raft_stereo_amd._lib import lib, Layout, RC_F32
    lay = Layout()
    lib.rc_pyramid_layout(1, 2, 64, 64, 4, RC_F32, lay)
    assert lib.rc_lookup_fwd(None, lay, None, 2, None, 4, 0, None) == -1
    assert lib.rc_build_pyramid(None, None, None, lay, 256, 0, None) == -1
    lay.pitch = 100                                                  # forged layout
    assert lib.rc_lookup_fwd(16, lay, 16, 2, 16, 4, 0, None) == -1
    assert b"rc_pyramid_layout" in lib.rc_last_error_string()


def test_block_row_index_is_a_bijection():
    """Every (w1, w2) of a level maps to its own cell of the group-skewed layout, a group of rb
    consecutive w2 is one 16-byte block, and for a fixed pixel consecutive groups are consecutive
    block rows (modulo the number of block rows)."""
    from raft_stereo_amd._lib import RC_WRAP_ROWS
    for W1, W2, i, rb in [(40, 45, 0, 8), (40, 45, 2, 4), (72, 200, 3, 8), (200, 45, 1, 4), (33, 33, 1, 8)]:
        Wi = W2 >> i
        WiP = (Wi + 7) // 8 * 8
        nb = WiP // rb
        pitch = (W1 + 63) // 64 * 64
        w1 = np.arange(W1)[:, None]
        w2 = np.arange(Wi)[None, :]
        sb = (w2 // rb - (w1 >> i) // rb + nb - RC_WRAP_ROWS // rb) % nb
        cell = (sb * pitch + w1) * rb + w2 % rb
        assert len(set(cell.ravel().tolist())) == W1 * Wi
        assert cell.max() < WiP * pitch
        assert np.all(cell[:, 1:][:, (w2[0, 1:] % rb) != 0] - cell[:, :-1][:, (w2[0, 1:] % rb) != 0] == 1)
        nxt = (w2[0, 1:] % rb) == 0                       # first element of the next group
        assert np.all((sb[:, 1:][:, nxt] - sb[:, :-1][:, nxt]) % nb == 1 % nb)


def test_dispatch_and_shards():
    import raft_stereo_amd as R
    assert set(R.HIP_CORR_CHOICES) == {"reg_hip", "alt_hip"} == set(R.CORR_IMPLEMENTATIONS)
    assert R.wants_autocast("reg_hip") and R.wants_autocast("reg_cuda") and not R.wants_autocast("reg")
    for n, ws in [(64, 8), (504, 8), (7, 3), (2, 4)]:
        spans = [R.shard_pairs(n, ws, r) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert R.shard_rows(504, 8, 3) == (189, 252)


def test_blocks_refuse_cpu_tensors():
    import torch
    import raft_stereo_amd as R
    f = torch.randn(1, 32, 2, 32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.CorrBlock1DHip(f, f)
    with pytest.raises(RuntimeError, match="no CPU path"):
        R.AltCorrBlock1DHip(f, f)


GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RC_ROOT"])
import numpy as np, torch, torch.distributed as dist
import raft_stereo_amd as R
from oracle import corr_oracle as O
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(3)
B, D, H, W = 5, 16, 6, 24
f1 = rng.standard_normal((B, D, H, W)).astype(np.float32)
f2 = rng.standard_normal((B, D, H, W)).astype(np.float32)
coords = O.synthetic_coords(B, H, W, seed=9)
full = O.lookup_exact(O.pool_pyramid(O.corr_volume(f1, f2), 3), coords, 2)
# pair sharding and row-slab sharding both reproduce the unsharded result with no exchange
lo, hi = R.shard_pairs(B, ws, rank)
part = O.lookup_exact(O.pool_pyramid(O.corr_volume(f1[lo:hi], f2[lo:hi]), 3), coords[lo:hi], 2)
assert np.array_equal(part, full[lo:hi])
r0, r1 = R.shard_rows(H, ws, rank)
slab = O.lookup_exact(O.pool_pyramid(O.corr_volume(f1[:, :, r0:r1], f2[:, :, r0:r1]), 3), coords[:, :, r0:r1], 2)
assert np.allclose(slab, full[:, :, r0:r1], rtol=0, atol=1e-6)
# the only cross-rank traffic of the bench: max-over-ranks of the elapsed time, sum of units
t = torch.tensor([float(rank + 1)]); dist.all_reduce(t, op=dist.ReduceOp.MAX)
n = torch.tensor([float(hi - lo)]); dist.all_reduce(n, op=dist.ReduceOp.SUM)
assert t.item() == ws and n.item() == B
dist.barrier()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, RC_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2


DDP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RC_ROOT"])
import numpy as np, torch, torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
import raft_stereo_amd as R
from oracle import update_oracle as U
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
N, H, W, n_pred = 4, 6, 10, 3
net = torch.nn.Sequential(torch.nn.Conv2d(2, 5, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(5, n_pred, 3, padding=1))
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 2, H, W, generator=g)
gt = torch.randn(N, 1, H, W, generator=g) * 3
valid = (torch.rand(N, H, W, generator=g) > 0.3).float()
valid[0] = 0.0                       # the ranks see very different numbers of valid pixels
valid[1, :, :3] = 0.0

def loss_of(model, xs, gts, valids):
    """sequence_loss in torch ops (train_stereo.py:35-70): gamma-weighted means of |pred - gt| over valid pixels"""
    preds = model(xs)
    mask = ((valids >= 0.5) & (gts[:, 0].abs() < 700)).unsqueeze(1)
    gamma = 0.9 ** (15 / (n_pred - 1))
    tot = 0.0
    for i in range(n_pred):
        tot = tot + gamma ** (n_pred - i - 1) * (preds[:, i:i + 1] - gts).abs()[mask].mean()
    return tot, int(mask.sum()), [preds[:, i:i + 1] for i in range(n_pred)]

ddp = DDP(net)
lo, hi = R.shard_pairs(N, ws, rank)
for case in ("uneven", "rank 0 has no valid pixel"):
    if case != "uneven":
        valid[:N // ws] = 0.0            # rank 0's local mean is over nothing (NaN): global_batch_loss must cope
    net.zero_grad(set_to_none=True)
    # one rank, the global batch (what nn.DataParallel + the loss on the gathered batch computes, train_stereo.py:186-193)
    ref_loss, _, ref_preds = loss_of(net, x, gt, valid)
    ref_grads = torch.autograd.grad(ref_loss, list(net.parameters()))
    # ... pinned on the numpy oracle of sequence_loss
    o_loss, _, _ = U.sequence_loss([p.detach().numpy() for p in ref_preds], gt.numpy(), valid.numpy())
    assert abs(float(ref_loss) - float(o_loss)) < 1e-5 * abs(float(o_loss))

    # two ranks, half a batch each, DDP + the two scalar all-reduces
    loss, cnt, _ = loss_of(ddp, x[lo:hi], gt[lo:hi], valid[lo:hi])
    assert (cnt == 0) == (case != "uneven" and rank == 0)
    loss_bwd, loss_global = R.global_batch_loss(loss, cnt)
    loss_bwd.backward()
    assert abs(float(loss_global) - float(ref_loss)) < 1e-5 * abs(float(ref_loss)), (case, float(loss_global), float(ref_loss))
    for p, gr in zip(net.parameters(), ref_grads):
        assert torch.isfinite(p.grad).all(), case
        assert torch.allclose(p.grad, gr, rtol=1e-4, atol=1e-6), (case, float((p.grad - gr).abs().max()))
    # per-rank means averaged with equal weight would NOT be the reference's loss
    naive = torch.tensor([float(torch.nan_to_num(loss.detach()))]); dist.all_reduce(naive); naive /= ws
    assert abs(float(naive) - float(ref_loss)) > 1e-3
seen = torch.ones(1); dist.all_reduce(seen)
assert int(seen.item()) == ws
dist.barrier()
print("rank", rank, "ddp ok")
'''


def test_two_rank_gloo_ddp_loss_and_gradient_parity(tmp_path):
    """DDP gradient parity, 1 rank on the global batch against 2 ranks on half batches with the
    valid-pixel-count normalisation of train_stereo.py:58 (global_batch_loss)."""
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER)
    env = dict(os.environ, RC_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("ddp ok") == 2


def test_bench_input_generator_matches_the_oracles():
    """bench.py carries its own copy of the synthetic-coordinate generator (only its cpu_baseline leg may touch
    oracle/); both copies must produce the same workload."""
    import importlib.util
    import numpy as np
    from oracle.corr_oracle import synthetic_coords as ref
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for adv in (False, True):
        assert np.array_equal(bench.synthetic_coords(2, 9, 40, seed=5, adversarial=adv), ref(2, 9, 40, seed=5, adversarial=adv))
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("from oracle") == 1 and "oracle.reg_torch" in src       # the cpu_baseline leg only


def test_no_kernel_went_to_scratch_memory():
    """The build keeps the compiler's per-kernel resource report (raft-stereo_amd/csrc/*.usage.txt, Makefile).
    Every kernel of the library is written to live in registers; a select between array elements or an array
    reached through a pointer silently turns into scratch memory (twice in round 2: the f32 lookup ran at two
    thirds of its speed, the bf16 lookup at half).  Allowed: the 16-bit build's variants for unaligned rows or
    D != 256, which spill a few registers by design."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raft-stereo_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.usage.txt")))
    sources = sorted(glob.glob(os.path.join(csrc, "*.hip")))
    if not files:
        pytest.skip("library was not built here by its Makefile (no resource reports)")
    assert len(files) == len(sources), "a source file has no resource report: rebuild with make -C raft-stereo_amd/csrc"
    allowed = re.compile(r"build16_kernelIDF16b?_?Li[01]E")      # FAST = 0 / 1
    n_kernels, offenders = 0, []
    for path in files:
        name = None
        for line in open(path):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                n_kernels += 1
                continue
            m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) > 0:
                if allowed.search(name) and int(m.group(1)) <= 64:
                    continue
                offenders.append((os.path.basename(path), name, int(m.group(1))))
    assert n_kernels > 300
    assert not offenders, offenders


def test_convgru_stacked_cache_survives_inference_mode_then_no_grad():
    """ConvGRUHip._stacked refreshes its cached stacked weights in place (so that a captured HIP graph keeps
    reading current values); buffers made under inference_mode cannot be written in place outside it and must be
    replaced instead."""
    import torch
    import raft_stereo_amd as R
    gru = R.ConvGRUHip(4, 6)
    with torch.inference_mode():
        w0, _ = gru._stacked(torch.float32)
    assert w0.is_inference()
    with torch.no_grad():
        gru.convz.weight.add_(1.0)          # bumps the version counter: the cache is stale
        w1, b1 = gru._stacked(torch.float32)
    assert torch.equal(w1, torch.cat([gru.convz.weight, gru.convr.weight], dim=0))
    assert torch.equal(b1, torch.cat([gru.convz.bias, gru.convr.bias], dim=0))
