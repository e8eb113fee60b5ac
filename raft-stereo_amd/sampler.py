"""Functional twin of the reference's `corr_sampler` extension module
(sampler/sampler.cpp:48-51): `forward(volume, coords, radius) -> [corr]` and
`backward(volume, coords, corr_grad, radius) -> [volume_grad]` on the caller's own row-major
single-level volume [B,H,W1,W2].  Same argument checks as sampler.cpp:20-22 (device tensor,
contiguous), raised as RuntimeError."""
import torch

from raft_stereo_amd import _lib
from raft_stereo_amd._lib import lib, check

_DT = {torch.float32: _lib.RC_F32, torch.float16: _lib.RC_F16, torch.bfloat16: _lib.RC_BF16}


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def forward(volume, coords, radius):
    _check_input(volume, "volume")
    _check_input(coords, "coords")
    if volume.dtype not in _DT:
        raise RuntimeError(f"corr_sampler: unsupported volume dtype {volume.dtype}")
    if coords.dtype != torch.float32:
        raise RuntimeError("coords must be float32")
    if volume.dim() != 4 or coords.dim() != 4:
        raise RuntimeError("volume and coords must be 4-dimensional")
    B, H, W1, W2 = volume.shape
    if coords.shape[0] != B or coords.shape[1] < 1 or coords.shape[2] != H or coords.shape[3] != W1:
        raise RuntimeError("coords shape does not match volume")
    if coords.device != volume.device:
        raise RuntimeError("volume and coords must live on the same device")
    corr = torch.empty((B, 2 * radius + 1, H, W1), dtype=volume.dtype, device=volume.device)
    with torch.cuda.device(volume.device):
        check(lib.rc_sampler_fwd(volume.data_ptr(), coords.data_ptr(), coords.shape[1], corr.data_ptr(),
                                 B, H, W1, W2, radius, _DT[volume.dtype], _stream(volume.device)),
              "rc_sampler_fwd")
    return [corr]


def backward(volume, coords, corr_grad, radius):
    _check_input(volume, "volume")
    _check_input(coords, "coords")
    _check_input(corr_grad, "corr_grad")
    if volume.dtype not in _DT:
        raise RuntimeError(f"corr_sampler: unsupported volume dtype {volume.dtype}")
    if corr_grad.dtype != volume.dtype:
        raise RuntimeError("corr_grad dtype must match volume dtype")
    if coords.dtype != torch.float32:
        raise RuntimeError("coords must be float32")
    if volume.dim() != 4 or coords.dim() != 4 or corr_grad.dim() != 4:
        raise RuntimeError("volume, coords and corr_grad must be 4-dimensional")
    B, H, W1, W2 = volume.shape
    if coords.shape[0] != B or coords.shape[1] < 1 or coords.shape[2] != H or coords.shape[3] != W1:
        raise RuntimeError("coords shape does not match volume")
    if tuple(corr_grad.shape) != (B, 2 * radius + 1, H, W1):
        raise RuntimeError(f"corr_grad must be [B, 2*radius+1, H, W1] = {(B, 2 * radius + 1, H, W1)}, "
                           f"got {tuple(corr_grad.shape)}")
    if coords.device != volume.device or corr_grad.device != volume.device:
        raise RuntimeError("volume, coords and corr_grad must live on the same device")
    grad = torch.empty_like(volume)
    with torch.cuda.device(volume.device):
        check(lib.rc_sampler_bwd(coords.data_ptr(), coords.shape[1], corr_grad.data_ptr(), grad.data_ptr(),
                                 B, H, W1, W2, radius, _DT[volume.dtype], _stream(volume.device)),
              "rc_sampler_bwd")
    return [grad]
