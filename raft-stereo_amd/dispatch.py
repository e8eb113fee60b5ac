"""Selection of the correlation implementation and work partitioning.

Counterpart of the dispatch in the reference's model driver (core/raft_stereo.py:90-100): a
string in `args.corr_implementation` picks the block class, `reg`/`alt` style blocks get fp32
feature maps, the native ones keep the autocast dtype.  The reference's four names are kept as
keys so that a caller can see what is (not) provided here; only the *_hip entries are
implemented by this package.
"""
import torch

from raft_stereo_amd.corr import CorrBlock1DHip, AltCorrBlock1DHip

HIP_CORR_CHOICES = ("reg_hip", "alt_hip")

# name -> (block class, cast feature maps to fp32 unless mixed precision is on)
CORR_IMPLEMENTATIONS = {
    "reg_hip": (CorrBlock1DHip, True),
    "alt_hip": (AltCorrBlock1DHip, True),
}


def wants_autocast(corr_implementation):
    """evaluate_stereo.py:230 enables autocast iff the name ends with `_cuda`; the HIP blocks
    take 16-bit feature maps the same way."""
    return corr_implementation.endswith("_cuda") or corr_implementation.endswith("_hip")


def make_corr_fn(corr_implementation, fmap1, fmap2, radius=4, num_levels=4, mixed_precision=False, out_dtype=None):
    """What core/raft_stereo.py:90-100 does for the names this package owns.  Unknown names
    raise (the reference falls through its if/elif and dies with UnboundLocalError).
    out_dtype (reg_hip with 16-bit feature maps only): the taps in the volume's dtype, as the reference's
    reg_cuda block returns them (core/corr.py:44-51), instead of fp32."""
    if corr_implementation not in CORR_IMPLEMENTATIONS:
        raise ValueError(f"corr_implementation {corr_implementation!r} is not provided by raft_stereo_amd; "
                         f"choices: {sorted(CORR_IMPLEMENTATIONS)}")
    cls, to_fp32 = CORR_IMPLEMENTATIONS[corr_implementation]
    if to_fp32 and not mixed_precision:
        fmap1, fmap2 = fmap1.float(), fmap2.float()
    if out_dtype is not None:
        if cls is not CorrBlock1DHip:
            raise ValueError(f"out_dtype is a reg_hip option, not one of {corr_implementation!r}")
        return cls(fmap1, fmap2, radius=radius, num_levels=num_levels, out_dtype=out_dtype)
    return cls(fmap1, fmap2, radius=radius, num_levels=num_levels)


def shard_pairs(n_pairs, world_size, rank):
    """Contiguous batch shard [lo, hi) of image pairs for one rank (no collective: pairs are
    independent)."""
    base, rem = divmod(n_pairs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(height, world_size, rank):
    """Row-slab [lo, hi) of the feature maps for one rank: 1-D correlation never crosses image
    rows (core/corr.py:120, sampler_kernel.cu:50), so slabs need no halo and no exchange."""
    return shard_pairs(height, world_size, rank)


def global_batch_loss(loss_local, count_local, group=None):
    """Training across ranks (one process per GPU, DistributedDataParallel over RCCL instead of the reference's
    nn.DataParallel, train_stereo.py:137): the reference computes sequence_loss on the gathered GLOBAL batch, a mean
    over all its valid pixels (train_stereo.py:58, :189), so ranks with different numbers of valid pixels must not be
    averaged with equal weight.  Given this rank's loss (a mean over its own `count_local` valid pixels) this
    all-reduces the pixel count and the L1 numerator (two scalars) and returns

        loss_for_backward  = loss_local * count_local / count_global * world_size
                             (DDP averages the ranks' gradients, i.e. divides by world_size again, so the averaged
                             gradient is exactly the gradient of the global-batch loss)
        loss_global        = sum over ranks of loss_local * count_local / count_global   (detached; for logging)

    A rank without a single valid pixel (its local mean is over nothing: NaN) contributes zero -- its loss is
    replaced by 0 * nan_to_num(loss), which keeps its autograd graph so that DDP still sees every parameter.
    With one rank (or torch.distributed not initialised) it returns the loss unchanged."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return loss_local, loss_local.detach()
    world = dist.get_world_size(group)
    cnt = torch.tensor([float(count_local)], dtype=torch.float64, device=loss_local.device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    share = float(count_local) / max(float(cnt.item()), 1.0)
    if count_local == 0:
        loss_local = torch.nan_to_num(loss_local, nan=0.0, posinf=0.0, neginf=0.0) * 0.0
    num = (loss_local.detach().double() * share).reshape(1).clone()
    dist.all_reduce(num, op=dist.ReduceOp.SUM, group=group)
    return loss_local * (share * world), num[0].to(loss_local.dtype)
