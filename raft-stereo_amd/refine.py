"""The iterative refinement loop that drives the correlation block: everything RAFTStereo.forward does after
its encoders (core/raft_stereo.py:100-141 of the reference), on this package's operators.

    refine_disparity(update_block, corr_fn, net_list, inp_list, args, iters=12, flow_init=None, test_mode=False)

`corr_fn` is a correlation block (`make_corr_fn(...)`), `update_block` a BasicMultiUpdateBlockHip (or the
reference's own block), `net_list` / `inp_list` the context encoder's hidden states and per-scale context
triples (core/raft_stereo.py:84-88).  Returns what the reference returns: the list of upsampled disparity
predictions, or `(coords1 - coords0, flow_up)` in test mode.  With a reg_hip block and a
BasicMultiUpdateBlockHip the lookup is fused with the motion encoder's first layer.
"""
import torch

from raft_stereo_amd.update import upsample_flow_hip
from raft_stereo_amd.update_block import BasicMultiUpdateBlockHip


def pixel_grid(batch, ht, wd, device):
    """coords0 of core/raft_stereo.py:46-53: channel 0 = x (column), channel 1 = y (row), float32."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def refine_disparity(update_block, corr_fn, net_list, inp_list, args, iters=12, flow_init=None, test_mode=False):
    n, _, h, w = net_list[0].shape
    device = net_list[0].device
    layers, slow_fast = args.n_gru_layers, args.slow_fast_gru
    amp = bool(getattr(args, "mixed_precision", False))
    coords0 = pixel_grid(n, h, w, device)
    coords1 = coords0.clone() if flow_init is None else coords0 + flow_init
    # reg_hip + this package's update block: the lookup is fused with the motion encoder's first layer, in test
    # mode and in training (the backward recomputes the lookup instead of keeping 36 channels per iteration)
    # (a DistributedDataParallel wrapper keeps the block under .module)
    fuse = isinstance(getattr(update_block, "module", update_block), BasicMultiUpdateBlockHip) \
        and hasattr(corr_fn, "lookup_convc1") and corr_fn.radius <= 7
    net_list = list(net_list)
    predictions, flow_up = [], None
    if fuse and test_mode and not torch.is_grad_enabled() and hasattr(corr_fn, "update_lookup_convc1"):
        # inference: an iteration's glue costs no launch of its own -- `coords1 += (delta_x, 0)` of the previous
        # iteration, `flow = coords1 - coords0` and relu(convc1(lookup(coords1))) are one kernel
        # (raft_stereo.py:109-111, :120-123; update.py:78), bit-identical to the separate steps
        enc = getattr(update_block, "module", update_block).encoder
        delta_flow = None
        for itr in range(iters):
            coords1, flow, act = corr_fn.update_lookup_convc1(coords1, delta_flow, enc.convc1.weight, enc.convc1.bias)
            with torch.autocast("cuda", enabled=amp):
                if layers == 3 and slow_fast:
                    net_list = update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
                if layers >= 2 and slow_fast:
                    net_list = update_block(net_list, inp_list, iter32=layers == 3, iter16=True, iter08=False,
                                            update=False)
                net_list, up_mask, delta_flow = update_block(net_list, inp_list, flow=flow, iter32=layers == 3,
                                                             iter16=layers >= 2, act=act)
        # the last iteration's update, in the reference's order of operations (once per pass)
        coords1 = coords1 + torch.cat([delta_flow[:, :1], torch.zeros_like(delta_flow[:, :1])], dim=1)
        final = coords1 - coords0
        return final, upsample_flow_hip(final, up_mask, args.n_downsample, channels=1)
    for itr in range(iters):
        coords1 = coords1.detach()                                     # raft_stereo.py:109
        flow = coords1 - coords0
        lookup = dict(corr_fn=corr_fn, coords=coords1) if fuse else dict(corr=corr_fn(coords1))
        with torch.autocast("cuda", enabled=amp):
            if layers == 3 and slow_fast:                              # low-resolution GRU alone
                net_list = update_block(net_list, inp_list, iter32=True, iter16=False, iter08=False, update=False)
            if layers >= 2 and slow_fast:                              # low- and mid-resolution GRUs
                net_list = update_block(net_list, inp_list, iter32=layers == 3, iter16=True, iter08=False,
                                        update=False)
            net_list, up_mask, delta_flow = update_block(net_list, inp_list, flow=flow, iter32=layers == 3,
                                                         iter16=layers >= 2, **lookup)
        # stereo: the update moves along the epipolar line only (raft_stereo.py:119-123)
        coords1 = coords1 + torch.cat([delta_flow[:, :1], torch.zeros_like(delta_flow[:, :1])], dim=1)
        if test_mode and itr < iters - 1:
            continue
        flow_up = upsample_flow_hip(coords1 - coords0, up_mask, args.n_downsample, channels=1)   # :133-134
        predictions.append(flow_up)
    if test_mode:
        return coords1 - coords0, flow_up
    return predictions
