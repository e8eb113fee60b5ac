"""The caller on the far side of the correlation lookup: the multi-scale update block, wired from this
package's kernels (core/update.py:64-138 of the reference).

    BasicMultiUpdateBlockHip(args, hidden_dims)     same constructor and parameter names as
                                                    BasicMultiUpdateBlock (a reference state_dict loads as is)
        .forward(net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True,
                 corr_fn=None, coords=None, act=None)

What runs in the HIP library: the three ConvGRUs' gate arithmetic (ConvGRUHip) and, when the caller hands
over the correlation block and the coordinates instead of a looked-up `corr` tensor, the lookup fused with
the motion encoder's first layer (`corr_fn.lookup_convc1`), and the pooling / bilinear resizing between the
scales.  The dense convolutions are torch's, as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from raft_stereo_amd.update import ConvGRUHip, _resample_ok, conv_bias_act, interp_hip, pool2x_hip


def _down2(x):
    if _resample_ok(x):                                                 # the library's kernel (and its adjoint)
        return pool2x_hip(x)
    return F.avg_pool2d(x, 3, stride=2, padding=1)                      # update.py:87-88


def _resize_like(x, ref):
    if _resample_ok(x):
        return interp_hip(x, ref)
    return F.interpolate(x, ref.shape[2:], mode="bilinear", align_corners=True)   # update.py:93-95


class MotionEncoderHip(nn.Module):
    """core/update.py:64-85.  `forward(flow, corr)` as the reference; `forward(flow, corr_fn=..., coords=...)`
    never materialises the L*(2r+1)-channel lookup result."""

    def __init__(self, args):
        super().__init__()
        planes = args.corr_levels * (2 * args.corr_radius + 1)
        self.convc1 = nn.Conv2d(planes, 64, 1)
        self.convc2 = nn.Conv2d(64, 64, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv = nn.Conv2d(128, 126, 3, padding=1)

    def forward(self, flow, corr=None, corr_fn=None, coords=None, act=None):
        """`act`: relu(convc1(lookup)) already computed by the caller (the refinement loop folds the coordinate
        update into the same kernel: CorrBlock1DHip.update_lookup_convc1)."""
        if corr is None:
            if act is not None:
                c = act
            elif corr_fn is None or coords is None:
                raise ValueError("pass either corr, act or (corr_fn, coords)")
            else:
                c = corr_fn.lookup_convc1(coords, self.convc1.weight, self.convc1.bias)
            if not torch.is_autocast_enabled():       # a 16-bit pyramid under an fp32 update block
                c = c.to(self.convc2.weight.dtype)
        else:
            c = F.relu(self.convc1(corr))
        c = conv_bias_act(self.convc2, c, relu=True)
        f = conv_bias_act(self.convf2, conv_bias_act(self.convf1, flow, relu=True), relu=True)
        both = conv_bias_act(self.conv, torch.cat([c, f.to(c.dtype)], dim=1), relu=True)
        # under autocast the reference's concatenation promotes to fp32 and the GRU's convolutions cast back;
        # casting the two flow planes instead gives the same values without the fp32 round trip of 126 planes
        return torch.cat([both, flow.to(both.dtype) if torch.is_autocast_enabled() else flow], dim=1)


class FlowHeadHip(nn.Module):
    """core/update.py:6-14 (two convolutions; nothing pointwise worth a kernel)."""

    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, output_dim, 3, padding=1)

    def forward(self, x):
        return conv_bias_act(self.conv2, conv_bias_act(self.conv1, x, relu=True))


class BasicMultiUpdateBlockHip(nn.Module):
    """core/update.py:97-138 with ConvGRUHip at the three scales."""

    def __init__(self, args, hidden_dims=()):
        super().__init__()
        self.args = args
        layers = args.n_gru_layers
        self.encoder = MotionEncoderHip(args)
        self.gru08 = ConvGRUHip(hidden_dims[2], 128 + hidden_dims[1] * (layers > 1))
        self.gru16 = ConvGRUHip(hidden_dims[1], hidden_dims[0] * (layers == 3) + hidden_dims[2])
        self.gru32 = ConvGRUHip(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHeadHip(hidden_dims[2], hidden_dim=256, output_dim=2)
        f = 2 ** args.n_downsample
        self.mask = nn.Sequential(nn.Conv2d(hidden_dims[2], 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, f * f * 9, 1, padding=0))

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True,
                corr_fn=None, coords=None, act=None):
        layers = self.args.n_gru_layers
        if iter32:
            net[2] = self.gru32(net[2], *inp[2], _down2(net[1]))
        if iter16:
            extra = [_down2(net[0])] + ([_resize_like(net[2], net[1])] if layers > 2 else [])
            net[1] = self.gru16(net[1], *inp[1], *extra)
        if iter08:
            motion = self.encoder(flow, corr, corr_fn=corr_fn, coords=coords, act=act)
            extra = [motion] + ([_resize_like(net[1], net[0])] if layers > 1 else [])
            net[0] = self.gru08(net[0], *inp[0], *extra)
        if not update:
            return net
        delta_flow = self.flow_head(net[0])
        mask = .25 * conv_bias_act(self.mask[2], conv_bias_act(self.mask[0], net[0], relu=True))   # update.py:136-137
        return net, mask, delta_flow
