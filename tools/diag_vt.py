"""Diagnostic: where rc_lookup_fwd_vt differs from rc_lookup_fwd rounded."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raft_stereo_amd as R
from oracle import corr_oracle as O
SHAPE = tuple(int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (2, 64, 3, 45, 45)
for dtype in (torch.float16, torch.bfloat16):
    for mode in ("exact", "gridsample"):
        B, D, H, W1, W2 = SHAPE
        g = torch.Generator(device="cuda").manual_seed(B * 1000 + W1)
        f1 = torch.randn((B, D, H, W1), generator=g, device="cuda").to(dtype)
        f2 = torch.randn((B, D, H, W2), generator=g, device="cuda").to(dtype)
        coords = torch.from_numpy(O.synthetic_coords(B, H, W1, seed=5)).cuda()
        coords[:, 0] = coords[:, 0] * (W2 / W1) + torch.rand(coords[:, 0].shape, generator=g, device="cuda") * 9 - 4
        coords[0, 0, 0, :8] = torch.tensor([-40., -1.5, -0.25, 0., W2 - 1.0, W2 - 0.5, W2 + 3.25, 1e6], device="cuda")
        ref = R.CorrBlock1DHip(f1, f2, coord_mode=mode)
        blk = R.CorrBlock1DHip(f1, f2, coord_mode=mode, out_dtype=dtype)
        b, a = blk(coords), ref(coords)
        print("volumes equal", torch.equal(ref.volume, blk.volume), "nan in a", int(a.isnan().sum()), "nan in b", int(b.isnan().sum()))
        ar = a.to(dtype)
        ne = (ar != b)
        print(dtype, mode, "mismatch", int(ne.sum()), "of", ne.numel())
        idx = ne.nonzero()[:6]
        for i in idx:
            i = tuple(i.tolist())
            print("  ", i, "x %.9g" % coords[i[0],0,i[2],i[3]].item(), "f32 %.9g" % a[i].item(), "rounded %.9g" % ar[i].item(), "vt %.9g" % b[i].item())
