#!/usr/bin/env python3
"""Debug aid: banded vs plane-resident Winograd chain on the same inputs; prints the first plane / channel / row that differs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
N, D, rows, cols = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 6, 16, 32
g = torch.Generator().manual_seed(0)
src4 = (torch.rand(N, 3, rows, cols, generator=g) * 2 - 1).cuda()
Hinc = torch.eye(3).repeat(N, D, 1, 1); Hinc[:, 1:, 0, 2] = 0.7; Hinc[:, 1:, 1, 2] = -0.4
H = torch.eye(3).repeat(N, D, 1, 1)
for d in range(1, D):
    H[:, d] = H[:, d - 1] @ Hinc[:, d]
F0 = torch.randn(N, 32, rows, cols, generator=g).cuda(); FL = torch.randn(N, 32, rows, cols, generator=g).cuda()
out = {}
for form in ("winograd", "banded"):
    net.options.chain_form = form
    c, m, f = eng.incremental_cost_volume(src4, H.cuda(), Hinc.cuda(), F0, FL, want_features=True)
    out[form] = (c.cpu(), m.cpu(), f.cpu())
print("status", eng.chain_status(), "mask equal", torch.equal(out["winograd"][1], out["banded"][1]))
fa, fb = out["winograd"][2], out["banded"][2]
for d in range(D):
    diff = (fa[0, :, d] - fb[0, :, d]).abs()
    print("plane", d, "max diff", float(diff.max()), "rows with diff > 1e-3:", sorted(set((diff > 1e-3).nonzero()[:, 1].tolist())),
          "channels:", sorted(set((diff > 1e-3).nonzero()[:, 0].tolist()))[:8], "cols:", sorted(set((diff > 1e-3).nonzero()[:, 2].tolist()))[:8])
