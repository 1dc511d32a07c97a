#!/usr/bin/env python3
"""Soak of the slab plan of the banded chain: REPS launches of N chains per grid, every output compared bit for bit with the
first launch's, the status word read after each (a hand-off that timed out would show here).
    python tools/slab_soak.py [reps] [chains]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
net.options.chain_form = "banded"
for rows, cols, D in ((30, 40, 96), (32, 64, 128)):
    g = torch.Generator().manual_seed(rows)
    src4 = (torch.rand(N, 3, rows, cols, generator=g) * 2 - 1).cuda()
    Hinc = torch.eye(3).repeat(N, D, 1, 1)
    Hinc[:, 1:, 0, 2] = torch.rand(N, D - 1, generator=g) * 1.6 - 0.8      # about a pixel per plane, both directions
    Hinc[:, 1:, 1, 2] = torch.rand(N, D - 1, generator=g) * 1.2 - 0.6
    H = torch.eye(3).repeat(N, D, 1, 1)
    for d in range(1, D):
        H[:, d] = H[:, d - 1] @ Hinc[:, d]
    F0 = torch.randn(N, 32, rows, cols, generator=g).cuda()
    FL = torch.randn(max(1, N // 2), 32, rows, cols, generator=g).cuda()
    H, Hinc = H.cuda(), Hinc.cuda()
    c0, m0, _ = eng.incremental_cost_volume(src4, H, Hinc, F0, FL)
    torch.cuda.synchronize()
    groups = eng.lib.mvsn_incremental_cost_volume_banded_groups(N, rows, cols)
    bad, t0 = 0, time.time()
    for i in range(reps):
        c, m, _ = eng.incremental_cost_volume(src4, H, Hinc, F0, FL)
        torch.cuda.synchronize()
        st = eng.chain_status()
        if st != 0 or not torch.equal(c, c0) or not torch.equal(m, m0):
            bad += 1
            print(f"  {rows}x{cols} launch {i}: status {st}, cost equal {torch.equal(c, c0)}")
    print(f"{rows}x{cols} D={D}: {reps} launches of {N} chains ({groups} workgroups per chain), {bad} deviating, finite "
          f"{bool(torch.isfinite(c0).all())}, {(time.time() - t0) / reps * 1e3:.2f} ms per launch incl. the comparison")
