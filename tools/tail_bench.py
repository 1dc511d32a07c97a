#!/usr/bin/env python3
"""Timing aid: the refiner's folded 32 -> 1 tail (mvsn_conv_to1_block) at level 0 / 1 sizes, median of 9."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
for (n, rows, cols) in ((128, 256, 512), (256, 128, 256), (256, 64, 128)):
    r = torch.randn(n, 32, rows, cols, device="cuda"); x = torch.randn(n, 32, rows, cols, device="cuda")
    st = torch.zeros(n, 4, 2, device="cuda"); st[:, :, 1] = 1
    prior = torch.rand(n, 1, rows, cols, device="cuda"); fx = torch.rand(n, device="cuda") * 50 + 10
    p = eng.refiners[0]; final = p["final"]; norm = p["res"][-1][1]
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.conv_to1_block(final, r, st, norm, x, prior, fx); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:]); ms = ts[len(ts) // 2]
    print("%dx%dx%d  median %.3f  min %.3f ms   %.2f TB/s algorithmic" %
          (n, rows, cols, ms, ts[0], (2 * r.numel() + 2 * prior.numel()) * 4 / ms / 1e9))
