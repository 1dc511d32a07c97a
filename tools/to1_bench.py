#!/usr/bin/env python3
"""Timing aid: the 32 -> 1 layers (3-D tap GEMM on the regulariser volume, 2-D with the folded last block)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=3):
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
c3 = eng.vf_convs[4]
for N in (2, 256):
    x = torch.randn(N, 32, 64, 16, 32, device="cuda")
    ms = timed(lambda: eng.conv_to1(c3, x))
    print("3-D %d chains: %.3f ms (%.0f GB/s)" % (N, ms, x.numel() * 4 / ms / 1e6))
p = eng.refiners[0]
for B in (1, 128):
    r = torch.randn(B, 32, 256, 512, device="cuda"); x = torch.randn_like(r)
    st = torch.zeros(B, 4, 2, device="cuda"); st[:, :, 1] = 1
    prior = torch.rand(B, 1, 256, 512, device="cuda"); fx = torch.full((B,), 300.0, device="cuda")
    ms = timed(lambda: eng.conv_to1_block(p["final"], r, st, p["res"][5][1], x, prior, fx))
    print("2-D folded block, batch %d: %.3f ms (%.0f GB/s)" % (B, ms, 2 * r.numel() * 4 / ms / 1e6))
    ms = timed(lambda: eng.conv_to1(p["final"], x, prior, fx))
    print("2-D plain, batch %d: %.3f ms (%.0f GB/s)" % (B, ms, r.numel() * 4 / ms / 1e6))
