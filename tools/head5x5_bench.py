#!/usr/bin/env python3
"""Timing aid: the extractor's first layer (5x5 stride 2, 3 -> 32) on the bench's 768 frames (256 images x 3 views)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=7):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best
for n in (768, 3):
    x = torch.randn(n, 3, 256, 512, device="cuda")
    ms = timed(lambda: eng.conv(eng.fe_down[0], x))
    fl = 2.0 * 75 * 32 * 128 * 256 * n
    print("%d frames: %.3f ms  %.1f TFLOP/s (%.2f of fp32 MFMA peak)  %.0f GB/s" % (n, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, (x.numel() + n * 32 * 128 * 256) * 4 / ms / 1e6))
