#!/usr/bin/env python3
"""Timing aid: the level-0 refiner head (4 -> 32 channels) and the level-1 head (36 -> 32), Winograd vs direct."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=4):
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
x0 = torch.randn(128, 4, 256, 512, device="cuda")
x1 = torch.randn(128, 36, 128, 256, device="cuda")
for wino in (True, False):
    eng.winograd = wino
    print("winograd" if wino else "direct  ", "level-0 head %.3f ms   level-1 head %.3f ms" % (
        timed(lambda: eng.conv(eng.refiners[0]["conv0"], x0, want_stats=True)),
        timed(lambda: eng.conv(eng.refiners[1]["conv0"], x1, want_stats=True))))
