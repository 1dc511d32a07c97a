#!/usr/bin/env python3
"""Timing aid: the extractor's 5x5 stride-2 3 -> 32 head on the bench's frames (median of 9, ms, TB/s, TFLOP/s)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
conv = eng.fe_down[0]
for N in [int(a) for a in sys.argv[1:]] or [768, 384, 6]:
    x = torch.randn(N, 3, 256, 512, device="cuda")
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = eng.conv(conv, x); b.record(); y = y[0] if isinstance(y, tuple) else y; torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:]); ms = ts[len(ts) // 2]
    ref = torch.nn.functional.conv2d(x[:2].double(), net.left_feature_extractor.conv0.weight.double(), None, 2, 2)
    err = (y[:2].double() - ref).abs().max().item()
    print("N=%4d  median %.3f  min %.3f ms   %.2f TB/s  %.1f TFLOP/s   max err vs fp64 %.2e" %
          (N, ms, ts[0], (x.numel() + y.numel()) * 4 / ms / 1e9, 2 * 75 * y.numel() / ms / 1e9, err))
