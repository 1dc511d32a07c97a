#!/usr/bin/env python3
"""Timing aid: the extractor's 5x5 stride-2 32 -> 32 layers (bench shapes), Winograd on the four phases against the direct kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
for lvl, (rows, cols) in enumerate(((128, 256), (64, 128), (32, 64)), 1):
    conv = eng.fe_down[lvl]
    x = torch.randn(N, 32, rows, cols, device="cuda")
    res = {}
    for wino in (True, False, True, False):
        eng.winograd_stride2 = wino
        ts = []
        for _ in range(9):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); y, _ = eng.conv(conv, x); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts = sorted(ts[2:]); ms = ts[len(ts) // 2]
        res[wino] = y
        fl = 2.0 * 800 * 32 * y[:, 0].numel()
        print("%dx%dx%d %s median %.3f min %.3f ms  direct-form %.1f TFLOP/s  executed %.1f" %
              (N, rows, cols, "phases-wino" if wino else "direct     ", ms, ts[0], fl / ms / 1e9, fl * (0.49 if wino else 1) / ms / 1e9))
    print("   max |wino - direct| = %.3e  (max |direct| %.3e)" % ((res[True] - res[False]).abs().max().item(), res[False].abs().max().item()))
eng.winograd_stride2 = True
