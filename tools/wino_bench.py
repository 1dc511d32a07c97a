#!/usr/bin/env python3
"""Timing aid: Winograd vs direct form of the level-0 3x3 32->32 layer (plain and with the fused input transform)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 32, 256, 512, device="cuda")
st = torch.zeros(B, 4, 2, device="cuda"); st[:, :, 1] = 1
def timed(fn, reps=3):
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
gf = 2.0 * 9 * 32 * 32 * x[:, 0].numel() / 1e9
eng.winograd_with_input_transform = True
for blk in (0, 1, 2, 3):            # dilations 1, 2, 4, 8
    conv, norm = eng.refiners[0]["res"][blk]
    for wino in (True, False):
        eng.winograd = wino
        ms = timed(lambda: eng.conv(conv, x, want_stats=True))
        ms1 = timed(lambda: eng.conv(conv, x, in_stats=st, in_norm=norm, want_stats=True))
        print("dilation %d" % conv.dilation, "winograd" if wino else "direct  ",
              "%.3f ms  %.1f algorithmic TFLOP/s;  with input transform %.3f ms" % (ms, gf / ms, ms1))
