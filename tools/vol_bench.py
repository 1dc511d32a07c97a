#!/usr/bin/env python3
"""Timing aid: the regulariser's 3x3x3 32->32 layer (volume Winograd form) and the level-0 2-D layers, median of 9."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=9):
    ts = []
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]
xv = torch.randn(256, 32, 64, 16, 32, device="cuda")
x2 = torch.randn(128, 32, 256, 512, device="cuda")
st = torch.zeros(128, 4, 2, device="cuda"); st[:, :, 1] = 1
print("volume   median %.3f  min %.3f ms" % timed(lambda: eng.conv(eng.vf_convs[1], xv, want_stats=True)))
for blk in (0, 1, 2, 3):
    conv, norm = eng.refiners[0]["res"][blk]
    print("2-D d%d   median %.3f  min %.3f ms" % ((conv.dilation,) + timed(lambda: eng.conv(conv, x2, want_stats=True))),
          "  with input transform  median %.3f  min %.3f ms" % timed(lambda: eng.conv(conv, x2, in_stats=st, in_norm=norm, want_stats=True)))
