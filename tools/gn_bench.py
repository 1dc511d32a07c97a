#!/usr/bin/env python3
"""Timing aid: the regulariser's in-place LReLU(GN(.)) pass on the bench's volume, from records and from finalised statistics."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(N, 32, 64, 16, 32, device="cuda")
for lazy in (True, False):
    y, st = eng.conv(eng.vf_convs[1], x, want_stats=True, lazy_stats=lazy)
    ts = []
    for _ in range(11):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.gn_lrelu(y, st, eng.vf_norms[0], out=y); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:]); ms = ts[len(ts) // 2]
    print("%s statistics (tiles %s): median %.3f  min %.3f ms   %.2f TB/s" %
          ("records  " if lazy else "finalised", getattr(st, "tiles", "-"), ms, ts[0], 8 * y.numel() / ms / 1e9))
