#!/usr/bin/env python3
"""Timing aid: the regulariser's 3x3x3 32->32 layer on 30x40 planes (BASELINE config 4: D = 96), 128 samples, median of 9;
executed-flop fraction of the fp32 MFMA peak beside it.   usage: python tools/vol30x40_bench.py [samples]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet  # noqa: E402
from multi_view_stereonet_amd.weights import load_weights  # noqa: E402

net = MultiViewStereoNet()
net.load_state_dict(load_weights("demon_45epochs"))
net = net.cuda().eval()
eng = net.engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timed(fn, reps=9):
    ts = []
    for _ in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


torch.manual_seed(0)
for (D, H, W) in ((96, 30, 40), (64, 16, 32)):
    x = torch.randn(N, 32, D, H, W, device="cuda")
    out, st = eng.conv(eng.vf_convs[1], x, want_stats=True)      # (checksums: A/B builds must agree bit for bit)
    print("  checksum out %d stats %d" % (int(out.view(torch.int32).to(torch.int64).sum()), int(st.view(torch.int32).to(torch.int64).sum())))
    med, mn = timed(lambda: eng.conv(eng.vf_convs[1], x, want_stats=True))
    direct = 2.0 * 27 * 32 * 32 * D * H * W * N
    print("%dx%dx%d x %d samples: median %.3f min %.3f ms   executed %.1f TFLOP/s = %.3f of 157.3" %
          (D, H, W, N, med, mn, direct * 4 / 9 / (med * 1e-3) / 1e12, direct * 4 / 9 / (med * 1e-3) / 1e12 / 157.3))
