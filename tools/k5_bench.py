#!/usr/bin/env python3
"""Time the extractor's three 32 -> 32 5x5 stride-2 layers at the bench batch (384 frames)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
tot = 0.0
for i, (r, c) in enumerate(((128, 256), (64, 128), (32, 64))):
    x = torch.randn(384, 32, r, c, device="cuda")
    for _ in range(3): eng.conv(eng.fe_down[i + 1], x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): eng.conv(eng.fe_down[i + 1], x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10; tot += ms
    print(f"{r}x{c}: {ms:.3f} ms  {2*32*25*32*(r//2)*(c//2)*384/ms/1e9:.1f} TFLOP/s")
print(f"total {tot:.3f} ms")
