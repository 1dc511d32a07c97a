#!/usr/bin/env python3
"""Time mvsn_homography_warp on full-resolution frames (3x256x512, one plane) with the homographies the forward really
uses (plane 0 of the bench's seeded frames), from 16 to 512 frames per call: up to ~128 frames the 256 MB memory-side cache
serves part of the traffic, beyond that the kernel runs at its HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
dev = torch.device("cuda")
_, inp = bench.make_inputs(8, 0, dev)
cap = {}
net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 64, True, [True] * 5, capture=cap)
H0 = cap["H_lvl0_plane0"]          # (S*B, 1, 3, 3)
print("H0[0]", H0[0, 0].cpu().numpy().round(4).tolist())
print("H0[9]", H0[9, 0].cpu().numpy().round(4).tolist())
img = torch.cat([p[0] for p in inp["right_image_pyr"]], 0)   # (16,3,256,512)
for B in (16, 32, 64, 128, 256, 512):
    im = img.repeat(B // 16, 1, 1, 1).contiguous(); H = H0.repeat(B // 16, 1, 1, 1).contiguous()
    Hid = torch.eye(3, device=dev).repeat(B, 1, 1, 1); Hid[:, :, 0, 2] = 3.3; Hid[:, :, 0, 1] = 0.01
    for name, HH in (("real", H),):
        for _ in range(3): eng.homography_warp(im, HH)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): out = eng.homography_warp(im, HH)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        nbytes = im.numel() * 8 + B * 256 * 512
        msk = out[1].float().mean().item() if isinstance(out, tuple) else float("nan")
        print(f"B={B} {name}: {ms*1e3:.1f} us, {nbytes/ms/1e9:.2f} TB/s algorithmic, outside fraction {msk:.3f}")
