#!/usr/bin/env python3
"""Tuning aid: phase stamps of one workgroup of the bf16x3 conv kernel (3-D and 2-D)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# needs a tuning build: MVSN_HIPCC_FLAGS=-DMVSN_BX_STAMPS python -m multi_view_stereonet_amd.build --force
import ctypes
from multi_view_stereonet_amd import _native
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
ctypes.CDLL(_native.library_path()).mvsn_debug_set_bx_stamps(ctypes.c_void_p(dbg.data_ptr()))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.multi_view_stereonet import _Conv
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine(); eng.conv_precision = "bf16x3"
for name, x, c in (("3d", torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32, 64, 16, 32, device="cuda"), eng.vf_convs[1]),
                   ("2d", torch.randn(int(sys.argv[2]) if len(sys.argv) > 2 else 32, 32, 256, 512, device="cuda"), eng.refiners[0]["res"][0][0])):
    for _ in range(2):
        dbg.zero_(); eng.conv(c, x, want_stats=True); torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    n = sum(1 for v in t if v)
    d = [t[i + 1] - t[i] for i in range(n - 1)]
    print(name, "stamps", n, "total", t[n - 1] - t[0], "prologue", d[0], "then [wait_prev, commit, barrier, mfma(+issue), (2-D: epilogue)] per stage, last two = tail:")
    for i in range(1, len(d), 5):
        print("   ", d[i:i + 5])
