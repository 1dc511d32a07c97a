#!/usr/bin/env python3
"""Tuning aid: s_memtime phase stamps of one mid-launch wave of the LDS-DMA fp32 conv kernel
(build with MVSN_HIPCC_FLAGS=-DMVSN_DMA_STAMPS).  Usage: dma_phases.py [batch] [rows] [cols] [dilation-index]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.weights import load_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 512
blk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
lib = ctypes.CDLL(_native.LIB_PATH) if hasattr(_native, "LIB_PATH") else eng.lib
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
lib.mvsn_debug_set_dma_stamps.argtypes = [ctypes.c_void_p]
assert lib.mvsn_debug_set_dma_stamps(dbg.data_ptr()) == 0
conv, norm = eng.refiners[0]["res"][blk]
x = torch.randn(B, 32, rows, cols, device="cuda")
for it in range(3):
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r, st = eng.conv(conv, x, want_stats=True); b.record(); torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    wall = (t[63] - t[62]) / 100.0   # us
    t = t[:60]
    n = sum(1 for v in t if v)
    d = [t[i + 1] - t[i] for i in range(n - 1)]
    print("shader clock %.0f MHz;" % ((t[n - 1] - t[0]) / wall), end=" ")
    print("launch %.3f ms, wave total %d cycles; prologue %d; per chunk [landed, barrier, issue, mfma]; epilogue %d" %
          (a.elapsed_time(b), t[n - 1] - t[0], d[0], d[-1]))
    for i in range(1, len(d) - 1, 4):
        print("   ", d[i:i + 4])
