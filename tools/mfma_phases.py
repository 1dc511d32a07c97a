#!/usr/bin/env python3
"""Tuning aid: per-phase s_memtime totals of one workgroup (thread 0) of the register-staged conv kernel on the
regulariser's 3-D layer in its direct form, or -- argument "5x5" -- the extractor's 32->32 5x5 stride-2 layer
(build with MVSN_HIPCC_FLAGS=-DMVSN_DMA_STAMPS).  Usage: mfma_phases.py [samples] [5x5]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
eng.lib.mvsn_debug_set_dma_stamps.argtypes = [ctypes.c_void_p]
assert eng.lib.mvsn_debug_set_dma_stamps(dbg.data_ptr()) == 0
eng.winograd_volume = False
five = len(sys.argv) > 2 and sys.argv[2] == "5x5"
x = torch.randn(384, 32, 128, 256, device="cuda") if five else torch.randn(N, 32, 64, 16, 32, device="cuda")
st = torch.zeros(N, 4, 2, device="cuda"); st[:, :, 1] = 1
names = ["prologue", "barrier 1", "transform + LDS writes", "barrier 2", "issue next loads", "MFMAs", "epilogue"]
for it in range(3):
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if five:
        a.record(); eng.conv(eng.fe_down[1], x); b.record()
    else:
        a.record(); eng.conv(eng.vf_convs[1], x, in_stats=st, in_norm=eng.vf_norms[0], want_stats=True); b.record()
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()[32:40]
    print("launch %.3f ms; thread 0 total %d cycles:" % (a.elapsed_time(b), t[7]), ", ".join("%s %d" % (n, v) for n, v in zip(names, t[:7])))
