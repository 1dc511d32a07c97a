#!/usr/bin/env python3
"""Tuning aid: the life of a ONE-TILE-PER-WORKGROUP Winograd launch (batch 1), s_memtime stamps of one workgroup
(build with MVSN_HIPCC_FLAGS=-DMVSN_WN_STAMPS): entry -> prologue -> 4 x [landed, barrier, multiplies] -> epilogue."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
eng.lib.mvsn_debug_set_wino_stamps.argtypes = [ctypes.c_void_p]
assert eng.lib.mvsn_debug_set_wino_stamps(dbg.data_ptr()) == 0
for (rows, cols), blk in (((256, 512), 0), ((64, 128), 0), ((256, 512), 3)):
    conv, norm = eng.refiners[0]["res"][blk]
    x = torch.randn(1, 32, rows, cols, device="cuda")
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    for it in range(3):
        big.zero_()                       # the layer's weights are cold in L2, as inside a forward
        dbg.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r, st = eng.conv(conv, x, want_stats=True); b.record()
        torch.cuda.synchronize()
        t = [v for v in dbg.cpu().tolist() if v]
        d = [t[i + 1] - t[i] for i in range(len(t) - 1)]
        print(f"{rows}x{cols} dilation {conv.dilation}: launch {a.elapsed_time(b) * 1e3:.1f} us, stamps {len(t)}, "
              f"in-kernel {t[-1] - t[0]} cycles: prologue {d[0]}; steps [landed, barrier, multiply] "
              f"{[d[i:i + 3] for i in range(1, len(d) - 1, 3)]}; tail {d[-1]}")
