#!/usr/bin/env python3
"""Tuning aid: s_memtime phase stamps of one mid-launch wave of the Winograd conv kernel
(build with MVSN_HIPCC_FLAGS=-DMVSN_WN_STAMPS).  Usage: wino_phases.py [batch] [vol|vol30]
(vol: the 3x3x3 regulariser layer on (2*batch, 32, 64, 16, 32) instead of the level-0 refiner layer)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
eng.lib.mvsn_debug_set_wino_stamps.argtypes = [ctypes.c_void_p]
assert eng.lib.mvsn_debug_set_wino_stamps(dbg.data_ptr()) == 0
if len(sys.argv) > 2 and sys.argv[2] == "vol":
    conv = eng.vf_convs[1]
    x = torch.randn(2 * B, 32, 64, 16, 32, device="cuda")
elif len(sys.argv) > 2 and sys.argv[2] == "vol30":      # the 30x40 planes of BASELINE config 4 (wide / rolling strips)
    conv = eng.vf_convs[1]
    x = torch.randn(B, 32, 96, 30, 40, device="cuda")
else:
    conv, norm = eng.refiners[0]["res"][0]
    x = torch.randn(B, 32, 256, 512, device="cuda")
for it in range(3):
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if "xf" in sys.argv:   # with the fused input transform (MODE 1)
        st_in = torch.zeros(x.shape[0], 4, 2, device="cuda"); st_in[:, :, 1] = 1
        a.record(); r, st = eng.conv(conv, x, in_stats=st_in, in_norm=eng.vf_norms[0], want_stats=True); b.record()
    else:
        a.record(); r, st = eng.conv(conv, x, want_stats=True); b.record()
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    n = sum(1 for v in t if v)
    d = [t[i + 1] - t[i] for i in range(n - 1)]
    nsteps = 12 if (len(sys.argv) > 2 and sys.argv[2].startswith("vol")) else 4
    print("launch %.3f ms; stamps from the workgroup's 4th tile on (kept in LDS, copied out at the end)" % a.elapsed_time(b))
    per = 3 * nsteps + 1   # (landed, barrier, multiplies + next transform) per step, then the tile's epilogue
    for k in range(2, len(d), per):
        tile = d[k:k + per]
        if len(tile) < per:
            break
        steps = [tile[i:i + 3] for i in range(0, 3 * nsteps, 3)]
        print("  tile: total %d cycles; epilogue %d; steps [landed, barrier, multiply]:" % (sum(tile), tile[-1]), steps)
