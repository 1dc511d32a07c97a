#!/usr/bin/env python3
"""Tuning aid: per-phase cycle counts of the slab plan of the banded chain (chain_slab_kernel: block 0, thread 0) for
steps 2..5.  Needs a tuning build: MVSN_HIPCC_FLAGS=-DMVSN_CHAIN_STAMPS python -m multi_view_stereonet_amd.build --force
    python tools/slab_phases.py [chains] [rows cols]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
dbg = torch.zeros(256, dtype=torch.int64, device="cuda")
ctypes.CDLL(_native.library_path()).mvsn_debug_set_chain_stamps(ctypes.c_void_p(dbg.data_ptr()))
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 64)
D = 32
g = torch.Generator().manual_seed(0)
src4 = (torch.rand(N, 3, rows, cols, generator=g) * 2 - 1).cuda()
H = torch.eye(3).repeat(N, D, 1, 1); H[:, :, 0, 2] = torch.linspace(0, 12, D)[None]
Hinc = torch.eye(3).repeat(N, D, 1, 1); Hinc[:, 1:, 0, 2] = 12.0 / (D - 1)
F0 = torch.randn(N, 32, rows, cols, generator=g).cuda(); FL = torch.randn(max(1, N // 2), 32, rows, cols, generator=g).cuda()
net.options.chain_form = "banded"
eng.lib.mvsn_debug_set_band_flags(16)
for _ in range(3):
    eng.incremental_cost_volume(src4, H.cuda(), Hinc.cuda(), F0, FL)
torch.cuda.synchronize()
print("status", eng.chain_status())
t = dbg.cpu()[:128].view(4, 32)
order = [(0, "step start"), (1, "A1 image + E1 collect"), (2, "Ba"), (3, "gather + E1b publish"), (4, "B1"),
         (5, "layout + E1b collect + U wait"), (6, "B2"), (7, "conv0"), (16, "sums + E2 publish"), (17, "B3 + dma issue"),
         (18, "E2 collect"), (8, "stats + apply"), (9, "U wait + B6"), (10, "conv1"), (20, "sums + E3 publish"),
         (21, "B7 + dma + plan(d+1)"), (22, "E3 collect"), (11, "stats + apply"), (12, "U wait + B10"),
         (13, "conv2 + left loads"), (14, "B11 + epilogue")]
for d in range(4):
    row = t[d]
    out, prev = [], int(row[0])
    for idx, name in order[1:]:
        v = int(row[idx])
        out.append(f"{name} {v - prev}")
        prev = v
    print("step", d + 2, "total", int(row[14] - row[0]), "|", "; ".join(out))
