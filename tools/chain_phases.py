#!/usr/bin/env python3
"""Tuning aid: print the chain kernel's per-phase cycle counts (block 0, wave 0) for steps 1..4."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
# needs a tuning build: MVSN_HIPCC_FLAGS=-DMVSN_CHAIN_STAMPS python -m multi_view_stereonet_amd.build --force
import ctypes
from multi_view_stereonet_amd import _native
dbg = torch.zeros(256, dtype=torch.int64, device="cuda")
ctypes.CDLL(_native.library_path()).mvsn_debug_set_chain_stamps(ctypes.c_void_p(dbg.data_ptr()))
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = synthetic.make_batch(256, 512, 2, batch=B, seed=7)
inp = snu.multi_view_unpack_batch(batch, torch.device("cuda"), 5)
for _ in range(2):
    net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 64, True, [True] * 5)
torch.cuda.synchronize()
t = dbg.cpu()[:64].view(4, 16)
wv = dbg.cpu()[64:128].view(8, 8)
names = ["A1+A2 gather", "B1", "A3 write+w0", "conv0 mfma", "B3", "gn0+write+w1", "conv1 mfma", "B7", "gn1+write+w2",
         "conv2 mfma", "B11", "epilogue", "B12"]
for d in range(4):
    row = t[d]
    deltas = [int(row[i + 1] - row[i]) for i in range(13)]
    print("step", d + 1, "total", int(row[13] - row[0]), {n: v for n, v in zip(names, deltas)})

if int(wv.abs().sum()):     # Winograd kernel, step 3: per-wave [conv0 start, conv0 end, after B3, GN0 done, U landed, after B6]
    t0 = int(wv[:, 0].min())
    for w in range(8):
        print("wave", w, [int(v - t0) for v in wv[w, :7]], "(last = DMA of the next layer's U issued)")
