#!/usr/bin/env python3
"""Batch-1 forward under rocprofv3 --kernel-trace: run as
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1 -o b1 -- python tools/b1_trace.py
and read b1_kernel_stats.csv (20 forwards after warm-up)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
# optional: rows cols hypotheses sources (default the headline 256 512 64 2; config 4: 480 640 96 1; config 5: 512 1024 128 4)
R, C, D, S = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else (256, 512, 64, 2)
inp = snu.multi_view_unpack_batch(synthetic.make_batch(R, C, S, batch=B, seed=7), torch.device("cuda"), 5)
run = lambda: net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5)
for _ in range(3):
    run()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
print("ms per forward", (time.perf_counter() - t0) / 20 * 1e3)
