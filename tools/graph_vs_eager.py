#!/usr/bin/env python3
"""Batch-1 forward of configs 4 / 5, eager against hipGraph replay (both device-bound: 8.5 / 12.9 ms either way)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.graphed import GraphedForward
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
for (R, C, D, S, w) in ((480, 640, 96, 1, "demon_45epochs"), (512, 1024, 128, 4, "gta_sfm_150epochs")):
    net = MultiViewStereoNet(); net.load_state_dict(load_weights(w)); net = net.cuda().eval()
    inp = snu.multi_view_unpack_batch(synthetic.make_batch(R, C, S, batch=1, seed=7), torch.device("cuda"), 5)
    a = (inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"])
    f = lambda: net(*a, D, True, [True] * 5)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 20 * 1e3
    g = GraphedForward(net, *a, D)
    for _ in range(3): g(*a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g(*a)
    torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{R}x{C} D={D} S={S}: eager {e:.2f} ms  graph {gr:.2f} ms")
