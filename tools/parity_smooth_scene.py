#!/usr/bin/env python3
"""The contract's figures on a SMOOTH synthetic scene (textured planes instead of uniform noise: the fixtures' frames are the
worst case for the bilinear warps), GPU forward against the CPU oracle at the headline and config-5 shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err_per_pixel, rel_err
from multi_view_stereonet_amd import MultiViewStereoNet, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
from oracle import mvsn_oracle as oracle
torch.set_grad_enabled(False)
w = load_weights("gta_sfm_150epochs")
net = MultiViewStereoNet(); net.load_state_dict(w); net = net.cuda().eval()
for (rows, cols, D, S) in ((256, 512, 64, 2), (512, 1024, 128, 4)):
    for smooth in (False, True):
        batch = synthetic.make_batch(rows, cols, S, batch=1, seed=31, smooth=smooth)
        cpu = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
        gpu = snu.multi_view_unpack_batch(batch, torch.device("cuda"), 5)
        ref = oracle.forward(w, cpu["left_image_pyr"], cpu["K_pyr"], cpu["T_right_in_left"], cpu["right_image_pyr"], D)["left_idepthmap_pyr"][0]
        got = net(gpu["left_image_pyr"], gpu["K_pyr"], gpu["T_right_in_left"], gpu["right_image_pyr"], D, True, [True] * 5)["left_idepthmap_pyr"][0].cpu()
        mx, p999 = rel_err_per_pixel(got, ref)
        mean_rel, max_rel = rel_err(got, ref)
        print(f"{cols}x{rows} D={D} S={S} {'smooth scene ' if smooth else 'uniform noise'}: per-pixel max {mx:.2e} p99.9 {p999:.2e}  mean-rel {mean_rel:.2e} max-rel {max_rel:.2e}")
