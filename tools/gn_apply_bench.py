#!/usr/bin/env python3
"""Timing aid: the stand-alone LeakyReLU(GroupNorm(.)) pass (mvsn_groupnorm_lrelu_apply) on the regulariser volume and
on a level-0 refiner tensor (the headline step runs it on 512 volumes: 4.3 GB in 0.63 ms = 6.9 TB/s)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
NORM = eng.vf_norms[0]
for shape in ((512, 32, 64, 16, 32), (256, 32, 64, 16, 32), (128, 32, 256, 512), (2, 32, 64, 16, 32)):
    r = torch.randn(*shape, device="cuda"); out = torch.empty_like(r); res = torch.randn_like(r)
    st = torch.zeros(shape[0], 4, 2, device="cuda"); st[:, :, 1] = 1
    ms = timed(lambda: eng.gn_lrelu(r, st, NORM, None, out))
    ms2 = timed(lambda: eng.gn_lrelu(r, st, NORM, res, out))
    ms3 = timed(lambda: eng.gn_lrelu(r, st, NORM, None, r))      # in place, as the regulariser calls it
    print("%s: plain %.3f ms (%.0f GB/s)   +residual %.3f ms (%.0f GB/s)   in place %.3f ms (%.0f GB/s)" % (
        shape, ms, 2 * r.numel() * 4 / ms / 1e6, ms2, 3 * r.numel() * 4 / ms2 / 1e6, ms3, 2 * r.numel() * 4 / ms3 / 1e6))
