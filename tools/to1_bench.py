import sys, os, torch
sys.path.insert(0, "/root/repo")
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
c = eng.vf_convs[4]
for N in (2, 256):
    x = torch.randn(N, 32, 64, 16, 32, device="cuda")
    for it in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = eng.conv_to1(c, x); b.record(); torch.cuda.synchronize()
    print(N, "chains: %.3f ms  (%.0f GB/s)" % (a.elapsed_time(b), x.numel() * 4 / a.elapsed_time(b) / 1e6))
