import sys, os, torch
sys.path.insert(0, os.getcwd())
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
def timed(fn, reps=4):
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
xv = torch.randn(256, 32, 64, 16, 32, device="cuda")
x2 = torch.randn(128, 32, 256, 512, device="cuda")
for ws in (True, False):
    print("want_stats", ws, "volume %.3f ms   2-D d1 %.3f ms   d2 %.3f ms" % (timed(lambda: eng.conv(eng.vf_convs[1], xv, want_stats=ws)),
      timed(lambda: eng.conv(eng.refiners[0]["res"][0][0], x2, want_stats=ws)),
      timed(lambda: eng.conv(eng.refiners[0]["res"][1][0], x2, want_stats=ws))))
