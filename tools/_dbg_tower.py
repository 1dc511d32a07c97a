import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs"), strict=True); net = net.to(dev).eval()
eng = net.engine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
g = torch.Generator().manual_seed(3)
xs = [torch.randn(N, 32, 16, 32, generator=g).to(dev) for _ in range(4)]
refs = [eng.tower_extractor_tail(x).clone() for x in xs]
torch.cuda.synchronize()
# (a) same buffers, back to back
bad = 0
outs = []
for i in range(reps):
    j = i % 4
    o = eng.tower_extractor_tail(xs[j])
    outs.append((j, o))
    if len(outs) == 64:
        for jj, oo in outs:
            if not torch.equal(oo, refs[jj]): bad += 1
        outs = []
print("(a) distinct input buffers:", reps, "launches, wrong", bad)
for mode in ("copy_", "aten_add", "mvsn_conv"):
    xb = torch.empty_like(xs[0])
    bad = 0
    outs = []
    conv = eng.fe_res[0][0] if hasattr(eng, "fe_res") else None
    for i in range(reps):
        j = (i * 7 + i // 5) % 4
        if mode == "copy_":
            xb.copy_(xs[j])
        elif mode == "aten_add":
            torch.add(xs[j], 0.0, out=xb)
        else:
            eng.copy_into(xb, xs[j])
        o = eng.tower_extractor_tail(xb)
        outs.append((j, o))
        if len(outs) == 64:
            for jj, oo in outs:
                if not torch.equal(oo, refs[jj]): bad += 1
            outs = []
    print("(b) one input buffer rewritten by %s:" % mode, reps, "launches, wrong", bad)
