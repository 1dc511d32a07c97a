#!/usr/bin/env python3
"""mvsn_tower_16x32 alone, launched densely through the C ABI: its input buffer is rewritten before every launch (by
mvsn_copy / by an elementwise library kernel), the output goes to a ring of 64 buffers that is compared with the
references every 64 launches.    python tools/tower_stress.py [samples] [launches]"""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs"), strict=True); net = net.to(dev).eval()
eng = net.engine(); lib = eng.lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
g = torch.Generator().manual_seed(3)
xs = [torch.randn(N, 32, 16, 32, generator=g).to(dev) for _ in range(4)]
refs = [eng.tower_extractor_tail(x).clone() for x in xs]
U, params, dils = eng._tower_pack("extractor")
xb = torch.empty_like(xs[0])
RING = 64
outs = [torch.empty_like(xs[0]) for _ in range(RING)]
descs = []
for o in outs:
    d = _native.TowerDesc()
    d.inp[0], d.channels[0], d.sample_mod[0] = xb.data_ptr(), 32, N
    d.head_chunks, d.n_blocks, d.tail_mode = 0, len(dils), 0
    for i, v in enumerate(dils): d.dilation[i] = v
    d.weights, d.params, d.out = U.data_ptr(), params.data_ptr(), o.data_ptr()
    descs.append(d)
stream = _native.stream()
nbytes = xb.numel() * 4
# a producer like the real one: the 5x5 stride-2 conv that writes the tower's input would need its own inputs; use
# mvsn_copy (hipMemcpyAsync) and, second, an elementwise library kernel (mvsn_idepth_scale: out = prior * fx) as writers
ones = torch.ones(N * 32, device=dev)
for mode in ("mvsn_copy", "scale_kernel"):
    bad, js = 0, []
    for i in range(reps):
        j = (i * 7 + i // 5) % 4
        if mode == "mvsn_copy":
            lib.mvsn_copy(xb.data_ptr(), xs[j].data_ptr(), nbytes, stream)
        else:
            lib.mvsn_idepth_scale(xs[j].data_ptr(), ones.data_ptr(), N * 32, 512, xb.data_ptr(), stream)
        lib.mvsn_tower_16x32(ctypes.byref(descs[i % RING]), N, stream)
        js.append(j)
        if len(js) == RING:
            torch.cuda.synchronize()
            for k, jj in enumerate(js):
                if not torch.equal(outs[k], refs[jj]): bad += 1
            js = []
    print("dense launches, input buffer rewritten by %s: %d launches, wrong %d" % (mode, reps, bad))
