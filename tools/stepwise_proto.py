#!/usr/bin/env python3
"""Prototype (tuning aid): the incremental feature chain as per-plane launches of the library's full-chip kernels
(warp, Winograd convolutions with GroupNorm on load, the two-raw-tensor pass) instead of the fused one-workgroup-per-chain
kernel -- for coarse grids without a Winograd chain plan (30x40, 32x64), where a chain's plane no longer fits one CU's LDS
and the fused direct kernel runs at one CU per chain.  Compares cost / mask with the fused kernel and times both
(hipGraph replay, so the host's enqueue time is not what is measured).
   python tools/stepwise_proto.py [batch] [rows cols D S]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R, C, D, S = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else (480, 640, 96, 1)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("demon_45epochs" if (R, C) == (480, 640) else "gta_sfm_150epochs"))
net = net.cuda().eval()
eng = net.engine(); lib = eng.lib
r = net.right_feature_extractor.refiner
conv0, bn0 = _Conv(lib, r.conv0.weight, r.conv0.bias), _Norm(r.bn0)
conv1, bn1 = _Conv(lib, r.res0.conv1.weight, r.res0.conv1.bias), _Norm(r.res0.bn1)
conv2 = _Conv(lib, r.conv_final.weight, r.conv_final.bias)
rows, cols = (R + 15) // 16, (C + 15) // 16
N, P = S * B, rows * cols
g = torch.Generator().manual_seed(0)
src4 = (torch.rand(N, 3, rows, cols, generator=g) * 2 - 1).cuda()
H = torch.eye(3).repeat(N, D, 1, 1); H[:, :, 0, 2] = torch.linspace(0, 12, D)[None]
Hinc = torch.eye(3).repeat(N, D, 1, 1); Hinc[:, 1:, 0, 2] = 12.0 / (D - 1)
F0 = torch.randn(N, 32, rows, cols, generator=g).cuda(); FL = torch.randn(B, 32, rows, cols, generator=g).cuda()
H, Hinc = H.cuda(), Hinc.cuda()


def stepwise():
    vol, mask = eng.homography_warp(src4, H)                       # (N,3,D,h,w), (N,D,h,w)
    img = vol.permute(2, 0, 1, 3, 4).contiguous()
    Ht = Hinc.transpose(0, 1).contiguous()                          # (D,N,3,3)
    cost = torch.empty(N, 32, D, rows, cols, device="cuda")
    keep = (~mask).float()
    Fp = F0
    cost[:, :, 0] = keep[:, 0, None] * (FL.repeat(S, 1, 1, 1) - Fp * keep[:, 0, None]).abs()
    FLr = FL.repeat(S, 1, 1, 1)
    for d in range(1, D):
        moved, _ = eng.homography_warp(Fp, Ht[d][:, None])
        moved = moved[:, :, 0]
        r0, st0 = eng.conv(conv0, [img[d], moved], want_stats=True)
        r1, st1 = eng.conv(conv1, r0, in_stats=st0, in_norm=bn0, want_stats=True)
        a1 = eng.gn_lrelu_add2(r1, st1, bn1, r0, st0, bn0, out=r1)
        delta, _ = eng.conv(conv2, a1)
        Fp = moved + delta
        k = keep[:, d, None]
        cost[:, :, d] = k * (FLr - Fp * k).abs()
    return cost, mask


def fused():
    c, m, _ = eng.incremental_cost_volume(src4, H, Hinc, F0, FL)
    return c, m


c1, m1 = stepwise(); c2, m2 = fused(); torch.cuda.synchronize()
err = (c1 - c2).abs()
print(f"grid {rows}x{cols} N={N} D={D}: masks equal {torch.equal(m1, m2)}  cost max abs diff {err.max().item():.3e}  "
      f"mean-rel {err.mean().item() / c2.abs().mean().item():.3e}")
for name, fn in (("fused", fused), ("stepwise", stepwise)):
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            fn()
    torch.cuda.synchronize()
    for _ in range(2): gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): gr.replay()
    b.record(); torch.cuda.synchronize()
    print(f"  {name:9s} {a.elapsed_time(b) / 5:8.3f} ms per chain set ({a.elapsed_time(b) / 5 / (D - 1) * 1e3:6.1f} us per step)")
