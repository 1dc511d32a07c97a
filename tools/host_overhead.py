#!/usr/bin/env python3
"""Tuning aid: host time to enqueue one forward vs device time, and a hipGraph replay of the same forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.set_grad_enabled(False)
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights(bench.WEIGHTS)); net = net.to(dev).eval()
net.stream_lanes = lanes
_, inp = bench.make_inputs(B, 7, dev)
for mode, cap, gr in (("eager (one ctypes call per launch from Python)", 0, False),
                      ("recorded plan, call list replayed from Python", 16, False),
                      ("recorded plan as one hipGraph launch (default at small batch)", 16, True)):
    net.options.plan_max_chains, net.options.plan_graph = cap, gr
    for _ in range(3):
        bench.run_forward(net, inp)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(10):
        t0 = time.perf_counter(); bench.run_forward(net, inp); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print(f"B={B} lanes={lanes} {mode}: host enqueue {1e3*min(hs):.3f} ms (median {1e3*sorted(hs)[5]:.3f}), until idle {1e3*min(ts):.3f} ms")
net.options.plan_max_chains = 0
# hipGraph capture of the whole forward
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    bench.run_forward(net, inp)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = bench.run_forward(net, inp)
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"  graph replay: {1e3*(t1-t0)/5:.2f} ms per forward -> {B*5/(t1-t0):.1f} depthmaps/s; finite={bool(torch.isfinite(out['left_idepthmap_pyr'][0]).all())}")
except Exception as e:
    print("  graph capture failed:", repr(e)[:300])
