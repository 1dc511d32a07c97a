#!/usr/bin/env python3
"""Per-call breakdown of one batch-B forward (device events around every library call, launch gaps included)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "headline"]      # optional: config2 .. config5
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights(cfg["weights"])); net = net.to(dev).eval()
_, inp, _ = bench.config_inputs(cfg, B, 0, dev)
for _ in range(3): bench.run_forward(net, inp, cfg["D"])
agg = bench.kernel_breakdown(net, inp, cfg["D"])
tot = sum(v["ms"] for v in agg.values())
print(f"B={B}: {sum(v['launches'] for v in agg.values())} calls, {tot:.3f} ms inside calls")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {v['ms']*1e3:8.1f} us {v['launches']:3d} x {v['ms']/v['launches']*1e3:7.1f}  {k}")
