#!/usr/bin/env python3
"""Headline forward at batch B against `carry_min_bytes` (which refiner levels run as two pipelined slices with
carried passes): python tools/carry_min_sweep.py [B]   (interleaved repeats, ms per forward)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = bench.CONFIGS["headline"]
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights(cfg["weights"])); net = net.to(dev).eval()
_, inp, _ = bench.config_inputs(cfg, B, 0, dev)


def run(it=6):
    for _ in range(2): bench.run_forward(net, inp, cfg["D"])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): bench.run_forward(net, inp, cfg["D"])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for rep in range(int(os.environ.get("REPS", "3"))):
    for mb in (16, 32, 48, 96, 192):
        net.options.carry_min_bytes = mb << 20
        ms = run()
        print(f"carry_min_bytes {mb:4d} MB: {ms:8.3f} ms  {B / ms * 1e3:7.1f} /s", flush=True)
