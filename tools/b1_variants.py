#!/usr/bin/env python3
"""Batch-1 (or B) forward latency under engine option variants; prints ms per forward and launches per forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda")
net = MultiViewStereoNet(); net.load_state_dict(load_weights(bench.WEIGHTS)); net = net.to(dev).eval()
_, inp = bench.make_inputs(B, 7, dev)
variants = [dict(), dict(lazy_stats_max_samples=0), dict(lazy_stats_max_records=512), dict(lazy_stats_max_records=8192),
            dict(fold_residual_blocks=True), dict(winograd=False), dict(winograd=False, fold_residual_blocks=True),
            dict(trim_tower_ends=False), dict(winograd_volume=False), dict(cat_free_heads=False), dict(chain_form="winograd")]
for opts in variants:
    old = {k: getattr(net.options, k) for k in opts}
    for k, v in opts.items(): setattr(net.options, k, v)
    try:
        for _ in range(3): bench.run_forward(net, inp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): bench.run_forward(net, inp)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        agg = bench.kernel_breakdown(net, inp)
        n = sum(v["launches"] for v in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:6]
        print(f"B={B} {str(opts):60s} {ms:7.3f} ms  {n} calls  " + "; ".join(f"{k[:38]} {v['launches']}x {v['ms']:.2f}" for k, v in top))
    finally:
        for k, v in old.items(): setattr(net.options, k, v)
