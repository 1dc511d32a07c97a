#!/usr/bin/env python3
"""How far is the REFERENCE from itself?  Build-container only (imports /root/reference the way
tests/golden/make_golden.py does; nothing here travels to the GPU box or is used by tests / bench).

The reference forms the incremental homography as `torch.inverse(H[:, d-1].unsqueeze(1)) @ H[:, d]`
(multi_view_stereonet.py:279-282).  For a batch of one that slice is contiguous and ATen takes the `linalg_solve_ex`
shortcut (LU of the transpose, transposed solve); for a batch of two the slice is strided, ATen factors a copy of the
matrix itself, and 40 % of the inverse's entries come out an ulp or more differently.  This script runs the same two
images through the reference once as a batch of two and once as two batches of one and reports how far the depth maps
are apart -- the scale against which this build's deviation from the batch-1 fixtures is read.

    python tools/reference_batch_spread.py [--config headline|config4] [--smooth]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(REPO, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
from multi_view_stereonet_amd import synthetic  # noqa: E402

CONFIGS = {"headline": (256, 512, 2, 64, "gta_sfm_150epochs"), "config2": (256, 512, 1, 64, "gta_sfm_150epochs"),
           "config4": (480, 640, 1, 96, "demon_45epochs")}


def split(batch, i):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v[i:i + 1].clone()
        elif isinstance(v, (list, tuple)):
            out[k] = [t[i:i + 1].clone() if torch.is_tensor(t) else t for t in v]
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--smooth", action="store_true")
    ap.add_argument("--seeds", type=int, default=3)
    args = ap.parse_args()
    rows, cols, S, D, wname = CONFIGS[args.config]
    torch.set_num_threads(8)
    net = mg.ref_net(wname)
    worst = 0.0
    for seed in range(args.seeds):
        batch = synthetic.make_batch(rows, cols, S, batch=2, seed=4100 + seed, smooth=args.smooth, pose_jitter=0.2)
        _, both, _ = mg.run_reference(net, batch, D, capture=False)
        two = both["left_idepthmap_pyr"][0]
        for i in range(2):
            _, one, _ = mg.run_reference(net, split(batch, i), D, capture=False)
            a, b = one["left_idepthmap_pyr"][0][0], two[i]
            rel = ((a - b).abs() / b.abs().clamp_min(1e-12))
            l4 = (one["left_idepthmap_pyr"][4][0] - both["left_idepthmap_pyr"][4][i]).abs() / \
                both["left_idepthmap_pyr"][4][i].abs().clamp_min(1e-12)
            print(f"{args.config} seed {seed} image {i}: batch-of-1 vs batch-of-2, level 0 per-pixel max rel {rel.max().item():.3e} "
                  f"mean rel {rel.mean().item():.3e}; level 4 max rel {l4.max().item():.3e}; "
                  f"bit-identical pixels {(a == b).float().mean().item():.3f}")
            worst = max(worst, rel.max().item())
    print(f"{args.config}{' smooth' if args.smooth else ''}: the reference against itself (batch 1 vs batch 2), per-pixel max rel {worst:.3e}")


if __name__ == "__main__":
    main()
