#!/usr/bin/env python3
"""Validation aid: the golden forwards under the engine option variants (sliced towers at every level, no alternation,
no carrying, sliced regulariser, chain forms) against the default options: carrying / slicing must be bit-identical, the
chain forms agree within rounding."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
import test_hip_parity as T
names = [("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs"), ("gc3_gta_512x256_d64_s5.npz", "gta_sfm_150epochs"),
         ("g3_demon_640x480_d96_s1.npz", "demon_45epochs"), ("g1b_gta_96x80_d8_s2_b2.npz", "gta_sfm_150epochs")]
for name, w in names:
    fix = load_golden(name)
    net = T.net_for(w)
    base = T._forward(net, fix)
    ref = [t.clone() for t in base["left_idepthmap_pyr"]]
    for opts in (dict(carry_min_bytes=0), dict(carry_min_bytes=0, carry_alternate=False), dict(carry_passes=False),
                 dict(carry_min_bytes=0, carry_volume_passes=True), dict(chain_form="stepwise"), dict(chain_form="direct")):
        old = {k: getattr(net.options, k) for k in opts}
        for k, v in opts.items(): setattr(net.options, k, v)
        try:
            out = T._forward(net, fix)
        finally:
            for k, v in old.items(): setattr(net.options, k, v)
        same = all(torch.equal(a, b) for a, b in zip(out["left_idepthmap_pyr"], ref))
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out["left_idepthmap_pyr"], ref))
        print(f"{name:34s} {str(opts):70s} identical {same}  max-rel vs default {err:.2e}")
