#!/usr/bin/env python3
"""Time the fused chain alone (mvsn_incremental_cost_volume) in its forms: ms per launch, us per step,
algorithmic TFLOP/s and GB/s.   python tools/chain_bench.py [N ...]   (N = chains per launch; default 2 and 256)
MVSN_GRID=rows,cols,D selects another coarse grid (default 16,32,64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
rows, cols, D = [int(v) for v in os.environ.get("MVSN_GRID", "16,32,64").split(",")]
cost16 = bool(int(os.environ.get("MVSN_COST_BF16", "0")))     # the cost volume stored as bf16 (bf16 feature tier)
only = [f for f in os.environ.get("MVSN_FORMS", "").split(",") if f]   # restrict to these tags
P = rows * cols
g = torch.Generator().manual_seed(0)
for N in [int(a) for a in sys.argv[1:]] or [2, 256]:
    B = max(1, N // 2)
    src4 = (torch.rand(N, 3, rows, cols, generator=g) * 2 - 1).cuda()
    H = torch.eye(3).repeat(N, D, 1, 1)
    H[:, :, 0, 2] = torch.linspace(0, 12, D)[None]          # ~0.2 px of disparity per plane
    Hinc = torch.eye(3).repeat(N, D, 1, 1); Hinc[:, 1:, 0, 2] = 12.0 / (D - 1)
    F0 = torch.randn(N, 32, rows, cols, generator=g).cuda(); FL = torch.randn(B, 32, rows, cols, generator=g).cuda()
    H, Hinc = H.cuda(), Hinc.cuda()
    for form in ("direct", "winograd", "stepwise", "banded", "banded4", "slab", "banded-auto", "banded-x"):
        if form == "banded-x":        # banded with MVSN_BAND_FLAGS (A/B switches of the kernel, e.g. 8 = no pre-spin)
            if "MVSN_BAND_FLAGS" not in os.environ:
                continue
            eng.lib.mvsn_debug_set_band_flags(int(os.environ["MVSN_BAND_FLAGS"]))
            form, tag = "banded", "banded-x"
        elif form == "banded-auto":     # the banded form's own choice (thin bands / slabs / slab passes + a thin tail pass)
            if (rows, cols) == (16, 32):
                continue
            eng.lib.mvsn_debug_set_band_flags(0)
            form, tag = "banded", "banded-auto"
        elif form == "slab":            # the slab plan of the banded form pinned (debug flag 16): few fat bands per chain
            eng.lib.mvsn_debug_set_band_flags(16)
            form, tag = "banded", "slab"
        elif form == "banded4":         # 16x32: the 4-band plan pinned (debug flag 4) against the 8-band half-split plan
            if (rows, cols) != (16, 32) or N > 32:
                continue
            eng.lib.mvsn_debug_set_band_flags(4)
            form, tag = "banded", "banded4"
        elif form == "banded":         # the thin-band plan pinned (debug flag 32)
            eng.lib.mvsn_debug_set_band_flags(32)
            tag = form
        else:
            eng.lib.mvsn_debug_set_band_flags(0)
            tag = form
        if form == "winograd" and (rows, cols) != (16, 32):
            continue
        if form == "stepwise" and (rows, cols) == (16, 32):
            continue
        if (only and tag not in only) or (cost16 and form == "stepwise"):
            continue
        net.options.chain_form = form
        for _ in range(3):
            eng.incremental_cost_volume(src4, H, Hinc, F0, FL, cost_bf16=cost16)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if N >= 32 else 20
        a.record()
        for _ in range(reps):
            eng.incremental_cost_volume(src4, H, Hinc, F0, FL, cost_bf16=cost16)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        flops = N * (D - 1) * 2.0 * 9 * 32 * 99 * P
        nbytes = N * (4.0 * 67 * P + (64.0 if cost16 else 128.0) * D * P + D * P)
        status = eng.chain_status()
        eng.lib.mvsn_debug_set_band_flags(0)
        print(f"N={N:4d} {tag + ('+bf16cost' if cost16 else ''):11s}: {ms:7.3f} ms/launch  {ms * 1e3 / (D - 1):6.1f} us/step  "
              f"{flops / ms / 1e9:6.1f} direct-form TFLOP/s  {nbytes / ms / 1e6:7.1f} GB/s algorithmic  status {status}")
