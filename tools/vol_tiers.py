#!/usr/bin/env python3
"""The regulariser (four 3x3x3 32->32 layers + the 32->1 tail) alone, per precision tier: ms per call, and -- under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) -- the HBM bytes of its launches.
    python tools/vol_tiers.py [chains] [rows cols D]         # timing
    python tools/vol_tiers.py json <dir with vt_{fetch,write}_summary.csv> > profiles/r05_bf16_feature_tier_pmc.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "json":
    from pmc_traffic_json import read_summary
    d = sys.argv[2]
    with open(os.path.join(ROOT, "multi_view_stereonet_amd", "libmvsn_hip.so.sources")) as f:
        digest = f.read().strip()
    fetch = read_summary(os.path.join(d, "vt_fetch_summary.csv"))
    write = {r["kernel"]: r for r in read_summary(os.path.join(d, "vt_write_summary.csv"))}
    out = {"_library_digest": digest,
           "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB) of `python tools/vol_tiers.py 128 32 64 128` "
                       "(config 5's regulariser: 128 chains of 32x128x32x64): bytes per launch and chain; FETCH_SIZE as read (the "
                       "kernels' 2- / 4-byte per-lane loads are not the half-counted 16-byte streaming form); `chain`: WRITE_SIZE of "
                       "`MVSN_GRID=32,64,128 MVSN_FORMS=banded-auto MVSN_COST_BF16=0|1 python tools/chain_bench.py 128` (Kernel A, "
                       "two slab passes of 64 chains -- their bytes include the four hand-offs' write-through granules, ~50 MB per chain -- and of `MVSN_FORMS=winograd ... chain_bench.py 512`, the headline's plane-resident kernel: cost volume written as fp32 / bf16, <.., false> / <.., true>)", "kernels": {}}
    chains = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    for r in fetch:
        k = r["kernel"]
        if "conv_bf16x3_kernel" not in k and "conv_wino_kernel" not in k:
            continue
        w = write.get(k)
        n = float(r["dispatches"])
        out["kernels"][k[:140]] = {"dispatches": int(n), "fetch_bytes_per_chain": round(float(r["FETCH_SIZE"]) * 1024 / n / chains),
                                   "write_bytes_per_chain": round(float(w["WRITE_SIZE"]) * 1024 / float(w["dispatches"]) / chains) if w else None}
    # Kernel A (the chain) with the cost volume stored as fp32 / bf16: WRITE_SIZE passes of tools/chain_bench.py (prof_feature_tier.sh)
    for c16 in (0, 1):
        for stem, per in (("chain_write", 64), ("chainw_write", 512)):   # (32x64: two slab passes of 64; 16x32: one launch of 512)
            f = os.path.join(d, f"{stem}{c16}_summary.csv")
            if not os.path.exists(f):
                continue
            for r in read_summary(f):
                if "chain_slab_kernel" in r["kernel"] or "chain_band_kernel" in r["kernel"] or "chain_wino_kernel" in r["kernel"]:
                    out.setdefault("chain", {})[r["kernel"][:110]] = {
                        "dispatches": int(float(r["dispatches"])), "chains_per_dispatch": per,
                        "write_bytes_per_chain": round(float(r["WRITE_SIZE"]) * 1024 / float(r["dispatches"]) / per)}
    print(json.dumps(out, indent=1))
    sys.exit(0)
import torch
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows, cols, D = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 64, 128)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine()
cost = torch.rand(N, 32, D, rows, cols, device="cuda")
for tier in ("fp32", "bf16", "bf16s"):
    net.options.conv_precision = tier
    for _ in range(2):
        out = eng.cost_volume_filter(cost)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        out = eng.cost_volume_filter(cost)
    b.record(); torch.cuda.synchronize()
    print(f"{tier:6s}: {a.elapsed_time(b) / 3:8.3f} ms per regulariser call ({N} chains of 32x{D}x{rows}x{cols}), finite {bool(torch.isfinite(out).all())}")
net.options.conv_precision = "fp32"
