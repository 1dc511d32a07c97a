#!/usr/bin/env python3
"""Refiner towers level by level: where the 35 ms of levels 0-2 go.

    python tools/level_profile.py run <level> [batch]      one level's sliced tower, timeline per kernel (device events)
    python tools/level_profile.py json <dir> [batch]       <dir>/L<level>_{fetch,write,sq}_summary.csv -> JSON on stdout

`run` is also the command the rocprofv3 --pmc passes wrap (tools/prof_levels.sh): with one level per process every
counter row belongs to that level.  `json` keys the counters like bench.py's timeline names
("mvsn_conv_forward[conv2d k3d2 32->32 wino L1 +pass]"), per launch, with the library digest the passes were taken with.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 3          # towers per process (the first also warms the allocator up)


def run(level, batch):
    import torch
    from multi_view_stereonet_amd import MultiViewStereoNet
    from multi_view_stereonet_amd.weights import load_weights
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    net = MultiViewStereoNet()
    net.load_state_dict(load_weights("gta_sfm_150epochs"), strict=True)
    net = net.to(dev).eval()
    eng = net.engine()
    rows, cols = 256 >> level, 512 >> level
    g = torch.Generator(device=dev).manual_seed(level)
    img = torch.rand(batch, 3, rows, cols, device=dev, generator=g) * 2 - 1
    guide = [img] if level == 0 else [img, torch.randn(batch, 32, rows, cols, device=dev, generator=g)]
    prior = torch.rand(batch, 1, rows, cols, device=dev, generator=g) * 0.5 + 0.2
    fx = torch.full((batch,), 0.8 * cols, device=dev)
    for rep in range(REPS):
        if rep == REPS - 1:
            eng.timeline = []
        out = eng.idepth_refiner(level, guide, prior, fx)
    torch.cuda.synchronize()
    tl, eng.timeline = eng.timeline, None
    agg = {}
    for name, a, b, flops, nbytes in tl:
        e = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        e["launches"] += 1
        e["ms"] += a.elapsed_time(b)
        e["flops"] += flops
        e["bytes"] += nbytes
    total = sum(e["ms"] for e in agg.values())
    print(json.dumps({"level": level, "batch": batch, "tower_ms": round(total, 3), "finite": bool(torch.isfinite(out).all()),
                      "kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                      "executed_TFLOPs": round(v["flops"] * (4 / 9 if " wino" in k else 1) / (v["ms"] * 1e-3) / 1e12, 1),
                                      "algorithmic_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9)}
                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}))


def timeline_name(kernel, level):
    """rocprof kernel name -> bench.py timeline name of the same launch at this level (None: not a tower launch)."""
    m = re.search(r"conv_wino_kernel<(\d), (\d), (\d), (\d)(?:, (true|false), (\d))?(?:, (?:true|false|\d))?>", kernel)
    if m:
        mode, ks, nstage, dil, vol, ride = m.groups()
        if vol == "true":
            return None
        if int(nstage) == 6 or (int(dil) == 1 and int(ks) == 2 and int(nstage) == 3 and not int(ride or 0)):
            cin = 4 if level == 0 else 36       # the tower's head (4-channel ring of six / 36-channel three-stage form)
            return f"mvsn_conv_forward_blocks[conv2d k3 {cin}->32 wino L{level}]"
        d = f"d{dil}" if int(dil) > 1 else ""
        return f"mvsn_conv_forward[conv2d k3{d} 32->32 wino L{level}" + (" +pass]" if int(ride or 0) else "]")
    if "conv_to1_2d" in kernel:
        return f"mvsn_conv_to1_block[L{level}]"
    if "gn_apply" in kernel:
        return f"mvsn_groupnorm_lrelu_apply[L{level}]"
    return None


def read_summary(path):
    """rows of a tools/pmc_summary.py file as dicts (kernel names contain unquoted commas: split from the right)."""
    with open(path) as f:
        header = f.readline().rstrip("\n").split(",")
        rows = []
        for ln in f:
            parts = ln.rstrip("\n").rsplit(",", len(header) - 1)
            if len(parts) == len(header):
                rows.append(dict(zip(header, parts)))
    return rows


def to_json(d, batch):
    with open(os.path.join(ROOT, "multi_view_stereonet_amd", "libmvsn_hip.so.sources")) as f:
        digest = f.read().strip()
    out = {"_library_digest": digest, "_batch": batch,
           "_comment": "per LAUNCH, from separate rocprofv3 --pmc passes of `tools/level_profile.py run <level> <batch>` "
                       "(one level per process; FETCH_SIZE / WRITE_SIZE in KiB -> bytes; fetch_bytes = 2 x raw: gfx950 counts "
                       "16-byte streaming reads at half, MI355X_MICROARCH.md HBM section); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                       "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); MODE 0 / 1 instantiations of a layer are merged"}
    for level in range(4):
        rows = {}
        for kind in ("fetch", "write", "sq"):
            path = os.path.join(d, f"L{level}_{kind}_summary.csv")
            if not os.path.exists(path):
                continue
            for r in read_summary(path):
                name = timeline_name(r["kernel"], level)
                if name is None:
                    continue
                e = rows.setdefault(name, {})
                n = float(r["dispatches"])
                e.setdefault("_n_" + kind, 0.0)
                e["_n_" + kind] += n
                for k, v in r.items():
                    if k not in ("kernel", "dispatches"):
                        e[k] = e.get(k, 0.0) + float(v)
        for name, e in rows.items():
            o = {}
            if "FETCH_SIZE" in e:
                o["fetch_bytes_raw"] = round(e["FETCH_SIZE"] * 1024 / e["_n_fetch"])
                o["fetch_bytes"] = 2 * o["fetch_bytes_raw"]
            if "WRITE_SIZE" in e:
                o["write_bytes"] = round(e["WRITE_SIZE"] * 1024 / e["_n_write"])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
                o["mfma_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
                o["gui_active_cycles_per_launch"] = round(e["GRBM_GUI_ACTIVE"] / e["_n_sq"])
                for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                    if k in e and e.get("SQ_WAVE_CYCLES"):
                        o[k.lower() + "_frac_of_wave_cycles"] = round(e[k] / e["SQ_WAVE_CYCLES"], 3)
            out[name] = o
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 256)
    else:
        to_json(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 256)
