#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of the bench command
(tools/prof_round.sh): bytes per chain and launch for the kernels bench.py's roofline block names, stamped with the
digest of the library the passes were taken with (bench.py refuses the file when the digest differs).

    python tools/pmc_traffic_json.py <dir with pmc_{fetch,write}_summary.csv> <chains per launch> [config4=<dir>:<chains> ...] > profiles/<round>_pmc_traffic.json
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernels_for(config):
    """timeline name -> (rocprof kernel-name fragments, algorithmic bytes per chain and launch, note, fetch doubled) for one
    of bench.py's configurations (a chain = one (image, source) cost volume of 32 x D x P fp32)."""
    D, P = {"headline": (64, 512), "config2": (64, 512), "config3": (64, 512), "config4": (96, 1200), "config5": (128, 2048)}[config]
    vol = {"headline": ["conv_wino_kernel<0, 2, 3, 1, true, 0", "conv_wino_kernel<1, 2, 3, 1, true, 0"],
           "config4": ["conv_wino_kernel<0, 2, 4, 1, true, 0, 2", "conv_wino_kernel<1, 2, 4, 1, true, 0, 2"]}
    chain = {"headline": ["chain_wino_kernel<16, 32"], "config4": ["chain_slab_kernel<mvsn::SlabGeo<30, 40"],
             "config5": ["chain_slab_kernel<mvsn::SlabGeo<32, 64"]}
    base = "headline" if config in ("config2", "config3") else config
    return {
        "mvsn_conv_forward[conv3d k3 32->32 wino]": (
            vol.get(base, vol["headline"]), 2 * 32 * D * P * 4,
            "volume Winograd kernel; tiles by 16-byte LDS-DMA: raw FETCH_SIZE doubled (gfx950 half-count); each input plane is "
            "fetched for three output planes, the re-fetches mostly L2 hits", True),
        "mvsn_incremental_cost_volume": (
            chain[base], 4 * 67 * P + 128 * D * P + D * P,
            "the fused chain kernel of this grid (plane-resident on 16x32; the slab plan on 30x40 / 32x64, whose hand-off "
            "granules are write-through traffic on top of the cost volume): left features per step as 8-byte pieces, transformed "
            "weights from L2 by LDS-DMA, cost slice straight from registers; raw FETCH_SIZE reading (4- / 8-byte accesses)",
            False),
    }


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


def read(path):
    return {r["kernel"]: r for r in read_summary(path)}


def collect(d, chains, config):
    fetch, write = read(os.path.join(d, "pmc_fetch_summary.csv")), read(os.path.join(d, "pmc_write_summary.csv"))
    out = {}
    for name, (frags, algo, note, doubled) in kernels_for(config).items():
        fb = wb = nf = nw = 0.0
        for k, r in fetch.items():
            if any(fr in k for fr in frags):
                fb += float(r["FETCH_SIZE"]) * 1024
                nf += float(r["dispatches"])
        for k, r in write.items():
            if any(fr in k for fr in frags):
                wb += float(r["WRITE_SIZE"]) * 1024
                nw += float(r["dispatches"])
        if not nf or not nw:
            continue
        raw = fb / nf / chains
        out[name] = {"_note": note, "fetch_bytes_per_chain_raw": round(raw),
                     "fetch_bytes_per_chain": round(raw * (2 if doubled else 1)), "fetch_doubled": bool(doubled),
                     "write_bytes_per_chain": round(wb / nw / chains), "algorithmic_bytes_per_chain": algo,
                     "dispatches": int(nf)}
    return out


def main():
    """pmc_traffic_json.py <headline dir> <chains per launch> [config4=<dir>:<chains> config5=<dir>:<chains> ...]"""
    d, chains = sys.argv[1], int(sys.argv[2])
    with open(os.path.join(ROOT, "multi_view_stereonet_amd", "libmvsn_hip.so.sources")) as f:
        digest = f.read().strip()
    out = {"_library_digest": digest, "_chains_per_launch": chains, "_source": d,
           "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counter unit KiB) of `python bench.py [--config C "
                       "--batch B] --steps 1 --warmup 1 --no-cpu-baseline --no-tiers`, summed over the kernel's dispatches, / "
                       "dispatches / chains per launch.  One chain = one (image, source) cost volume, 32 x D x rows4 x cols4 "
                       "fp32 (headline: 32x64x16x32).  `configs`: the same for bench.py's other configurations, each from "
                       "passes of its own command."}
    out.update(collect(d, chains, "headline"))
    out["configs"] = {}
    for arg in sys.argv[3:]:
        name, rest = arg.split("=", 1)
        cdir, cchains = rest.rsplit(":", 1)
        entry = collect(cdir, int(cchains), name)
        entry["_chains_per_launch"], entry["_source"] = int(cchains), cdir
        out["configs"][name] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
