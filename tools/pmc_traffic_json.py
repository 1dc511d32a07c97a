#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of the bench command
(tools/prof_round.sh): bytes per chain and launch for the kernels bench.py's roofline block names, stamped with the
digest of the library the passes were taken with (bench.py refuses the file when the digest differs).

    python tools/pmc_traffic_json.py <dir with pmc_{fetch,write}_summary.csv> <chains per launch> > profiles/r04_pmc_traffic.json
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {   # timeline name -> (rocprof kernel-name fragments, algorithmic bytes per chain and launch, note)
    "mvsn_conv_forward[conv3d k3 32->32 wino]": (
        ["conv_wino_kernel<0, 2, 3, 1, true, 0", "conv_wino_kernel<1, 2, 3, 1, true, 0"], 2 * 32 * 64 * 512 * 4,
        "volume Winograd kernel; tiles by 16-byte LDS-DMA: raw FETCH_SIZE doubled (gfx950 half-count); each input plane is "
        "fetched for three output planes, the re-fetches mostly L2 hits", True),
    "mvsn_incremental_cost_volume": (
        ["chain_wino_kernel<16, 32>"], 4 * 67 * 512 + 128 * 64 * 512 + 64 * 512,
        "chain_wino_kernel<16,32>: left features per step as 8-byte pieces, transformed weights from L2 by LDS-DMA (L2 hits do "
        "not reach the memory-side counter), cost slice straight from registers; raw FETCH_SIZE reading (4- / 8-byte accesses)",
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


def main():
    d, chains = sys.argv[1], int(sys.argv[2])
    fetch, write = read(os.path.join(d, "pmc_fetch_summary.csv")), read(os.path.join(d, "pmc_write_summary.csv"))
    with open(os.path.join(ROOT, "multi_view_stereonet_amd", "libmvsn_hip.so.sources")) as f:
        digest = f.read().strip()
    out = {"_library_digest": digest, "_chains_per_launch": chains, "_source": d,
           "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counter unit KiB) of `python bench.py --steps 1 "
                       "--warmup 1 --no-cpu-baseline --no-tiers`, summed over the kernel's dispatches, / dispatches / chains per "
                       "launch.  One chain = one (image, source) cost volume, 32x64x16x32 fp32."}
    for name, (frags, algo, note, doubled) in KERNELS.items():
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
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
