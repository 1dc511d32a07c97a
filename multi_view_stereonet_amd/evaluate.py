#!/usr/bin/env python3
"""Evaluation entry point with the reference's ``test.py`` command line (test.py:318-412):

    python -m multi_view_stereonet_amd.evaluate <weights> <data_dir> <test_file> [--split gta_sfm|demon]
    torchrun --nproc-per-node 8 -m multi_view_stereonet_amd.evaluate ...       # batches sharded over GPUs

``weights`` is a shipped model name (gta_sfm_150epochs | demon_45epochs) or a .safetensors state dict;
``params`` default to the shipped yaml values (size 480x640, 12 idepth samples, filter and refiners on).
Writes avg_depth_metrics.txt (+ per-rank rows are all-gathered first) into --output_dir.
"""
import argparse
import json
import os

import torch

from . import MultiViewStereoNet, datasets, metrics
from . import distributed as mdist
from .weights import load_weights


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("weights")
    ap.add_argument("data_dir")
    ap.add_argument("test_file")
    ap.add_argument("--split", default=None, help="gta_sfm | demon (default: inferred from the weights name)")
    ap.add_argument("--size", type=int, nargs=2, default=[480, 640], metavar=("ROWS", "COLS"))
    ap.add_argument("--num_idepth_samples", type=int, default=12)
    ap.add_argument("--num_right_images", type=int, default=1, help="DeMoN only")
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--output_dir", default="output")
    args = ap.parse_args(argv)

    rank, world, local = mdist.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    split = args.split or ("demon" if "demon" in args.weights else "gta_sfm")
    params = {"size": args.size, "num_idepth_samples": args.num_idepth_samples, "cost_volume_filter": True,
              "refiners": [True] * 5}
    tf = datasets.get_testing_transforms(params)
    if split == "demon":
        data = datasets.DeMoNDataset(args.data_dir, args.test_file, num_right_images=args.num_right_images,
                                     transform=tf, shuffle_on_read=False)
    else:
        data = datasets.GTASfMMultiViewStereoDataset(args.data_dir, args.test_file, transform=tf,
                                                     load_groundtruth_depthmaps=True, shuffle_on_read=False)
    # shard at the dataset level: each rank reads, decodes and resizes only its own images
    mine = mdist.shard_indices(len(data), rank, world)
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(data, mine), batch_size=args.batch_size, shuffle=False)
    net = MultiViewStereoNet()
    net.load_state_dict(load_weights(args.weights), strict=True)
    net = net.to(dev).eval()
    avg = metrics.evaluate(net, loader, params, split, dev, rank=rank, world=world, image_indices=mine)
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, "avg_depth_metrics.txt"), "w") as f:
            keys = [k for k in avg if k != "num_samples"]
            f.write(" ".join(keys) + "\n" + " ".join(str(avg[k]) for k in keys) + "\n")
        print(json.dumps(avg))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
