#!/usr/bin/env python3
"""Write <out_dir>/stereo_network.pt: the TorchScript archive the reference's test.py loads (test.py:308-314), for the
MI355X build.   python tools/make_archive.py gta_sfm_150epochs /tmp/gta_weights"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd.torchscript import export_archive  # noqa: E402
from multi_view_stereonet_amd.weights import load_weights  # noqa: E402

if __name__ == "__main__":
    name, out_dir = sys.argv[1], sys.argv[2]
    os.makedirs(out_dir, exist_ok=True)
    print(export_archive(load_weights(name), os.path.join(out_dir, "stereo_network.pt")))
