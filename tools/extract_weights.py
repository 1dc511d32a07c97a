#!/usr/bin/env python3
"""Pull the 226 fp32 tensors out of the reference's TorchScript archives into plain
tensor containers (safetensors) under ``weights/``.

Runs ONLY in the build container (it reads ``/root/reference/pretrained``); the output
files are data and are committed.  The archives themselves are never copied: their
``code/`` directory is serialised reference source.

Why a hand-rolled unpickler: ``torch.jit.load`` of the GTA archive fails on torch 2.10
 and the DeMoN archive is an older 4-argument export.  Both share the
same parameter layout, which is all we need.

Usage:  python tools/extract_weights.py
"""
import io
import os
import pickle
import sys
import zipfile

import torch
from safetensors.torch import save_file

REF = "/root/reference/pretrained"
ARCHIVES = {
    "gta_sfm_150epochs": f"{REF}/gta_sfm_150epochs/checkpoints/epoch0149/stereo_network.pt",
    "demon_45epochs": f"{REF}/demon_45epochs/checkpoints/epoch0044/stereo_network.pt",
}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights")


class _Blob:
    """Stand-in for every scripted class in the pickle; keeps whatever state it is given."""

    def __init__(self, *a, **k):
        self.state = {}

    def __setstate__(self, st):
        self.state = st


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, zf, root):
        super().__init__(f)
        self.zf, self.root = zf, root

    def find_class(self, module, name):
        if module.startswith("__torch__"):
            return _Blob
        return super().find_class(module, name)

    def persistent_load(self, pid):
        kind, stype, key, _loc, numel = pid
        assert kind == "storage"
        raw = self.zf.read(f"{self.root}/data/{key}")
        dtype = stype.dtype
        un = torch.UntypedStorage.from_buffer(raw, dtype=torch.uint8) if False else None
        buf = bytearray(raw)
        un = torch.frombuffer(buf, dtype=torch.uint8).untyped_storage()
        return torch.storage.TypedStorage(wrap_storage=un, dtype=dtype, _internal=True)


def _flatten(obj, prefix, out):
    st = obj.state if isinstance(obj, _Blob) else obj
    if not isinstance(st, dict):
        return
    for k, v in st.items():
        name = f"{prefix}.{k}" if prefix else k
        if isinstance(v, torch.Tensor):
            out[name] = v
        elif isinstance(v, (_Blob, dict)):
            _flatten(v, name, out)


def extract(path):
    zf = zipfile.ZipFile(path)
    root = zf.namelist()[0].split("/")[0]
    top = _Unpickler(io.BytesIO(zf.read(f"{root}/data.pkl")), zf, root).load()
    tensors = {}
    _flatten(top, "", tensors)
    # keep parameters only (drop python scalars such as `training`, `min_idepth`)
    return {k: v.detach().clone().contiguous().float() for k, v in tensors.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, path in ARCHIVES.items():
        sd = extract(path)
        # The source-view extractor re-registers the left extractor: same storage under a
        # second name.  Store each tensor once; the loader re-creates the alias.
        uniq = {k: v for k, v in sd.items()
                if not k.startswith("right_feature_extractor.feature_extractor.")}
        for k in sd:
            if k.startswith("right_feature_extractor.feature_extractor."):
                twin = "left_feature_extractor." + k.split(".", 2)[2]
                assert torch.equal(sd[k], sd[twin]), k
        n = sum(v.numel() for v in sd.values())
        print(f"{name}: {len(sd)} keys ({len(uniq)} stored), {n} elements")
        save_file(uniq, os.path.join(OUT, f"{name}.safetensors"),
                  metadata={"source": f"pretrained/{name} (TorchScript archive, tensors only)",
                            "keys_with_alias": str(len(sd))})


if __name__ == "__main__":
    sys.exit(main())
