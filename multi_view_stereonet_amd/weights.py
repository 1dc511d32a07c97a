"""Checkpoint I/O: the reference's 226-key state_dict layout as plain tensors.

``weights/*.safetensors`` hold the tensors extracted from the reference's TorchScript
archives (tools/extract_weights.py).  The source-view extractor shares the left extractor's
parameters under a second name (multi_view_stereonet.py:506-507, :243); the container stores
each tensor once and ``load_weights`` re-creates the alias keys.
"""
import os
from typing import Dict

import torch

_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights")
PRETRAINED = ("gta_sfm_150epochs", "demon_45epochs")
_ALIAS_SRC = "left_feature_extractor."
_ALIAS_DST = "right_feature_extractor.feature_extractor."


def with_alias_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith(_ALIAS_SRC):
            out[_ALIAS_DST + k[len(_ALIAS_SRC):]] = v
    return out


def load_weights(name_or_path: str, device="cpu") -> Dict[str, torch.Tensor]:
    """Return the full 226-key dict for a shipped model name or a .safetensors path."""
    from safetensors.torch import load_file
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(_ROOT, name_or_path + ".safetensors")
    sd = load_file(path, device=str(device))
    return with_alias_keys(sd)


def default_init_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights with the reference's statistics: conv weights N(0, 0.01), biases 0,
    GroupNorm affine (1, 0) (multi_view_stereonet.py:40,47,68,308).  NOT bit-identical to the
    reference's own seeded init (the draw order differs); parity runs use the golden files."""
    from .params import parameter_shapes
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}
    for k, shape in parameter_shapes().items():
        if k.startswith(_ALIAS_DST):
            continue
        if ".bn" in k:
            sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
        elif k.endswith("bias"):
            sd[k] = torch.zeros(shape)
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.01
    return with_alias_keys(sd)
