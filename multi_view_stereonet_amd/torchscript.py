"""TorchScript face of the MI355X build: a `stereo_network.pt` that `torch.jit.load` accepts.

The reference's evaluation script loads its network with
``torch.jit.load(os.path.join(weights_dir, "stereo_network.pt"))`` (test.py:308-314), moves it with ``.to(device)``,
reads ``stereo_network.num_levels`` (test.py:199) and calls it with the seven positional arguments of
``MultiViewStereoNet.forward`` (multi_view_stereonet_utils.py:647-654).  This module provides exactly that object:

* ``mvsn::plane_sweep_forward`` -- ONE registered operator (torch.library) whose implementation is the HIP launch
  sequence of multi_view_stereonet.PlaneSweepEngine over libmvsn_hip.so.  It takes the module's parameters as a
  tensor list, so the archive carries the checkpoint and the operator carries no state of its own.
* ``ScriptedMultiViewStereoNet`` -- a scriptable module with the reference's 226-key parameter tree, ``num_levels``
  and the reference's ``forward`` signature and return type; its graph is one call of the operator.
* ``export_archive`` -- script + save; ``tools/make_archive.py`` is the command line.

An archive's graph refers to the operator by name, so the operator must be registered in the process before
``torch.jit.load`` runs: ``import multi_view_stereonet_amd.torchscript`` does it; ``python -m
multi_view_stereonet_amd.run_script test.py ...`` does it and then runs an UNCHANGED script.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .params import build_parameter_tree

OP_SCHEMA = ("plane_sweep_forward(Tensor[] params, Tensor[] left_image_pyr, Tensor[] K_pyr, Tensor[] T_right_in_lefts, "
             "Tensor[] right_image_pyrs_flat, int num_idepth_samples, bool do_cost_volume_filter, "
             "bool[] do_refiners) -> Tensor[]")
NUM_LEVELS = 5


def parameter_names() -> List[str]:
    """The 202 distinct parameters in registration order (the shared extractor appears once)."""
    holder = nn.Module()
    build_parameter_tree(holder)
    return [k for k, _ in holder.named_parameters()]


_lib = torch.library.Library("mvsn", "DEF")
_lib.define(OP_SCHEMA)
_networks: Dict[tuple, object] = {}


def _network_for(params: List[Tensor]):
    """The eager HIP module whose parameters ALIAS `params` (no copy); cached per parameter storage."""
    from .multi_view_stereonet import MultiViewStereoNet
    # storage address AND version counter: the cached module aliases the caller's storages (which also keeps them
    # alive, so an address cannot be recycled under a cached key), but its own version counters do not follow in-place
    # updates of the caller's tensors (load_state_dict on the loaded archive) -- those show up here
    key = tuple((p.data_ptr(), p._version) for p in params)
    net = _networks.get(key)
    if net is None:
        names = parameter_names()
        if len(params) != len(names):
            raise RuntimeError(f"mvsn::plane_sweep_forward: {len(params)} parameter tensors, expected {len(names)}")
        net = MultiViewStereoNet()
        own = dict(net.named_parameters())
        for name, p in zip(names, params):
            if own[name].shape != p.shape:
                raise RuntimeError(f"mvsn::plane_sweep_forward: parameter {name} has shape {tuple(p.shape)}")
            own[name].data = p.detach()
        net.eval()
        if len(_networks) > 8:
            _networks.clear()
        _networks[key] = net
    return net


def _forward_hip(params, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs_flat, num_idepth_samples,
                 do_cost_volume_filter, do_refiners):
    S = len(T_right_in_lefts)
    if S == 0 or len(right_image_pyrs_flat) % S:
        raise RuntimeError("mvsn::plane_sweep_forward: right_image_pyrs_flat must hold S pyramids of equal length")
    levels = len(right_image_pyrs_flat) // S
    right = [list(right_image_pyrs_flat[s * levels:(s + 1) * levels]) for s in range(S)]
    out = _network_for(list(params))(list(left_image_pyr), list(K_pyr), list(T_right_in_lefts), right,
                                     int(num_idepth_samples), bool(do_cost_volume_filter), list(do_refiners))
    return out["left_idepthmap_pyr"] + out["left_idepthmap_raw_pyr"] + out["left_idepthmap_mask_pyr"]


def _forward_cpu(*args):
    raise RuntimeError("MultiViewStereoNet (MI355X build) runs on HIP devices only: move the module and its inputs "
                       "to 'cuda'; there is no CPU implementation of the plane-sweep path")


_lib.impl("plane_sweep_forward", _forward_hip, "CUDA")
_lib.impl("plane_sweep_forward", _forward_cpu, "CPU")


def _scripted_class():
    """The parameter list is spelled out attribute by attribute (TorchScript cannot iterate parameters())."""
    plist = ",\n            ".join("self." + n for n in parameter_names())
    src = f'''
class ScriptedMultiViewStereoNet(nn.Module):
    """Scriptable twin of multi_view_stereonet.MultiViewStereoNet (reference :494-695): same constructor, same
    parameter tree, same forward signature and return type; the graph is one call of mvsn::plane_sweep_forward."""

    def __init__(self):
        super().__init__()
        self.num_levels = {NUM_LEVELS}
        self.min_idepth = 0.0
        build_parameter_tree(self)

    def forward(self, left_image_pyr: List[Tensor], K_pyr: List[Tensor], T_right_in_lefts: List[Tensor],
                right_image_pyrs: List[List[Tensor]], num_idepth_samples: int, do_cost_volume_filter: bool,
                do_refiners: List[bool]) -> Dict[str, List[Optional[Tensor]]]:
        assert len(K_pyr) == self.num_levels
        assert len(left_image_pyr) == self.num_levels
        assert len(T_right_in_lefts) == len(right_image_pyrs)
        assert len(do_refiners) == self.num_levels
        params = [
            {plist}]
        flat: List[Tensor] = []
        for pyr in right_image_pyrs:
            assert len(pyr) == self.num_levels
            for x in pyr:
                flat.append(x)
        out = torch.ops.mvsn.plane_sweep_forward(params, left_image_pyr, K_pyr, T_right_in_lefts, flat,
                                                 num_idepth_samples, do_cost_volume_filter, do_refiners)
        idepth: List[Optional[Tensor]] = []
        raw: List[Optional[Tensor]] = []
        mask: List[Optional[Tensor]] = []
        for lvl in range(self.num_levels):
            idepth.append(out[lvl])
            raw.append(out[self.num_levels + lvl])
            mask.append(out[2 * self.num_levels + lvl])
        return {{"left_idepthmap_pyr": idepth, "left_idepthmap_raw_pyr": raw, "left_idepthmap_mask_pyr": mask}}
'''
    import linecache
    fname = "<multi_view_stereonet_amd.torchscript.ScriptedMultiViewStereoNet>"
    linecache.cache[fname] = (len(src), None, src.splitlines(True), fname)   # TorchScript reads the source back
    ns = {"nn": nn, "torch": torch, "Tensor": Tensor, "List": List, "Dict": Dict, "Optional": Optional,
          "build_parameter_tree": build_parameter_tree, "__name__": __name__}
    exec(compile(src, fname, "exec"), ns)
    return ns["ScriptedMultiViewStereoNet"]


ScriptedMultiViewStereoNet = _scripted_class()


def script_network(state_dict: Optional[Dict[str, Tensor]] = None) -> torch.jit.ScriptModule:
    net = ScriptedMultiViewStereoNet()
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    return torch.jit.script(net.eval())


def export_archive(state_dict: Dict[str, Tensor], path: str) -> str:
    """Write a `stereo_network.pt` that torch.jit.load() accepts (with this module imported first)."""
    script_network(state_dict).save(path)
    return path
