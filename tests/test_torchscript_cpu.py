"""The `stereo_network.pt` drop-in: a TorchScript archive whose graph calls the registered operator
mvsn::plane_sweep_forward, loadable the way the reference's test.py loads its network (test.py:308-314)."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT
from multi_view_stereonet_amd import synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd import torchscript as ts
from multi_view_stereonet_amd.params import parameter_shapes
from multi_view_stereonet_amd.weights import load_weights


def test_archive_round_trip_keeps_the_reference_interface(tmp_path):
    sd = load_weights("demon_45epochs")
    path = ts.export_archive(sd, str(tmp_path / "stereo_network.pt"))
    assert os.path.getsize(path) < 4 * 2 ** 20
    net = torch.jit.load(path)
    net = net.to(torch.device("cpu"))
    net.eval()
    assert net.num_levels == 5                                       # test.py:199
    loaded = net.state_dict()
    assert list(loaded) == list(parameter_shapes())                   # the 226 checkpoint keys, same order
    assert all(torch.equal(loaded[k], sd[k]) for k in sd)
    assert "mvsn::plane_sweep_forward" in str(net.graph)
    schema = str(net.forward.schema)
    assert "Tensor[][] right_image_pyrs, int num_idepth_samples, bool do_cost_volume_filter, bool[] do_refiners" in schema
    assert schema.endswith("-> Dict(str, Tensor?[])")                # Dict[str, List[Optional[Tensor]]] (:545)
    assert len(ts.parameter_names()) == 202
    # the seven-argument positional call of multi_view_forward (multi_view_stereonet_utils.py:647-654) reaches the
    # operator; on CPU tensors it refuses exactly as the eager module does (no CPU path)
    inp = snu.multi_view_unpack_batch(synthetic.make_batch(64, 128, 1), torch.device("cpu"), 5)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        snu.multi_view_forward(net, inp, {"num_idepth_samples": 8})
    with pytest.raises(Exception):                                    # wrong pyramid length -> scripted assert
        net(inp["left_image_pyr"][:4], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 8, True,
            [True] * 5)


LOADER = """
import os, sys, torch
weights_dir = sys.argv[1]
stereo_network = torch.jit.load(os.path.join(weights_dir, "stereo_network.pt"))   # test.py:311, verbatim
stereo_network = stereo_network.to(torch.device("cpu"))
stereo_network.eval()
print("LOADED", stereo_network.num_levels, len(stereo_network.state_dict()))
"""


def test_unchanged_script_runs_under_the_launcher(tmp_path):
    ts.export_archive(load_weights("gta_sfm_150epochs"), str(tmp_path / "stereo_network.pt"))
    script = tmp_path / "load_models.py"
    script.write_text(LOADER)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    ok = subprocess.run([sys.executable, "-m", "multi_view_stereonet_amd.run_script", str(script), str(tmp_path)],
                        capture_output=True, text=True, env=env, timeout=600)
    assert ok.returncode == 0 and "LOADED 5 226" in ok.stdout, ok.stderr[-1500:]
    # without the operator registered the archive cannot resolve its one call: the launcher (or one import) is needed
    bare = subprocess.run([sys.executable, str(script), str(tmp_path)], capture_output=True, text=True, env=env,
                          timeout=600)
    assert bare.returncode != 0 and "mvsn" in bare.stderr
