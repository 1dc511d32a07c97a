"""The reference's UNCHANGED test.py, run through `python -m multi_view_stereonet_amd.run_script` against an archive
from this build ("drops into test.py unchanged").

Build-container only (skipped where /root/reference is absent, e.g. on the GPU box): test.py parses its arguments, reads
params.yaml, builds the reference's own GTA-SfM reader over a miniature tree (the files of g10_datasets.npz),
`torch.jit.load`s OUR stereo_network.pt, moves it to the device, unpacks the first batch with the reference's own
`multi_view_unpack_batch` and calls the network with the seven positional arguments -- at which point this build's
operator runs.  There is no GPU here, so the operator refuses CPU tensors: the test asserts that the script got exactly
that far (on the GPU, `test_torchscript_archive_matches_eager` covers the numbers).
The reference imports torchvision / pyquaternion, absent from this image: two stand-in packages with the entry points
its evaluation path touches are put on PYTHONPATH for the subprocess (same stand-ins as tests/golden/make_dataset_golden.py).
"""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT, load_golden

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "test.py")),
                                reason="the reference tree is only present in the build container")

TORCHVISION = textwrap.dedent('''
    import types
    import numpy as np
    import torch
    from PIL import Image

    class _Compose:
        def __init__(self, transforms): self.transforms = transforms
        def __call__(self, x):
            for t in self.transforms: x = t(x)
            return x

    class _Lambda:
        def __init__(self, lambd): self.lambd = lambd
        def __call__(self, x): return self.lambd(x)

    def _resize(img, size): return img.resize((size[1], size[0]), Image.BILINEAR)

    def _to_tensor(pic):
        if isinstance(pic, np.ndarray):
            if pic.ndim == 2: pic = pic[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(pic.transpose((2, 0, 1))))
            return t.float().div(255) if t.dtype == torch.uint8 else t
        arr = np.array(pic.convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(arr.transpose((2, 0, 1)).copy()).float().div(255)

    def _normalize(tensor, mean, std):
        m = torch.as_tensor(mean, dtype=tensor.dtype).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=tensor.dtype).view(-1, 1, 1)
        return (tensor - m) / s

    transforms = types.ModuleType("torchvision.transforms")
    transforms.Compose, transforms.Lambda = _Compose, _Lambda
    transforms.functional = types.ModuleType("torchvision.transforms.functional")
    transforms.functional.resize, transforms.functional.to_tensor, transforms.functional.normalize = _resize, _to_tensor, _normalize
    utils = types.ModuleType("torchvision.utils")
''')


def test_unchanged_test_py_reaches_the_operator(tmp_path):
    from multi_view_stereonet_amd import torchscript as ts
    from multi_view_stereonet_amd.weights import load_weights
    # stand-in packages + the reference's namespace `datasets` directory ahead of any installed package of that name
    stubs = tmp_path / "stubs"
    (stubs / "torchvision").mkdir(parents=True)
    (stubs / "torchvision" / "__init__.py").write_text(TORCHVISION)
    (stubs / "pyquaternion").mkdir()
    (stubs / "pyquaternion" / "__init__.py").write_text("class Quaternion:\n    pass\n")
    (stubs / "datasets").mkdir()
    (stubs / "datasets" / "__init__.py").write_text(f"__path__ = [{os.path.join(REFERENCE, 'datasets')!r}]\n")
    # <model>/checkpoints/epoch/stereo_network.pt with params.yaml two levels up, as test.py expects (:338)
    model = tmp_path / "model"
    weights_dir = model / "checkpoints" / "epoch0149"
    weights_dir.mkdir(parents=True)
    ts.export_archive(load_weights("gta_sfm_150epochs"), str(weights_dir / "stereo_network.pt"))
    (model / "params.yaml").write_text("size: [32, 48]\nnum_idepth_samples: 8\ncost_volume_filter: True\n"
                                       "refiners: [True, True, True, True, True]\nsupervision_factor: 1.0\n"
                                       "left_right_factor: 0.0\nreconstruction_factor: 0.0\n"
                                       "estimate_right_idepthmap: False\n")
    # the miniature GTA-SfM tree recorded in the dataset fixture
    fix = load_golden("g10_datasets.npz")
    for i, name in enumerate(fix["file_names"]):
        path = tmp_path / str(name)
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_bytes(fix[f"file_{i}"].tobytes())
    split = tmp_path / "gta_sfm_mini_test.txt"            # test.py picks the reader by "gta_sfm" in the file name
    split.write_text((tmp_path / "gta_split.txt").read_text())
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(stubs), REFERENCE, ROOT]), MPLBACKEND="Agg",
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-m", "multi_view_stereonet_amd.run_script", os.path.join(REFERENCE, "test.py"),
                        str(weights_dir), str(tmp_path / "gta"), str(split)], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert "DEFAULTING TO CPU!" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
    # ... data loaded, archive loaded and moved, first batch unpacked, network called: our operator answered
    assert p.returncode != 0 and "HIP devices only" in p.stderr, p.stderr[-2000:]
    assert "multi_view_forward" in p.stderr            # raised from inside the reference's own wrapper (:647-654)
