"""CPU-side checks of the drop-in boundary: checkpoint layout, the C-ABI library and its header,
host plumbing.  No compute calls into the library here (no GPU)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from multi_view_stereonet_amd import MultiViewStereoNet, _native, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.params import parameter_shapes
from multi_view_stereonet_amd.weights import load_weights, PRETRAINED


def test_state_dict_layout_matches_reference_checkpoints():
    net = MultiViewStereoNet()
    assert net.num_levels == 5
    sd = net.state_dict()
    assert len(sd) == 226
    assert sum(v.numel() for v in sd.values()) == 752742
    for name in PRETRAINED:
        w = load_weights(name)
        assert set(w) == set(sd)
        missing, unexpected = net.load_state_dict(w, strict=True)
        assert not missing and not unexpected
    # the source extractor shares the left extractor's storage
    a = net.left_feature_extractor.conv0.weight
    b = net.right_feature_extractor.feature_extractor.conv0.weight
    assert a.data_ptr() == b.data_ptr()
    assert len(list(net.parameters())) == 202


def test_parameter_shapes_examples():
    ps = parameter_shapes()
    assert ps["right_feature_extractor.refiner.conv0.weight"] == (32, 35, 3, 3)
    assert ps["volume_filter4.conv4.weight"] == (1, 32, 3, 3, 3)
    assert ps["refiner0.conv0.weight"] == (32, 4, 3, 3)
    assert ps["refiner3.conv0.weight"] == (32, 36, 3, 3)
    assert "left_feature_extractor.conv0.bias" not in ps and "left_feature_extractor.conv_final.bias" in ps


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mvsn_hip.h")).read()
    declared = set(re.findall(r"\b(mvsn_[a-z0-9_]+)\s*\(", header))
    declared -= {"mvsn_stream_t"}
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_native.library_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mvsn_hip.h but not exported"
    assert declared == set(_native.SIGNATURES), (declared ^ set(_native.SIGNATURES))
    typed = _native.load()
    assert typed.mvsn_abi_version() == _native.ABI_VERSION == 5
    # direct form + the chain's Winograd U + the same three layers in the convolution kernels' layout (stepwise form)
    assert typed.mvsn_feature_refiner_packed_floats() == (9 * 9 * 128 + 2 * 9 * 8 * 128 + 7 * 32) + 25 * 2048 + 25 * 2048
    assert typed.mvsn_incremental_cost_volume_form(16, 32) == _native.CHAIN_WINOGRAD
    assert typed.mvsn_incremental_cost_volume_form(4, 8) == _native.CHAIN_WINOGRAD
    assert typed.mvsn_incremental_cost_volume_form(30, 40) == _native.CHAIN_DIRECT      # planes + U exceed 160 KB
    assert typed.mvsn_incremental_cost_volume_form(5, 6) == _native.CHAIN_DIRECT        # odd rows
    # the stepwise form: workspace grows with the planes (image volume) and is 0 for the plane-resident Winograd form
    ws = typed.mvsn_incremental_cost_volume_workspace_bytes_for
    assert ws(2, 64, 16, 32, _native.CHAIN_WINOGRAD) == 0
    assert ws(1, 96, 30, 40, _native.CHAIN_STEPWISE) > ws(1, 48, 30, 40, _native.CHAIN_STEPWISE) > 0
    assert ws(1, 96, 30, 42, _native.CHAIN_STEPWISE) == 0                               # cols % 4 != 0: no such form
    assert ws(4, 96, 30, 40, _native.CHAIN_DIRECT) == typed.mvsn_incremental_cost_volume_workspace_bytes(4, 30, 40)


def test_conv_planning_is_host_side_and_validates():
    lib = _native.load()
    d = _native.ConvDesc(2, 32, 32, 64, 16, 32, 3, 3, 3, 1, 1)
    assert lib.mvsn_conv_num_tiles(ctypes.byref(d)) == 32 * 2 * 1 * 16     # partial records: tiles x 4 waves x 4 lane rows
    dw = _native.ConvDesc(2, 32, 32, 64, 16, 32, 3, 3, 3, 1, 1, _native.CONV_FP32_WINO)
    assert lib.mvsn_conv_winograd_supported(ctypes.byref(dw)) == 1          # volume form: planes x tiles x 8 waves x 4 rows
    assert lib.mvsn_conv_num_tiles(ctypes.byref(dw)) == 64 * 1 * 32
    assert lib.mvsn_conv_packed_floats(ctypes.byref(dw)) == 3 * 8 * 16 * 128
    assert lib.mvsn_conv_packed_floats(ctypes.byref(d)) == 4 * 27 * 256
    bad = _native.ConvDesc(2, 32, 33, 1, 16, 32, 1, 3, 3, 1, 1)      # c_out > 32
    assert lib.mvsn_conv_packed_floats(ctypes.byref(bad)) == 0
    even = _native.ConvDesc(2, 32, 32, 1, 16, 32, 1, 4, 4, 1, 1)     # even kernels unsupported
    assert lib.mvsn_conv_num_tiles(ctypes.byref(even)) == 0
    # argument validation happens before any device work
    rc = lib.mvsn_soft_argmin(None, None, 1, 4, 16, None, None)
    assert rc == -1 and b"null" in lib.mvsn_last_error()
    # direct form: the step's moved features always travel through the workspace (32 x P floats per chain, ABI 3) ...
    assert lib.mvsn_incremental_cost_volume_workspace_bytes(4, 16, 32) == 4 * 32 * 512 * 4      # ... alone (planes in LDS)
    assert lib.mvsn_incremental_cost_volume_workspace_bytes(4, 30, 40) > 4 * 32 * 1200 * 4      # ... + global planes
    assert lib.mvsn_incremental_cost_volume_repair_workspace_bytes(4, 16, 32) == 0   # the repair form there is Winograd
    assert lib.mvsn_incremental_cost_volume_repair_workspace_bytes(4, 30, 40) == \
        lib.mvsn_incremental_cost_volume_workspace_bytes(4, 30, 40)
    assert lib.mvsn_incremental_cost_volume_workspace_bytes(1, 32, 64) > 0        # config 5 grid


def test_banded_plan_selection_host_logic():
    """Which plan a banded call runs, and what it needs, from the host side alone (no device: the library assumes 256
    CUs): thin bands while the chains fit one of their passes, slabs beyond (mvsn_chain_slab.hip); passes of equal size, or
    full slab passes + one thin pass for a small remainder;
    the status word behind the granules of the largest pass."""
    lib = _native.load()
    groups = lib.mvsn_incremental_cost_volume_banded_groups
    assert [groups(n, 30, 40) for n in (1, 17, 18, 128, 1000)] == [15, 15, 3, 3, 3]
    assert [groups(n, 32, 64) for n in (1, 16, 17, 64, 300)] == [16, 16, 4, 4, 4]
    assert [groups(n, 16, 32) for n in (1, 32, 33, 64, 512)] == [8, 8, 4, 4, 4]       # 16x32: thin bands only
    assert groups(4, 24, 40) == 0 and groups(0, 30, 40) == 0
    form_for = lib.mvsn_incremental_cost_volume_form_for
    assert all(form_for(n, r, c) == _native.CHAIN_BANDED for n in (1, 20, 128, 1000) for r, c in ((30, 40), (32, 64)))
    assert form_for(64, 16, 32) == _native.CHAIN_BANDED and form_for(65, 16, 32) == _native.CHAIN_WINOGRAD
    ws = lib.mvsn_incremental_cost_volume_workspace_bytes_for
    off = lib.mvsn_incremental_cost_volume_status_offset
    for (rows, cols) in ((30, 40), (32, 64)):
        cap = 256 // groups(1000, rows, cols)
        # one pass: the workspace grows with the chains; several passes: it is the largest pass's (equal sizes)
        b18, b19 = ws(18, 96, rows, cols, _native.CHAIN_BANDED), ws(19, 96, rows, cols, _native.CHAIN_BANDED)
        assert 0 < b18 < b19 and off(18, rows, cols) + 64 == b18
        n = 2 * cap + 40                      # three passes of (n + 2) // 3 chains
        per = -(-n // 3)
        assert ws(n, 96, rows, cols, _native.CHAIN_BANDED) == ws(per, 96, rows, cols, _native.CHAIN_BANDED)
        assert off(n, rows, cols) == off(per, rows, cols)
        # a remainder that fits ONE thin-band pass: full slab passes + that thin pass, its granules inside the slab
        # passes' workspace (status block where a full pass has it)
        n = 2 * cap + 2
        assert ws(n, 96, rows, cols, _native.CHAIN_BANDED) == ws(cap, 96, rows, cols, _native.CHAIN_BANDED)
        assert off(n, rows, cols) == off(cap, rows, cols) and groups(n, rows, cols) == groups(1000, rows, cols)


def test_forward_refuses_cpu_tensors():
    net = MultiViewStereoNet()
    batch = synthetic.make_batch(64, 128, 1)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 8, True, [True] * 5)


def test_unpack_batch_semantics():
    batch = synthetic.make_batch(60, 90, 3, batch=2, seed=4, pose_jitter=0.2)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    sizes = [tuple(p.shape[-2:]) for p in inp["left_image_pyr"]]
    assert sizes == [(60, 90), (30, 45), (15, 23), (8, 12), (4, 6)]
    # first source has unit baseline after normalisation, every pose shares that scale
    t0 = inp["T_right_in_left"][0][:, :3, 3].norm(dim=1)
    assert torch.allclose(t0, torch.ones(2), atol=1e-6)
    raw1 = batch["T_right_in_left"][1].squeeze(1)[:, :3, 3]
    assert torch.allclose(inp["T_right_in_left"][1][:, :3, 3], raw1 / inp["baseline"][:, None], atol=1e-6)
    for T, Tinv in zip(inp["T_right_in_left"], inp["T_left_in_right"]):
        assert torch.allclose(T @ Tinv, torch.eye(4).expand(2, 4, 4), atol=1e-5)
    K4 = inp["K_pyr"][4]
    sx = 6.0 / 90.0
    assert torch.allclose(K4[:, 0, 0], inp["K_pyr"][0][:, 0, 0] * sx)
    assert torch.allclose(K4[:, 0, 2], sx * (inp["K_pyr"][0][:, 0, 2] + 0.5) - 0.5)
    # the caller's batch is not modified
    assert batch["T_right_in_left"][0].shape == (2, 1, 4, 4)


def test_multi_view_forward_wrapper_passes_params():
    seen = {}

    def fake(lp, kp, ts, rp, D, flt, refs):
        seen.update(D=D, flt=flt, refs=refs)
        z = [torch.zeros(1)] * 5
        return {"left_idepthmap_pyr": z, "left_idepthmap_raw_pyr": z, "left_idepthmap_mask_pyr": z}

    batch = synthetic.make_batch(32, 64, 1)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    out = snu.multi_view_forward(fake, inp, {"num_idepth_samples": 12})        # DeMoN yaml lacks the flags
    assert seen == {"D": 12, "flt": True, "refs": [True] * 5} and out["stereo_time_ms"] >= 0.0
    snu.multi_view_forward(fake, inp, {"num_idepth_samples": 7, "cost_volume_filter": False,
                                       "refiners": [False, True, True, True, False]})
    assert seen == {"D": 7, "flt": False, "refs": [False, True, True, True, False]}


def test_module_copies_and_pickles_with_a_live_engine():
    """The cached engine holds ctypes function pointers (not picklable); copies drop it and keep the options."""
    import copy
    import io
    import pickle
    net = MultiViewStereoNet()
    net.load_state_dict(load_weights("gta_sfm_150epochs"), strict=True)
    net.options.conv_precision = "bf16x3"

    class FakeEngine:                     # stands in for PlaneSweepEngine (which needs a device to pack weights)
        def __init__(self):
            self.lib = _native.load()
            self.fn = self.lib.mvsn_abi_version

    net._engine, net._engine_key = FakeEngine(), ("stale",)
    with pytest.raises(Exception):
        pickle.dumps(net._engine)
    for dup in (copy.deepcopy(net), pickle.loads(pickle.dumps(net))):
        assert dup._engine is None and dup._engine_key is None
        assert dup.options.conv_precision == "bf16x3" and dup.options is not net.options
        assert torch.equal(dup.refiner0.conv0.weight, net.refiner0.conv0.weight)
        assert dup.left_feature_extractor.conv0.weight.data_ptr() == \
            dup.right_feature_extractor.feature_extractor.conv0.weight.data_ptr()      # sharing survives the copy
    buf = io.BytesIO()
    torch.save(net, buf)
    assert net._engine is not None        # the original keeps its engine


def test_library_is_the_build_of_the_sources_beside_it():
    """__graft_entry__.build() rebuilds on a content mismatch (sha256 of the .hip / .h files in libmvsn_hip.so.sources),
    so a binary that travelled with a snapshot is known to correspond to the snapshot's sources."""
    from multi_view_stereonet_amd import build
    assert build.built_from_current_sources(), "libmvsn_hip.so does not match csrc/: run __graft_entry__.build()"
    assert len(build.source_digest()) == 64
