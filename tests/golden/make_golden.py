#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only: imports the eager reference from /root/reference (with a stub
``torchvision``, which the hot path imports but never uses), loads the extracted pretrained
tensors, runs seeded synthetic inputs (multi_view_stereonet_amd.synthetic) through it and
records inputs, named intermediates and outputs as .npz files.  The fixtures are data; the
reference itself never travels to the GPU box.

    python tests/golden/make_golden.py

Fixtures:
  g1_gta_128x64_d16_s1.npz      config 1 with pretrained GTA weights, every intermediate
  g1_init_128x64_d16_s1.npz     config 1 with default-statistics random weights (weights.default_init_weights(0))
  g1b_gta_96x80_d8_s2_b2.npz    batch 2, two sources, per-element poses, odd-ish pyramid sizes
  g2_gta_512x256_d64_s2.npz     headline config: outputs only (inputs regenerated from the seed)
  g2s_gta_512x256_d64_s2.npz    same on the smooth synthetic scene
  g3_demon_640x480_d96_s1.npz   config 4 shape with DeMoN weights: outputs only
  g4_units.npz                  single-op pins on tiny tensors
  g6_flags_128x64.npz           do_cost_volume_filter=False / refiners off variants
  g7_depth_metrics.npz          test.py's depth metrics on a synthetic truth/estimate pair
  g8_two_view_128x64_d12.npz    2-view twins (unpack_batch / forward) with the right-view estimate
  g9_two_view_consistency.npz   occlusion masks + left/right consistency loss of the reference's losses.py, 5 levels
  gc2_gta_512x256_d64_s1.npz    BASELINE config 2 (one source view): outputs only
  gc3_gta_512x256_d64_s5.npz    BASELINE config 3 (five source views): outputs only
  gc5_gta_1024x512_d128_s4.npz  BASELINE config 5 geometry (fp32 reference): idepth_0 (fp32), idepth_4, mask_4
  g11_incremental_homographies.npz  every homography the reference hands its warper during a forward (H0, H family, H_inc)

    python tests/golden/make_golden.py [name-prefix ...]     # e.g. "gc" regenerates only the gc* fixtures
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
sys.path.insert(0, "/root/reference")

from multi_view_stereonet.multi_view_stereonet import (  # noqa: E402  (reference)
    MultiViewStereoNet, FeatureRefiner, CostVolumeFilter, IDepthmapRefiner, MaskUpsampler,
    extract_idepthmap, create_idepth_samples, create_plane_sweep_homographies, PlaneSweepWarper)
import multi_view_stereonet.multi_view_stereonet as ref_mod  # noqa: E402
from multi_view_stereonet import multi_view_stereonet_utils as ref_snu  # noqa: E402
from stereo import image_predictor as ref_ip  # noqa: E402
from utils import image_utils as ref_iu  # noqa: E402

from multi_view_stereonet_amd import synthetic  # noqa: E402
from multi_view_stereonet_amd.weights import load_weights, default_init_weights  # noqa: E402
from multi_view_stereonet_amd import multi_view_stereonet_utils as my_snu  # noqa: E402

torch.set_grad_enabled(False)


def npy(t):
    return t.detach().cpu().numpy()


def ref_net(weights=None, seed=0):
    """Reference eager module with either extracted pretrained tensors or the build's seeded
    default-statistics init (regenerable from the seed, so never stored in a fixture)."""
    net = MultiViewStereoNet().eval()
    sd = load_weights(weights) if weights is not None else default_init_weights(seed)
    net.load_state_dict(sd, strict=True)
    return net


def run_reference(net, batch, D, do_filter=True, refiners=(True,) * 5, capture=True):
    """Unpack with the REFERENCE's unpacker, run forward, optionally hook intermediates."""
    import copy
    b = copy.deepcopy(batch)
    inputs = ref_snu.multi_view_unpack_batch(b, torch.device("cpu"), net.num_levels)
    rec = {"samples": [], "H": [], "rfe_out": [], "vf_in": [], "vf_out": [], "fe_out": [], "warp": []}
    hooks = []
    if capture:
        orig_samples = ref_mod.create_idepth_samples
        orig_H = ref_mod.create_plane_sweep_homographies

        def samples_spy(*a, **k):
            r = orig_samples(*a, **k)
            rec["samples"].append(r.clone())
            return r

        def H_spy(*a, **k):
            r = orig_H(*a, **k)
            rec["H"].append(r.clone())
            return r

        ref_mod.create_idepth_samples = samples_spy
        ref_mod.create_plane_sweep_homographies = H_spy
        def on_rfe(m, i, o):
            rec["rfe_out"].append((o[0].clone(), o[1].clone()))

        def on_vf(m, i, o):
            rec["vf_in"].append(i[0].clone())
            rec["vf_out"].append(o.clone())

        def on_fe(m, i, o):
            rec["fe_out"].append([t.clone() for t in o])

        def on_warp(m, i, o):
            if len(rec["warp"]) < 4:
                rec["warp"].append((o[0].clone(), o[1].clone()))

        hooks.append(net.right_feature_extractor.register_forward_hook(on_rfe))
        hooks.append(net.volume_filter4.register_forward_hook(on_vf))
        hooks.append(net.left_feature_extractor.register_forward_hook(on_fe))
        hooks.append(net.right_feature_extractor.warper.register_forward_hook(on_warp))
    try:
        out = net(inputs["left_image_pyr"], inputs["K_pyr"], inputs["T_right_in_left"],
                  inputs["right_image_pyr"], D, do_filter, list(refiners))
    finally:
        for h in hooks:
            h.remove()
        if capture:
            ref_mod.create_idepth_samples = orig_samples
            ref_mod.create_plane_sweep_homographies = orig_H
    return inputs, out, rec


def pack_outputs(d, out):
    for lvl in range(5):
        d[f"idepth_{lvl}"] = npy(out["left_idepthmap_pyr"][lvl])
        d[f"raw_{lvl}"] = npy(out["left_idepthmap_raw_pyr"][lvl])
    return d


def check_unpack(batch, inputs, levels=5):
    """The build's unpacker must reproduce the reference's inputs bit-for-bit (row a14)."""
    mine = my_snu.multi_view_unpack_batch(batch, torch.device("cpu"), levels)
    for a, b in zip(mine["K_pyr"], inputs["K_pyr"]):
        assert torch.equal(a, b)
    for a, b in zip(mine["left_image_pyr"], inputs["left_image_pyr"]):
        assert torch.equal(a, b)
    for a, b in zip(mine["T_right_in_left"], inputs["T_right_in_left"]):
        assert torch.equal(a, b)
    assert torch.equal(mine["baseline"], inputs["baseline"])


def full_capture(name, weights, rows, cols, D, S, B=1, seed=1, jitter=0.0, init_seed=0, store_weights=False):
    net = ref_net(weights, init_seed)
    batch = synthetic.make_batch(rows, cols, S, batch=B, seed=seed, pose_jitter=jitter)
    inputs, out, rec = run_reference(net, batch, D)
    check_unpack(batch, inputs)
    d = {"meta": np.array([rows, cols, D, S, B, seed], dtype=np.int64), "jitter": np.float32(jitter)}
    d["left_image"] = npy(batch["left_image"])
    d["K"] = npy(batch["K"])
    for s in range(S):
        d[f"right_image_{s}"] = npy(batch["right_image"][s])
        d[f"T_{s}"] = npy(batch["T_right_in_left"][s])
        d[f"T_unpacked_{s}"] = npy(inputs["T_right_in_left"][s])
        d[f"idepth_samples_{s}"] = npy(rec["samples"][s])
        d[f"H_lvl0_plane0_{s}"] = npy(rec["H"][2 * s])
        d[f"H_{s}"] = npy(rec["H"][2 * s + 1])
        d[f"feature_volume_{s}"] = npy(rec["rfe_out"][s][0])
        d[f"mask_volume_{s}"] = npy(rec["rfe_out"][s][1])
        d[f"cost_volume_{s}"] = npy(rec["vf_in"][s])
        d[f"filtered_cost_{s}"] = npy(rec["vf_out"][s])
        d[f"plane0_features_{s}"] = npy(rec["fe_out"][1 + s][-1])
    # warper calls of source 0: [0] full-res plane 0, [1] L4 image volume, [2..] incremental
    d["warped_fullres_0"] = npy(rec["warp"][0][0])
    d["image_volume_0"] = npy(rec["warp"][1][0])
    d["image_volume_mask_0"] = npy(rec["warp"][1][1])
    d["baseline"] = npy(inputs["baseline"])
    for lvl in range(5):
        d[f"K_pyr_{lvl}"] = npy(inputs["K_pyr"][lvl])
        d[f"left_feat_{lvl}"] = npy(rec["fe_out"][0][lvl]) if lvl > 0 else np.zeros(1, np.float32)
        d[f"mask_{lvl}"] = np.packbits(npy(out["left_idepthmap_mask_pyr"][lvl]))
        d[f"mask_shape_{lvl}"] = np.array(out["left_idepthmap_mask_pyr"][lvl].shape, dtype=np.int64)
    d["left_image_lvl4"] = npy(inputs["left_image_pyr"][4])
    pack_outputs(d, out)
    if store_weights:
        for k, v in net.state_dict().items():
            if not k.startswith("right_feature_extractor.feature_extractor."):
                d["w:" + k] = npy(v)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")


def outputs_only(name, weights, rows, cols, D, S, seed, smooth=False, slim=False):
    """`slim` (large configs): no per-source filtered cost / feature planes, only the output pins."""
    net = ref_net(weights)
    batch = synthetic.make_batch(rows, cols, S, batch=1, seed=seed, smooth=smooth)
    inputs, out, rec = run_reference(net, batch, D)
    check_unpack(batch, inputs)
    d = {"meta": np.array([rows, cols, D, S, 1, seed], dtype=np.int64), "smooth": np.int64(smooth)}
    for s in range(S):
        d[f"idepth_samples_{s}"] = npy(rec["samples"][s])
        d[f"H_{s}"] = npy(rec["H"][2 * s + 1])
        d[f"mask_volume_count_{s}"] = np.int64(rec["rfe_out"][s][1].sum().item())
        if slim:
            continue
        d[f"filtered_cost_{s}"] = npy(rec["vf_out"][s]).astype(np.float32)
        fv = rec["rfe_out"][s][0]
        d[f"feature_volume_last_plane_{s}"] = npy(fv[:, :, -1])
    for lvl in range(5):
        d[f"mask_count_{lvl}"] = np.int64(out["left_idepthmap_mask_pyr"][lvl].sum().item())
    d["mask_4"] = np.packbits(npy(out["left_idepthmap_mask_pyr"][4]))
    d["idepth_0"] = npy(out["left_idepthmap_pyr"][0])
    d["idepth_4"] = npy(out["left_idepthmap_pyr"][4])
    d["raw_4"] = npy(out["left_idepthmap_raw_pyr"][4])
    if not slim:
        d["raw_0"] = npy(out["left_idepthmap_raw_pyr"][0]).astype(np.float16)  # coarse pin only
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")


def flags_capture(name):
    net = ref_net("gta_sfm_150epochs")
    batch = synthetic.make_batch(64, 128, 2, batch=1, seed=5)
    d = {"meta": np.array([64, 128, 16, 2, 1, 5], dtype=np.int64)}
    variants = {"nofilter": (False, (True,) * 5),
                "norefine4": (True, (True, True, True, True, False)),
                "norefine_all": (True, (False,) * 5),
                "norefine_0_2": (True, (False, True, False, True, True))}
    for key, (flt, refs) in variants.items():
        _, out, _ = run_reference(net, batch, 16, flt, refs, capture=False)
        for lvl in (0, 4):
            d[f"{key}:idepth_{lvl}"] = npy(out["left_idepthmap_pyr"][lvl])
            d[f"{key}:raw_{lvl}"] = npy(out["left_idepthmap_raw_pyr"][lvl])
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok")


def unit_pins(name):
    g = torch.Generator().manual_seed(11)
    d = {}
    sd = load_weights("gta_sfm_150epochs")
    # --- HomographyImagePredictor on special homographies ---------------------------------
    img = torch.rand(4, 5, 6, 9, generator=g) * 2 - 1
    Hs = torch.eye(3).repeat(4, 1, 1)
    Hs[1, 0, 2] = 0.5                                   # half-pixel shift in x
    Hs[2] = torch.tensor([[1.0, 0.0, 5.0], [0.0, 1.0, -2.0], [0.0, 0.0, 1.0]])   # half the image OOB
    Hs[3] = torch.tensor([[1.0, 0.1, 0.3], [-0.05, 1.1, 0.2], [0.01, 0.002, -1.0]])  # z < 0
    pred, mask = ref_ip.HomographyImagePredictor()(Hs, img)
    d.update(hip_image=npy(img), hip_H=npy(Hs), hip_pred=npy(pred), hip_mask=npy(mask))
    # --- PlaneSweepWarper: (B,C,h,w) x (B,n,3,3) -> volume + mask, with zeroing ----------
    img2 = torch.rand(2, 3, 7, 10, generator=g)
    H2 = torch.eye(3).repeat(2, 3, 1, 1) + 0.05 * (torch.rand(2, 3, 3, 3, generator=g) - 0.5)
    H2[..., 2, :2] *= 0.05
    vol, vmask = PlaneSweepWarper()(img2, H2)
    d.update(psw_image=npy(img2), psw_H=npy(H2), psw_volume=npy(vol), psw_mask=npy(vmask))
    # --- get_fronto_parallel_homography / create_plane_sweep_homographies ----------------
    batch = synthetic.make_batch(48, 80, 2, batch=2, seed=3, pose_jitter=0.3)
    inputs = ref_snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    T = inputs["T_right_in_left"][1]
    K4 = inputs["K_pyr"][4]
    samples = create_idepth_samples(T.clone(), K4, 3, 5, 7)
    Hfam = create_plane_sweep_homographies(T, K4, samples, [3, 5])
    d.update(fph_T=npy(T), fph_K=npy(K4), fph_samples=npy(samples), fph_H=npy(Hfam))
    disp = 6.0 * torch.ones(2, 1, 3, 5)
    d["d2i_idepth"] = npy(ref_ip.disparity_to_idepth(K4, T, disp))
    # --- FeatureRefiner one step ---------------------------------------------------------
    fr = FeatureRefiner(32).eval()
    fr.load_state_dict({k.split("refiner.", 1)[1]: v for k, v in sd.items()
                        if k.startswith("right_feature_extractor.refiner.")})
    fimg = torch.rand(2, 3, 6, 8, generator=g) * 2 - 1
    ffeat = torch.randn(2, 32, 6, 8, generator=g)
    d.update(fr_image=npy(fimg), fr_feat=npy(ffeat), fr_out=npy(fr(fimg, ffeat)))
    # --- CostVolumeFilter on a small volume ----------------------------------------------
    cvf = CostVolumeFilter(32).eval()
    cvf.load_state_dict({k.split("volume_filter4.", 1)[1]: v for k, v in sd.items()
                         if k.startswith("volume_filter4.")})
    vol_in = torch.rand(2, 32, 8, 4, 8, generator=g)
    vout = cvf(vol_in)
    d.update(cvf_in=npy(vol_in), cvf_out=npy(vout))
    # --- extract_idepthmap ---------------------------------------------------------------
    idv = torch.linspace(0, 1.5, 8).view(1, 8, 1, 1).repeat(2, 1, 4, 8)
    d.update(sm_idepth=npy(idv[:, :, 0, 0]), sm_out=npy(extract_idepthmap(vout, idv)))
    # --- IDepthmapRefiner level 1 (36 ch) and level 0 (4 ch) ----------------------------
    for lvl, ch in ((1, 35), (0, 3)):
        r = IDepthmapRefiner(ch, 1.0).eval()
        r.load_state_dict({k.split(f"refiner{lvl}.", 1)[1]: v for k, v in sd.items()
                           if k.startswith(f"refiner{lvl}.")})
        guide = torch.randn(2, ch, 20, 24, generator=g)
        prior = torch.rand(2, 1, 20, 24, generator=g) * 30.0
        d[f"idr{lvl}_guide"] = npy(guide)
        d[f"idr{lvl}_prior"] = npy(prior)
        d[f"idr{lvl}_out"] = npy(r(guide, prior))
    # --- MaskUpsampler / bilinear upsample on odd sizes ---------------------------------
    m = torch.rand(2, 5, 8, 15, generator=g) > 0.5
    d.update(mu_in=npy(m), mu_out=npy(MaskUpsampler()(m, [15, 30])))
    x = torch.rand(2, 1, 8, 15, generator=g)
    d.update(up_in=npy(x), up_out=npy(torch.nn.functional.interpolate(x, size=[15, 30], mode="bilinear",
                                                                     align_corners=False)))
    # --- build_image_pyramid on odd sizes ------------------------------------------------
    im = torch.rand(1, 3, 30, 45, generator=g)
    pyr = ref_iu.build_image_pyramid(im, 4)
    d["pyr_in"] = npy(im)
    for i, p in enumerate(pyr):
        d[f"pyr_{i}"] = npy(p)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok")


def metric_pins(name):
    """Depth metrics: run the reference's own get_depth_prediction_metrics (test.py:41-71).  test.py
    imports the dataset stack (torchvision, pyquaternion: absent), so only that function is compiled
    out of its source, here in the build container."""
    import ast
    mod = ast.parse(open("/root/reference/test.py").read())
    fn = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name == "get_depth_prediction_metrics"][0]
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "ref_test.py", "exec"), ns)
    rng = np.random.default_rng(5)
    t_ = rng.uniform(0.6, 9.0, size=(40, 50)).astype(np.float32)
    e_ = (t_ * rng.uniform(0.7, 1.5, size=t_.shape)).astype(np.float32)
    m = ns["get_depth_prediction_metrics"](t_.reshape(-1), e_.reshape(-1))
    np.savez_compressed(os.path.join(HERE, name), depth_true=t_, depth_est=e_,
                        **{k: np.float64(v) for k, v in m.items()})
    print(name, "ok")


def two_view_pins(name):
    """The 2-view twins snu.unpack_batch / snu.forward with estimate_right_idepthmap (section 8f rank 4)."""
    import copy
    net = ref_net("gta_sfm_150epochs")
    mv = synthetic.make_batch(64, 128, 1, batch=2, seed=13, pose_jitter=0.2)
    batch = {"left_image": mv["left_image"], "right_image": mv["right_image"][0], "K": mv["K"],
             "T_right_in_left": mv["T_right_in_left"][0], "left_filename": ["l"] * 2, "right_filename": ["r"] * 2}
    inputs = ref_snu.unpack_batch(copy.deepcopy(batch), torch.device("cpu"), 5)
    params = {"num_idepth_samples": 12, "cost_volume_filter": True, "refiners": [True] * 5,
              "estimate_right_idepthmap": True}
    out = ref_snu.forward(net, inputs, params)
    d = {"T_right_in_left": npy(inputs["T_right_in_left"]), "T_left_in_right": npy(inputs["T_left_in_right"]),
         "baseline": npy(inputs["baseline"]),
         "left_idepth_0": npy(out["left_idepthmap_pyr"][0]), "right_idepth_0": npy(out["right_idepthmap_pyr"][0]),
         "left_idepth_4": npy(out["left_idepthmap_pyr"][4]), "right_idepth_4": npy(out["right_idepthmap_pyr"][4])}
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok")


def consistency_pins(name):
    """Occlusion masks and the left/right consistency loss (section 8f rank 4): the reference's own losses.py
    (:42-82, :112-160) over its IDepthmapProjector, every pyramid level.  The idepth maps are seeded smooth maps in a
    geometrically sensible range (0.02 .. 0.1 at unit baseline: the network's estimates on white-noise frames project
    out of the other image everywhere and pin nothing); poses / intrinsics are the two-view batch of g8."""
    import copy
    from multi_view_stereonet import losses as ref_losses
    mv = synthetic.make_batch(64, 128, 1, batch=2, seed=13, pose_jitter=0.2)
    batch = {"left_image": mv["left_image"], "right_image": mv["right_image"][0], "K": mv["K"],
             "T_right_in_left": mv["T_right_in_left"][0], "left_filename": ["l"] * 2, "right_filename": ["r"] * 2}
    inputs = ref_snu.unpack_batch(copy.deepcopy(batch), torch.device("cpu"), 5)
    gen = torch.Generator().manual_seed(99)
    d = {"T_right_in_left": npy(inputs["T_right_in_left"]), "T_left_in_right": npy(inputs["T_left_in_right"])}
    Ls, Rs, left_occ, right_occ = [], [], [], []
    for lvl in range(5):
        rows, cols = inputs["left_image_pyr"][lvl].shape[-2:]
        L = 0.06 + 0.04 * synthetic._smooth_image(gen, 2, rows, cols)[:, :1]
        R = 0.06 + 0.04 * synthetic._smooth_image(gen, 2, rows, cols)[:, :1]
        R[:, :, : rows // 3] += 0.05                      # a nearer band: real occlusions
        dummy = torch.zeros(2, 1, rows, cols, dtype=torch.bool)
        lo = ref_losses.get_occlusion_mask(inputs["K_pyr"][lvl], inputs["T_right_in_left"], L, dummy, R, dummy)
        ro = ref_losses.get_occlusion_mask(inputs["K_pyr"][lvl], inputs["T_left_in_right"], R, dummy, L, dummy)
        Ls.append(L), Rs.append(R), left_occ.append(lo), right_occ.append(ro)
        d[f"K_{lvl}"] = npy(inputs["K_pyr"][lvl])
        d[f"left_idepth_{lvl}"], d[f"right_idepth_{lvl}"] = npy(L), npy(R)
        d[f"left_occlusion_{lvl}"], d[f"right_occlusion_{lvl}"] = npy(lo), npy(ro)
        if lvl in (2, 4):
            uv, idp, inv = ref_ip.IDepthmapProjector()(inputs["K_pyr"][lvl], inputs["T_right_in_left"], L)
            d[f"proj_uv_{lvl}"], d[f"proj_idepth_{lvl}"], d[f"proj_invalid_{lvl}"] = npy(uv), npy(idp), npy(inv)
    loss = ref_losses.left_right_idepthmap_consistency_losses(
        inputs["T_right_in_left"], inputs["T_left_in_right"], inputs["K_pyr"], Ls, left_occ, Rs, right_occ)
    d["left_right_loss"] = np.float64(loss.item())
    per_level = []
    for lvl in range(5):
        pick = lambda pyr: [t if i == lvl else None for i, t in enumerate(pyr)]   # noqa: E731
        per_level.append(ref_losses.left_right_idepthmap_consistency_losses(
            inputs["T_right_in_left"], inputs["T_left_in_right"], inputs["K_pyr"], pick(Ls), left_occ, pick(Rs),
            right_occ).item())
    d["left_right_loss_per_level"] = np.array(per_level, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok", "loss", d["left_right_loss"], [(int(m.sum()), m.numel()) for m in left_occ])


INC_CASES = [  # (tag, rows, cols, D, S, B, seed, jitter): the golden forwards' geometry + two batched, jittered families
    ("headline", 256, 512, 64, 2, 1, 7, 0.0), ("config4", 480, 640, 96, 1, 1, 9, 0.0), ("config5", 512, 1024, 128, 4, 1, 25, 0.0),
    ("config2", 256, 512, 64, 1, 1, 21, 0.0), ("config3", 256, 512, 64, 5, 1, 23, 0.0),
    ("b2_jitter", 80, 96, 8, 2, 2, 2, 0.3), ("b2_wide_jitter", 256, 512, 64, 2, 2, 40, 0.5)]


def incremental_homography_pins(name):
    """Every homography the reference's PlaneSweepWarper is HANDED during a forward (multi_view_stereonet.py:254-283):
    call 0 of a source = the level-0 plane-0 matrix, call 1 = the level-4 family, calls 2.. = the incremental
    `inverse(H[d-1]) @ H[d]` of every step -- captured by a pre-hook on the warper, so the values are the reference's own
    (its tensor layouts, hence its ATen paths, included).  Inputs are regenerated from the seeds by the tests."""
    net = ref_net("gta_sfm_150epochs")
    d = {}
    for tag, rows, cols, D, S, B, seed, jitter in INC_CASES:
        batch = synthetic.make_batch(rows, cols, S, batch=B, seed=seed, pose_jitter=jitter)
        seen = []
        hook = net.right_feature_extractor.warper.register_forward_pre_hook(lambda m, i: seen.append(i[1].clone()))
        spy = {}
        orig_samples = ref_mod.create_idepth_samples

        def samples_spy(*a, **k):
            r = orig_samples(*a, **k)
            spy.setdefault("samples", []).append(r.clone())
            return r
        ref_mod.create_idepth_samples = samples_spy
        try:
            inputs, _, _ = run_reference(net, batch, D, capture=False)
        finally:
            hook.remove()
            ref_mod.create_idepth_samples = orig_samples
        assert len(seen) == S * (D + 1), (len(seen), S, D)
        for s in range(S):
            calls = seen[s * (D + 1):(s + 1) * (D + 1)]
            d[f"{tag}_H0_{s}"] = npy(calls[0])                                  # (B,1,3,3)
            d[f"{tag}_H4_{s}"] = npy(calls[1])                                  # (B,D,3,3)
            d[f"{tag}_Hinc_{s}"] = npy(torch.cat(calls[2:], 1))                 # (B,D-1,3,3)
            d[f"{tag}_samples_{s}"] = npy(spy["samples"][s])                    # (B,D)
            # the pose as the reference's unpacker hands it over (translations over the first source's baseline,
            # multi_view_stereonet_utils.py:597-604): torch's reduction there rounds differently from host to host, so
            # the tests take it from here instead of re-deriving it on whatever host they run on
            d[f"{tag}_T_{s}"] = npy(inputs["T_right_in_left"][s])               # (B,4,4)
        d[f"{tag}_meta"] = np.array([rows, cols, D, S, B, seed, int(round(jitter * 100))], np.int64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, "ok")


def main():
    torch.set_num_threads(8)
    G = "gta_sfm_150epochs"
    jobs = [
        ("g8_two_view_128x64_d12.npz", lambda n: two_view_pins(n)),
        ("g7_depth_metrics.npz", lambda n: metric_pins(n)),
        ("g1_gta_128x64_d16_s1.npz", lambda n: full_capture(n, G, 64, 128, 16, 1, seed=1)),
        ("g1_init_128x64_d16_s1.npz", lambda n: full_capture(n, None, 64, 128, 16, 1, seed=1, init_seed=0)),
        ("g1b_gta_96x80_d8_s2_b2.npz", lambda n: full_capture(n, G, 80, 96, 8, 2, B=2, seed=2, jitter=0.3)),
        ("g2_gta_512x256_d64_s2.npz", lambda n: outputs_only(n, G, 256, 512, 64, 2, seed=7)),
        ("g2s_gta_512x256_d64_s2.npz", lambda n: outputs_only(n, G, 256, 512, 64, 2, seed=7, smooth=True)),
        ("g3_demon_640x480_d96_s1.npz", lambda n: outputs_only(n, "demon_45epochs", 480, 640, 96, 1, seed=9)),
        ("g6_flags_128x64.npz", lambda n: flags_capture(n)),
        ("g4_units.npz", lambda n: unit_pins(n)),
        # BASELINE configs 2, 3 and 5 at their stated sizes (reference loop multi_view_stereonet.py:564-627)
        ("gc2_gta_512x256_d64_s1.npz", lambda n: outputs_only(n, G, 256, 512, 64, 1, seed=21, slim=True)),
        ("gc3_gta_512x256_d64_s5.npz", lambda n: outputs_only(n, G, 256, 512, 64, 5, seed=23, slim=True)),
        ("gc5_gta_1024x512_d128_s4.npz", lambda n: outputs_only(n, G, 512, 1024, 128, 4, seed=25, slim=True)),
        ("g9_two_view_consistency.npz", lambda n: consistency_pins(n)),
        ("g11_incremental_homographies.npz", lambda n: incremental_homography_pins(n)),
    ]
    want = sys.argv[1:]
    for name, job in jobs:
        if not want or any(name.startswith(w) for w in want):
            job(name)


if __name__ == "__main__":
    main()
