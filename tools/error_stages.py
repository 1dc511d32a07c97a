#!/usr/bin/env python3
"""Which stage contributes the forward's deviation from the CPU oracle?  (1) the GPU's own intermediates against the oracle's
(end-to-end, errors accumulate), (2) each GPU stage run on the ORACLE's input for that stage (the stage's own error):
chain (cost volume from the oracle's plane-0 / left features), regulariser (on the oracle's cost volume), soft-argmin (on the
oracle's filtered cost).  rel = rms(difference) / rms(reference tensor)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, batch_from_meta
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
from oracle import mvsn_oracle as oracle
torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "g2_gta_512x256_d64_s2.npz"
fix = load_golden(name)
w = load_weights("demon_45epochs" if "demon" in name else "gta_sfm_150epochs")
net = MultiViewStereoNet(); net.load_state_dict(w); net = net.cuda().eval()
eng = net.engine()
batch, D = batch_from_meta(fix["meta"], fix.get("jitter", 0.0), bool(fix["smooth"]) if "smooth" in fix else False)
inp = snu.multi_view_unpack_batch(batch, torch.device("cuda"), 5)
cpu = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
gcap, ocap = {}, {}
out = net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5, capture=gcap)
orc = oracle.forward(w, cpu["left_image_pyr"], cpu["K_pyr"], cpu["T_right_in_left"], cpu["right_image_pyr"], D, capture=ocap)
S = len(ocap["sources"])


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


cat = lambda key: torch.cat([ocap["sources"][s][key] for s in range(S)], 0)
print("end to end (GPU intermediates vs the oracle's):")
print("  left features   %.2e" % rel(gcap["left_features"][-1], ocap["left_features"][-1]))
print("  idepth samples  %.2e" % rel(gcap["idepth_samples"], cat("idepth_samples")))
print("  feature volume  %.2e   (last plane %.2e)" % (rel(gcap["feature_volume"], cat("feature_volume")),
                                                      rel(gcap["feature_volume"][:, :, -1], cat("feature_volume")[:, :, -1])))
print("  cost volume     %.2e" % rel(gcap["cost_volume"], cat("cost_volume")))
print("  filtered cost   %.2e" % rel(gcap["filtered_cost"], cat("filtered_cost")))
oraw = torch.cat([oracle.soft_argmin(ocap["sources"][s]["filtered_cost"], ocap["sources"][s]["idepth_samples"]) for s in range(S)], 0)
print("  raw per chain   %.2e" % rel(gcap["raw_per_chain"], oraw))
print("  final idepth    %.2e" % rel(out["left_idepthmap_pyr"][0], orc["left_idepthmap_pyr"][0]))
print("each GPU stage on the ORACLE's input:")
ocost = cat("cost_volume").cuda().contiguous()
gf = eng.cost_volume_filter(ocost.clone())
print("  regulariser     %.2e" % rel(gf, cat("filtered_cost")))
of = cat("filtered_cost").cuda().contiguous()
graw = eng.soft_argmin(of, cat("idepth_samples").cuda().contiguous())
print("  soft-argmin     %.2e" % rel(graw, oraw))
graw2 = eng.soft_argmin(gf, cat("idepth_samples").cuda().contiguous())
print("  regulariser + soft-argmin -> raw   %.2e" % rel(graw2, oraw))
# sensitivity of the raw depth to the regulariser's input: the ORACLE's regulariser on the GPU's cost volume
of2 = torch.cat([oracle.cost_volume_filter(w, "volume_filter4", gcap["cost_volume"][s:s + 1].cpu()) for s in range(S)], 0)
oraw2 = torch.cat([oracle.soft_argmin(of2[s:s + 1], ocap["sources"][s]["idepth_samples"]) for s in range(S)], 0)
print("oracle's regulariser + soft-argmin on the GPU's cost volume -> raw   %.2e   (the chain's share of the raw map's deviation)" % rel(oraw2, oraw))
# ---- the chain's deviation plane by plane (unmasked voxels only), per chain form
fo = cat("feature_volume")
mo = cat("mask_volume")
print("feature volume, rel deviation per plane d (chain 0):")
for form in ("auto", "direct", "winograd"):
    net.options.chain_form = form
    cap2 = {}
    net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5, capture=cap2)
    fg = cap2["feature_volume"].cpu()
    row = []
    for d in (0, 1, 2, 4, 8, 16, 24, 32, 40, 48, 56, D - 2):
        a, b = fg[0, :, d].double(), fo[0, :, d].double()
        row.append("%d:%.1e" % (d, float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))))
    print("  %-9s" % form, " ".join(row))
net.options.chain_form = "auto"
# ---- plane 0: where does its deviation come from?  H0, the full-resolution warp, the extractor on the warped frame
oc = ocap["sources"]
oH0 = torch.cat([oc[s]["H_lvl0_plane0"] for s in range(S)], 0)
print("H (level 0, plane 0)  max |d| %.2e" % float((gcap["H_lvl0_plane0"].cpu().reshape(-1, 9) - oH0.reshape(-1, 9)).abs().max()))
ow = torch.cat([oracle.homography_warp(cpu["right_image_pyr"][s][0], oc[s]["H_lvl0_plane0"])[0][:, :, 0] for s in range(S)], 0)
gw = gcap["warped_fullres"]
gw = gw if gw.dim() == 4 else gw[:, :, 0]
d = (gw.cpu().double() - ow.double())
print("warped full-res frame  rel %.2e   max |d| %.2e   pixels with |d| > 1e-5: %d of %d" % (rel(gw, ow), float(d.abs().max()), int((d.abs() > 1e-5).sum()), d.numel()))
print("plane-0 features       rel %.2e" % rel(gcap["plane0_features"], torch.cat([oc[s]["plane0_features"] for s in range(S)], 0)))
# the extractor alone: GPU extractor on the ORACLE's warped frame
gf0 = eng.feature_network(ow.cuda().contiguous())[-1]
print("GPU extractor on the oracle's warped frame: rel %.2e" % rel(gf0, torch.cat([oc[s]["plane0_features"] for s in range(S)], 0)))
