#!/usr/bin/env python3
"""How much of the forward's deviation from the CPU oracle / the reference fixture is the plane-sweep GEOMETRY (idepth samples,
H at levels 0 and 4, H_inc: up to round 5 the library computed them in fp64 and rounded once, the reference chains fp32
operations -- a 1-ulp difference in H feeds the full-resolution warp of a random-noise frame; since round 6 the set-up kernel
follows the reference's fp32 chain, `-DMVSN_SETUP_FP64_H=1` restores the old values)?  The same GPU forward twice: with the library's
geometry, and with the oracle's H0 / H4 / H_inc handed to the kernels instead."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, batch_from_meta, rel_err_per_pixel, rel_err
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
from oracle import mvsn_oracle as oracle
torch.set_grad_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "g2_gta_512x256_d64_s2.npz"
fix = load_golden(name)
w = load_weights("demon_45epochs" if "demon" in name else "gta_sfm_150epochs")
net = MultiViewStereoNet(); net.load_state_dict(w); net = net.cuda().eval()
net.options.plan_max_chains = 0          # eager: the patched set-up below must run every time
for item in filter(None, os.environ.get("MVSN_OPTS", "").split(",")):      # engine option overrides, "name=value,..."
    k, v = item.split("=")
    cur = getattr(net.options, k)
    setattr(net.options, k, type(cur)(int(v)) if isinstance(cur, (bool, int)) else type(cur)(v))
eng = net.engine()
batch, D = batch_from_meta(fix["meta"], fix.get("jitter", 0.0), bool(fix["smooth"]) if "smooth" in fix else False)
inp = snu.multi_view_unpack_batch(batch, torch.device("cuda"), 5)
cpu = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
ocap = {}
orc = oracle.forward(w, cpu["left_image_pyr"], cpu["K_pyr"], cpu["T_right_in_left"], cpu["right_image_pyr"], D, capture=ocap)
S = len(ocap["sources"])
ref = torch.from_numpy(fix["idepth_0"])
run = lambda: net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5)


def report(tag, out):
    got = out["left_idepthmap_pyr"][0].cpu()
    for what, r in (("reference fixture", ref), ("CPU oracle", orc["left_idepthmap_pyr"][0])):
        mx, p999 = rel_err_per_pixel(got, r)
        mean_rel, max_rel = rel_err(got, r)
        print(f"{tag:30s} vs {what:17s}: per-pixel max {mx:.2e} p99.9 {p999:.2e}  mean-rel {mean_rel:.2e} max-rel {max_rel:.2e}")


report("library geometry", run())
oH0 = torch.cat([ocap["sources"][s]["H_lvl0_plane0"] for s in range(S)], 0).cuda().contiguous()
oH4 = torch.cat([ocap["sources"][s]["H"] for s in range(S)], 0).cuda().contiguous()
oHinc = torch.eye(3).repeat(oH4.shape[0], D, 1, 1)
H4c = oH4.cpu()
for d in range(1, D):
    oHinc[:, d] = oracle.inv3x3(H4c[:, d - 1]) @ H4c[:, d]
oHinc = oHinc.cuda().contiguous()
for which in ("H0", "H0 + H4 + H_inc"):
    for fn_name in ("plane_sweep_setup", "plane_sweep_setup_sources"):
        orig = getattr(type(eng), fn_name)

        def patched(self, *a, _orig=orig, _which=which, **k):
            samples, H4, Hinc, H0, base = _orig(self, *a, **k)
            if _which == "H0":
                return samples, H4, Hinc, oH0.reshape(H0.shape), base
            return samples, oH4.reshape(H4.shape), oHinc.reshape(Hinc.shape), oH0.reshape(H0.shape), base
        setattr(eng, fn_name, patched.__get__(eng))
    report(f"the oracle's {which}", run())
