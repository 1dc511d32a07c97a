"""GPU parity: every entry point of libmvsn_hip.so against the CPU oracle on the same seeded inputs,
and the whole forward against the reference-generated golden fixtures.

Tolerances (fp32 path; the reference's own MKLDNN on/off spread is ~6e-6 max-rel):
  single ops        rtol 1e-4 / atol 1e-5, masks bit-exact
  chain (63 steps)  mean-rel 1e-5, max-rel (vs max |ref|) 1e-4 (measured 6e-7 / 1e-6 in both forms)
  intermediates and every output level: mean-rel 2e-4, max-rel 1e-3
  final idepth      mean-rel <= 1e-3 (the north-star contract), checked at 2e-4
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, t, unpack_mask, batch_from_meta, rel_err, rel_err_per_pixel


def assert_contract(got, ref, what, limit=1e-3):
    """BASELINE.json's contract on the final depth map, per pixel: |a - ref| / max(|ref|, 1e-3 mean|ref|) < 1e-3 wherever
    the reference depth is positive (max and p99.9 printed)."""
    mx, p999 = rel_err_per_pixel(got, ref)
    print(f"{what}: per-pixel relative error max {mx:.3e} p99.9 {p999:.3e} (contract {limit:g})")
    assert mx < limit, (what, mx, p999)
from multi_view_stereonet_amd import MultiViewStereoNet, _native, synthetic
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights, default_init_weights
from oracle import mvsn_oracle as oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
_NETS = {}


def net_for(wname):
    if wname not in _NETS:
        net = MultiViewStereoNet()
        net.load_state_dict(load_weights(wname) if wname else default_init_weights(0), strict=True)
        _NETS[wname] = net.to(DEV).eval()
    return _NETS[wname]


def to_dev(inp):
    mv = lambda x: x.to(DEV)
    return ([mv(x) for x in inp["left_image_pyr"]], [mv(x) for x in inp["K_pyr"]],
            [mv(x) for x in inp["T_right_in_left"]], [[mv(x) for x in p] for p in inp["right_image_pyr"]])


def close(a, b, rtol=1e-4, atol=1e-5):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    ok = torch.allclose(a, b, rtol=rtol, atol=atol)
    assert ok, f"max abs diff {(a - b).abs().max().item():.3e} (ref max {b.abs().max().item():.3e})"


# ------------------------------------------------------------------------------------------
def test_mfma_fragment_mapping():
    lib = _native.load()
    assert lib.mvsn_selftest_mfma(_native.stream()) == 0, lib.mvsn_last_error()


@pytest.mark.parametrize("rows,cols,S,B,D,skew", [(64, 128, 1, 1, 16, 0.0), (256, 512, 2, 2, 64, 0.0), (480, 640, 1, 1, 96, 0.0),
                                                   (80, 96, 2, 3, 8, 0.0), (256, 512, 2, 2, 64, 0.7), (80, 96, 2, 3, 8, 0.3)])
def test_plane_sweep_setup(rows, cols, S, B, D, skew):
    """`skew` != 0: intrinsics with a shear term are not of the form the reference-order path (ref32, csrc/mvsn_setup.hip)
    covers -- the kernel's double-precision evaluation, rounded once, must serve them inside the same tolerances."""
    eng = net_for("gta_sfm_150epochs").engine()
    batch = synthetic.make_batch(rows, cols, S, batch=B, seed=3, pose_jitter=0.3)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    r4, c4 = inp["left_image_pyr"][4].shape[-2:]
    T = torch.cat(inp["T_right_in_left"], 0)
    for lvl in (0, 4):
        inp["K_pyr"][lvl] = inp["K_pyr"][lvl].clone()
        inp["K_pyr"][lvl][:, 0, 1] = skew
    K0, K4 = inp["K_pyr"][0].repeat(S, 1, 1), inp["K_pyr"][4].repeat(S, 1, 1)
    samples, H4, Hinc, H0, base = eng.plane_sweep_setup(T.to(DEV), K0.to(DEV), K4.to(DEV), r4, c4, D)
    # oracle: renormalise per source, then the reference pipeline
    Tn = T.clone()
    b = Tn[:, :3, 3].pow(2).sum(1).sqrt()
    Tn[:, :3, 3] /= b[:, None]
    close(base, b, rtol=1e-6, atol=0)
    s_ref = oracle.idepth_samples(Tn, K4, r4, c4, D)
    close(samples, s_ref, rtol=2e-5, atol=1e-7)
    close(H4, oracle.plane_sweep_homographies(Tn, K4, s_ref), rtol=1e-4, atol=2e-5)
    close(H0[:, 0], oracle.plane_sweep_homographies(Tn, K0, s_ref[:, :1])[:, 0], rtol=1e-4, atol=1e-4)
    H64 = H4.cpu().double()
    inc_ref = torch.linalg.inv(H64[:, :-1]) @ H64[:, 1:]
    close(Hinc[:, 1:], inc_ref, rtol=1e-5, atol=1e-6)
    close(Hinc[:, 0], torch.eye(3).expand(S * B, 3, 3), rtol=0, atol=0)
    # the same straight from the per-source pose tensors and the batch's intrinsics (no cat / repeat): identical bits
    got = eng.plane_sweep_setup_sources([t.to(DEV) for t in inp["T_right_in_left"]], inp["K_pyr"][0].to(DEV),
                                        inp["K_pyr"][4].to(DEV), r4, c4, D)
    for a, b2 in zip(got, (samples, H4, Hinc, H0, base)):
        assert torch.equal(a, b2)
    fx = eng.focal_pyramid([k.to(DEV) for k in inp["K_pyr"]]).cpu()
    assert torch.equal(fx, torch.stack([k[:, 0, 0] for k in inp["K_pyr"]]))


@pytest.mark.parametrize("rows,cols,S,D,jitter", [(256, 512, 2, 64, 0.0), (512, 1024, 4, 128, 0.3), (480, 640, 1, 96, 0.3),
                                                    (256, 512, 5, 64, 0.5), (64, 128, 1, 16, 0.0)])
def test_plane_sweep_homographies_follow_the_reference_fp32_chain(rows, cols, S, D, jitter):
    """Round 6: the homographies the kernels consume (H at levels 0 and 4) are formed by the reference's own fp32 chain --
    torch's CPU inverse of the pose (MKL's pivoted LU of the transpose + trans solve), the strti2 inverse of the
    intrinsics, ATen's naive 3x3 products -- instead of an fp64 evaluation rounded once: one ulp of the level-0
    translation entries was the whole forward's deviation from the reference on noise frames (profiles/r06_parity/).
    Against the oracle (the same torch ops as the reference): the entries agree BIT FOR BIT except cancellation residues
    (entries that are 0 in exact arithmetic, ~1e-9), over several seeds / pose jitters / source counts."""
    eng = net_for("gta_sfm_150epochs").engine()
    total = exact = 0
    worst = worst4 = 0.0
    for seed in range(6):
        batch = synthetic.make_batch(rows, cols, S, batch=2, seed=40 + seed, pose_jitter=jitter)
        inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
        r4, c4 = inp["left_image_pyr"][4].shape[-2:]
        T = torch.cat(inp["T_right_in_left"], 0)
        K0, K4 = inp["K_pyr"][0].repeat(S, 1, 1), inp["K_pyr"][4].repeat(S, 1, 1)
        samples, H4, _, H0, _ = eng.plane_sweep_setup(T.to(DEV), K0.to(DEV), K4.to(DEV), r4, c4, D)
        Tn = T.clone()
        Tn[:, :3, 3] /= Tn[:, :3, 3].pow(2).sum(1).sqrt()[:, None]
        s_dev = samples.cpu()                                   # (the library's own samples: the H chain is what is pinned)
        # level 0, plane 0 (idepth 0: rotation and intrinsics only) -- the matrix the full-resolution warp consumes
        got, ref = H0.cpu()[:, 0], oracle.plane_sweep_homographies(Tn, K0, s_dev[:, :1])[:, 0]
        same = got.view(torch.int32) == ref.view(torch.int32)
        total += same.numel()
        exact += int(same.sum())
        d = (got.double() - ref.double()).abs()
        worst = max(worst, float(d.max()))
        # (poses that are a rotation about one axis -- the fixtures', the bench's --: a differing entry is a residue; a
        # general rotation's inverse matches MKL's triangular solve to the last bit in most entries only: one ulp there)
        assert float(d.max()) <= (1e-7 if jitter == 0.0 else 2.5e-7) * float(ref.abs().max()), (seed, float(d.max()))
        # level 4, every plane: the translation of the inverted pose enters scaled by the idepth, and MKL's triangular solve
        # is matched to the last bit only in its large entries -- a few ulps of the largest entry at most
        got4, ref4 = H4.cpu(), oracle.plane_sweep_homographies(Tn, K4, s_dev)
        d4 = (got4.double() - ref4.double()).abs().flatten(2).max(2).values / ref4.abs().flatten(2).max(2).values.double()
        worst4 = max(worst4, float(d4.max()))
        assert float(d4.max()) <= (5e-7 if jitter == 0.0 else 5e-6), (seed, float(d4.max()))
    print(f"H0 entries equal bit for bit: {exact} of {total} ({exact / total:.4f}), largest difference {worst:.2e}; "
          f"H4: largest difference relative to the matrix's largest entry {worst4:.2e}")
    assert exact >= (0.93 if jitter == 0.0 else 0.6) * total


def test_plane_sweep_setup_reproduces_the_captured_reference_homographies():
    """Host-independent pin of the whole geometry chain (rows a3 / a4): the idepth samples and every matrix the REFERENCE
    handed its warper during seven forwards (tests/golden/g11_incremental_homographies.npz: level-0 plane-0 H, the level-4
    family, every incremental `inverse(H[d-1]) @ H[d]`; the golden configs, batches of one and two, jittered poses) against
    what `mvsn_plane_sweep_setup` emits for the same inputs -- bit for bit, every chain."""
    eng = net_for("gta_sfm_150epochs").engine()
    fix = load_golden("g11_incremental_homographies.npz")
    report = []
    for key in sorted(k for k in fix if k.endswith("_meta")):
        tag = key[:-5]
        rows, cols, D, S, B, seed, jit = (int(v) for v in fix[key])
        batch = synthetic.make_batch(rows, cols, S, batch=B, seed=seed, pose_jitter=jit / 100.0)
        inp = snu.multi_view_unpack_batch(batch, DEV, 5)
        r4, c4 = inp["left_image_pyr"][4].shape[-2:]
        # the poses as the reference's unpacker produced them (the fixture's): the host-side normalisation is a torch
        # reduction that rounds differently from host to host; the DEVICE-side unpacker (what the module uses) must give
        # exactly these
        T = torch.cat([t(fix[f"{tag}_T_{s}"]) for s in range(S)], 0)
        T_dev = torch.cat(inp["T_right_in_left"], 0).cpu()
        assert torch.equal(T_dev.view(torch.int32), T.view(torch.int32)), (tag, "device-side unpack", float((T_dev - T).abs().max()))
        K0, K4 = inp["K_pyr"][0].repeat(S, 1, 1), inp["K_pyr"][4].repeat(S, 1, 1)
        samples, H4, Hinc, H0, _ = eng.plane_sweep_setup(T.to(DEV), K0.to(DEV), K4.to(DEV), r4, c4, D)
        want = {k: np.concatenate([fix[f"{tag}_{k}_{s}"] for s in range(S)], 0) for k in ("samples", "H4", "Hinc", "H0")}
        got = {"samples": samples.cpu().numpy(), "H4": H4.cpu().numpy(), "Hinc": Hinc.cpu().numpy()[:, 1:],
               "H0": H0.cpu().numpy().reshape(S * B, 1, 3, 3)}
        assert np.array_equal(Hinc.cpu().numpy()[:, 0], np.broadcast_to(np.eye(3, dtype=np.float32), (S * B, 3, 3)))
        # the idepth samples: the reference's fp32 tensor program per level-4 pixel and torch's own summation order
        # (ref32::max_idepth_pixel and the cascade after it) -- every chain of the capture, bit for bit
        chain_ok = (got["samples"].view(np.int32) == want["samples"].view(np.int32)).all(1)
        report.append(f"{tag}: samples of {int(chain_ok.sum())} of {S * B} chains equal")
        assert chain_ok.all(), (tag, "samples", chain_ok, float(np.abs(got["samples"] - want["samples"]).max()))
        for k in ("H0", "H4", "Hinc"):
            rows_ok = chain_ok if k != "H0" else np.ones(S * B, bool)     # (plane 0: idepth 0 whatever the samples)
            same = (got[k].view(np.int32) == want[k].view(np.int32))[rows_ok]
            report.append(f"{k} {int(same.sum())} of {same.size}")
            assert same.all(), (tag, k, int((~same).sum()), same.size,
                                float(np.abs(got[k].astype(np.float64) - want[k])[rows_ok].max()))
    print("; ".join(report))


@pytest.mark.parametrize("name,wname,limit", [("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", 2e-5),
                                              ("g3_demon_640x480_d96_s1.npz", "demon_45epochs", 3e-5),
                                              ("gc5_gta_1024x512_d128_s4.npz", "gta_sfm_150epochs", 2e-5)])
def test_forward_headroom_against_the_reference(name, wname, limit):
    """A regression guard on the HEADROOM, not the contract (1e-3, asserted by the golden tests): with every homography
    the kernels consume formed in the reference's own fp32 operation order (H at levels 0 and 4, and the incremental
    `inverse(H[d-1]) @ H[d]` -- tests/golden/g11 pins them bit for bit) the per-pixel maximum against the reference's own
    depth map is 5.8e-6 / 8.9e-6 / 5.9e-6 on the headline / config 4 / config 5 fixtures -- the size of the reference's
    own MKLDNN-on/off spread (6e-6).  Rounds 1-5: 1.64e-4 / 2.38e-4 / 4.97e-4 (one fp64 evaluation,
    rounded once); H0 / H4 only: 5.6e-5 / 6.9e-5 / 1.06e-4."""
    fix = load_golden(name)
    out = _forward(net_for(wname), fix)
    mx, p999 = rel_err_per_pixel(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"{name}: per-pixel max {mx:.2e} p99.9 {p999:.2e} (guard {limit:g}, contract 1e-3)")
    assert mx < limit


@pytest.mark.parametrize("B,C,n,rows,cols", [(2, 3, 1, 64, 128), (1, 3, 16, 4, 8), (2, 32, 3, 16, 32),
                                              (1, 5, 4, 7, 9), (1, 3, 1, 256, 512)])
def test_homography_warp(B, C, n, rows, cols):
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * cols + n)
    img = torch.rand(B, C, rows, cols, generator=g) * 2 - 1
    H = torch.eye(3).repeat(B, n, 1, 1) + 0.04 * (torch.rand(B, n, 3, 3, generator=g) - 0.5)
    H[..., 0, 2] += (torch.rand(B, n, generator=g) - 0.5) * cols * 0.5
    H[..., 1, 2] += (torch.rand(B, n, generator=g) - 0.5) * rows * 0.5
    H[..., 2, :2] *= 0.02
    vol, mask = eng.homography_warp(img.to(DEV), H.to(DEV))
    vref, mref = oracle.homography_warp(img, H)
    # mask voxels may only differ where the normalised coordinate sits within a few fp32 ulps of the |n| > 1
    # predicate (the oracle forms H @ grid as a matmul, the kernel as the reference's scalar expression): every
    # mismatching voxel is located, its coordinate recomputed in float64, and its distance to the predicate asserted
    diff = (mask.cpu() != mref).nonzero()
    print(f"homography_warp {B}x{C}x{n}x{rows}x{cols}: {diff.shape[0]} of {mref.numel()} mask voxels differ")
    for b, d, yy, xx in diff.tolist():
        u = H[b, d].double() @ torch.tensor([float(xx), float(yy), 1.0], dtype=torch.float64)
        nx = ((u[0] / u[2] + 0.5) * 2.0) / cols - 1.0
        ny = ((u[1] / u[2] + 0.5) * 2.0) / rows - 1.0
        edge = min(abs(abs(float(nx)) - 1.0), abs(abs(float(ny)) - 1.0))
        assert edge < 16 * 2.0 ** -23, f"mask voxel {(b, d, yy, xx)} differs {edge:.3e} away from the predicate"
    assert diff.shape[0] <= max(1, mref.numel() // 20000)
    agree = (mask.cpu() == mref)[:, None].expand_as(vref)
    # white-noise frames: one ulp of a ~cols-sized coordinate times a gradient of up to 2/px
    close(vol.cpu()[agree], vref[agree], rtol=1e-4, atol=cols * 2.0 ** -23 * 8)


@pytest.mark.parametrize("B,C,n,rows,cols", [(40, 3, 1, 256, 512), (24, 5, 3, 128, 512), (40, 32, 96, 30, 40)])
def test_homography_warp_many_frames_form_is_bit_identical(B, C, n, rows, cols):
    """With enough pixels in flight mvsn_homography_warp runs four pixels per thread (16-byte streaming stores, the mask
    bytes as dwords); the coordinate algebra and the products per pixel are the one-pixel kernel's, so a large call must
    equal the same frames warped in calls small enough for the one-pixel form, bit for bit -- including borders (pairs
    shifted at the right edge), NaN / Inf texels and the mask."""
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(B * 131 + cols)
    img = torch.rand(B, C, rows, cols, generator=g) * 2 - 1
    img[0, 0, 3, 5], img[B - 1, C - 1, rows - 1, cols - 1] = float("nan"), float("inf")
    H = torch.eye(3).repeat(B, n, 1, 1) + 0.04 * (torch.rand(B, n, 3, 3, generator=g) - 0.5)
    H[..., 0, 2] += (torch.rand(B, n, generator=g) - 0.5) * cols * 0.5
    H[..., 1, 2] += (torch.rand(B, n, generator=g) - 0.5) * rows * 0.5
    H[..., 2, :2] *= 0.02
    img, H = img.to(DEV), H.to(DEV)
    assert B * n * rows * cols >= 4 * 16 * 64 * 4 * 256      # (the four-pixel form's threshold on 256 CUs)
    vol, mask = eng.homography_warp(img, H)
    for b in range(B):       # one frame and one plane at a time: far below the threshold
        for d in range(0, n, max(1, n // 3)):
            v1, m1 = eng.homography_warp(img[b:b + 1], H[b:b + 1, d:d + 1].contiguous())
            assert torch.equal(m1[0, 0], mask[b, d])
            assert torch.equal(v1[0, :, 0].view(torch.int32), vol[b, :, d].view(torch.int32)), (b, d)


def test_homography_warp_golden_units():
    fix = load_golden("g4_units.npz")
    eng = net_for("gta_sfm_150epochs").engine()
    vol, mask = eng.homography_warp(t(fix["psw_image"]).to(DEV), t(fix["psw_H"]).to(DEV))
    assert torch.equal(mask.cpu(), t(fix["psw_mask"])[:, 0])
    close(vol, fix["psw_volume"], rtol=1e-5, atol=2e-6)
    img, H = t(fix["hip_image"]), t(fix["hip_H"])
    vol, mask = eng.homography_warp(img.to(DEV), H[:, None].contiguous().to(DEV))
    ref_mask = t(fix["hip_mask"])[:, 0]
    assert torch.equal(mask.cpu()[:, 0], ref_mask)
    close(vol[:, :, 0], t(fix["hip_pred"]) * (~ref_mask).float()[:, None], rtol=1e-5, atol=2e-6)


CONV_CASES = [
    # cin, cout, k, stride, dil, rows, cols, n
    (32, 32, 3, 1, 1, 16, 32, 2), (32, 32, 3, 1, 2, 20, 24, 1), (32, 32, 3, 1, 4, 33, 47, 1),
    (32, 32, 3, 1, 8, 64, 96, 1), (3, 32, 5, 2, 1, 64, 128, 2), (32, 32, 5, 2, 1, 30, 45, 1),
    (32, 32, 5, 2, 1, 15, 23, 1), (36, 32, 3, 1, 1, 8, 12, 2), (4, 32, 3, 1, 1, 40, 70, 1),
    (35, 32, 3, 1, 1, 4, 8, 1), (32, 1, 3, 1, 1, 16, 32, 2), (32, 1, 3, 1, 1, 9, 5, 1),
    # the extractor head's dedicated kernel (3 -> 32, 5x5, stride 2, cols % 8 == 0): partial tiles, odd rows, one row
    (3, 32, 5, 2, 1, 37, 72, 2), (3, 32, 5, 2, 1, 64, 128, 3), (3, 32, 5, 2, 1, 1, 8, 1), (3, 32, 5, 2, 1, 50, 136, 1),
]


@pytest.mark.parametrize("cin,cout,k,stride,dil,rows,cols,n", CONV_CASES)
def test_conv2d(cin, cout, k, stride, dil, rows, cols, n):
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(cin * 1000 + rows)
    w = torch.randn(cout, cin, k, k, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(n, cin, rows, cols, generator=g)
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV), stride=stride, dilation=dil)
    out, stats = eng.conv(c, x.to(DEV), want_stats=(cout == 32))
    ref = F.conv2d(x, w, b, stride=stride, padding=dil * (k // 2), dilation=dil)
    close(out, ref, rtol=1e-4, atol=1e-4)
    if cout == 32:
        rg = ref.reshape(n, 4, -1).double()
        close(stats[:, :, 0], rg.mean(2), rtol=1e-4, atol=1e-5)
        close(stats[:, :, 1], 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt(), rtol=1e-4, atol=1e-5)
        # fused GroupNorm + LeakyReLU (+ residual)
        gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1

        class P:  # parameter holder
            weight, bias = gamma.to(DEV), beta.to(DEV)
        y = eng.gn_lrelu(out, stats, _Norm(P))
        yref = F.leaky_relu(F.group_norm(ref, 4, gamma, beta, 1e-5), 0.2)
        close(y, yref, rtol=1e-4, atol=1e-4)
        y2 = eng.gn_lrelu(out, stats, _Norm(P), residual=out)
        close(y2, yref + ref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("cin,rows,cols,n,dil", [(32, 8, 32, 2, 1), (32, 16, 32, 1, 1), (32, 37, 68, 2, 1), (36, 40, 72, 1, 1), (36, 128, 256, 3, 1), (35, 16, 32, 2, 1),
                                                  (4, 24, 40, 3, 1), (32, 256, 512, 3, 1), (32, 5, 4, 2, 1),
                                                  (32, 64, 128, 37, 1), (4, 256, 512, 2, 1),
                                                  (32, 16, 32, 2, 2), (32, 37, 68, 2, 2), (32, 128, 256, 5, 2),
                                                  (32, 16, 32, 2, 4), (32, 41, 76, 3, 4), (32, 256, 512, 2, 4),
                                                  (32, 16, 32, 2, 8), (32, 37, 68, 2, 8), (32, 256, 512, 2, 8),
                                                  (32, 9, 12, 1, 8)])
def test_conv_winograd_form(cin, rows, cols, n, dil):
    """Winograd F(2x2,3x3) form of the 3x3 layers (MVSN_CONV_FP32_WINO) against ATen and against the direct
    fp32 kernel: values, GroupNorm statistics, and the fused input transform (32-channel inputs)."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(cin * 10 + rows + dil)
    w = torch.randn(32, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g) * 0.1
    x = torch.randn(n, cin, rows, cols, generator=g)
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV), dilation=dil)
    if cin > 36 or (cin > 32 and dil > 2):
        return
    assert c.packed_wino is not None
    eng.winograd = True
    out, stats = eng.conv(c, x.to(DEV), want_stats=True)
    eng.winograd = False
    out_d, stats_d = eng.conv(c, x.to(DEV), want_stats=True)
    eng.winograd = True
    ref = F.conv2d(x, w, b, padding=dil, dilation=dil)
    close(out, ref, rtol=1e-4, atol=1e-4)
    assert rel_err(out.cpu(), out_d.cpu())[0] < 2e-6      # mean-rel vs the direct kernel: rounding only
    rg = ref.reshape(n, 4, -1).double()
    close(stats[:, :, 0], rg.mean(2), rtol=1e-4, atol=1e-5)
    close(stats[:, :, 1], 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt(), rtol=1e-4, atol=1e-5)
    if cin == 32:
        gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1

        class P:
            weight, bias = gamma.to(DEV), beta.to(DEV)
        xg = x.reshape(n, 4, -1).double()
        st_in = torch.stack([xg.mean(2), 1.0 / (xg.var(2, unbiased=False) + 1e-5).sqrt()], 2).float().contiguous()
        out2, _ = eng.conv(c, x.to(DEV), in_stats=st_in.to(DEV), in_norm=_Norm(P))
        ref2 = F.conv2d(F.leaky_relu(F.group_norm(x, 4, gamma, beta, 1e-5), 0.2), w, b, padding=dil, dilation=dil)
        close(out2, ref2, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("depth,rows,cols,n", [(8, 16, 32, 2), (5, 9, 36, 1), (64, 16, 32, 1), (3, 37, 68, 2), (1, 16, 32, 2),
                                               (2, 8, 12, 3), (16, 16, 32, 20),
                                               # 32 < cols <= 40: 10-row strips of the whole width (WIDE tiles): the 30x40 grid
                                               # of BASELINE config 4, a ragged last strip, a narrower plane, many items
                                               (6, 30, 40, 2), (3, 23, 36, 1), (96, 30, 40, 1), (4, 30, 40, 70),
                                               # round 6, rolling strips (six patch rows at a time through the sample's
                                               # planes): odd depth (a short last item), odd heights (two zero rows
                                               # between the parts), PR = 6 / 7 / 8 (no split / every split position),
                                               # one plane, and heights below six patch rows (stay on the 10-row strips)
                                               (9, 30, 40, 2), (5, 13, 40, 1), (7, 12, 36, 3), (11, 15, 40, 1), (1, 30, 40, 2),
                                               (4, 10, 40, 1), (13, 29, 36, 1)])
def test_conv_winograd_volume_form(depth, rows, cols, n):
    """3x3x3 32->32 layers as 2-D Winograd products summed over the depth tap (MVSN_CONV_FP32_WINO with kd = 3):
    values and GroupNorm statistics against ATen and the direct fp32 kernel, plus the fused input transform."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(depth * 100 + rows)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.06
    b = torch.randn(32, generator=g) * 0.1
    x = torch.randn(n, 32, depth, rows, cols, generator=g)
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV))
    assert c.packed_wino is not None
    eng.winograd_volume = True
    out, stats = eng.conv(c, x.to(DEV), want_stats=True)
    eng.winograd_volume = False
    out_d, stats_d = eng.conv(c, x.to(DEV), want_stats=True)
    eng.winograd_volume = True
    ref = F.conv3d(x, w, b, padding=1)
    close(out, ref, rtol=1e-4, atol=1e-4)
    assert rel_err(out.cpu(), out_d.cpu())[0] < 2e-6      # mean-rel vs the direct kernel: rounding only
    rg = ref.reshape(n, 4, -1).double()
    close(stats[:, :, 0], rg.mean(2), rtol=1e-4, atol=1e-5)
    close(stats[:, :, 1], 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt(), rtol=1e-4, atol=1e-5)
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1

    class P:
        weight, bias = gamma.to(DEV), beta.to(DEV)
    xg = x.reshape(n, 4, -1).double()
    st_in = torch.stack([xg.mean(2), 1.0 / (xg.var(2, unbiased=False) + 1e-5).sqrt()], 2).float().contiguous()
    out2, _ = eng.conv(c, x.to(DEV), in_stats=st_in.to(DEV), in_norm=_Norm(P))
    ref2 = F.conv3d(F.leaky_relu(F.group_norm(x, 4, gamma, beta, 1e-5), 0.2), w, b, padding=1)
    close(out2, ref2, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("split,rows,cols,n", [((3, 32, 1), 40, 72, 2), ((3, 1), 64, 128, 3), ((35, 1), 16, 32, 2),
                                               ((3, 32, 1), 128, 256, 5), ((36,), 24, 40, 1), ((4, 4, 4), 33, 52, 2)])
def test_conv_channel_blocks(split, rows, cols, n):
    """mvsn_conv_forward_blocks: the layer on [image, features, idepth] handed over as separate tensors is
    bit-identical to the same kernel on their torch.cat (same arithmetic, only the fetch addresses differ)."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    cin = sum(split)
    g = torch.Generator().manual_seed(cin + rows)
    w = torch.randn(32, cin, 3, 3, generator=g) * 0.1
    b = torch.randn(32, generator=g) * 0.1
    blocks = [torch.randn(n, c_, rows, cols, generator=g).to(DEV) for c_ in split]
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV))
    got = eng.conv_blocks(c, blocks, want_stats=True)
    assert got is not None
    out, stats = got
    out_c, stats_c = eng.conv(c, torch.cat(blocks, 1), want_stats=True)
    assert torch.equal(out, out_c) and torch.equal(stats, stats_c)
    ref = F.conv2d(torch.cat([t.cpu() for t in blocks], 1), w, b, padding=1)
    close(out, ref, rtol=1e-4, atol=1e-4)
    # conv() takes the list too, and falls back to the concatenation where the layer has no Winograd form
    out_l, _ = eng.conv(c, blocks, want_stats=True)
    assert torch.equal(out_l, out)
    eng.cat_free_heads = False
    out_f, _ = eng.conv(c, blocks, want_stats=True)
    eng.cat_free_heads = True
    assert torch.equal(out_f, out)


def test_conv_channel_blocks_rejects_bad_arguments():
    import ctypes
    from multi_view_stereonet_amd import _native
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    lib = eng.lib
    c = _Conv(lib, torch.randn(32, 36, 3, 3).to(DEV), None)
    a, b_ = torch.zeros(1, 35, 8, 16, device=DEV), torch.zeros(1, 1, 8, 16, device=DEV)
    out = torch.empty(1, 32, 8, 16, device=DEV)
    d = c.desc(1, 1, 8, 16, _native.CONV_FP32_WINO)
    ptrs = (ctypes.c_void_p * 2)(a.data_ptr(), b_.data_ptr())
    for chans, prec in (((35, 2), _native.CONV_FP32_WINO), ((35, 1), _native.CONV_FP32)):
        d.precision = prec
        rc = lib.mvsn_conv_forward_blocks(ctypes.byref(d), ptrs, (ctypes.c_int * 2)(*chans), 2,
                                          _native.ptr(c.packed_wino), None, _native.ptr(out), None, _native.stream())
        assert rc != 0 and lib.mvsn_last_error()
    d.precision = _native.CONV_FP32_WINO
    rc = lib.mvsn_conv_forward_blocks(ctypes.byref(d), ptrs, (ctypes.c_int * 2)(35, 1), 4,
                                      _native.ptr(c.packed_wino), None, _native.ptr(out), None, _native.stream())
    assert rc != 0


def test_conv_with_folded_groupnorm_input():
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 6, 10, 14, generator=g) * 2 + 0.5
    w1 = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    b1 = torch.randn(32, generator=g) * 0.1
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    w2 = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
    b2 = torch.randn(32, generator=g) * 0.1

    class P:
        weight, bias = gamma.to(DEV), beta.to(DEV)
    c1 = _Conv(eng.lib, w1.to(DEV), b1.to(DEV))
    c2 = _Conv(eng.lib, w2.to(DEV), b2.to(DEV))
    r1, st = eng.conv(c1, x.to(DEV), want_stats=True)
    r2, _ = eng.conv(c2, r1, in_stats=st, in_norm=_Norm(P))
    ref1 = F.conv3d(x, w1, b1, padding=1)
    close(r1, ref1, rtol=1e-4, atol=1e-4)
    ref2 = F.conv3d(F.leaky_relu(F.group_norm(ref1, 4, gamma, beta, 1e-5), 0.2), w2, b2, padding=1)
    close(r2, ref2, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("rows,cols,dil,cout", [(16, 32, 1, 32), (40, 50, 2, 32), (70, 33, 8, 32), (9, 20, 1, 1)])
def test_conv_with_folded_residual_block(rows, cols, dil, cout):
    """x' = x + LReLU(GN(r)) computed inside the next conv's tile load and written out once."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows + dil)
    r = torch.randn(2, 32, rows, cols, generator=g) * 1.5 + 0.3
    x = torch.randn(2, 32, rows, cols, generator=g)
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    w = torch.randn(cout, 32, 3, 3, generator=g) * 0.05
    b = torch.randn(cout, generator=g) * 0.1

    class P:
        weight, bias = gamma.to(DEV), beta.to(DEV)
    rg = r.reshape(2, 4, -1).double()
    stats = torch.stack([rg.mean(2), 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt()], 2).float().contiguous()
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV), dilation=dil)
    xs_ref = x + F.leaky_relu(F.group_norm(r, 4, gamma, beta, 1e-5), 0.2)
    ref = F.conv2d(xs_ref, w, b, padding=dil, dilation=dil)
    out, _, staged = eng.conv(c, r.to(DEV), in_stats=stats.to(DEV), in_norm=_Norm(P), in_residual=x.to(DEV),
                              write_staged=True)
    close(staged, xs_ref, rtol=1e-4, atol=1e-4)
    close(out, ref, rtol=1e-4, atol=2e-4)
    # without the residual, with the write-out (first block of a refiner)
    out2, _, staged2 = eng.conv(c, r.to(DEV), in_stats=stats.to(DEV), in_norm=_Norm(P), write_staged=True)
    x0_ref = F.leaky_relu(F.group_norm(r, 4, gamma, beta, 1e-5), 0.2)
    close(staged2, x0_ref, rtol=1e-4, atol=1e-4)
    close(out2, F.conv2d(x0_ref, w, b, padding=dil, dilation=dil), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("dims,depth,rows,cols,n", [(2, 1, 16, 32, 2), (2, 1, 37, 68, 1), (2, 1, 256, 512, 1), (2, 1, 5, 4, 3),
                                                    (3, 8, 4, 8, 2), (3, 12, 16, 32, 1), (3, 5, 30, 40, 1),
                                                    (3, 64, 16, 32, 2), (3, 96, 30, 40, 1), (3, 7, 32, 64, 1), (3, 1, 16, 32, 1),
                                                    (3, 33, 17, 36, 1)])
def test_conv_to1_tap_gemm(dims, depth, rows, cols, n):
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * 7 + depth)
    shape = (n, 32, depth, rows, cols) if dims == 3 else (n, 32, rows, cols)
    x = torch.randn(shape, generator=g)
    w = torch.randn((1, 32) + (3,) * dims, generator=g) * 0.05
    b = torch.randn(1, generator=g) * 0.1
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV))
    out = eng.conv_to1(c, x.to(DEV))
    ref = (F.conv3d if dims == 3 else F.conv2d)(x, w, b, padding=1)
    assert out is not None
    close(out, ref, rtol=1e-4, atol=1e-4)
    if dims == 2:
        prior = torch.rand(n, 1, rows, cols, generator=g) * 3.0
        fx = torch.rand(n, generator=g) * 50 + 10
        got = eng.conv_to1(c, x.to(DEV), prior.to(DEV), fx.to(DEV))
        s = fx.view(-1, 1, 1, 1)
        close(got, torch.relu(prior * s + ref) / s, rtol=1e-4, atol=1e-5)
    assert eng.conv_to1(c, torch.zeros(1, 32, 6, 45, device=DEV)) is None if dims == 2 else True   # ragged width -> MFMA kernel


@pytest.mark.parametrize("rows,cols,n,with_res,with_prior", [(16, 32, 2, True, True), (37, 68, 1, True, False),
                                                             (256, 512, 1, True, True), (5, 4, 3, False, True),
                                                             (30, 40, 2, True, True)])
def test_conv_to1_block_folds_last_residual_block(rows, cols, n, with_res, with_prior):
    """32 -> 1 layer reading x + LReLU(GN(r)) formed in registers (mvsn_conv_to1_block) against ATen."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * 3 + cols)
    r = torch.randn(n, 32, rows, cols, generator=g) * 1.5 + 0.3
    x = torch.randn(n, 32, rows, cols, generator=g) if with_res else None
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    w = torch.randn(1, 32, 3, 3, generator=g) * 0.05
    b = torch.randn(1, generator=g) * 0.1

    class P:
        weight, bias = gamma.to(DEV), beta.to(DEV)
    rg = r.reshape(n, 4, -1).double()
    stats = torch.stack([rg.mean(2), 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt()], 2).float().contiguous()
    y = F.leaky_relu(F.group_norm(r, 4, gamma, beta, 1e-5), 0.2)
    if with_res:
        y = y + x
    ref = F.conv2d(y, w, b, padding=1)
    prior = fx = None
    if with_prior:
        prior = torch.rand(n, 1, rows, cols, generator=g) * 3.0
        fx = torch.rand(n, generator=g) * 50 + 10
        sc = fx.view(-1, 1, 1, 1)
        ref = torch.relu(prior * sc + ref) / sc
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV))
    got = eng.conv_to1_block(c, r.to(DEV), stats.to(DEV), _Norm(P), x.to(DEV) if with_res else None,
                             prior.to(DEV) if with_prior else None, fx.to(DEV) if with_prior else None)
    close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,rows,cols,with_bias", [(2, 37, 72, True), (1, 1, 8, False), (10, 256, 512, False), (70, 50, 136, True)])
def test_conv_5x5_stride2_head_persistent_kernel(n, rows, cols, with_bias):
    """The extractor's 3 -> 32 head (conv5x5s2_head_kernel: persistent workgroups, two-slot tile ring by LDS-DMA,
    descriptor stores) against ATen -- partial tiles, one row, and more tiles than resident workgroups (1280 / 840:
    the ring's second slot and the walk over (image, tile) ids)."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * 11 + cols)
    x = torch.randn(n, 3, rows, cols, generator=g)
    w = torch.randn(32, 3, 5, 5, generator=g) * 0.1
    b = torch.randn(32, generator=g) * 0.1 if with_bias else None
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV) if with_bias else None, stride=2)
    got, _ = eng.conv(c, x.to(DEV))
    ref = F.conv2d(x, w, b, stride=2, padding=2)
    close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,rows,cols,with_bias", [(3, 64, 128, False), (2, 37, 72, True), (2, 100, 200, False),
                                                   (1, 256, 512, False), (5, 8, 16, True),
                                                   (150, 64, 128, False), (70, 37, 72, True)])   # > 256 tiles: several per workgroup
def test_conv_5x5_stride2_winograd_on_phases(n, rows, cols, with_bias):
    """The extractor's 5x5 stride-2 32 -> 32 layers as Winograd F(2x2,3x3) on the input's four stride-2 phases
    (conv_wino_s2_kernel) against ATen and against the direct kernel (ragged rows / columns, partial tiles, bias)."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * 5 + cols)
    x = torch.randn(n, 32, rows, cols, generator=g)
    w = torch.randn(32, 32, 5, 5, generator=g) * 0.05
    b = torch.randn(32, generator=g) * 0.1 if with_bias else None
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV) if with_bias else None, stride=2)
    assert c.packed_wino is not None
    ref = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, stride=2, padding=2).float()
    eng.timeline = []
    try:
        eng.winograd_stride2 = True
        got, _ = eng.conv(c, x.to(DEV))
        assert any("k5s2 32->32 wino" in t[0] for t in eng.timeline)
        eng.winograd_stride2 = False
        direct, _ = eng.conv(c, x.to(DEV))
    finally:
        eng.winograd_stride2 = True
        eng.timeline = None
    close(direct, ref, rtol=1e-4, atol=1e-4)
    close(got, ref, rtol=1e-4, atol=1e-4)
    assert (got.cpu() - ref).abs().max() <= 4 * (direct.cpu() - ref).abs().max() + 1e-5


def test_refiner_tower_end_trimming_is_equivalent():
    """The tower with the head activation and the last block never materialised (gn_lrelu_add2 +
    conv_to1_block) against the one-pass-per-block form, on every refiner of the pretrained weights."""
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(11)
    for lvl, (rows, cols) in zip((4, 2, 0), ((16, 32), (64, 128), (256, 512))):
        cin = eng.refiners[lvl]["conv0"].cin
        guide = torch.rand(2, cin - 1, rows, cols, generator=g).to(DEV)
        prior = (torch.rand(2, 1, rows, cols, generator=g) * 0.5).to(DEV)
        fx = torch.tensor([300.0 / 2 ** lvl, 260.0 / 2 ** lvl], device=DEV)
        eng.trim_tower_ends = True
        a = eng.idepth_refiner(lvl, guide, prior, fx)
        eng.trim_tower_ends = False
        b = eng.idepth_refiner(lvl, guide, prior, fx)
        eng.trim_tower_ends = True
        close(a, b.cpu(), rtol=2e-5, atol=2e-6)


def test_statistics_formed_by_the_consumer_are_bit_identical():
    """Small batches: the normalise/activate passes and the folded 32 -> 1 tail are handed the producing convolution's
    GroupNorm RECORDS and finalise them inside their own launch (mvsn_groupnorm_lrelu_apply_records,
    mvsn_conv_to1_block_records) -- the same bits as with the stand-alone mvsn_groupnorm_finalize launch, fewer
    launches; the regulariser's in-place passes likewise."""
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(21)
    keep = (eng.lazy_stats_max_samples, eng.lazy_stats_max_records)
    try:
        for lvl, (rows, cols) in zip((3, 2, 1, 0), ((32, 64), (64, 128), (128, 256), (256, 512))):
            cin = eng.refiners[lvl]["conv0"].cin
            guide = torch.rand(2, cin - 1, rows, cols, generator=g).to(DEV)
            prior = (torch.rand(2, 1, rows, cols, generator=g) * 0.5).to(DEV)
            fx = torch.tensor([300.0 / 2 ** lvl, 260.0 / 2 ** lvl], device=DEV)
            got = {}
            for mode, (ms, mr) in (("records", (8, 1 << 20)), ("finalize", (0, 0))):
                eng.lazy_stats_max_samples, eng.lazy_stats_max_records = ms, mr
                eng.timeline = []
                got[mode] = eng.idepth_refiner(lvl, guide, prior, fx)
                got[mode + "_launches"] = sum(1 for e in eng.timeline if e[0] == "mvsn_groupnorm_finalize")
            eng.timeline = None
            assert torch.equal(got["records"], got["finalize"])
            # (only the head's launch stays -- except where a layer leaves more records per sample than ONE workgroup
            # reduces (level 0: 256 tiles x 32 = 8192 > 2048): there the stand-alone launch uses the sliced order, so the
            # consumers are never handed records whatever lazy_stats_max_records says -- one order per record count)
            sliced = eng.lib.mvsn_groupnorm_finalize_split_workspace_bytes(2, (rows // 16) * (cols // 32) * 32) > 0
            assert sliced == (lvl == 0)
            assert got["records_launches"] == (7 if sliced else 1) and got["finalize_launches"] == 7, got
        cost = torch.randn(2, 32, 16, 16, 32, generator=g).to(DEV)
        out = {}
        for mode, (ms, mr) in (("records", (8, 1 << 20)), ("finalize", (0, 0))):
            eng.lazy_stats_max_samples, eng.lazy_stats_max_records = ms, mr
            out[mode] = eng.cost_volume_filter(cost.clone())
        assert torch.equal(out["records"], out["finalize"])
    finally:
        eng.lazy_stats_max_samples, eng.lazy_stats_max_records = keep
        eng.timeline = None


@pytest.mark.parametrize("n,records", [(1, 2049), (3, 8192), (1, 19200), (2, 32768), (5, 40001), (4, 2048)])
def test_groupnorm_finalize_in_slices(n, records):
    """mvsn_groupnorm_finalize_split (many records per sample: up to 16 workgroups per sample, their double sums added
    in slice order) against the one-workgroup finalize and against the definition: equal to rounding, deterministic,
    and independent of the batch a sample travels in (the slice count is a function of the record count alone)."""
    lib = net_for("gta_sfm_150epochs").engine().lib
    g = torch.Generator().manual_seed(records)
    cnt = torch.randint(0, 3, (n, records, 4, 1), generator=g).float() * 64.0
    mean = torch.randn(n, records, 4, 1, generator=g) * 0.5 + 0.3
    m2 = torch.rand(n, records, 4, 1, generator=g) * cnt * 0.7
    part = torch.cat([cnt, mean, m2], 3).contiguous().to(DEV)
    one = torch.empty(n, 4, 2, device=DEV)
    assert lib.mvsn_groupnorm_finalize(_native.ptr(part), n, records, _native.ptr(one), _native.stream()) == 0

    def split(p):
        k = p.shape[0]
        out = torch.empty(k, 4, 2, device=DEV)
        nb = lib.mvsn_groupnorm_finalize_split_workspace_bytes(k, records)
        assert (nb == 0) == (records <= 2048)
        ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=DEV)
        assert lib.mvsn_groupnorm_finalize_split(_native.ptr(p), k, records, _native.ptr(out), _native.ptr(ws), nb,
                                                 _native.stream()) == 0, lib.mvsn_last_error()
        return out
    a, b = split(part), split(part)
    assert torch.equal(a, b)
    for i in range(n):
        assert torch.equal(split(part[i:i + 1].contiguous())[0], a[i]), i
    if records <= 2048:
        assert torch.equal(a, one)
    c, mu, q = cnt.double().squeeze(3), mean.double().squeeze(3), m2.double().squeeze(3)
    N, S, Q = c.sum(1), (c * mu).sum(1), (q + c * mu * mu).sum(1)
    ref_mean = S / N
    ref_rstd = 1.0 / ((Q / N - ref_mean ** 2).clamp_min(0) + 1e-5).sqrt()
    for got in (a, one):
        close(got[:, :, 0], ref_mean, rtol=2e-6, atol=1e-7)
        close(got[:, :, 1], ref_rstd, rtol=2e-6, atol=1e-7)


def test_refiner_heads_without_concatenation_are_identical():
    """Refiner input handed to the head conv as [image, features, idepth] blocks vs. one torch.cat: bit-identical,
    for every refiner of the pretrained weights (the level-0 head has no feature block: 3 + 1 channels)."""
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(12)
    for lvl, (rows, cols) in zip((3, 1, 0), ((32, 64), (128, 256), (256, 512))):
        cin = eng.refiners[lvl]["conv0"].cin
        guide = [torch.rand(2, 3, rows, cols, generator=g).to(DEV)]
        if cin > 4:
            guide.append(torch.rand(2, cin - 4, rows, cols, generator=g).to(DEV))
        prior = (torch.rand(2, 1, rows, cols, generator=g) * 0.5).to(DEV)
        fx = torch.tensor([300.0 / 2 ** lvl, 260.0 / 2 ** lvl], device=DEV)
        a = eng.idepth_refiner(lvl, guide, prior, fx)
        eng.cat_free_heads = False
        b = eng.idepth_refiner(lvl, guide, prior, fx)
        eng.cat_free_heads = True
        assert torch.equal(a, b)


@pytest.mark.parametrize("dims,depth,rows,cols,dil,n", [(2, 1, 16, 32, 1, 2), (2, 1, 37, 70, 1, 1), (2, 1, 24, 40, 2, 1),
                                                        (2, 1, 64, 96, 2, 1), (3, 8, 16, 32, 1, 2), (3, 5, 9, 35, 1, 1),
                                                        (3, 64, 16, 32, 1, 1)])
def test_conv_bf16x3_split(dims, depth, rows, cols, dil, n):
    """3 x bf16 split tier: fp32-equivalent to ~2^-16 per product (error measured against fp64)."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv, _Norm
    eng = net_for("gta_sfm_150epochs").engine()
    g = torch.Generator().manual_seed(rows * 3 + depth)
    shape = (n, 32, depth, rows, cols) if dims == 3 else (n, 32, rows, cols)
    x = torch.randn(shape, generator=g) * 1.5 + 0.2
    w = torch.randn((32, 32) + (3,) * dims, generator=g) * 0.06
    b = torch.randn(32, generator=g) * 0.1
    conv = F.conv3d if dims == 3 else F.conv2d
    ref = conv(x.double(), w.double(), b.double(), padding=dil, dilation=dil)
    c = _Conv(eng.lib, w.to(DEV), b.to(DEV), dilation=dil)
    assert c.packed_bx is not None
    eng.conv_precision = "bf16x3"
    try:
        out, stats = eng.conv(c, x.to(DEV), want_stats=True)
        mean_rel, max_rel = rel_err(out.cpu(), ref)
        assert mean_rel < 3e-5 and max_rel < 1e-4, (mean_rel, max_rel)
        rg = ref.reshape(n, 4, -1)
        close(stats[:, :, 0], rg.mean(2), rtol=1e-4, atol=2e-5)
        close(stats[:, :, 1], 1.0 / (rg.var(2, unbiased=False) + 1e-5).sqrt(), rtol=1e-4, atol=1e-5)
        # folded GroupNorm + LeakyReLU on the input
        gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1

        class P:
            weight, bias = gamma.to(DEV), beta.to(DEV)
        out2, _ = eng.conv(c, out, in_stats=stats, in_norm=_Norm(P))
        xin = F.leaky_relu(F.group_norm(ref, 4, gamma.double(), beta.double(), 1e-5), 0.2)
        ref2 = conv(xin, w.double(), b.double(), padding=dil, dilation=dil)
        mean_rel, max_rel = rel_err(out2.cpu(), ref2)
        assert mean_rel < 1e-4 and max_rel < 3e-4, (mean_rel, max_rel)
    finally:
        eng.conv_precision = "fp32"
    fp32_out, _ = eng.conv(c, x.to(DEV))
    m32, _ = rel_err(fp32_out.cpu(), ref)
    assert m32 < 2e-6          # the default tier is exact fp32


@pytest.mark.parametrize("name,wname,smooth", [("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", False),
                                               ("g2s_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", True),
                                               ("g3_demon_640x480_d96_s1.npz", "demon_45epochs", False),
                                               ("g1_gta_128x64_d16_s1.npz", "gta_sfm_150epochs", False)])
def test_forward_bf16x3_tier_within_contract(name, wname, smooth):
    """The split tier against the reference's depth maps: inside the 1e-3 contract (reported, not hidden)."""
    fix = load_golden(name)
    net = net_for(wname)
    eng = net.engine()
    eng.conv_precision = "bf16x3"
    try:
        out = _forward(net, fix, smooth=smooth)
    finally:
        eng.conv_precision = "fp32"
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"bf16x3 {name}: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
    assert mean_rel < 5e-4 and max_rel < 1e-3, (mean_rel, max_rel)
    assert_contract(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"], f"bf16x3 {name}")


@pytest.mark.parametrize("name,wname,smooth", [("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", False),
                                               ("g2s_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", True)])
def test_forward_bf16_operand_tier_is_a_speed_tier(name, wname, smooth):
    """Plain bf16 operands on the 32 -> 32 3x3 layers (BASELINE config 5's tier): finite, close, and -- as the
    survey measured for bf16 conv3d operands -- NOT inside the 1e-3 contract the other two forms meet."""
    fix = load_golden(name)
    net = net_for(wname)
    eng = net.engine()
    errs = {}
    for tier in ("bf16x3", "bf16"):
        eng.conv_precision = tier
        try:
            out = _forward(net, fix, smooth=smooth)
        finally:
            eng.conv_precision = "fp32"
        errs[tier] = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"{name}: bf16x3 {errs['bf16x3'][0]:.3e} / {errs['bf16x3'][1]:.3e}, bf16 {errs['bf16'][0]:.3e} / {errs['bf16'][1]:.3e}")
    assert errs["bf16"][0] < 2e-2 and errs["bf16"][1] < 2e-1
    assert errs["bf16"][0] > 5 * errs["bf16x3"][0]     # the split buys at least that much


def test_flag_branch_elementwise_kernels():
    """The three elementwise entry points behind the flag branches (no ATen compute left on the path):
    channel L2 norm (torch.norm(.., dim=1), :598), prior*fx and relu(prior*fx + delta)/fx (:482, :607-611)."""
    lib = _native.load()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 32, 5, 7, 9, generator=g)
    out = torch.empty(3, 5, 7, 9, device=DEV)
    xd = x.to(DEV)
    _native.check(lib.mvsn_channel_l2_norm(_native.ptr(xd), 3, 32, 5 * 7 * 9, _native.ptr(out), _native.stream()), "l2")
    close(out, torch.norm(x, dim=1), rtol=1e-6, atol=1e-7)
    prior = torch.rand(4, 1, 11, 13, generator=g) * 2
    delta = torch.randn(4, 1, 11, 13, generator=g) * 40
    fx = torch.tensor([410.0, 25.6, 51.2, 102.4])
    o1, o2 = torch.empty(4, 1, 11, 13, device=DEV), torch.empty(4, 1, 11, 13, device=DEV)
    pd, fd, dd = prior.to(DEV), fx.to(DEV), delta.to(DEV)      # kept alive across the raw-pointer calls
    _native.check(lib.mvsn_idepth_scale(_native.ptr(pd), _native.ptr(fd), 4, 143, _native.ptr(o1), _native.stream()),
                  "scale")
    _native.check(lib.mvsn_refiner_epilogue(_native.ptr(pd), _native.ptr(fd), _native.ptr(dd), 4, 143, _native.ptr(o2),
                                            _native.stream()), "epi")
    sc = fx.view(-1, 1, 1, 1)
    assert torch.equal(o1.cpu(), prior * sc)
    assert torch.equal(o2.cpu(), torch.relu(prior * sc + delta) / sc)
    assert (o2 == 0).any() and (o2 > 0).any()


def test_groupnorm_apply_beyond_one_grid():
    """More than 2047 samples: the wrappers walk the batch in grid-sized chunks."""
    eng = net_for("gta_sfm_150epochs").engine()
    from multi_view_stereonet_amd.multi_view_stereonet import _Norm
    g = torch.Generator().manual_seed(5)
    n = 2100
    r = torch.randn(n, 32, 4, 8, generator=g)
    res = torch.randn(n, 32, 4, 8, generator=g)
    stats = torch.stack([torch.randn(n, 4, generator=g) * 0.1, torch.rand(n, 4, generator=g) + 0.5], -1)
    P = type("P", (), {})()
    P.weight, P.bias = torch.rand(32, generator=g).to(DEV) + 0.5, torch.randn(32, generator=g).to(DEV)
    y = eng.gn_lrelu(r.to(DEV), stats.to(DEV), _Norm(P), residual=res.to(DEV))
    mean = stats[..., 0].repeat_interleave(8, 1)[:, :, None, None]
    rstd = stats[..., 1].repeat_interleave(8, 1)[:, :, None, None]
    want = res + F.leaky_relu((r - mean) * rstd * P.weight.cpu()[None, :, None, None] + P.bias.cpu()[None, :, None, None], 0.2)
    close(y, want, rtol=1e-5, atol=1e-5)
    y2 = eng.gn_lrelu_add2(r.to(DEV), stats.to(DEV), _Norm(P), res.to(DEV), stats.to(DEV), _Norm(P))
    want2 = want - res + F.leaky_relu((res - mean) * rstd * P.weight.cpu()[None, :, None, None] +
                                      P.bias.cpu()[None, :, None, None], 0.2)
    close(y2, want2, rtol=1e-5, atol=1e-5)


def test_conv_to1_volume_with_groupnorm_on_load():
    """mvsn_conv_to1_volume_norm == (LReLU(GN(raw)) materialised, then the 32 -> 1 volume layer)."""
    eng = net_for("gta_sfm_150epochs").engine()
    lib = eng.lib
    g = torch.Generator().manual_seed(8)
    n, D, rows, cols = 3, 9, 12, 20
    raw = torch.randn(n, 32, D, rows, cols, generator=g).to(DEV)
    stats = torch.stack([torch.randn(n, 4, generator=g) * 0.2, torch.rand(n, 4, generator=g) + 0.5], -1).to(DEV)
    last, nrm = eng.vf_convs[4], eng.vf_norms[3]
    want = eng.conv_to1(last, eng.gn_lrelu(raw, stats, nrm))
    got = torch.empty(n, D, rows, cols, device=DEV)
    _native.check(lib.mvsn_conv_to1_volume_norm(_native.ptr(raw), _native.ptr(stats), _native.ptr(nrm.gamma),
                                                _native.ptr(nrm.beta), _native.ptr(last.weight), _native.ptr(last.bias),
                                                n, D, rows, cols, _native.ptr(got), _native.stream()), "volume_norm")
    close(got, want[:, 0].cpu(), rtol=1e-5, atol=1e-5)


def test_cost_volume_filter_and_soft_argmin_golden_unit():
    fix = load_golden("g4_units.npz")
    eng = net_for("gta_sfm_150epochs").engine()
    out = eng.cost_volume_filter(t(fix["cvf_in"]).to(DEV))
    close(out, fix["cvf_out"], rtol=1e-4, atol=2e-5)
    raw = eng.soft_argmin(out, t(fix["sm_idepth"]).to(DEV))
    close(raw, fix["sm_out"], rtol=1e-4, atol=1e-5)
    # constant cost -> mean of the samples
    samples = torch.linspace(0, 1.7, 9)[None].to(DEV)
    flat = eng.soft_argmin(torch.full((1, 9, 2, 2), 3.0, device=DEV), samples)
    close(flat, samples.mean().cpu().expand(1, 1, 2, 2), rtol=1e-6, atol=1e-6)


def test_idepth_refiner_golden_units():
    fix = load_golden("g4_units.npz")
    eng = net_for("gta_sfm_150epochs").engine()
    ones = torch.ones(2, device=DEV)
    for lvl in (1, 0):
        out = eng.idepth_refiner(lvl, t(fix[f"idr{lvl}_guide"]).to(DEV), t(fix[f"idr{lvl}_prior"]).to(DEV), ones)
        close(out, fix[f"idr{lvl}_out"], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("wname", ["gta_sfm_150epochs", "demon_45epochs", None])
def test_plane_resident_towers_vs_oracle(wname):
    """mvsn_tower_16x32 (one persistent workgroup per sample, activations resident in LDS): the extractor's residual
    stack + conv_final and the level-4 refiner (head over [image | features | prior*fx], dilations 1,2,4,8,1,1, the
    32 -> 1 tail with the gain epilogue) against the oracle's layer-by-layer form AND against this library's own
    launch-per-layer path, for chains that share reference images (N = S * B)."""
    w = load_weights(wname) if wname else default_init_weights(0)
    net = net_for(wname)
    eng = net.engine()
    g = torch.Generator().manual_seed(17)
    # extractor tail
    x = torch.randn(5, 32, 16, 32, generator=g)
    ref = x
    for i in range(6):
        ref = oracle._res_block(w, f"left_feature_extractor.res{i}", ref)
    ref = oracle._conv(w, "left_feature_extractor.conv_final", ref)
    got = eng.tower_extractor_tail(x.to(DEV))
    assert got is not None
    mean_rel, max_rel = rel_err(got.cpu(), ref)
    print(f"tower[extractor, {wname}]: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
    assert mean_rel < 2e-5 and max_rel < 2e-4
    want, _ = eng.residual_tower_unfused(x.to(DEV), None, eng.fe_res, eng.fe_final)
    mean_rel, max_rel = rel_err(got.cpu(), want.cpu())
    assert mean_rel < 2e-5 and max_rel < 2e-4
    # level-4 refiner, S = 3 chains per reference image
    B, S = 2, 3
    img = torch.rand(B, 3, 16, 32, generator=g) * 2 - 1
    feats = torch.randn(B, 32, 16, 32, generator=g)
    prior = 0.02 + 0.2 * torch.rand(S * B, 1, 16, 32, generator=g)
    fx = 20.0 + 10.0 * torch.rand(B, generator=g)
    fxn = fx.repeat(S).view(-1, 1, 1, 1)
    guide = torch.cat([img, feats], 1).repeat(S, 1, 1, 1)
    ref = oracle.idepth_refiner(w, "refiner4", guide, prior * fxn) / fxn
    got = eng.tower_refiner4(img.to(DEV), feats.to(DEV), prior.to(DEV), fx.to(DEV))
    assert got is not None
    mean_rel, max_rel = rel_err(got.cpu(), ref)
    print(f"tower[refiner 4, {wname}]: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
    assert mean_rel < 2e-5 and max_rel < 2e-4
    want = eng.idepth_refiner(4, guide.to(DEV), prior.to(DEV), fx.repeat(S).to(DEV))
    mean_rel, max_rel = rel_err(got.cpu(), want.cpu())
    assert mean_rel < 2e-5 and max_rel < 2e-4


def test_upsamplers_golden_units():
    fix = load_golden("g4_units.npz")
    eng = net_for("gta_sfm_150epochs").engine()
    assert torch.equal(eng.upsample_mask(t(fix["mu_in"]).to(DEV), (15, 30)).cpu(), t(fix["mu_out"]))
    close(eng.upsample(t(fix["up_in"]).to(DEV), (15, 30)), fix["up_out"], rtol=1e-6, atol=1e-6)
    g = torch.Generator().manual_seed(8)
    m = torch.rand(1, 64, 16, 32, generator=g) > 0.45
    assert torch.equal(eng.upsample_mask(m.to(DEV), (32, 64)).cpu(), oracle.upsample_mask(m, (32, 64)))
    for (h, w, H, W) in ((30, 40, 60, 80), (8, 12, 15, 23), (4, 6, 8, 12), (5, 7, 9, 13), (64, 128, 128, 256), (1, 8, 2, 16)):
        m = torch.rand(2, 12, h, w, generator=g) > 0.5
        assert torch.equal(eng.upsample_mask(m.to(DEV), (H, W)).cpu(), oracle.upsample_mask(m, (H, W)))


@pytest.mark.parametrize("rows,cols,D,S,B,wname", [(64, 128, 16, 1, 1, "gta_sfm_150epochs"),
                                                   (256, 512, 64, 2, 1, "gta_sfm_150epochs"),
                                                   (80, 96, 8, 2, 2, "gta_sfm_150epochs"),
                                                   (192, 320, 10, 1, 2, "gta_sfm_150epochs"),    # 12x20: 60 patches, 4 tiles, last partial
                                                   (128, 256, 9, 2, 1, "gta_sfm_150epochs"),     # 8x16: two full tiles
                                                   (480, 640, 12, 1, 1, "demon_45epochs"),
                                                   (512, 1024, 6, 1, 1, "gta_sfm_150epochs")])
@pytest.mark.parametrize("form", ["direct", "winograd", "stepwise", "banded"])
def test_incremental_chain_vs_oracle(rows, cols, D, S, B, wname, form):
    """The fused chain (features, cost, mask) against the oracle's step-by-step recurrence, fed
    with the SAME plane-0 features and homographies so only the chain itself is compared.  Both forms of the
    three 3x3 convolutions: direct implicit GEMM (any grid) and Winograd F(2x2,3x3) (where the grid has a plan)."""
    w = load_weights(wname)
    net = net_for(wname)
    eng = net.engine()
    r4, c4 = (rows + 15) // 16, (cols + 15) // 16
    if form == "winograd" and eng.lib.mvsn_incremental_cost_volume_form(r4, c4) != _native.CHAIN_WINOGRAD:
        pytest.skip(f"no Winograd plan for a {r4}x{c4} coarse grid")
    if form == "stepwise" and c4 % 4 != 0:
        pytest.skip(f"the stepwise form needs cols % 4 == 0 ({r4}x{c4})")
    if form == "banded" and (r4, c4) not in ((16, 32), (30, 40), (32, 64)):
        pytest.skip(f"the banded form has plans for 16x32, 30x40 and 32x64 ({r4}x{c4})")
    net.options.chain_form = form
    try:
        _chain_vs_oracle(w, eng, rows, cols, D, S, B, form)
    finally:
        net.options.chain_form = "auto"


def _chain_vs_oracle(w, eng, rows, cols, D, S, B, form):
    batch = synthetic.make_batch(rows, cols, S, batch=B, seed=11, pose_jitter=0.2)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    r4, c4 = inp["left_image_pyr"][4].shape[-2:]
    g = torch.Generator().manual_seed(2)
    FL = torch.randn(B, 32, r4, c4, generator=g)
    for s in range(S):
        T = inp["T_right_in_left"][s].clone()
        T[:, :3, 3] /= T[:, :3, 3].pow(2).sum(1).sqrt()[:, None]
        samples = oracle.idepth_samples(T, inp["K_pyr"][4], r4, c4, D)
        H = oracle.plane_sweep_homographies(T, inp["K_pyr"][4], samples)
        Hinc = torch.eye(3).repeat(B, D, 1, 1)
        Hinc[:, 1:] = torch.linalg.inv(H[:, :-1]) @ H[:, 1:]
        F0 = torch.randn(B, 32, r4, c4, generator=g)
        src4 = inp["right_image_pyr"][s][4]
        # oracle recurrence
        image_vol, mask_ref = oracle.homography_warp(src4, H)
        planes = [F0]
        for d in range(1, D):
            moved, _ = oracle.homography_warp(planes[-1], Hinc[:, d:d + 1])
            planes.append(oracle.feature_refiner(w, "right_feature_extractor.refiner", image_vol[:, :, d], moved[:, :, 0]))
        fvol_ref = torch.stack(planes, 2) * (~mask_ref).float()[:, None]
        cost_ref = (~mask_ref).float()[:, None] * (FL[:, :, None] - fvol_ref).abs()
        cost, mask, fvol = eng.incremental_cost_volume(src4.to(DEV), H.to(DEV), Hinc.to(DEV), F0.to(DEV), FL.to(DEV),
                                                       want_features=True)
        assert eng.chain_status() == 0, "a banded-chain hand-off timed out"
        assert int((mask.cpu() != mask_ref).sum()) == 0
        for name, a, b in (("features", fvol, fvol_ref), ("cost", cost, cost_ref)):
            mean_rel, max_rel = rel_err(a.cpu(), b)
            print(f"chain[{form}] {r4}x{c4} D={D} source {s} {name}: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
            assert mean_rel < 1e-5 and max_rel < 1e-4, (name, s, mean_rel, max_rel)


def _oracle_chain(w, src4, H, Hinc, F0, FL):
    """The oracle's recurrence (multi_view_stereonet.py:270-300 + the cost build :553,587-592) on explicit
    homography families."""
    D = H.shape[1]
    image_vol, mask_ref = oracle.homography_warp(src4, H)
    planes = [F0]
    for d in range(1, D):
        moved, _ = oracle.homography_warp(planes[-1], Hinc[:, d:d + 1])
        planes.append(oracle.feature_refiner(w, "right_feature_extractor.refiner", image_vol[:, :, d], moved[:, :, 0]))
    fvol_ref = torch.stack(planes, 2) * (~mask_ref).float()[:, None]
    cost_ref = (~mask_ref).float()[:, None] * (FL[:, :, None] - fvol_ref).abs()
    return fvol_ref, cost_ref, mask_ref


def _motion_family(N, D, kind, seed):
    """Incremental homographies with a prescribed inter-plane motion (H = their running product)."""
    g = torch.Generator().manual_seed(seed)
    Hinc = torch.eye(3).repeat(N, D, 1, 1)
    for n in range(N):
        for d in range(1, D):
            if kind == "small":            # within the banded form's gather window (about a pixel per plane, any direction)
                Hinc[n, d, 0, 2] = float(torch.rand(1, generator=g)) * 2 - 1
                Hinc[n, d, 1, 2] = float(torch.rand(1, generator=g)) * 2 - 1
            elif kind == "vertical":       # 3-6 rows per plane: beyond the window, every tap from the hand-off granules
                Hinc[n, d, 1, 2] = (3 + 3 * float(torch.rand(1, generator=g))) * (1 if d % 2 else -1)
                Hinc[n, d, 0, 2] = float(torch.rand(1, generator=g)) - 0.5
            elif kind == "mixed":          # alternates between the two paths, with shear / scale and projective terms
                big = d % 3 == 0
                Hinc[n, d, 1, 2] = (5.0 if big else 0.7) * (1 if d % 2 else -1)
                Hinc[n, d, 0, 1] = 0.05 * (float(torch.rand(1, generator=g)) - 0.5)
                Hinc[n, d, 1, 1] = 1.0 + 0.1 * (float(torch.rand(1, generator=g)) - 0.5)
                Hinc[n, d, 2, 0] = 1e-3 * (float(torch.rand(1, generator=g)) - 0.5)
    H = torch.eye(3).repeat(N, D, 1, 1)
    for d in range(1, D):
        H[:, d] = H[:, d - 1] @ Hinc[:, d]
    return H, Hinc


@pytest.mark.parametrize("kind,N,D,grid", [("small", 1, 12, (16, 32)), ("vertical", 2, 10, (16, 32)),
                                            ("mixed", 3, 13, (16, 32)), ("small", 8, 6, (16, 32)),
                                            ("small", 1, 9, (30, 40)), ("mixed", 2, 8, (30, 40)), ("vertical", 3, 7, (30, 40)),
                                            ("small", 2, 8, (32, 64)), ("mixed", 1, 9, (32, 64)), ("vertical", 4, 6, (32, 64))])
def test_banded_chain_gather_paths_vs_oracle(kind, N, D, grid):
    """The banded form's two gather paths (rows fetched into the LDS window / every tap from the granules) against the
    oracle's recurrence AND against another form of this library on the same inputs (the plane-resident Winograd kernel
    on 16x32, the stepwise form on the grids that do not fit one CU), on all three banded geometries: 4 bands of 4 rows
    (16x32), 15 of 2 (30x40: a partly filled patch tile, two halo items per thread), 16 of 2 with a +-1-row window
    (32x64)."""
    w = load_weights("gta_sfm_150epochs")
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    g = torch.Generator().manual_seed(5)
    H, Hinc = _motion_family(N, D, kind, seed=3)
    r4, c4 = grid
    src4 = torch.rand(N, 3, r4, c4, generator=g) * 2 - 1
    F0 = torch.randn(N, 32, r4, c4, generator=g)
    FL = torch.randn(N, 32, r4, c4, generator=g)
    fvol_ref, cost_ref, mask_ref = _oracle_chain(w, src4, H, Hinc, F0, FL)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    got = {}
    other = "winograd" if grid == (16, 32) else "stepwise"
    try:
        for form in ("banded", other):
            net.options.chain_form = form
            cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
            assert eng.chain_status() == 0
            got[form] = (cost.cpu(), mask.cpu(), fvol.cpu())
    finally:
        net.options.chain_form = "auto"
    for form, (cost, mask, fvol) in got.items():
        assert int((mask != mask_ref).sum()) == 0, form
        for name, a, b in (("features", fvol, fvol_ref), ("cost", cost, cost_ref)):
            mean_rel, max_rel = rel_err(a, b)
            print(f"chain[{form}] {kind} N={N} D={D} {name}: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
            assert mean_rel < 1e-5 and max_rel < 1e-4, (form, name, mean_rel, max_rel)
    assert torch.equal(got["banded"][1], got[other][1])


SLAB = 16     # mvsn_debug_set_band_flags: pin the slab plan of the banded form (few fat bands per chain, 512 threads)


@pytest.mark.parametrize("kind,N,D,grid,extra", [
    ("small", 2, 9, (16, 32), 0), ("mixed", 3, 10, (16, 32), 0), ("vertical", 2, 7, (16, 32), 0),
    ("small", 3, 9, (30, 40), 0), ("mixed", 2, 8, (30, 40), 0), ("vertical", 2, 6, (30, 40), 0),
    ("small", 2, 8, (32, 64), 0), ("mixed", 2, 9, (32, 64), 0), ("vertical", 3, 6, (32, 64), 0),
    ("small", 2, 8, (32, 64), 1), ("small", 3, 7, (30, 40), 1), ("small", 1, 1, (32, 64), 0), ("small", 2, 2, (30, 40), 0)])
def test_slab_chain_gather_paths_vs_oracle(kind, N, D, grid, extra):
    """The SLAB plan of the banded form (mvsn_chain_slab.hip: 3 / 4 fat bands per chain resident in LDS, what AUTO runs
    on the 30x40 / 32x64 coarse grids once many chains are in flight; on 16x32 two bands, for this test) against the
    oracle's recurrence (multi_view_stereonet.py:279-290, :424-440) and against another form of this library: the gather
    from the planes ("small": every tap within one row of the band), every tap from the granules ("vertical": 3-6 rows per
    plane; `extra` = 1 forces that path on small motion too), steps that alternate ("mixed"); D = 1 / 2 (no step / one)."""
    w = load_weights("gta_sfm_150epochs")
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    g = torch.Generator().manual_seed(5)
    H, Hinc = _motion_family(N, D, kind, seed=3)
    r4, c4 = grid
    src4 = torch.rand(N, 3, r4, c4, generator=g) * 2 - 1
    F0 = torch.randn(N, 32, r4, c4, generator=g)
    FL = torch.randn(N, 32, r4, c4, generator=g)
    fvol_ref, cost_ref, mask_ref = _oracle_chain(w, src4, H, Hinc, F0, FL)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    got = {}
    other = "winograd" if grid == (16, 32) else "stepwise"
    try:
        for form in ("banded", other):
            net.options.chain_form = form
            eng.lib.mvsn_debug_set_band_flags((SLAB | extra) if form == "banded" else 0)
            cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
            torch.cuda.synchronize()
            assert eng.chain_status() == 0, form
            got[form] = (cost.cpu(), mask.cpu(), fvol.cpu())
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"
    for form, (cost, mask, fvol) in got.items():
        assert int((mask != mask_ref).sum()) == 0, form
        for name, a, b in (("features", fvol, fvol_ref), ("cost", cost, cost_ref)):
            mean_rel, max_rel = rel_err(a, b)
            print(f"chain[{form}{'/slab' if form == 'banded' else ''}] {kind} N={N} D={D} {name}: mean-rel {mean_rel:.3e} "
                  f"max-rel {max_rel:.3e}")
            assert mean_rel < 1e-5 and max_rel < 1e-4, (form, name, mean_rel, max_rel)
    assert torch.equal(got["banded"][1], got[other][1])


@pytest.mark.parametrize("grid,N", [((30, 40), 24), ((32, 64), 20), ((30, 40), 12), ((16, 32), 80)])
def test_plain_entry_auto_stays_off_the_slab_plan(grid, N):
    """the PLAIN C entry point has no repair launch behind a banded call, so its MVSN_CHAIN_AUTO must not pick
    the multi-pass slab plan (a time-out there leaves NaN): beyond one thin-band pass it runs the stepwise form, word for
    word what an explicit MVSN_CHAIN_STEPWISE call returns; within one pass it is the thin bands; and the AUTO workspace
    figure covers whichever entry point resolves it."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    lib = eng.lib
    r4, c4 = grid
    D = 5
    g = torch.Generator().manual_seed(41)
    H, Hinc = _motion_family(N, D, "small", seed=4)
    src, H, Hinc, f0, fl = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                                                 torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    thin_cap = 256 // lib.mvsn_incremental_cost_volume_banded_groups(1, r4, c4)
    if grid == (16, 32):
        want = _native.CHAIN_WINOGRAD                      # (plane-resident plan: never the slab question)
    else:
        want = _native.CHAIN_BANDED if N <= thin_cap else _native.CHAIN_STEPWISE
        assert lib.mvsn_incremental_cost_volume_form_for(N, r4, c4) == _native.CHAIN_BANDED     # the guarded entries' AUTO
    need = {f: lib.mvsn_incremental_cost_volume_workspace_bytes_for(N, D, r4, c4, f) for f in (0, want, _native.CHAIN_BANDED)}
    assert need[0] >= need[want] and (grid == (16, 32) or need[0] >= need[_native.CHAIN_BANDED])

    def run(form):
        cost = torch.full((N, 32, D, r4, c4), float("nan"), device=DEV)
        mask = torch.zeros((N, D, r4, c4), dtype=torch.bool, device=DEV)
        nbytes = lib.mvsn_incremental_cost_volume_workspace_bytes_for(N, D, r4, c4, form)
        ws = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=DEV)
        rc = lib.mvsn_incremental_cost_volume(_native.ptr(src), _native.ptr(H), _native.ptr(Hinc), _native.ptr(f0),
                                              _native.ptr(fl), _native.ptr(eng.refiner_packed), N, N, D, r4, c4,
                                              _native.ptr(cost), _native.ptr(mask), None, _native.ptr(ws), nbytes, form,
                                              _native.stream())
        _native.check(rc, "mvsn_incremental_cost_volume")
        torch.cuda.synchronize()
        return cost, mask

    auto_cost, auto_mask = run(_native.CHAIN_AUTO)
    same_cost, same_mask = run(want)
    assert bool(torch.isfinite(auto_cost).all())
    assert torch.equal(auto_cost, same_cost) and torch.equal(auto_mask, same_mask)
    if want == _native.CHAIN_STEPWISE:                       # the slab plan (explicit BANDED) agrees to rounding only
        slab_cost, slab_mask = run(_native.CHAIN_BANDED)
        assert torch.equal(slab_mask, auto_mask)
        mean_rel, max_rel = rel_err(slab_cost.cpu(), auto_cost.cpu())
        assert mean_rel < 1e-5 and max_rel < 1e-3, (mean_rel, max_rel)


@pytest.mark.parametrize("grid,N,D", [((30, 40), 90, 4), ((32, 64), 70, 4), ((30, 40), 40, 5)])
def test_slab_chain_selected_for_many_chains_and_in_passes(grid, N, D):
    """Beyond two passes of the thin-band plan (34 / 32 chains on 256 CUs) the banded form runs the slab plan, and more
    chains than fit the chip at one workgroup per band (85 / 64) run as consecutive passes of equal size: every chain
    of a multi-pass call equals the same chain run in a small call of its own, bit for bit (the chains are independent)."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    g = torch.Generator().manual_seed(23)
    H, Hinc = _motion_family(N, D, "small", seed=9)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    assert eng.lib.mvsn_incremental_cost_volume_form_for(N, r4, c4) == _native.CHAIN_BANDED      # AUTO's choice
    net.options.chain_form = "auto"
    cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
    torch.cuda.synchronize()
    assert eng.last_chain_form == _native.CHAIN_BANDED and eng.chain_status() == 0
    # (a remainder of at most one thin-band pass behind full slab passes runs as that thin pass: 90 = 85 + 5 on 30x40)
    cap = 256 // eng.lib.mvsn_incremental_cost_volume_banded_groups(1000, r4, c4)
    thin_cap = 256 // eng.lib.mvsn_incremental_cost_volume_banded_groups(1, r4, c4)
    n_slab = N - N % cap if N > cap and 0 < N % cap <= thin_cap else N
    try:
        net.options.chain_form = "banded"
        for a in range(0, N, 7):
            for lo, hi, flag in ((a, min(a + 3, n_slab), SLAB), (max(a, n_slab), min(a + 3, N), 32)):
                if lo >= hi:
                    continue
                eng.lib.mvsn_debug_set_band_flags(flag)
                sl = slice(lo, hi)
                c1, m1, f1 = eng.incremental_cost_volume(*[x[sl].contiguous() for x in dev], want_features=True)
                torch.cuda.synchronize()
                assert eng.chain_status() == 0
                assert torch.equal(c1, cost[sl]) and torch.equal(m1, mask[sl]) and torch.equal(f1, fvol[sl]), (lo, hi)
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"


@pytest.mark.parametrize("grid,N,D", [((30, 40), 5, 24), ((32, 64), 4, 20), ((30, 40), 85, 6), ((32, 64), 64, 6)])
def test_slab_chain_hand_offs_under_uneven_load(grid, N, D):
    """The slab plan's four hand-offs per step (boundary rows of F, of the moved features, of the two raw convolution
    outputs with the GroupNorm sums) under UNEVEN load from a second stream, consumers L1-warm: 12 repeats, every word of
    every output compared with the first run's."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    g = torch.Generator().manual_seed(29)
    H, Hinc = _motion_family(N, D, "mixed" if N < 10 else "small", seed=4)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    try:
        net.options.chain_form = "banded"
        eng.lib.mvsn_debug_set_band_flags(SLAB)
        cost0, mask0, fvol0 = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        side = torch.cuda.Stream()
        big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
        for it in range(12):
            if N < 10:      # (a full-chip launch has no CU to spare: the side stream would take bands' CUs away)
                with torch.cuda.stream(side):
                    for k in range(1 + it % 4):
                        n = (8 + 8 * ((it + k) % 7)) << 20
                        big[:n].add_(1.0)
            cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
            torch.cuda.synchronize()
            assert eng.chain_status() == 0
            assert torch.equal(cost, cost0) and torch.equal(mask, mask0) and torch.equal(fvol, fvol0), it
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"


@pytest.mark.parametrize("grid,N,D", [((30, 40), 2, 8), ((32, 64), 3, 6)])
def test_slab_chain_missing_band_is_repaired_in_stream(grid, N, D):
    """A band that never runs (test hook; what a shared device can do to a launch that needs co-residency): the others'
    waits are bounded, the status word is set, and the gated direct-form launch behind the banded one recomputes the
    call in-stream -- finite outputs equal to the direct form's."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    g = torch.Generator().manual_seed(31)
    H, Hinc = _motion_family(N, D, "small", seed=6)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    net.reset_device_status()
    net.check_device_status()
    net.reset_device_status()
    try:
        net.options.chain_form = "direct"
        alone, mask_alone, _ = eng.incremental_cost_volume(*dev)
        net.options.chain_form = "banded"
        eng.lib.mvsn_debug_set_band_flags(SLAB | 2 | (10 << 8))
        fixed, mask_fixed, _ = eng.incremental_cost_volume(*dev)
        torch.cuda.synchronize()
        assert eng.chain_status() != 0 and bool(torch.isfinite(fixed).all())
        assert torch.equal(fixed, alone) and torch.equal(mask_fixed, mask_alone)
        with pytest.warns(RuntimeWarning, match="hand-off"):
            assert net.check_device_status() == 1
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"
        net.reset_device_status()


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("grid", [(16, 32), (30, 40), (32, 64)])
def test_chain_with_one_to_three_planes_all_forms(grid, D):
    """Edge of the recurrence (multi_view_stereonet.py:279-290 runs D - 1 steps): D = 1 is plane 0 alone (no step, no
    hand-off, the cost slice straight from the extractor's features), D = 2 one step, D = 3 the first step whose
    gather reads a plane this launch produced.  Every form the grid has a plan for, against the oracle."""
    w = load_weights("gta_sfm_150epochs")
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    N = 3
    g = torch.Generator().manual_seed(41)
    H, Hinc = _motion_family(N, D, "mixed", seed=12)
    src4 = torch.rand(N, 3, r4, c4, generator=g) * 2 - 1
    F0, FL = torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g)
    fvol_ref, cost_ref, mask_ref = _oracle_chain(w, src4, H, Hinc, F0, FL)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    forms = ["direct", "banded"] + (["winograd"] if grid == (16, 32) else ["stepwise"])
    try:
        for form in forms:
            net.options.chain_form = form
            cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
            torch.cuda.synchronize()
            assert eng.chain_status() == 0, form
            assert cost.shape == (N, 32, D, r4, c4) and mask.shape == (N, D, r4, c4)
            assert int((mask.cpu() != mask_ref).sum()) == 0, form
            for name, a, b in (("features", fvol.cpu(), fvol_ref), ("cost", cost.cpu(), cost_ref)):
                mean_rel, max_rel = rel_err(a, b)
                assert mean_rel < 1e-5 and max_rel < 1e-4, (form, D, name, mean_rel, max_rel)
    finally:
        net.options.chain_form = "auto"


@pytest.mark.parametrize("kind,N,D", [("small", 1, 64), ("mixed", 5, 20), ("vertical", 2, 9), ("small", 32, 6), ("mixed", 17, 7)])
def test_banded_chain_half_split_is_bit_identical_to_four_bands(kind, N, D):
    """16x32 has two banded plans: up to CUs / 8 chains run on 8 bands of 2 rows whose waves split every layer by
    transform-row half (the halves are added through LDS in the one-wave order), more on 4 bands of 4 rows.  Same
    arithmetic per output, same GroupNorm records in the same order: cost, mask and feature volumes must agree bit for
    bit (debug flag 4 pins the 4-band plan), on both gather paths."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    g = torch.Generator().manual_seed(31)
    H, Hinc = _motion_family(N, D, kind, seed=8)
    B = max(1, N // 2)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, 16, 32, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, 16, 32, generator=g), torch.randn(B, 32, 16, 32, generator=g))]
    net.options.chain_form = "banded"
    try:
        got = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        eng.lib.mvsn_debug_set_band_flags(4)
        try:
            ref = eng.incremental_cost_volume(*dev, want_features=True)
            torch.cuda.synchronize()
            assert eng.chain_status() == 0
        finally:
            eng.lib.mvsn_debug_set_band_flags(0)
        for name, a, b in zip(("cost", "mask", "features"), got, ref):
            assert torch.equal(a, b), (name, int((a != b).sum()))
        assert bool(torch.isfinite(got[0]).all())
    finally:
        net.options.chain_form = "auto"


@pytest.mark.parametrize("form", ["winograd", "banded"])
@pytest.mark.parametrize("case,D", [("long", 128), ("long", 256), ("steps", 48), ("offset", 24)])
def test_chain_groupnorm_single_pass_stays_accurate(case, D, form):
    """The Winograd chain forms take GroupNorm's variance in ONE shifted pass, var = E[(x-c)^2] - E[x-c]^2 with c = the
    previous plane's mean (mvsn_chain_wino.hip wino_groupnorm_lrelu, mvsn_chain_band.hip exchange): exact enough while
    |mean - c| stays within a few deviations, cancelling if it did not.  Driven here where the shortcut is weakest:
    D = 128 / 256 planes on 16x32 (error accumulation over a long recurrence), a source whose planes jump between a
    half that is +40 and one that is -40 from plane to plane (the largest plane-to-plane change of the statistics an
    input can cause), and plane-0 features with a common offset of 300 (the first step runs with c = 0).  Measured:
    max |err| <= 8e-6 of the features' spread in every case -- the zero padding of the 3x3 convolution bounds
    |mean| / sigma of its output by the plane's border fraction (~0.2-1.0 on 16x32 whatever offset the input carries),
    so the one-pass form never leaves its accurate range on this network; the assertions pin that."""
    w = load_weights("gta_sfm_150epochs")
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    N = 2
    g = torch.Generator().manual_seed(D)
    src4 = torch.rand(N, 3, 16, 32, generator=g) * 2 - 1
    F0 = torch.randn(N, 32, 16, 32, generator=g)
    FL = torch.randn(N, 32, 16, 32, generator=g)
    if case == "long":
        H, Hinc = _motion_family(N, D, "small", seed=D)
        # keep the sweep inside the image: small zero-mean steps
        Hinc[:, 1:, 0, 2] *= 0.15
        Hinc[:, 1:, 1, 2] *= 0.1
        H = torch.eye(3).repeat(N, D, 1, 1)
        for d in range(1, D):
            H[:, d] = H[:, d - 1] @ Hinc[:, d]
    else:
        # image planes alternate between the two halves of a source that is +40 on the left and -40 on the right
        src4[:, :, :, :16] += 40.0
        src4[:, :, :, 16:] -= 40.0
        H = torch.eye(3).repeat(N, D, 1, 1)
        for d in range(D):
            H[:, d, 0, 0] = 0.45                                   # the plane shows half of the source ...
            H[:, d, 0, 2] = 0.5 if d % 2 == 0 else 16.5           # ... the bright half on even planes, the dark on odd
        Hinc = torch.eye(3).repeat(N, D, 1, 1)
        Hinc[:, 1:, 0, 2] = 0.3
        if case == "offset":
            F0 = F0 * 0.05 + 300.0
    fvol_ref, cost_ref, mask_ref = _oracle_chain(w, src4, H, Hinc, F0, FL)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    net.options.chain_form = form
    try:
        cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
        assert eng.chain_status() == 0
    finally:
        net.options.chain_form = "auto"
    assert int((mask.cpu() != mask_ref).sum()) == 0
    assert bool(torch.isfinite(fvol).all())
    mean_rel, max_rel = rel_err(fvol.cpu(), fvol_ref)
    # per plane as well: a drift would grow along the recurrence
    per_plane = [(fvol.cpu()[:, :, d] - fvol_ref[:, :, d]).abs().mean() / fvol_ref[:, :, d].abs().mean().clamp_min(1e-6)
                 for d in (1, D // 2, D - 1)]
    # against the features' spread, not their size: with the common offset the values are ~300 and a wrong variance
    # would only move them by a fraction of their deviation
    valid = (~mask_ref)[:, None].expand_as(fvol_ref)
    spread = float(fvol_ref[valid].std())
    vs_spread = float((fvol.cpu() - fvol_ref).abs().max()) / spread
    print(f"chain[{form}] GN stress {case} D={D}: features mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}; "
          f"planes 1 / D/2 / D-1: {[f'{float(v):.2e}' for v in per_plane]}; max |err| / spread {vs_spread:.2e} "
          f"(spread {spread:.3g})")
    assert mean_rel < 2e-5 and max_rel < 2e-4, (case, D, form, mean_rel, max_rel)
    assert vs_spread < 1e-3, (case, D, form, vs_spread)


@pytest.mark.parametrize("grid,N,D", [((30, 40), 20, 5), ((32, 64), 19, 5), ((16, 32), 70, 4)])
def test_banded_chain_in_passes(grid, N, D):
    """More chains than fit the chip at one workgroup per band: the banded call runs consecutive passes over one
    workspace (17 / 16 / 64 chains per pass on an MI355X).  Every chain must come out exactly as when it runs in a call
    of its own (chains are independent: `torch.equal` -- on 16x32 that call of 6 chains runs the 8-band half-split plan
    while the 70-chain call runs 4 bands: the two plans must agree bit for bit), agree with another form of the library,
    and leave status 0;
    AUTO picks the banded form for up to two passes where no plane-resident plan exists."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    B = 5                               # left features are shared: chain n reads batch item n % B
    g = torch.Generator().manual_seed(23)
    H, Hinc = _motion_family(N, D, "mixed", seed=6)
    src4 = torch.rand(N, 3, r4, c4, generator=g) * 2 - 1
    F0, FL = torch.randn(N, 32, r4, c4, generator=g), torch.randn(B, 32, r4, c4, generator=g)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    cap = 256 // {16: 4, 30: 15, 32: 16}[r4]
    assert N > cap
    lib = eng.lib
    want = _native.CHAIN_BANDED if r4 != 16 else _native.CHAIN_WINOGRAD
    assert lib.mvsn_incremental_cost_volume_form_for(N, r4, c4) == want
    assert lib.mvsn_incremental_cost_volume_form_for(cap, r4, c4) == _native.CHAIN_BANDED
    # beyond the thin-band pass: 16x32 has the plane-resident kernel; 30x40 / 32x64 stay banded (its slab plan takes over)
    assert lib.mvsn_incremental_cost_volume_form_for(2 * cap + 1, r4, c4) == (
        _native.CHAIN_BANDED if r4 != 16 else _native.CHAIN_WINOGRAD)
    net.options.chain_form = "banded"
    try:
        eng.lib.mvsn_debug_set_band_flags(32)      # the thin-band plan pinned (30x40 / 32x64 would pick slabs beyond one pass)
        cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        # the second pass's chains on their own: batch index n % B must follow the chain's GLOBAL index
        tail = list(range(cap, N))
        idx = torch.tensor([n % B for n in tail])
        c2, m2, f2 = eng.incremental_cost_volume(dev[0][cap:], dev[1][cap:], dev[2][cap:], dev[3][cap:],
                                                 FL[idx].to(DEV), want_features=True)
        assert eng.chain_status() == 0
        assert torch.equal(cost[cap:], c2) and torch.equal(mask[cap:], m2) and torch.equal(fvol[cap:], f2)
        net.options.chain_form = "winograd" if r4 == 16 else "stepwise"
        c3, m3, f3 = eng.incremental_cost_volume(*dev, want_features=True)
        assert torch.equal(mask, m3)
        for name, a, b in (("features", fvol, f3), ("cost", cost, c3)):
            mean_rel, max_rel = rel_err(a.cpu(), b.cpu())
            assert mean_rel < 1e-5 and max_rel < 2e-4, (name, mean_rel, max_rel)
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"


@pytest.mark.parametrize("grid,N,D,form,flag", [
    ((16, 32), 70, 5, "winograd", 0), ((16, 32), 3, 6, "banded", 0), ((16, 32), 40, 4, "banded", 0),
    ((30, 40), 3, 5, "banded", 0), ((32, 64), 2, 5, "banded", 0), ((30, 40), 20, 4, "banded", 0), ((32, 64), 66, 3, "banded", 0),
    ((30, 40), 4, 4, "direct", 0), ((12, 20), 5, 6, "winograd", 0), ((9, 14), 3, 5, "direct", 0), ((32, 64), 3, 3, "direct", 0)])
def test_chain_bf16_cost_volume_is_the_fp32_volume_rounded(grid, N, D, form, flag):
    """mvsn_incremental_cost_volume_bf16 (the bf16 feature tier's Kernel A: cost volume stored as bf16) in every form that
    has the variant -- plane-resident Winograd, thin bands, slabs (20 / 66 chains: beyond one thin pass; 66 = one slab
    pass + a thin tail pass), direct: the bf16 volume is the fp32 call's volume rounded to nearest-even, word for word;
    masks and the fp32 feature volume are unchanged; status 0.  The stepwise form falls back to fp32 + conversion."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    B = min(N, 3)
    g = torch.Generator().manual_seed(41 + N)
    H, Hinc = _motion_family(N, D, "mixed", seed=3)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(B, 32, r4, c4, generator=g))]
    net.options.chain_form = form
    try:
        eng.lib.mvsn_debug_set_band_flags(flag)
        c32, m32, f32 = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        c16, m16, f16 = eng.incremental_cost_volume(*dev, want_features=True, cost_bf16=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        assert c16.dtype == torch.bfloat16 and c16.shape == c32.shape
        assert torch.equal(c16, c32.to(torch.bfloat16)), float((c16.float() - c32).abs().max())
        assert torch.equal(m16, m32) and torch.equal(f16, f32)
        if r4 % 2 == 0 and c4 % 4 == 0 and r4 != 16:
            net.options.chain_form = "stepwise"      # no bf16 variant: the fp32 call + a conversion
            c3, m3, _ = eng.incremental_cost_volume(*dev, cost_bf16=True)
            assert c3.dtype == torch.bfloat16 and torch.equal(m3, m32)
            mean_rel, max_rel = rel_err(c3.float().cpu(), c32.cpu())
            assert mean_rel < 3e-3, mean_rel
    finally:
        eng.lib.mvsn_debug_set_band_flags(0)
        net.options.chain_form = "auto"


def test_forward_bf16_feature_tier_cost_volume_storage_changes_no_bit():
    """Under `conv_precision = "bf16s"` the chain stores the cost volume as bf16 and the first regulariser layer reads it
    as such; with a capture dict the cost volume stays fp32 (the captured tensors are the fp32 ones) and that layer rounds
    it to bf16 itself: the same bits enter the multiplies, so the two forwards must agree bit for bit."""
    net = net_for("gta_sfm_150epochs")
    fix = load_golden("g2_gta_512x256_d64_s2.npz")
    net.options.conv_precision = "bf16s"
    try:
        eng = net.engine()
        out_a = _forward(net, fix)
        assert eng.last_cost_dtype == torch.bfloat16
        out_a = {k: [None if t is None else t.clone() for t in v] for k, v in out_a.items()}
        out_b = _forward(net, fix, capture={})
        assert eng.last_cost_dtype == torch.float32
        out_c = _forward(net, fix)                     # (the recorded plan replayed: the bf16 volume is one of its buffers)
        for a, b, c in zip(out_a["left_idepthmap_pyr"], out_b["left_idepthmap_pyr"], out_c["left_idepthmap_pyr"]):
            assert torch.equal(a, b) and torch.equal(a, c)
    finally:
        net.options.conv_precision = "fp32"


@pytest.mark.parametrize("grid,extra,D", [((30, 40), 1, 3), ((30, 40), 17, 2), ((32, 64), 2, 3)])
def test_banded_chain_slab_passes_with_thin_tail(grid, extra, D):
    """A call of k x (chains per slab pass) + a few chains runs full slab passes and ONE thin-band pass for the remainder
    (256 chains on 30x40: 85 + 85 + 85 on 3 slabs each, 1 on 15 thin bands) over one workspace with one status block.
    The slab chains must equal a call of their own (slab plan, one pass), the remainder a call of its own (thin plan),
    word for word; status 0; the call as a whole agrees with another form."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    lib = eng.lib
    r4, c4 = grid
    cap = 256 // lib.mvsn_incremental_cost_volume_banded_groups(1000, r4, c4)
    N, B = cap + extra, 5
    assert lib.mvsn_incremental_cost_volume_status_offset(N, r4, c4) == lib.mvsn_incremental_cost_volume_status_offset(cap, r4, c4)
    g = torch.Generator().manual_seed(31)
    H, Hinc = _motion_family(N, D, "mixed", seed=8)
    src4 = torch.rand(N, 3, r4, c4, generator=g) * 2 - 1
    F0, FL = torch.randn(N, 32, r4, c4, generator=g), torch.randn(B, 32, r4, c4, generator=g)
    dev = [x.to(DEV) for x in (src4, H, Hinc, F0, FL)]
    net.options.chain_form = "banded"
    try:
        cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        c1, m1, f1 = eng.incremental_cost_volume(dev[0][:cap], dev[1][:cap], dev[2][:cap], dev[3][:cap], dev[4],
                                                 want_features=True)
        assert eng.chain_status() == 0
        assert torch.equal(cost[:cap], c1) and torch.equal(mask[:cap], m1) and torch.equal(fvol[:cap], f1)
        idx = torch.tensor([n % B for n in range(cap, N)])
        c2, m2, f2 = eng.incremental_cost_volume(dev[0][cap:], dev[1][cap:], dev[2][cap:], dev[3][cap:],
                                                 FL[idx].to(DEV), want_features=True)
        assert eng.chain_status() == 0
        assert torch.equal(cost[cap:], c2) and torch.equal(mask[cap:], m2) and torch.equal(fvol[cap:], f2)
        net.options.chain_form = "stepwise"
        c3, m3, f3 = eng.incremental_cost_volume(*dev, want_features=True)
        assert torch.equal(mask, m3)
        for name, a, b in (("features", fvol, f3), ("cost", cost, c3)):
            mean_rel, max_rel = rel_err(a.cpu(), b.cpu())
            assert mean_rel < 1e-5 and max_rel < 2e-4, (name, mean_rel, max_rel)
    finally:
        net.options.chain_form = "auto"


@pytest.mark.parametrize("grid,N,D", [((16, 32), 5, 64), ((30, 40), 3, 24), ((32, 64), 4, 20),
                                      ((16, 32), 32, 12),    # 8 bands x 32 chains: every CU of the chip takes part
                                      ((16, 32), 64, 8)])    # 4 bands x 64 chains: the same
def test_banded_chain_hand_offs_under_uneven_load(grid, N, D):
    """The inter-workgroup hand-offs (tagged granules) must not depend on timing or placement: the same launch
    repeated while a second stream keeps part of the chip busy with streaming copies of varying size must return
    bit-identical volumes every time (every word compared), and no wait may time out."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    g = torch.Generator().manual_seed(9)
    H, Hinc = _motion_family(N, D, "small", seed=4)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    net.options.chain_form = "banded"
    try:
        cost0, mask0, fvol0 = eng.incremental_cost_volume(*dev, want_features=True)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0
        side = torch.cuda.Stream()
        big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
        for it in range(12):
            with torch.cuda.stream(side):
                for k in range(1 + it % 4):
                    n = (8 + 8 * ((it + k) % 7)) << 20
                    big[:n].add_(1.0)
            cost, mask, fvol = eng.incremental_cost_volume(*dev, want_features=True)
            torch.cuda.synchronize()
            assert eng.chain_status() == 0
            assert torch.equal(cost, cost0) and torch.equal(mask, mask0) and torch.equal(fvol, fvol0), it
    finally:
        net.options.chain_form = "auto"


@pytest.mark.parametrize("grid,N,D", [((16, 32), 2, 12), ((32, 64), 1, 8)])
def test_banded_chain_missing_band_falls_back_and_matches_oracle(grid, N, D):
    """The banded chain needs its workgroups co-resident.  If one never runs (test hook: the last band of every chain
    returns at once, what a device shared with other work can do), the others' waits are BOUNDED and the status word
    names the hand-off.  Default (`banded_repair`): the gated single-launch form behind the banded launch recomputes the
    chain IN-STREAM -- the call's outputs are finite and equal what that form computes alone; the sticky words count the
    repair, check_device_status reports it (warning once) and latches AUTO off the banded form.  Without the repair
    launch the cost slice carries NaN and check_device_status raises.  The next clean launch is bit-identical again."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    r4, c4 = grid
    g = torch.Generator().manual_seed(19)
    H, Hinc = _motion_family(N, D, "small", seed=2)
    dev = [x.to(DEV) for x in (torch.rand(N, 3, r4, c4, generator=g) * 2 - 1, H, Hinc,
                               torch.randn(N, 32, r4, c4, generator=g), torch.randn(N, 32, r4, c4, generator=g))]
    net.reset_device_status()
    net.check_device_status()          # (drain what earlier tests may have left in the sticky words)
    net.reset_device_status()
    try:
        net.options.chain_form = "winograd" if grid == (16, 32) else "direct"
        alone, mask_alone, _ = eng.incremental_cost_volume(*dev)      # the form the repair launch runs
        net.options.chain_form = "banded"
        good, mask_good, _ = eng.incremental_cost_volume(*dev)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0 and bool(torch.isfinite(good).all())
        assert net.check_device_status() == 0
        eng.lib.mvsn_debug_set_band_flags(2 | (10 << 8))       # last band absent, 1024 polls per hand-off
        try:
            fixed, mask_fixed, _ = eng.incremental_cost_volume(*dev)
            torch.cuda.synchronize()
            assert eng.chain_status() != 0                      # the banded launch did time out ...
            assert bool(torch.isfinite(fixed).all())            # ... and the gated launch behind it repaired the call
            assert torch.equal(fixed, alone) and torch.equal(mask_fixed, mask_alone)
            mean_rel, max_rel = rel_err(fixed.cpu(), good.cpu())
            assert mean_rel < 1e-5 and max_rel < 1e-4, (mean_rel, max_rel)
            with pytest.warns(RuntimeWarning, match="hand-off"):
                assert net.check_device_status() == 1
            assert net.__dict__["_shared_state"].banded_latched
            net.options.chain_form = "auto"
            eng.incremental_cost_volume(*dev)
            assert eng.last_chain_form != _native.CHAIN_BANDED      # latched: AUTO stays off the banded form
            net.reset_device_status()
            # without the repair launch: NaN in the cost slice, and the wrappers' check raises
            net.options.chain_form, net.options.banded_repair = "banded", False
            bad, _, _ = eng.incremental_cost_volume(*dev)
            torch.cuda.synchronize()
            assert eng.chain_status() != 0
            assert bool(torch.isnan(bad).any()), "a timed-out chain must poison its cost slice"
            with pytest.raises(RuntimeError, match="hand-off"):
                net.check_device_status()
        finally:
            eng.lib.mvsn_debug_set_band_flags(0)
            net.options.banded_repair = True
        net.options.chain_form = "banded"
        again, _, _ = eng.incremental_cost_volume(*dev)
        torch.cuda.synchronize()
        assert eng.chain_status() == 0 and torch.equal(again, good)
        assert net.check_device_status() == 0
    finally:
        net.options.chain_form = "auto"
        net.reset_device_status()
    if grid == (16, 32):
        # ... and end to end, through plain net(...) calls as torch.jit.load + the reference's multi_view_forward make
        # them (no status check by the caller): finite depth maps inside the contract on the forward whose banded chain
        # timed out, and the NEXT forward no longer picks the banded form
        fix = load_golden("g2_gta_512x256_d64_s2.npz")
        net.options.plan_max_chains = 0        # (eager: a hipGraph recorded by an earlier test has its flags frozen in)
        eng.lib.mvsn_debug_set_band_flags(2 | (10 << 8))
        try:
            with pytest.warns(RuntimeWarning, match="hand-off"):
                out = _forward(net, fix)
                assert net.engine().last_chain_form == _native.CHAIN_BANDED
                torch.cuda.synchronize()
                got = out["left_idepthmap_pyr"][0].cpu()
                out2 = _forward(net, fix)                      # polls the sticky words: latched
            assert net.engine().last_chain_form != _native.CHAIN_BANDED
        finally:
            eng.lib.mvsn_debug_set_band_flags(0)
            net.reset_device_status()
            net.options.plan_max_chains = 16
        assert bool(torch.isfinite(got).all())
        mean_rel, max_rel = rel_err(got, fix["idepth_0"])
        assert mean_rel < 2e-4 and max_rel < 1e-3, (mean_rel, max_rel)
        assert_contract(got, fix["idepth_0"], "repaired forward")
        mean_rel, max_rel = rel_err(out2["left_idepthmap_pyr"][0].cpu(), got)
        assert mean_rel < 1e-5, mean_rel
        out3 = _forward(net, fix)                              # latch reset: banded again, clean
        assert net.engine().last_chain_form == _native.CHAIN_BANDED and net.check_device_status() == 0


@pytest.mark.parametrize("chain_form,plan_graph,reps", [("auto", True, 600), ("winograd", False, 1500)])
def test_graph_replays_on_changing_inputs_match_eager(chain_form, plan_graph, reps):
    """A recorded forward replayed as one hipGraph launch on inputs that CHANGE from call to call must give what the
    eager forward gives on each of them, bit for bit.  (Regression: the banded chain's buffers were visible to the
    runtime only inside a by-value struct; under graph replay nothing ordered the caches between it and its
    neighbours, and a fraction of a percent of the forwards read lines of the previous replay -- invisible while the
    inputs repeat.  tools/soak.py is the long form of this test; csrc/mvsn_common.h MVSN_VIS10 the rule.)
    Second parameter set: the plane-resident chain with the plan replayed as a call list -- the set-up in which the
    tower kernel's zero-slot race (HISTORY 11.5) showed up about once per thousand forwards."""
    net = net_for("gta_sfm_150epochs")
    fix = load_golden("gc3_gta_512x256_d64_s5.npz")
    sets = []
    for k in range(3):
        meta = fix["meta"].copy()
        meta[5] = int(meta[5]) + 11 * k                      # the fixture's seed, then two others
        batch, D = batch_from_meta(meta, fix.get("jitter", 0.0), False)
        sets.append(to_dev(snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)))
    keep = (net.options.plan_max_chains, net.options.chain_form, net.options.plan_graph)
    net.options.chain_form, net.options.plan_graph = chain_form, plan_graph
    try:
        net.options.plan_max_chains = 0
        refs = [net(*x, D, True, [True] * 5)["left_idepthmap_pyr"][0].clone() for x in sets]
        net.options.plan_max_chains = keep[0]
        assert not torch.equal(refs[0], refs[1])
        before = net.engine().replays
        for i in range(reps):
            j = (i * 5 + i // 7) % 3
            got = net(*sets[j], D, True, [True] * 5)["left_idepthmap_pyr"][0]
            assert torch.equal(got, refs[j]), (i, j, float((got - refs[j]).abs().max()))
        assert net.engine().replays - before >= reps - 10 and net.engine().chain_status() == 0
    finally:
        net.options.plan_max_chains, net.options.chain_form, net.options.plan_graph = keep


@pytest.mark.parametrize("opt,val", [("fold_residual_blocks", True), ("winograd", False), ("towers", False),
                                     ("trim_tower_ends", False), ("conv_precision", "bf16x3")])
def test_planned_forward_matches_eager_under_option_variants(opt, val):
    """Every launch-sequence option must give the same bits planned (recorded once, replayed) as eager -- or mark its
    forward as not replayable.  (Regression: with fold_residual_blocks the refiner input was assembled by an unrecorded
    torch.cat, and every replay of that plan returned the first call's depth maps; found by tools/soak.py.)"""
    net = net_for("gta_sfm_150epochs")
    fix = load_golden("g2_gta_512x256_d64_s2.npz")
    sets = []
    for k in range(2):
        meta = fix["meta"].copy()
        meta[5] = int(meta[5]) + 7 * k
        batch, D = batch_from_meta(meta, fix.get("jitter", 0.0), False)
        sets.append(to_dev(snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)))
    keep = (getattr(net.options, opt), net.options.plan_max_chains)
    setattr(net.options, opt, val)
    try:
        net.options.plan_max_chains = 0
        refs = [net(*x, D, True, [True] * 5)["left_idepthmap_pyr"][0].clone() for x in sets]
        net.options.plan_max_chains = keep[1]
        for i in range(8):
            got = net(*sets[i % 2], D, True, [True] * 5)["left_idepthmap_pyr"][0]
            assert torch.equal(got, refs[i % 2]), (opt, i, float((got - refs[i % 2]).abs().max()))
    finally:
        setattr(net.options, opt, keep[0])
        net.options.plan_max_chains = keep[1]


def test_wrapper_status_check_across_graph_replays():
    """multi_view_forward (the reference's wrapper: timer with synchronize, then check_device_status) on a batch-1
    forward that goes eager -> recorded list -> hipGraph: the banded chain's status word must read 0 after every one of
    them (regression: a 64-byte hipMemset node in front of the chain left garbage in it on graph replay, and the
    wrapper raised on a healthy forward), and the outputs stay bit-identical."""
    net = net_for("gta_sfm_150epochs")
    fix = load_golden("g2_gta_512x256_d64_s2.npz")
    batch, D = batch_from_meta(fix["meta"], fix.get("jitter", 0.0), False)
    inp = snu.multi_view_unpack_batch(batch, DEV, 5)
    params = {"num_idepth_samples": D}
    first = None
    for i in range(8):
        out = snu.multi_view_forward(net, inp, params)          # raises if the status word is not 0
        assert net.engine().last_chain_form == _native.CHAIN_BANDED and net.engine().chain_status() == 0, i
        if first is None:
            first = out["left_idepthmap_pyr"][0].clone()
        assert torch.equal(out["left_idepthmap_pyr"][0], first), i
    assert net.engine().replays >= 6


def _forward(net, fix, smooth=False, **kw):
    batch, D = batch_from_meta(fix["meta"], fix.get("jitter", 0.0), smooth)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    lp, kp, ts, rp = to_dev(inp)
    return net(lp, kp, ts, rp, D, kw.get("flt", True), kw.get("refs", [True] * 5), capture=kw.get("capture"))


@pytest.mark.parametrize("name,wname", [("g1_gta_128x64_d16_s1.npz", "gta_sfm_150epochs"),
                                        ("g1_init_128x64_d16_s1.npz", None),
                                        ("g1b_gta_96x80_d8_s2_b2.npz", "gta_sfm_150epochs")])
def test_forward_full_capture_golden(name, wname):
    """Config 1 (and a batch-2 / two-source variant): every named intermediate of the reference."""
    fix = load_golden(name)
    cap = {}
    out = _forward(net_for(wname), fix, capture=cap)
    S, B = int(fix["meta"][3]), int(fix["meta"][4])
    for s in range(S):
        sl = slice(s * B, (s + 1) * B)
        close(cap["idepth_samples"][sl], fix[f"idepth_samples_{s}"], rtol=2e-5, atol=1e-7)
        close(cap["H"][sl], fix[f"H_{s}"], rtol=1e-4, atol=2e-5)
        mean_rel, max_rel = rel_err(cap["plane0_features"][sl].cpu(), fix[f"plane0_features_{s}"])
        assert mean_rel < 1e-4 and max_rel < 1e-4, ("plane0", s, mean_rel, max_rel)
        assert int((cap["mask_volume"][sl].cpu() != t(fix[f"mask_volume_{s}"])).sum()) == 0
        for key in ("feature_volume", "cost_volume", "filtered_cost"):
            mean_rel, max_rel = rel_err(cap[key][sl].cpu(), fix[f"{key}_{s}"])
            assert mean_rel < 2e-4 and max_rel < 1e-3, (key, s, mean_rel, max_rel)
    cols = int(fix["meta"][1])
    close(cap["warped_fullres"][:B], fix["warped_fullres_0"], rtol=1e-4, atol=cols * 2.0 ** -23 * 8)
    for lvl in range(1, 5):
        mean_rel, max_rel = rel_err(cap["left_features"][lvl].cpu(), fix[f"left_feat_{lvl}"])
        assert mean_rel < 1e-4 and max_rel < 1e-4, ("left features", lvl, mean_rel, max_rel)
    for lvl in range(5):
        for kind, key in (("idepth", "left_idepthmap_pyr"), ("raw", "left_idepthmap_raw_pyr")):
            mean_rel, max_rel = rel_err(out[key][lvl].cpu(), fix[f"{kind}_{lvl}"])
            assert mean_rel < 2e-4 and max_rel < 1e-3, (kind, lvl, mean_rel, max_rel)
        m = out["left_idepthmap_mask_pyr"][lvl]
        assert m.dtype == torch.bool and np.array_equal(m.cpu().numpy(), unpack_mask(fix, lvl))
    assert_contract(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"], name)


@pytest.mark.parametrize("name,wname,smooth", [("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", False),
                                               ("g2s_gta_512x256_d64_s2.npz", "gta_sfm_150epochs", True),
                                               ("g3_demon_640x480_d96_s1.npz", "demon_45epochs", False)])
def test_forward_headline_golden(name, wname, smooth):
    """BASELINE configs 2-4 shapes with pretrained weights: final depth within the 1e-3 contract."""
    fix = load_golden(name)
    cap = {}
    out = _forward(net_for(wname), fix, smooth=smooth, capture=cap)
    S = int(fix["meta"][3])
    for s in range(S):
        close(cap["idepth_samples"][s:s + 1], fix[f"idepth_samples_{s}"], rtol=2e-5, atol=1e-7)
        assert abs(int(cap["mask_volume"][s].sum()) - int(fix[f"mask_volume_count_{s}"])) <= 2
        mean_rel, _ = rel_err(cap["feature_volume"][s:s + 1, :, -1].cpu(), fix[f"feature_volume_last_plane_{s}"])
        assert mean_rel < 3e-4, ("last plane", s, mean_rel)
        mean_rel, _ = rel_err(cap["filtered_cost"][s:s + 1].cpu(), fix[f"filtered_cost_{s}"])
        assert mean_rel < 3e-4, ("filtered", s, mean_rel)
    l1 = (out["left_idepthmap_pyr"][0].cpu() - t(fix["idepth_0"])).abs().mean().item()
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"{name}: L1 {l1:.3e} mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
    assert mean_rel < 2e-4 and max_rel < 1e-3
    assert_contract(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"], name)
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][4].cpu(), fix["idepth_4"])
    assert mean_rel < 2e-4 and max_rel < 1e-3
    for lvl in range(5):
        assert abs(int(out["left_idepthmap_mask_pyr"][lvl].sum()) - int(fix[f"mask_count_{lvl}"])) <= 2 * 4 ** (4 - lvl)
    _check_mask4(name, out, fix)


def test_forward_headline_launch_geometry_vs_golden_and_oracle():
    """The HEADLINE launch geometry inside pytest (the golden forwards all ran at B = 1): 128 reference
    images x 2 sources = 256 chains -- one plane-resident chain per CU (`chain_wino_kernel<16,32>`), the refiner towers
    on two batch slices with carried passes, the regulariser on two slices.  Image 0 is g2's input and is held against the
    reference's own depth map; the FIRST image of slice B (index 64) and the LAST image (127: slice B, last round of
    every persistent kernel) are held against the oracle on the same seeded inputs -- contract per pixel < 1e-3, masks
    of every level bit for bit (multi_view_stereonet.py:538-695)."""
    fix = load_golden("g2_gta_512x256_d64_s2.npz")
    rows, cols, D, S, _, seed0 = (int(x) for x in fix["meta"])
    B = 128
    net = net_for("gta_sfm_150epochs")
    parts = [synthetic.make_batch(rows, cols, S, batch=1, seed=seed0 + i) for i in range(B)]
    merged = {"left_image": torch.cat([p["left_image"] for p in parts], 0),
              "right_image": [torch.cat([p["right_image"][s] for p in parts], 0) for s in range(S)],
              "K": torch.cat([p["K"] for p in parts], 0),
              "T_right_in_left": [torch.cat([p["T_right_in_left"][s] for p in parts], 0) for s in range(S)]}
    inp = snu.multi_view_unpack_batch(merged, torch.device(DEV), 5)
    out = net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5)
    eng = net.engine()
    assert eng.last_chain_form == _native.CHAIN_WINOGRAD and eng.last_chain_shape[0] == B * S
    assert bool(torch.isfinite(out["left_idepthmap_pyr"][0]).all())
    got0 = out["left_idepthmap_pyr"][0][:1].cpu()
    assert_contract(got0, fix["idepth_0"], "B=128 image 0 vs g2")
    w = load_weights("gta_sfm_150epochs")
    for idx in (B // 2, B - 1):
        cpu_in = snu.multi_view_unpack_batch(parts[idx], torch.device("cpu"), 5)
        ref = oracle.forward(w, cpu_in["left_image_pyr"], cpu_in["K_pyr"], cpu_in["T_right_in_left"],
                             cpu_in["right_image_pyr"], D)
        for lvl in range(5):
            got = out["left_idepthmap_pyr"][lvl][idx:idx + 1].cpu()
            mean_rel, max_rel = rel_err(got, ref["left_idepthmap_pyr"][lvl])
            assert mean_rel < 2e-4 and max_rel < 1e-3, (idx, lvl, mean_rel, max_rel)
            m = out["left_idepthmap_mask_pyr"][lvl][idx:idx + 1].cpu()
            # (masks derive from the level-4 predicate; the oracle evaluates the same fp32 expression order)
            assert int((m != ref["left_idepthmap_mask_pyr"][lvl]).sum()) <= 2 * 4 ** (4 - lvl), (idx, lvl)
        assert_contract(out["left_idepthmap_pyr"][0][idx:idx + 1].cpu(), ref["left_idepthmap_pyr"][0],
                        f"B=128 image {idx} vs oracle")


def test_bench_real_multi_rank_line_on_one_gpu():
    """The REAL N > 1 line of bench.py (not the launcher self-test): two ranks under torch.distributed.run that both use
    cuda:0 and exchange over gloo (`--single-device-selftest`) -- forward, timed region with barriers and max over
    ranks, all-gather of the per-image metric rows, per-rank roofline fractions, parity of image 0 (test.py:146-164,
    188-280).  RCCL itself needs N GPUs; everything else of the N-rank path runs here."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device-selftest",
                        "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-tiers"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert p.stdout.rstrip("\n").splitlines()[-1] == lines[0] and len(lines[0]) < 4096      # the driver parses the LAST line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == 16 and len(line["per_rank_ms_per_step"]) == 2
    assert len(line["roofline"]["frac_per_rank"]) == 2 and all(0.0 < f < 1.0 for f in line["roofline"]["frac_per_rank"])
    l1 = line["l1_vs_ref"]
    assert np.isfinite(l1["l1"]) and l1["max_rel_per_pixel"] < 1e-3, l1
    assert np.isfinite(line["mean_idepth"]) and line["value"] > 0 and "selftest" in line
    print("bench --gpus 2 --single-device-selftest:", lines[0][:400])


_SHARED_GPU_WORKER = r"""
import os, sys, time, warnings
import numpy as np, torch
root, tag, tmp, forwards = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from conftest import load_golden, batch_from_meta, rel_err_per_pixel
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd import multi_view_stereonet_utils as snu
from multi_view_stereonet_amd.weights import load_weights
from oracle import mvsn_oracle as oracle
torch.set_grad_enabled(False)
warnings.simplefilter("error", RuntimeWarning)                  # a repair warns: that must not happen here
w = load_weights("gta_sfm_150epochs")
net = MultiViewStereoNet(); net.load_state_dict(w, strict=True); net = net.to("cuda").eval()
fix = load_golden("g2_gta_512x256_d64_s2.npz")
batch, D = batch_from_meta(fix["meta"])
cpu = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
want = oracle.forward(w, cpu["left_image_pyr"], cpu["K_pyr"], cpu["T_right_in_left"], cpu["right_image_pyr"], D)["left_idepthmap_pyr"][0]
inp = snu.multi_view_unpack_batch(batch, torch.device("cuda"), 5)
run = lambda: net(inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D, True, [True] * 5)
out = run(); torch.cuda.synchronize()                           # (the first forward asks for the device's co-resident right)
form = net.engine().last_chain_form
open(os.path.join(tmp, tag + ".ready"), "w").close()
t0 = time.time()
while not all(os.path.exists(os.path.join(tmp, x + ".ready")) for x in ("a", "b")) and time.time() - t0 < 300:
    time.sleep(0.01)
worst = 0.0
for i in range(forwards):                                        # both processes hammer cuda:0 at the same time
    out = run()
    if i % 8 == 7 or i == forwards - 1:
        got = out["left_idepthmap_pyr"][0].cpu()
        assert bool(torch.isfinite(got).all())
        worst = max(worst, rel_err_per_pixel(got, want)[0], rel_err_per_pixel(got, torch.from_numpy(fix["idepth_0"]))[0])
torch.cuda.synchronize()
repairs = net.check_device_status()
assert net.engine().last_chain_form == form
print("SHARED", tag, form, repairs, "%.3e" % worst, int(_native.coresident_right(torch.cuda.current_device())), flush=True)
open(os.path.join(tmp, tag + ".done"), "w").close()
while not all(os.path.exists(os.path.join(tmp, x + ".done")) for x in ("a", "b")) and time.time() - t0 < 600:
    time.sleep(0.01)                                             # (keep the lock until the other one is through)
"""


def test_two_processes_on_one_gpu_share_it_without_repairs(tmp_path):
    """"one banded call per device" is ENFORCED across processes (an advisory lock keyed by the GPU's
    identity, `_native.coresident_right`), not documented.  Two processes run batch-1 headline forwards on cuda:0 at the same
    time: the one that asked first keeps the banded chain, the other gets the single-launch plane-resident form; both stay
    inside the contract against the oracle AND the reference fixture on every checked forward, with 0 repaired forwards
    (before: two banded launches could each hold part of the chip and time out into the repair launch)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    _native.release_coresident_right()            # this pytest process may hold the right from earlier tests
    try:
        env = dict(os.environ, MVSN_LOCK_DIR=str(tmp_path))
        env.pop("MVSN_CORESIDENT_LOCK", None)
        procs = {}
        for tag in ("a", "b"):
            procs[tag] = subprocess.Popen([sys.executable, "-c", _SHARED_GPU_WORKER, ROOT, tag, str(tmp_path), "120"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if tag == "a":                        # a asks first (its ready file follows its first forward)
                import time
                t0 = time.time()
                while not os.path.exists(os.path.join(str(tmp_path), "a.ready")):
                    assert procs["a"].poll() is None and time.time() - t0 < 600, procs["a"].communicate()[1][-3000:]
                    time.sleep(0.05)
        res = {}
        for tag, p in procs.items():
            out, err = p.communicate(timeout=900)
            assert p.returncode == 0, (tag, out[-1000:], err[-3000:])
            ln = [x for x in out.splitlines() if x.startswith("SHARED")][-1].split()
            res[tag] = dict(form=int(ln[2]), repairs=int(ln[3]), worst=float(ln[4]), right=int(ln[5]))
        print("two processes on cuda:0:", res)
        assert res["a"]["right"] == 1 and res["b"]["right"] == 0
        assert res["a"]["form"] == _native.CHAIN_BANDED and res["b"]["form"] == _native.CHAIN_WINOGRAD
        for tag in ("a", "b"):
            assert res[tag]["repairs"] == 0 and res[tag]["worst"] < 1e-3, res
    finally:
        _native.release_coresident_right()


_RCCL_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from multi_view_stereonet_amd import distributed as mdist
rank, world, local = mdist.init_from_env(backend="nccl", force=True)      # world size 1: loads librccl, sets the device
assert (rank, world, local) == (0, 1, 0) and dist.get_backend() == "nccl" and torch.cuda.current_device() == 0
dev = torch.device("cuda", 0)
seen = []
real = dist.all_gather
dist.all_gather = lambda out, x, *a, **k: (seen.append((tuple(x.shape), str(x.dtype), x.device.type)), real(out, x, *a, **k))[1]
rows = torch.tensor([[0.11, 0.02, 1.5, 0.3, 0.9, 0.97, 0.99, 2.46, 8.9e-6], [0.12, 0.03, 1.6, 0.31, 0.91, 0.98, 0.995, 2.5, 9.1e-6],
                     [0.10, 0.01, 1.4, 0.29, 0.89, 0.96, 0.985, 2.4, 8.7e-6]], dtype=torch.float32, device=dev)
idx = torch.tensor([7, 3, 5], dtype=torch.int64, device=dev)
all_rows, all_idx = mdist.gather_metric_rows(rows, idx)
dist.all_gather = real
torch.cuda.synchronize()
assert seen == [((1,), "torch.int64", "cuda"), ((3, 10), "torch.float64", "cuda")], seen
assert all_idx.tolist() == [3, 5, 7] and all_rows.device.type == "cuda" and all_rows.dtype == torch.float32
assert torch.equal(all_rows.cpu(), rows.cpu()[[1, 2, 0]])
empty_rows, empty_idx = mdist.gather_metric_rows(rows[:0], idx[:0])          # a rank whose images were all skipped
assert empty_rows.shape == (0, 9) and empty_idx.numel() == 0
dist.barrier()
loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln]
assert loaded, "no RCCL library mapped into the process"
print("RCCL_OK", os.path.basename(loaded[0]), torch.cuda.nccl.version())
dist.destroy_process_group()
"""


def _clean_rank_env():
    import os
    from conftest import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_rccl_single_rank_group_carries_the_metric_rows():
    """RCCL had never been LOADED in any round (every multi-rank test is gloo).  A world-size-1 `nccl`
    process group on cuda:0, and `gather_metric_rows`' exact payload -- the int64 count, then float64 rows + index column, as
    DEVICE tensors -- through `dist.all_gather`; `init_from_env`'s `torch.cuda.set_device(local)` runs too (test.py:146-164
    is what the rows feed)."""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, "-c", _RCCL_WORKER, ROOT], env=_clean_rank_env(), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "RCCL_OK" in p.stdout, p.stdout[-1500:]
    print(p.stdout.strip().splitlines()[-1])


def test_bench_single_gpu_line_over_rccl():
    """`MVSN_BENCH_BACKEND=nccl python bench.py --gpus 1`: the measuring path itself with a world-size-1 RCCL group --
    barriers around the timed region, the per-rank timing / roofline gathers and the metric rows all go through RCCL on
    device tensors; the line says `backend: nccl`."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = _clean_rank_env()
    env["MVSN_BENCH_BACKEND"] = "nccl"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--no-cpu-baseline", "--no-tiers"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = p.stdout.rstrip("\n").splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0] and len(lines[0]) < 4096, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["backend"] == "nccl" and line["world_size"] == 1 and line["n_gpus"] == 1
    assert line["l1_vs_ref"]["max_rel_per_pixel"] < 1e-3 and np.isfinite(line["mean_idepth"]) and line["value"] > 0
    assert len(line["roofline"]["frac_per_rank"]) == 1 and len(line["per_rank_ms_per_step"]) == 1
    print("bench --gpus 1 over RCCL:", lines[0][:300])


def test_forward_headline_golden_direct_chain_form():
    """The headline fixture once more with the chain's convolutions in their DIRECT form (the path of the coarse grids
    the Winograd plan does not cover): both forms sit inside the contract, and agree with each other far below it."""
    fix = load_golden("g2_gta_512x256_d64_s2.npz")
    net = net_for("gta_sfm_150epochs")
    wino = _forward(net, fix)["left_idepthmap_pyr"][0]
    net.options.chain_form = "direct"
    try:
        out = _forward(net, fix)
    finally:
        net.options.chain_form = "auto"
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    assert mean_rel < 2e-4 and max_rel < 1e-3, (mean_rel, max_rel)
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), wino.cpu())
    print(f"direct vs Winograd chain, final idepth: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
    assert mean_rel < 2e-5 and max_rel < 2e-4
    _check_mask4("g2 (direct chain)", out, fix)


def _check_mask4(name, out, fix):
    """The stored level-4 mask of the reference, voxel by voxel: mismatches are counted and printed, and only
    voxels whose normalised coordinate sits within an ulp of the |n| > 1 predicate may differ."""
    m = out["left_idepthmap_mask_pyr"][4].cpu().numpy()
    ref = np.unpackbits(fix["mask_4"])[:m.size].reshape(m.shape).astype(bool)
    bad = int((m != ref).sum())
    print(f"{name}: level-4 mask mismatches {bad} of {m.size}")
    assert bad == 0, (name, bad)


CONFIG_FIXTURES = [("gc2_gta_512x256_d64_s1.npz", "config 2: 512x256, D=64, S=1"),
                   ("gc3_gta_512x256_d64_s5.npz", "config 3: 512x256, D=64, S=5"),
                   ("gc5_gta_1024x512_d128_s4.npz", "config 5: 1024x512, D=128, S=4")]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name,what", CONFIG_FIXTURES)
def test_forward_baseline_configs_golden(name, what, precision):
    """BASELINE configs 2, 3 and 5 at their stated sizes against the reference's own outputs
    (multi_view_stereonet.py:564-627: the per-source loop and the fusion), in exact fp32 and with the
    3 x bf16 split tier -- both inside the 1e-3 contract."""
    fix = load_golden(name)
    net = net_for("gta_sfm_150epochs")
    net.options.conv_precision = precision
    try:
        cap = {}
        out = _forward(net, fix, capture=cap)
    finally:
        net.options.conv_precision = "fp32"
    S = int(fix["meta"][3])
    for s in range(S):
        close(cap["idepth_samples"][s:s + 1], fix[f"idepth_samples_{s}"], rtol=2e-5, atol=1e-7)
        close(cap["H"][s:s + 1], fix[f"H_{s}"], rtol=1e-4, atol=2e-5)
        assert int(cap["mask_volume"][s].sum()) == int(fix[f"mask_volume_count_{s}"])
    for lvl in (0, 4):
        got = out["left_idepthmap_pyr"][lvl].cpu()
        ref = t(fix[f"idepth_{lvl}"])
        l1 = (got - ref).abs().mean().item()
        mean_rel, max_rel = rel_err(got, ref)
        print(f"{what} [{precision}] level {lvl}: L1 {l1:.3e} mean-rel {mean_rel:.3e} max-rel {max_rel:.3e}")
        assert mean_rel < 2e-4 and max_rel < 1e-3, (what, precision, lvl, mean_rel, max_rel)
        assert_contract(got, ref, f"{what} [{precision}] level {lvl}")
    mean_rel, max_rel = rel_err(out["left_idepthmap_raw_pyr"][4].cpu(), fix["raw_4"])
    assert mean_rel < 2e-4 and max_rel < 1e-3
    for lvl in range(5):
        assert abs(int(out["left_idepthmap_mask_pyr"][lvl].sum()) - int(fix[f"mask_count_{lvl}"])) <= 2 * 4 ** (4 - lvl)
    _check_mask4(name, out, fix)


# BASELINE config 5's speed tier (bf16 operands, fp32 accumulate on the 32 -> 32 3x3[x3] layers) is outside the 1e-3
# contract by construction; its OWN error budget against the reference's depth maps, asserted here
# and quoted by bench.py's `bf16_operand_tier`:
BF16_TIER_BUDGET = {"mean_rel": 5e-3, "p999_rel_per_pixel": 2e-2, "max_rel_per_pixel": 5e-2}


def test_forward_config5_bf16_operand_tier_reported():
    """BASELINE config 5 names a bf16 speed tier.  Plain bf16 operands on the 32->32 3x3[x3] layers are outside
    the 1e-3 contract (SURVEY section 7 measured it), so the error against the reference is REPORTED here and only
    bounded loosely; the masks do not depend on the tier and stay bit-exact."""
    name = "gc5_gta_1024x512_d128_s4.npz"
    fix = load_golden(name)
    net = net_for("gta_sfm_150epochs")
    net.options.conv_precision = "bf16"
    try:
        out = _forward(net, fix)
    finally:
        net.options.conv_precision = "fp32"
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    mx, p999 = rel_err_per_pixel(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"config 5 [bf16 operands] level 0: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e} per-pixel max {mx:.3e} "
          f"p99.9 {p999:.3e} (contract 1e-3: outside; the tier's own budget: {BF16_TIER_BUDGET})")
    assert mean_rel < BF16_TIER_BUDGET["mean_rel"] and p999 < BF16_TIER_BUDGET["p999_rel_per_pixel"] and \
        mx < BF16_TIER_BUDGET["max_rel_per_pixel"], (mean_rel, p999, mx)
    _check_mask4(name, out, fix)


# ... and with bf16 STORAGE of the regulariser's intermediate volumes on top (config 5's "bf16 features", `bf16s`): its
# own, wider budget -- every voxel of three of the four layers' outputs is rounded to 8 mantissa bits once more (measured on
# config 5: mean-rel 8.7e-3, per-pixel p99.9 2.6e-2, max 3.1e-2; headline shapes: see the test's printout)
BF16_FEATURE_TIER_BUDGET = {"mean_rel": 1.5e-2, "p999_rel_per_pixel": 4e-2, "max_rel_per_pixel": 8e-2}


@pytest.mark.parametrize("name,wname", [("gc5_gta_1024x512_d128_s4.npz", "gta_sfm_150epochs"),
                                        ("g2_gta_512x256_d64_s2.npz", "gta_sfm_150epochs")])
def test_forward_bf16_feature_tier_reported(name, wname):
    """BASELINE config 5's tier as written -- bf16 FEATURES: the regulariser's intermediate volumes stored as bf16
    (`conv_precision = "bf16s"`: mvsn_conv_forward_bf16_storage; fp32 accumulation and GroupNorm statistics), on top of
    the bf16-operand kernels.  Outside the 1e-3 contract like the operand tier: reported, bounded by its own budget;
    the masks do not depend on the tier and stay bit-exact; the storage layers did run."""
    fix = load_golden(name)
    net = net_for(wname)
    net.options.conv_precision = "bf16s"
    try:
        eng = net.engine()
        eng.timeline = []
        out = _forward(net, fix)
        torch.cuda.synchronize()
        ran = [t[0] for t in eng.timeline]
        eng.timeline = None
    finally:
        net.options.conv_precision = "fp32"
    S = int(fix["meta"][3])
    assert sum("bf16 storage" in k for k in ran) == 4, ran[:40]          # (all S sources run as one batch of chains)
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    mx, p999 = rel_err_per_pixel(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    print(f"{name} [bf16 features, S={S}] level 0: mean-rel {mean_rel:.3e} max-rel {max_rel:.3e} per-pixel max {mx:.3e} "
          f"p99.9 {p999:.3e} (the tier's own budget: {BF16_FEATURE_TIER_BUDGET})")
    assert mean_rel < BF16_FEATURE_TIER_BUDGET["mean_rel"] and p999 < BF16_FEATURE_TIER_BUDGET["p999_rel_per_pixel"] and \
        mx < BF16_FEATURE_TIER_BUDGET["max_rel_per_pixel"], (mean_rel, p999, mx)
    _check_mask4(name, out, fix)


@pytest.mark.parametrize("n,depth,rows,cols", [(2, 8, 16, 32), (1, 5, 9, 36), (3, 6, 30, 40), (1, 4, 17, 30)])
def test_conv3d_bf16_storage_is_the_bf16_kernel_with_rounded_tensors(n, depth, rows, cols):
    """mvsn_conv_forward_bf16_storage against the bf16-operand kernel it extends (MVSN_CONV_BF16, fp32 tensors), bit for
    bit: a bf16 INPUT tensor gives what the fp32 kernel gives on the same values widened; a bf16 OUTPUT tensor is the
    fp32 kernel's output rounded to nearest-even; the GroupNorm records are those of the unrounded accumulators; with
    and without the previous layer's LeakyReLU(GroupNorm(.)) applied on load.  And against ATen on the rounded operands."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Conv
    eng = net_for("gta_sfm_150epochs").engine()
    lib = eng.lib
    g = torch.Generator().manual_seed(depth * 31 + cols)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.06
    b = torch.randn(32, generator=g) * 0.1
    x = torch.randn(n, 32, depth, rows, cols, generator=g)
    c = _Conv(lib, w.to(DEV), b.to(DEV))
    assert c.packed_bx is not None
    d = c.desc(n, depth, rows, cols, _native.CONV_BF16)
    assert lib.mvsn_conv_bf16x3_supported(ctypes.byref(d))
    tiles = lib.mvsn_conv_num_tiles(ctypes.byref(d))
    x16 = x.to(DEV).to(torch.bfloat16).contiguous()
    xw = x16.float().contiguous()                       # the same values, fp32 storage
    gamma, beta = (torch.rand(32, generator=g) + 0.5).to(DEV), (torch.randn(32, generator=g) * 0.1).to(DEV)
    xg = xw.reshape(n, 4, -1).double()
    st = torch.stack([xg.mean(2), 1.0 / (xg.var(2, unbiased=False) + 1e-5).sqrt()], 2).float().contiguous()

    def plain(inp, stats):
        out = torch.empty(n, 32, depth, rows, cols, device=DEV)
        part = torch.empty(n, tiles, 4, 3, device=DEV)
        rc = lib.mvsn_conv_forward(ctypes.byref(d), _native.ptr(inp), _native.ptr(c.packed_bx), _native.ptr(c.bias),
                                   _native.ptr(stats), _native.ptr(gamma) if stats is not None else None,
                                   _native.ptr(beta) if stats is not None else None, None, None, _native.ptr(out),
                                   _native.ptr(part), _native.stream())
        assert rc == 0, lib.mvsn_last_error()
        return out, part

    def storage(inp, in16, out16, stats):
        out = torch.empty(n, 32, depth, rows, cols, device=DEV, dtype=torch.bfloat16 if out16 else torch.float32)
        part = torch.empty(n, tiles, 4, 3, device=DEV)
        rc = lib.mvsn_conv_forward_bf16_storage(ctypes.byref(d), _native.ptr(inp), int(in16), _native.ptr(c.packed_bx),
                                                _native.ptr(c.bias), _native.ptr(stats),
                                                _native.ptr(gamma) if stats is not None else None,
                                                _native.ptr(beta) if stats is not None else None, _native.ptr(out),
                                                int(out16), _native.ptr(part), _native.stream())
        assert rc == 0, lib.mvsn_last_error()
        return out, part

    for stats in (None, st):
        ref, ref_part = plain(xw, stats)
        for in16, out16 in ((False, True), (True, True), (True, False)):
            got, part = storage(x16 if in16 else xw, in16, out16, stats)
            want = ref.to(torch.bfloat16) if out16 else ref
            assert torch.equal(got, want), (in16, out16, stats is not None, float((got.float() - want.float()).abs().max()))
            assert torch.equal(part, ref_part), (in16, out16)
    a = xw.cpu()
    aten = F.conv3d(a.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, padding=1)
    close(plain(xw, None)[0], aten, rtol=2e-3, atol=2e-3)
    # rejected: neither tensor bf16; a 2-D layer
    rc = lib.mvsn_conv_forward_bf16_storage(ctypes.byref(d), _native.ptr(xw), 0, _native.ptr(c.packed_bx), _native.ptr(c.bias),
                                            None, None, None, _native.ptr(ref), 0, None, _native.stream())
    assert rc != 0


def test_forward_flag_variants_golden():
    fix = load_golden("g6_flags_128x64.npz")
    net = net_for("gta_sfm_150epochs")
    variants = {"nofilter": (False, [True] * 5),
                "norefine4": (True, [True, True, True, True, False]),
                "norefine_all": (True, [False] * 5),
                "norefine_0_2": (True, [False, True, False, True, True])}
    for key, (flt, refs) in variants.items():
        out = _forward(net, fix, flt=flt, refs=refs)
        for lvl in (0, 4):
            for kind, okey in (("idepth", "left_idepthmap_pyr"), ("raw", "left_idepthmap_raw_pyr")):
                mean_rel, max_rel = rel_err(out[okey][lvl].cpu(), fix[f"{key}:{kind}_{lvl}"])
                assert mean_rel < 2e-4 and max_rel < 1e-3, (key, kind, lvl, mean_rel, max_rel)


def test_unpack_batch_on_device_matches_host():
    """pyramids built by the HIP kernel inside multi_view_unpack_batch."""
    fix = load_golden("g4_units.npz")
    pyr = snu.build_image_pyramid(t(fix["pyr_in"]).to(DEV), 4)
    for i, p in enumerate(pyr):
        close(p, fix[f"pyr_{i}"], rtol=1e-6, atol=1e-7)
    batch = synthetic.make_batch(60, 90, 2, batch=2, seed=4, pose_jitter=0.2)
    host = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    dev = snu.multi_view_unpack_batch(batch, torch.device(DEV), 5)
    for a, b in zip(dev["left_image_pyr"], host["left_image_pyr"]):
        assert a.shape == b.shape
        close(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(dev["right_image_pyr"][1], host["right_image_pyr"][1]):
        close(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(dev["K_pyr"], host["K_pyr"]):
        close(a, b, rtol=1e-6, atol=1e-6)
    close(dev["T_right_in_left"][1], host["T_right_in_left"][1], rtol=1e-6, atol=1e-7)
    close(dev["T_left_in_right"][1], host["T_left_in_right"][1], rtol=1e-5, atol=1e-6)
    close(dev["baseline"], host["baseline"], rtol=1e-6, atol=0)
    # sizes that halve exactly: all levels in ONE launch (mvsn_image_pyramid) -- bit-identical to the host pyramid
    # (exact 2x2 means), and the K pyramid / normalised poses of mvsn_prepare_cameras bit-identical too
    batch = synthetic.make_batch(64, 160, 3, batch=3, seed=9, pose_jitter=0.3)
    host = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    dev = snu.multi_view_unpack_batch(batch, torch.device(DEV), 5)
    assert _native.load().mvsn_image_pyramid_supported(64, 160, 5) == 1
    for a, b in zip(dev["left_image_pyr"] + dev["right_image_pyr"][2], host["left_image_pyr"] + host["right_image_pyr"][2]):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)
    for a, b in zip(dev["K_pyr"], host["K_pyr"]):
        assert torch.equal(a.cpu(), b)
    for s_ in range(3):
        assert torch.equal(dev["T_right_in_left"][s_].cpu(), host["T_right_in_left"][s_])
        close(dev["T_left_in_right"][s_], host["T_left_in_right"][s_], rtol=1e-5, atol=1e-6)
    assert torch.equal(dev["baseline"].cpu(), host["baseline"])


def test_forward_properties_at_headline_size():
    """Size-independent properties at 512x256 / D=64 / S=2 (no oracle needed):
    * the order of the source views does not matter (the fusion is a mean);
    * scaling every translation by k scales the level-4 idepth (raw and refined) by exactly 1/k: each source
      is renormalised to unit baseline inside the forward and divided by its baseline afterwards
      (:566-571, :615-619).  The finer levels refine in those units and are not scale-equivariant, which is
      why multi_view_unpack_batch normalises the poses first."""
    net = net_for("gta_sfm_150epochs")
    batch = synthetic.make_batch(256, 512, 2, batch=2, seed=77, pose_jitter=0.15, smooth=True)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    lp, kp, ts, rp = to_dev(inp)
    base = net(lp, kp, ts, rp, 64, True, [True] * 5)
    swapped = net(lp, kp, ts[::-1], rp[::-1], 64, True, [True] * 5)
    close(swapped["left_idepthmap_pyr"][0], base["left_idepthmap_pyr"][0].cpu(), rtol=1e-5, atol=1e-6)
    assert torch.equal(swapped["left_idepthmap_mask_pyr"][0], base["left_idepthmap_mask_pyr"][0])
    k = 2.5
    ts_k = [t_.clone() for t_ in ts]
    for t_ in ts_k:
        t_[:, :3, 3] *= k
    scaled = net(lp, kp, ts_k, rp, 64, True, [True] * 5)
    close(scaled["left_idepthmap_pyr"][4] * k, base["left_idepthmap_pyr"][4].cpu(), rtol=2e-5, atol=2e-6)
    close(scaled["left_idepthmap_raw_pyr"][4] * k, base["left_idepthmap_raw_pyr"][4].cpu(), rtol=2e-5, atol=2e-6)
    assert torch.equal(scaled["left_idepthmap_mask_pyr"][4], base["left_idepthmap_mask_pyr"][4])


def test_forward_batch_independence_and_determinism():
    """Images are independent units: a batch of 3 equals three batches of 1, bit for
    bit except GroupNorm partial-combination order (none here: same tiles), and reruns agree."""
    net = net_for("gta_sfm_150epochs")
    batch = synthetic.make_batch(128, 256, 2, batch=3, seed=21, pose_jitter=0.2)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    lp, kp, ts, rp = to_dev(inp)
    full = net(lp, kp, ts, rp, 24, True, [True] * 5)
    again = net(lp, kp, ts, rp, 24, True, [True] * 5)
    assert torch.equal(full["left_idepthmap_pyr"][0], again["left_idepthmap_pyr"][0])
    for b in range(3):
        one = net([x[b:b + 1] for x in lp], [x[b:b + 1] for x in kp], [x[b:b + 1] for x in ts],
                  [[x[b:b + 1] for x in p] for p in rp], 24, True, [True] * 5)
        close(one["left_idepthmap_pyr"][0], full["left_idepthmap_pyr"][0][b:b + 1].cpu(), rtol=1e-5, atol=1e-6)
        assert torch.equal(one["left_idepthmap_mask_pyr"][0], full["left_idepthmap_mask_pyr"][0][b:b + 1])


def test_forward_config5_shape_vs_oracle():
    """BASELINE config 5's geometry (1024x512 frames, 32x64 coarse grid, 4 sources) in fp32; fewer
    hypotheses than 128 so the host oracle finishes in seconds."""
    wname = "gta_sfm_150epochs"
    w = load_weights(wname)
    batch = synthetic.make_batch(512, 1024, 4, batch=1, seed=44, smooth=True)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    ref = oracle.forward(w, inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 16)
    lp, kp, ts, rp = to_dev(inp)
    out = net_for(wname)(lp, kp, ts, rp, 16, True, [True] * 5)
    for lvl in (0, 4):
        mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][lvl].cpu(), ref["left_idepthmap_pyr"][lvl])
        assert mean_rel < 2e-4 and max_rel < 1e-3, (lvl, mean_rel, max_rel)
        diff = int((out["left_idepthmap_mask_pyr"][lvl].cpu() != ref["left_idepthmap_mask_pyr"][lvl]).sum())
        assert diff <= 2 * 4 ** (4 - lvl), (lvl, diff)


def test_evaluate_cli_on_miniature_dataset(tmp_path):
    """datasets -> multi_view_unpack_batch (device) -> HIP forward -> depth metrics, end to end."""
    import json
    import numpy as np
    from PIL import Image
    from multi_view_stereonet_amd import evaluate
    root = tmp_path / "gta"
    seq = root / "scene" / "0000"
    (seq / "images").mkdir(parents=True)
    (seq / "depth").mkdir()
    rng = np.random.default_rng(0)
    with open(seq / "intrinsics.txt", "w") as f, open(seq / "poses.txt", "w") as g:
        f.write("id K\n")
        g.write("id T\n")
        for i in range(3):
            T = np.eye(4, dtype=np.float32)
            T[0, 3] = 0.3 * i
            f.write(f"{i} 200 0 128.5 0 200 64.5 0 0 1\n")
            g.write(str(i) + " " + " ".join(str(v) for v in T.reshape(-1)) + "\n")
            Image.fromarray(rng.integers(0, 255, (128, 256, 3), dtype=np.uint8), "RGB").save(seq / "images" / f"{i:04d}.jpg")
            np.save(seq / "depth" / f"{i:04d}.npy", np.full((128, 256), 5.0, np.float32))
    split = tmp_path / "split.txt"
    split.write_text("scene/0000/images/0001.jpg scene/0000/images/0000.jpg scene/0000/images/0002.jpg\n"
                     "scene/0000/images/0000.jpg scene/0000/images/0001.jpg scene/0000/images/0002.jpg\n")
    out_dir = tmp_path / "out"
    evaluate.main(["gta_sfm_150epochs", str(root), str(split), "--size", "128", "256", "--num_idepth_samples", "8",
                   "--output_dir", str(out_dir)])
    txt = (out_dir / "avg_depth_metrics.txt").read_text().split("\n")
    vals = dict(zip(txt[0].split(), (float(v) for v in txt[1].split())))
    assert set(["abs_rel", "rmse", "a1", "runtime_ms"]) <= set(vals) and all(np.isfinite(list(vals.values())))
    # the actual numbers: the same two samples through the dataset reader, the blocking unpack, one forward each and
    # the reference's host-side metric arithmetic (numpy, image by image) must give the averages the CLI wrote
    from multi_view_stereonet_amd import datasets, metrics
    params = {"size": [128, 256], "num_idepth_samples": 8, "cost_volume_filter": True, "refiners": [True] * 5}
    data = datasets.GTASfMMultiViewStereoDataset(str(root), str(split), transform=datasets.get_testing_transforms(params),
                                                 load_groundtruth_depthmaps=True, shuffle_on_read=False)
    net = net_for("gta_sfm_150epochs")
    rows = []
    for batch in torch.utils.data.DataLoader(data, batch_size=1, shuffle=False):
        inputs = snu.multi_view_unpack_batch(batch, torch.device(DEV), 5)
        out = snu.multi_view_forward(net, inputs, params)
        est = metrics.idepth_to_depth(out["left_idepthmap_pyr"][0], inputs["baseline"])[0, 0].cpu().numpy()
        rows.append(metrics.image_metric_row(batch["left_depthmap_true"][0, 0].numpy(), est, *metrics.depth_range("gta_sfm")))
    want = metrics.compute_avg_metrics(rows)
    assert want["num_samples"] == 2
    for k in metrics.METRIC_KEYS:
        assert abs(vals[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, vals[k], want[k])
    assert 0.0 < vals["abs_rel"] < 10.0 and vals["rmse"] > 0.0      # random frames against a constant 5 m truth


def test_depth_metrics_on_device_match_the_host_arithmetic():
    """mvsn_depth_metrics (idepth -> depth, range masks, seven means per image; test.py:41-71, 210-235) against the
    numpy functions pinned to the reference (g7): every metric to 1e-6 relative, counts exact; an image without valid
    truth, one without a selected pixel, negative / zero idepths and the DeMoN range included."""
    from multi_view_stereonet_amd import metrics
    fix = load_golden("g7_depth_metrics.npz")
    g = torch.Generator().manual_seed(3)
    for (lo, hi), shape in (((0.0, 1e3), (5, 1, 96, 160)), ((0.5, 10.0), (3, 1, 37, 53)), ((0.0, 1e3), (2, 1, 256, 512))):
        B = shape[0]
        truth = torch.rand(shape, generator=g) * 12.0
        truth[truth < 1.0] = 0.0                                  # holes in the ground truth
        idepth = 1.0 / (truth.clamp_min(0.3) * (0.7 + 0.6 * torch.rand(shape, generator=g)))
        idepth[:, :, ::7, ::5] = 0.0                              # non-positive estimates stay as they are
        idepth[:, :, 1::9, 2::4] = -0.25
        if B >= 3:
            truth[1] = 0.0                                        # no valid truth at all -> n_truth = 0 (skipped)
            idepth[2] = -1.0                                      # truth but no valid estimate -> NaN metrics
        baseline = 0.5 + torch.rand(B, generator=g)
        got = metrics.depth_metric_rows(idepth.to(DEV), truth.to(DEV), baseline.to(DEV), lo, hi).cpu()
        want = metrics.depth_metric_rows(idepth, truth, baseline, lo, hi)
        assert torch.equal(got[:, :2], want[:, :2]), (got[:, :2], want[:, :2])
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        ok = ~torch.isnan(want)
        rel = ((got[ok] - want[ok]).abs() / want[ok].abs().clamp_min(1e-12)).max()
        assert float(rel) < 1e-6, float(rel)
    # the reference's own numbers (fixture of test.py's function): already-masked positive depths, baseline 1
    t = torch.from_numpy(fix["depth_true"]).reshape(1, 1, 1, -1).float()
    e = torch.from_numpy(fix["depth_est"]).reshape(1, 1, 1, -1).float()
    row = metrics.depth_metric_rows((1.0 / e).to(DEV), t.to(DEV), torch.ones(1, device=DEV), 0.0, 1e9).cpu()[0]
    for i, k in enumerate(metrics.METRIC_KEYS):
        assert abs(float(row[2 + i]) - float(fix[k])) <= 2e-6 * max(1.0, abs(float(fix[k]))), k


def test_evaluate_with_prefetcher_matches_blocking_loop():
    """metrics.evaluate on a GPU (Prefetcher + device metric rows + deferred checks, no per-image host copy) gives the
    averages of the reference-style loop: blocking unpack, synchronised forward, numpy metrics image by image."""
    from multi_view_stereonet_amd import metrics
    net = net_for("gta_sfm_150epochs")
    params = {"num_idepth_samples": 8, "cost_volume_filter": True, "refiners": [True] * 5}
    batches = []
    for i in range(5):
        b = synthetic.make_batch(64, 128, 2, batch=2 if i % 2 else 1, seed=40 + i, smooth=True)
        n = b["left_image"].shape[0]
        depth = 2.0 + 6.0 * torch.rand(n, 1, 64, 128, generator=torch.Generator().manual_seed(i))
        if i == 2:
            depth[0] = 0.0                      # an image without ground truth: skipped by both loops
        b["left_depthmap_true"] = depth
        b["right_depthmap_true"] = [depth.clone(), depth.clone()]
        batches.append(b)
    got = metrics.evaluate(net, batches, params, "gta_sfm", torch.device(DEV))
    rows = []
    for b in batches:
        inputs = snu.multi_view_unpack_batch(b, torch.device(DEV), 5)
        out = snu.multi_view_forward(net, inputs, params)
        est = metrics.idepth_to_depth(out["left_idepthmap_pyr"][0], inputs["baseline"]).cpu().numpy()
        for k in range(est.shape[0]):
            r = metrics.image_metric_row(b["left_depthmap_true"][k, 0].numpy(), est[k, 0], 0.0, 1e3)
            if r is not None:
                rows.append(r)
    want = metrics.compute_avg_metrics(rows)
    assert got["num_samples"] == want["num_samples"] == 6
    for k in metrics.METRIC_KEYS:
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got[k], want[k])
    assert got["runtime_ms"] > 0.0 and got["batch_runtime_ms"] >= got["runtime_ms"]


def test_two_view_bidirectional_golden():
    """Two-view path with the right-view estimate (views swapped, pose inverted) on the HIP network."""
    from test_oracle_golden import _two_view_batch
    fix = load_golden("g8_two_view_128x64_d12.npz")
    inputs = snu.unpack_batch(_two_view_batch(), torch.device(DEV), 5)
    out = snu.forward(net_for("gta_sfm_150epochs"), inputs, {"num_idepth_samples": 12, "estimate_right_idepthmap": True})
    for key, lvl in (("left", 0), ("right", 0), ("left", 4), ("right", 4)):
        mean_rel, max_rel = rel_err(out[f"{key}_idepthmap_pyr"][lvl].cpu(), fix[f"{key}_idepth_{lvl}"])
        assert mean_rel < 2e-4 and max_rel < 1e-3, (key, lvl, mean_rel, max_rel)
    # the occlusion pyramids and the consistency loss the reference derives from such a pair (:711-753), HIP vs oracle
    from multi_view_stereonet_amd import losses
    occ = snu.occlusion_masks(inputs, out)
    cpu = lambda x: x.cpu()   # noqa: E731
    for lvl in range(5):
        ref = oracle.get_occlusion_mask(cpu(inputs["K_pyr"][lvl]), cpu(inputs["T_right_in_left"]),
                                        cpu(out["left_idepthmap_pyr"][lvl]), cpu(out["right_idepthmap_pyr"][lvl]))
        assert int((occ["left_occlusion_mask_pyr"][lvl].cpu() != ref).sum()) <= max(2, ref.numel() // 2000)
    loss = losses.left_right_idepthmap_consistency_losses(
        inputs["T_right_in_left"], inputs["T_left_in_right"], inputs["K_pyr"], out["left_idepthmap_pyr"],
        occ["left_occlusion_mask_pyr"], out["right_idepthmap_pyr"], occ["right_occlusion_mask_pyr"])
    assert loss.is_cuda and loss.dim() == 0    # (NaN when every pixel is occluded, as in the reference)


def test_torchscript_archive_matches_eager(tmp_path):
    """The literal test.py flow (test.py:308-314, multi_view_stereonet_utils.py:647-654): torch.jit.load of a
    stereo_network.pt, .to(device), .eval(), seven positional arguments -- same tensors as the eager module."""
    from multi_view_stereonet_amd import torchscript as ts
    sd = load_weights("gta_sfm_150epochs")
    ts.export_archive(sd, str(tmp_path / "stereo_network.pt"))
    stereo_network = torch.jit.load(str(tmp_path / "stereo_network.pt"))
    stereo_network = stereo_network.to(torch.device(DEV))
    stereo_network.eval()
    fix = load_golden("g1_gta_128x64_d16_s1.npz")
    batch, D = batch_from_meta(fix["meta"])
    inputs = snu.multi_view_unpack_batch(batch, torch.device(DEV), stereo_network.num_levels)
    params = {"num_idepth_samples": D, "cost_volume_filter": True, "refiners": [True] * 5}
    out = snu.multi_view_forward(stereo_network, inputs, params)
    eager = snu.multi_view_forward(net_for("gta_sfm_150epochs"), inputs, params)
    for key in ("left_idepthmap_pyr", "left_idepthmap_raw_pyr", "left_idepthmap_mask_pyr"):
        for a, b in zip(out[key], eager[key]):
            assert a.dtype == b.dtype and torch.equal(a, b)
    mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][0].cpu(), fix["idepth_0"])
    assert mean_rel < 2e-4 and max_rel < 1e-3
    # a second call reuses the cached engine (same parameter storage), a flag variant goes through the same operator
    again = snu.multi_view_forward(stereo_network, inputs, dict(params, refiners=[False, True, True, True, True]))
    assert torch.equal(again["left_idepthmap_pyr"][1], out["left_idepthmap_pyr"][1])
    # weights changed in place on the loaded archive (load_state_dict): the operator must not serve stale packed copies
    stereo_network.load_state_dict({k: v.to(DEV) for k, v in load_weights("demon_45epochs").items()})
    other = snu.multi_view_forward(stereo_network, inputs, params)
    want = snu.multi_view_forward(net_for("demon_45epochs"), inputs, params)
    assert torch.equal(other["left_idepthmap_pyr"][0], want["left_idepthmap_pyr"][0])
    assert not torch.equal(other["left_idepthmap_pyr"][0], out["left_idepthmap_pyr"][0])


def test_two_view_consistency_ops_golden():
    """SURVEY 8f rank 4, loss side: get_occlusion_mask and left_right_idepthmap_consistency_losses
    (multi_view_stereonet/losses.py:42-160) on the HIP path against the reference's own outputs (g9)."""
    from test_oracle_golden import _consistency_inputs
    from multi_view_stereonet_amd import losses
    fix = load_golden("g9_two_view_consistency.npz")
    T, Ti, Ks, L, R = _consistency_inputs(fix, DEV)
    lo, ro = [], []
    for lvl in range(5):
        lo.append(losses.get_occlusion_mask(Ks[lvl], T, L[lvl], None, R[lvl], None))
        ro.append(losses.get_occlusion_mask(Ks[lvl], Ti, R[lvl], None, L[lvl], None))
        bad = int((lo[-1].cpu() != t(fix[f"left_occlusion_{lvl}"])).sum()) + \
            int((ro[-1].cpu() != t(fix[f"right_occlusion_{lvl}"])).sum())
        print(f"level {lvl}: occlusion-mask mismatches {bad} of {2 * lo[-1].numel()}")
        assert lo[-1].dtype == torch.bool and bad <= max(2, lo[-1].numel() // 2000)   # pixels on the threshold
    for lvl in (2, 4):
        uv, idp, inv = losses.idepthmap_projector(Ks[lvl], T, L[lvl])
        close(uv, fix[f"proj_uv_{lvl}"], rtol=1e-5, atol=2e-6)
        close(idp, fix[f"proj_idepth_{lvl}"], rtol=1e-5, atol=1e-7)
        assert int((inv.cpu() != t(fix[f"proj_invalid_{lvl}"])).sum()) <= 1
    # the loss with the REFERENCE's masks (isolates the loss kernels), then end to end with the HIP masks
    ref_lo = [t(fix[f"left_occlusion_{lvl}"]).to(DEV) for lvl in range(5)]
    ref_ro = [t(fix[f"right_occlusion_{lvl}"]).to(DEV) for lvl in range(5)]
    loss = losses.left_right_idepthmap_consistency_losses(T, Ti, Ks, L, ref_lo, R, ref_ro)
    assert abs(float(loss) - float(fix["left_right_loss"])) < 2e-6 * 10
    for lvl in range(5):
        pick = lambda pyr: [x if i == lvl else None for i, x in enumerate(pyr)]   # noqa: E731
        one = losses.left_right_idepthmap_consistency_losses(T, Ti, Ks, pick(L), ref_lo, pick(R), ref_ro)
        assert abs(float(one) - float(fix["left_right_loss_per_level"][lvl])) < 1e-6
    loss2 = losses.left_right_idepthmap_consistency_losses(T, Ti, Ks, L, lo, R, ro)
    assert abs(float(loss2) - float(fix["left_right_loss"])) < 1e-3 * float(fix["left_right_loss"])
    with pytest.raises(RuntimeError):
        losses.get_occlusion_mask(Ks[0].cpu(), T.cpu(), L[0].cpu(), None, R[0].cpu(), None)     # no CPU path


def test_forward_plan_replay_is_bit_identical():
    """Small batches run from a recorded plan (ForwardPlan: the library calls of the first forward of a shape, replayed
    with their recorded arguments).  The replay must equal the eager forward bit for bit on NEW input values, its
    outputs must survive the next forward, and a different shape must get its own plan."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    D = 16

    def inputs(seed, rows=64, cols=128, S=2, B=1):
        return to_dev(snu.multi_view_unpack_batch(synthetic.make_batch(rows, cols, S, batch=B, seed=seed, smooth=True),
                                                  torch.device("cpu"), 5))

    def run(inp):
        return net(*inp, D, True, [True] * 5)

    def same(a, b):
        return all((x is None and y is None) or torch.equal(x, y) for k in a for x, y in zip(a[k], b[k]))

    old = net.options.plan_max_chains
    try:
        net.options.plan_max_chains = 0
        eager = [run(inputs(s)) for s in (1, 2, 3)]
        eager_other = run(inputs(4, rows=96, cols=160, S=1, B=2))
        net.options.plan_max_chains = 16
        before = eng.replays
        first = run(inputs(1))                 # records
        second = run(inputs(2))                # replays with new values
        third = run(inputs(3))
        assert eng.replays - before == 2
        assert same(first, eager[0]) and same(second, eager[1]) and same(third, eager[2])
        assert same(second, eager[1])          # (not clobbered by the third forward)
        other = run(inputs(4, rows=96, cols=160, S=1, B=2))     # another shape: its own plan
        other2 = run(inputs(4, rows=96, cols=160, S=1, B=2))
        assert same(other, eager_other) and same(other2, eager_other)
        assert same(run(inputs(2)), eager[1])  # and back (by now the call list runs as a hipGraph)
        assert same(run(inputs(3)), eager[2])
        assert any(p is not None and p.graph for p in eng.plans.values())
        # a caller that captures the forward into its own graph (GraphedForward) while a plan with a graph exists
        from multi_view_stereonet_amd.graphed import GraphedForward
        gf = GraphedForward(net, *inputs(2), D)
        assert same(gf(*inputs(3)), eager[2])
    finally:
        net.options.plan_max_chains = old


def test_graph_replay_matches_eager():
    """hipGraph capture of the whole forward: replay on new inputs equals the eager launch sequence."""
    from multi_view_stereonet_amd.graphed import GraphedForward
    net = net_for("gta_sfm_150epochs")
    inp_a = snu.multi_view_unpack_batch(synthetic.make_batch(128, 256, 2, batch=2, seed=51), torch.device(DEV), 5)
    inp_b = snu.multi_view_unpack_batch(synthetic.make_batch(128, 256, 2, batch=2, seed=52, smooth=True),
                                        torch.device(DEV), 5)
    pack = lambda i: (i["left_image_pyr"], i["K_pyr"], i["T_right_in_left"], i["right_image_pyr"])
    g = GraphedForward(net, *pack(inp_a), 16)
    eager_b = net(*pack(inp_b), 16, True, [True] * 5)
    out_b = g(*pack(inp_b))
    assert torch.equal(out_b["left_idepthmap_pyr"][0], eager_b["left_idepthmap_pyr"][0])
    assert torch.equal(out_b["left_idepthmap_mask_pyr"][0], eager_b["left_idepthmap_mask_pyr"][0])
    eager_a = net(*pack(inp_a), 16, True, [True] * 5)
    out_a = g(*pack(inp_a))
    assert torch.equal(out_a["left_idepthmap_pyr"][0], eager_a["left_idepthmap_pyr"][0])


def test_forward_vs_oracle_small_batch():
    """Fresh seeds (no fixture): HIP forward vs the oracle run here on the host."""
    wname = "gta_sfm_150epochs"
    w = load_weights(wname)
    batch = synthetic.make_batch(96, 160, 3, batch=2, seed=33, pose_jitter=0.25, smooth=True)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    ref = oracle.forward(w, inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], 20)
    lp, kp, ts, rp = to_dev(inp)
    out = net_for(wname)(lp, kp, ts, rp, 20, True, [True] * 5)
    for lvl in range(5):
        mean_rel, max_rel = rel_err(out["left_idepthmap_pyr"][lvl].cpu(), ref["left_idepthmap_pyr"][lvl])
        assert mean_rel < 2e-4 and max_rel < 1e-3, (lvl, mean_rel, max_rel)
        diff = int((out["left_idepthmap_mask_pyr"][lvl].cpu() != ref["left_idepthmap_mask_pyr"][lvl]).sum())
        assert diff <= 2 * 4 ** (4 - lvl), (lvl, diff)


@pytest.mark.parametrize("rows,cols,S,D", [(250, 500, 1, 12), (131, 277, 2, 8), (72, 200, 1, 16)])
def test_forward_ragged_sizes_vs_oracle(rows, cols, S, D):
    """Sizes whose pyramid levels are not multiples of 4 / 16 / 32: the layers fall back from the LDS-DMA,
    Winograd and tap-GEMM kernels to the register-staged ones level by level; the result must not care."""
    wname = "gta_sfm_150epochs"
    w = load_weights(wname)
    batch = synthetic.make_batch(rows, cols, S, batch=2, seed=rows + cols, pose_jitter=0.2, smooth=True)
    inp = snu.multi_view_unpack_batch(batch, torch.device("cpu"), 5)
    ref = oracle.forward(w, inp["left_image_pyr"], inp["K_pyr"], inp["T_right_in_left"], inp["right_image_pyr"], D)
    lp, kp, ts, rp = to_dev(inp)
    out = net_for(wname)(lp, kp, ts, rp, D, True, [True] * 5)
    for lvl in range(5):
        got = out["left_idepthmap_pyr"][lvl].cpu()
        assert got.shape == ref["left_idepthmap_pyr"][lvl].shape
        mean_rel, max_rel = rel_err(got, ref["left_idepthmap_pyr"][lvl])
        assert mean_rel < 2e-4 and max_rel < 1e-3, (lvl, mean_rel, max_rel)


def _gn_stats(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randn(n, 4, generator=g) * 0.1, 1.0 + torch.rand(n, 4, generator=g)], -1).contiguous().to(DEV)


@pytest.mark.parametrize("block,mode1,add2,n,jn,rows,cols,expect", [
    (0, False, False, 2, 2, 32, 64, 1),      # dilation 1, plain residual pass, job = the layer's own size
    (1, False, True, 2, 2, 32, 64, 1),       # dilation 2, the two-raw-tensor form (a refiner's first block)
    (3, False, False, 3, 2, 48, 64, 1),      # dilation 8 (one k-step per step), job smaller than the capacity
    (0, True, True, 2, 1, 32, 64, 1),        # carrier applies the previous layer's LReLU(GN(.)) on load
    (4, False, False, 2, 2, 40, 72, 1),      # partial tiles (40 x 72 pixels = 5.6 units of 512: 2880 % 512 != 0 -> own launch)
    (2, False, False, 2, 2, 32, 64, 1),      # dilation 4: carried by the one-k-step form of the layer
    (0, False, False, 1, 2, 32, 64, 0),      # job larger than the carrying layer
    (5, False, False, 5, 5, 16, 32, 1),      # 512-pixel planes: one unit pair per plane
])
def test_conv_forward_carry_is_bit_identical(block, mode1, add2, n, jn, rows, cols, expect):
    """mvsn_conv_forward_carry = mvsn_conv_forward + the stand-alone pass, bit for bit, whether the job travels
    inside the convolution's launch or not (`carried` says which)."""
    eng = net_for("gta_sfm_150epochs").engine()
    lib = eng.lib
    conv, norm = eng.refiners[0]["res"][block]
    norm0 = eng.refiners[0]["bn0"]
    g = torch.Generator().manual_seed(100 + block)
    x = torch.randn(n, 32, rows, cols, generator=g).to(DEV)
    jr = torch.randn(jn, 32, rows, cols, generator=g).to(DEV)
    jres = torch.randn(jn, 32, rows, cols, generator=g).to(DEV)
    st, st0, ist = _gn_stats(jn, 1), _gn_stats(jn, 2), _gn_stats(n, 3)
    if (rows * cols) % 512 != 0:
        expect = 0
    # reference: the two calls
    want_out, want_st = eng.conv(conv, x, in_stats=ist if mode1 else None, in_norm=norm0 if mode1 else None, want_stats=True)
    if add2:
        want_job = eng.gn_lrelu_add2(jr, st, norm, jres, st0, norm0)
    else:
        want_job = eng.gn_lrelu(jr, st, norm, residual=jres)
    from multi_view_stereonet_amd.multi_view_stereonet import _Job
    jr2 = jr.clone()
    job = _Job(jr2, st, norm, jres, st0 if add2 else None, norm0 if add2 else None)     # in place, as the towers use it
    before = eng.carried_jobs
    got_out, got_st = eng.conv(conv, x, in_stats=ist if mode1 else None, in_norm=norm0 if mode1 else None,
                               want_stats=True, carry=job)
    torch.cuda.synchronize()
    assert eng.carried_jobs - before == expect
    assert torch.equal(got_out, want_out) and torch.equal(got_st, want_st)
    assert torch.equal(jr2, want_job)


@pytest.mark.parametrize("mode1,add2,n,jn,rows,cols,reverse", [
    (False, False, 2, 2, 32, 64, 0), (False, False, 2, 2, 32, 64, 1),      # both walk directions
    (True, True, 2, 2, 32, 64, 0), (True, True, 3, 3, 32, 64, 1),          # input transform + two-raw-tensor pass
    (False, False, 5, 5, 16, 32, 0),                                        # one tile per image, 5 images on 256 CUs
    (False, False, 3, 2, 48, 96, 1),                                        # ragged tile counts: 3 x 3 tiles, job smaller
    (False, True, 7, 7, 64, 32, 0),                                         # one tile column, odd batch
    (False, False, 1, 1, 256, 512, 1),                                      # a level-0 plane: 128 tiles of one image
    (False, False, 9, 4, 80, 160, 0),                                       # 5 x 5 tiles x 9 images: > 256 work items
])
def test_carried_dilation1_launches_stress_bit_identical(mode1, add2, n, jn, rows, cols, reverse):
    """The dilation-1 carrying launches issue their LDS-DMA as inline assembly (invisible to the compiler's waitcnt pass:
    the ring and the carried loads are guarded by hand-counted s_waitcnt vmcnt(N), mvsn_conv_wino.hip wn_dma16 /
    wait_landed).  A wrong count shows up as a stale LDS read, i.e. as a rare wrong word -- so: 1000 launches per shape
    (nine shapes: both walk directions, ragged tile counts, more work items than CUs, both pass forms, the input
    transform), every output word and every word of the carried pass compared with the two-call reference each time."""
    eng = net_for("gta_sfm_150epochs").engine()
    conv, norm = eng.refiners[0]["res"][0]          # dilation 1
    norm0 = eng.refiners[0]["bn0"]
    g = torch.Generator().manual_seed(rows * 7 + cols + n)
    x = torch.randn(n, 32, rows, cols, generator=g).to(DEV)
    jr = torch.randn(jn, 32, rows, cols, generator=g).to(DEV)
    jres = torch.randn(jn, 32, rows, cols, generator=g).to(DEV)
    st, st0, ist = _gn_stats(jn, 1), _gn_stats(jn, 2), _gn_stats(n, 3)
    kw = dict(in_stats=ist if mode1 else None, in_norm=norm0 if mode1 else None, want_stats=True)
    want_out, want_st = eng.conv(conv, x, **kw)
    want_job = eng.gn_lrelu_add2(jr, st, norm, jres, st0, norm0) if add2 else eng.gn_lrelu(jr, st, norm, residual=jres)
    from multi_view_stereonet_amd.multi_view_stereonet import _Job
    jr2 = torch.empty_like(jr)
    out = torch.empty_like(want_out)
    before = eng.carried_jobs
    bad = 0
    iters = 1000
    for it in range(iters):
        jr2.copy_(jr)
        job = _Job(jr2, st, norm, jres, st0 if add2 else None, norm0 if add2 else None)
        job.job.reverse = reverse if it % 3 else 1 - reverse       # mostly one direction, every third launch the other
        _, got_st = eng.conv(conv, x, carry=job, out=out, **kw)
        ok = torch.equal(out, want_out) & torch.equal(got_st, want_st) & torch.equal(jr2, want_job)   # (syncs)
        bad += 0 if ok else 1
    assert eng.carried_jobs - before == iters, "the pass did not travel inside the launch: nothing was stressed"
    assert bad == 0, f"{bad} of {iters} launches differed from the two-call reference"


def test_sliced_refiner_tower_is_bit_identical():
    """The two-slice software-pipelined refiner tower (carried passes) against the one-batch tower: same kernels per
    sample, so the same bits -- at an even and an odd batch (odd: one slice's jobs exceed the other's capacity)."""
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    g = torch.Generator().manual_seed(5)
    old = (net.options.carry_passes, net.options.carry_min_bytes)
    try:
        for n, lvl, rows, cols in ((4, 0, 64, 128), (3, 1, 32, 64), (2, 2, 48, 64)):
            img = torch.rand(n, 3, rows, cols, generator=g).to(DEV)
            guide = [img] if lvl == 0 else [img, torch.randn(n, 32, rows, cols, generator=g).to(DEV)]
            prior = (0.1 + torch.rand(n, 1, rows, cols, generator=g)).to(DEV)
            fx = (100.0 + 50.0 * torch.rand(n, generator=g)).to(DEV)
            net.options.carry_passes = False
            want = eng.idepth_refiner(lvl, guide, prior, fx)
            net.options.carry_passes, net.options.carry_min_bytes = True, 0
            before = eng.carried_jobs
            got = eng.idepth_refiner(lvl, guide, prior, fx)
            torch.cuda.synchronize()
            assert eng.carried_jobs > before, "no pass travelled inside a convolution launch"
            assert torch.equal(got, want), (n, lvl)
    finally:
        net.options.carry_passes, net.options.carry_min_bytes = old


@pytest.mark.parametrize("mode1,n,jn,depth,rows,cols,expect", [(False, 2, 2, 8, 16, 32, 1), (True, 2, 2, 5, 16, 32, 1),
                                                                (False, 3, 2, 4, 12, 24, 0),     # 4*12*24 % 256 != 0
                                                                (False, 2, 2, 6, 32, 64, 1),
                                                                (False, 2, 2, 32, 30, 40, 1), (True, 3, 3, 32, 30, 40, 1),   # WIDE tiles
                                                                (True, 2, 2, 16, 30, 40, 1), (False, 1, 1, 64, 23, 36, 1)])  # (rolling strips)
def test_conv3d_forward_carry_is_bit_identical(mode1, n, jn, depth, rows, cols, expect):
    """The volume form (3x3x3 regulariser layer) carrying an in-place LReLU(GN(.)) pass over another volume."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Job
    eng = net_for("gta_sfm_150epochs").engine()
    conv, norm = eng.vf_convs[1], eng.vf_norms[0]
    g = torch.Generator().manual_seed(31)
    x = torch.randn(n, 32, depth, rows, cols, generator=g).to(DEV)
    jr = torch.randn(jn, 32, depth, rows, cols, generator=g).to(DEV)
    st, ist = _gn_stats(jn, 4), _gn_stats(n, 5)
    kw = dict(in_stats=ist, in_norm=norm) if mode1 else {}
    want_out, want_st = eng.conv(conv, x, want_stats=True, **kw)
    want_job = eng.gn_lrelu(jr, st, norm)
    jr2 = jr.clone()
    before = eng.carried_jobs
    got_out, got_st = eng.conv(conv, x, want_stats=True, carry=_Job(jr2, st, norm), **kw)
    torch.cuda.synchronize()
    assert eng.carried_jobs - before == expect
    assert torch.equal(got_out, want_out) and torch.equal(got_st, want_st) and torch.equal(jr2, want_job)


@pytest.mark.parametrize("mode1,n,depth,rows,cols", [(False, 3, 32, 30, 40), (True, 2, 16, 30, 40), (False, 1, 64, 23, 36)])
def test_rolling_strip_launches_stress_bit_identical(mode1, n, depth, rows, cols):
    """The rolling-strip form of the volume kernel (six patch rows at a time through a sample's planes; one descriptor over two
    planes per step, per-item lane geometry) issues its LDS-DMA by hand-counted waits when it carries a pass: 300 launches
    per shape, plain and carrying alternately, both walk directions, every output word, every GroupNorm record-derived
    statistic and every word of the carried pass compared with the first launch / the two-call reference each time."""
    from multi_view_stereonet_amd.multi_view_stereonet import _Job
    eng = net_for("gta_sfm_150epochs").engine()
    conv, norm = eng.vf_convs[1], eng.vf_norms[0]
    g = torch.Generator().manual_seed(rows * 3 + depth)
    x = torch.randn(n, 32, depth, rows, cols, generator=g).to(DEV)
    jr = torch.randn(n, 32, depth, rows, cols, generator=g).to(DEV)
    st, ist = _gn_stats(n, 4), _gn_stats(n, 5)
    kw = dict(in_stats=ist, in_norm=norm) if mode1 else {}
    want_out, want_st = eng.conv(conv, x, want_stats=True, **kw)
    want_job = eng.gn_lrelu(jr, st, norm)
    can_carry = (depth * rows * cols) % 256 == 0
    jr2 = torch.empty_like(jr)
    bad, carried0 = 0, eng.carried_jobs
    for it in range(300):
        if it % 2 and can_carry:
            jr2.copy_(jr)
            job = _Job(jr2, st, norm)
            job.job.reverse = (it // 2) % 2
            out, got_st = eng.conv(conv, x, want_stats=True, carry=job, **kw)
            ok = torch.equal(out, want_out) & torch.equal(got_st, want_st) & torch.equal(jr2, want_job)
        else:
            out, got_st = eng.conv(conv, x, want_stats=True, **kw)
            ok = torch.equal(out, want_out) & torch.equal(got_st, want_st)
        bad += 0 if ok else 1
    assert bad == 0, f"{bad} of 300 launches deviated"
    assert (eng.carried_jobs - carried0 == 150) == can_carry


def test_sliced_regulariser_is_bit_identical():
    net = net_for("gta_sfm_150epochs")
    eng = net.engine()
    g = torch.Generator().manual_seed(6)
    old = (net.options.carry_passes, net.options.carry_min_bytes, net.options.carry_volume_passes)
    try:
        net.options.carry_volume_passes = True
        for n, depth in ((4, 8), (3, 16)):
            cost = torch.rand(n, 32, depth, 16, 32, generator=g).to(DEV)
            net.options.carry_passes = False
            want = eng.cost_volume_filter(cost)
            net.options.carry_passes, net.options.carry_min_bytes = True, 0
            before = eng.carried_jobs
            got = eng.cost_volume_filter(cost)
            torch.cuda.synchronize()
            assert eng.carried_jobs > before and torch.equal(got, want), (n, depth)
    finally:
        net.options.carry_passes, net.options.carry_min_bytes, net.options.carry_volume_passes = old


def test_forward_option_variants_agree():
    """The whole forward on a batch-2 golden case under the engine's scheduling options: slicing / carrying (at every
    level, with and without alternation, the regulariser too) must not change a bit, the chain forms agree within
    rounding."""
    fix = load_golden("g1b_gta_96x80_d8_s2_b2.npz")
    net = net_for("gta_sfm_150epochs")
    ref = [t.clone() for t in _forward(net, fix)["left_idepthmap_pyr"]]
    for opts, exact in ((dict(carry_min_bytes=0), True), (dict(carry_min_bytes=0, carry_alternate=False), True),
                        (dict(carry_passes=False), True), (dict(carry_min_bytes=0, carry_volume_passes=True), True),
                        (dict(chain_form="stepwise"), False), (dict(chain_form="direct"), False)):
        old = {k: getattr(net.options, k) for k in opts}
        for k, v in opts.items():
            setattr(net.options, k, v)
        try:
            out = _forward(net, fix)["left_idepthmap_pyr"]
        finally:
            for k, v in old.items():
                setattr(net.options, k, v)
        for a, b in zip(out, ref):
            if exact:
                assert torch.equal(a, b), opts
            else:
                assert float((a - b).abs().max() / b.abs().max()) < 1e-4, opts
