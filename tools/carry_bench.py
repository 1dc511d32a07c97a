#!/usr/bin/env python3
"""A refiner-block convolution carrying another slice's normalise/activate/add pass (mvsn_conv_forward_carry):
time of conv alone, pass alone, both in sequence, and the carrying launch; results compared bit for bit."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.weights import load_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 512)
net = MultiViewStereoNet(); net.load_state_dict(load_weights("gta_sfm_150epochs")); net = net.cuda().eval()
eng = net.engine(); lib = eng.lib
torch.manual_seed(0)
x = torch.randn(n, 32, rows, cols, device="cuda")
jr = torch.randn(n, 32, rows, cols, device="cuda")
jres = torch.randn(n, 32, rows, cols, device="cuda")
jr0 = torch.randn(n, 32, rows, cols, device="cuda")
stats = torch.stack([torch.randn(n, 4, device="cuda") * 0.1, 1.0 + torch.rand(n, 4, device="cuda")], -1).contiguous()
stats0 = torch.stack([torch.randn(n, 4, device="cuda") * 0.1, 1.0 + torch.rand(n, 4, device="cuda")], -1).contiguous()
in_stats = torch.stack([torch.randn(n, 4, device="cuda") * 0.1, 1.0 + torch.rand(n, 4, device="cuda")], -1).contiguous()


def timed(fn, it=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


for bi, mode1, add2 in ((0, False, False), (1, False, False), (2, False, False), (3, False, False), (0, True, True), (1, False, True)):
    conv, norm = eng.refiners[0]["res"][bi]
    norm0 = eng.refiners[0]["bn0"]
    d = conv.desc(n, 1, rows, cols, _native.CONV_FP32_WINO)
    tiles = lib.mvsn_conv_num_tiles(ctypes.byref(d))
    out = torch.empty(n, 32, rows, cols, device="cuda"); out2 = torch.empty_like(out)
    part = torch.empty(n, tiles, 4, 3, device="cuda"); part2 = torch.empty_like(part)
    jout = torch.empty_like(jr); jout2 = torch.empty_like(jr)
    P = _native.ptr
    ist = (P(in_stats), P(norm0.gamma), P(norm0.beta)) if mode1 else (None, None, None)

    def run_conv(o=out, p=part):
        _native.check(lib.mvsn_conv_forward(ctypes.byref(d), P(x), P(conv.packed_wino), P(conv.bias), *ist, None, None,
                                            P(o), P(p), _native.stream()), "conv")

    def run_apply(o=jout):
        if add2:
            _native.check(lib.mvsn_groupnorm_lrelu_add2(P(jr), P(stats), P(norm.gamma), P(norm.beta), P(jr0), P(stats0),
                                                        P(norm0.gamma), P(norm0.beta), n, rows * cols, P(o),
                                                        _native.stream()), "add2")
        else:
            _native.check(lib.mvsn_groupnorm_lrelu_apply(P(jr), P(stats), P(norm.gamma), P(norm.beta), P(jres), n,
                                                         rows * cols, P(o), _native.stream()), "apply")

    job = _native.ApplyJob(P(jr), P(stats), P(norm.gamma), P(norm.beta), P(jr0) if add2 else P(jres),
                           P(stats0) if add2 else None, P(norm0.gamma) if add2 else None,
                           P(norm0.beta) if add2 else None, P(jout2), n, int(os.environ.get("CARRY_REV", "0")), rows * cols)
    carried = ctypes.c_int(-1)

    def run_carry():
        _native.check(lib.mvsn_conv_forward_carry(ctypes.byref(d), P(x), P(conv.packed_wino), P(conv.bias), *ist,
                                                  P(out2), P(part2), ctypes.byref(job), ctypes.byref(carried),
                                                  _native.stream()), "carry")

    t_seq = timed(lambda: (run_conv(), run_apply()))
    t_carry = timed(run_carry)
    t_conv, t_apply = timed(run_conv), timed(run_apply)
    t_seq = 0.5 * (t_seq + timed(lambda: (run_conv(), run_apply())))
    t_carry = 0.5 * (t_carry + timed(run_carry))
    ok = torch.equal(out, out2) and torch.equal(part, part2) and torch.equal(jout, jout2)
    print(f"block {bi} dil {conv.dilation} mode1 {int(mode1)} add2 {int(add2)}: conv {t_conv:.3f} apply {t_apply:.3f} "
          f"seq {t_seq:.3f} carry {t_carry:.3f} ms  carried {carried.value}  bit-identical {ok}", flush=True)
