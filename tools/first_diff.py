#!/usr/bin/env python3
"""Which intermediate deviates first when a planned forward goes wrong?  Every tensor the recorded plan keeps alive is
snapshotted after a good replay of each input set; after a replay whose outputs differ from the eager reference the kept
tensors are diffed in allocation (= execution) order and the first few that differ are printed with the recorded calls
that touch them.  (Round 3: this is what pointed at the extractor tower, HISTORY 11.5.)
    python tools/first_diff.py <config> <batch> <reps> [option=value ...]"""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
dev = torch.device("cuda")
name, b, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = dict(kv.split("=") for kv in sys.argv[4:])
cfg = bench.CONFIGS[name]
net = MultiViewStereoNet(); net.load_state_dict(load_weights(cfg["weights"]), strict=True); net = net.to(dev).eval()
for k, v in opts.items():
    setattr(net.options, k, int(v) if v.lstrip("-").isdigit() else v)
K = 4
inps = [bench.config_inputs(cfg, b, r, dev)[1] for r in range(K)]
def flat(o): return list(o["left_idepthmap_pyr"]) + list(o["left_idepthmap_raw_pyr"]) + list(o["left_idepthmap_mask_pyr"])
keep = net.options.plan_max_chains
net.options.plan_max_chains = 0
refs = [[t.clone() for t in flat(bench.run_forward(net, x, cfg["D"]))] for x in inps]
net.options.plan_max_chains = keep
run = lambda j: flat(bench.run_forward(net, inps[j], cfg["D"]))
run(0); run(0)
plan = [p for p in net.engine().plans.values() if p is not None][0]
print("plan: calls", len(plan.calls), "kept tensors", len(plan.keep), "MB", sum(t.numel() * t.element_size() for t in plan.keep) / 1e6)
snaps = []
for j in range(K):
    for attempt in range(5):
        got = run(j)
        if all(torch.equal(a, r) for a, r in zip(got, refs[j])):
            break
    else:
        raise SystemExit("no good replay for set %d" % j)
    torch.cuda.synchronize()
    snaps.append([t.clone() for t in plan.keep])
def owner(ptr, nbytes):
    names = []
    for fn, args, nm in plan.calls:
        for a in args:
            v = a.value if isinstance(a, ctypes.c_void_p) else a
            if isinstance(v, int) and ptr <= v < ptr + nbytes:
                names.append(nm); break
    return names
found = 0
for i in range(reps):
    j = (i * 7 + i // 5) % K
    got = run(j)
    if not all(torch.equal(a, r) for a, r in zip(got, refs[j])):
        torch.cuda.synchronize()
        found += 1
        print("=== wrong forward at iter", i, "set", j)
        shown = 0
        for idx, (t, s) in enumerate(zip(plan.keep, snaps[j])):
            if not torch.equal(t, s):
                d = (t.float() - s.float()).abs()
                nz = (d > 0).nonzero()
                print("  keep[%d] shape %s dtype %s: %d of %d differ, max %.3e, first idx %s last idx %s; calls touching it: %s"
                      % (idx, tuple(t.shape), t.dtype, int((d > 0).sum()), d.numel(), float(d.max()), nz[0].tolist(), nz[-1].tolist(),
                         owner(t.data_ptr(), t.numel() * t.element_size())[:4]))
                shown += 1
                if shown >= 8: break
        if found >= 3: break
print("done, wrong forwards:", found)
