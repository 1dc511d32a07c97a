import os, sys, torch, time
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
K = 4
inps = [bench.config_inputs(cfg, b, r, dev)[1] for r in range(K)]          # K different input sets
def flat(o): return list(o["left_idepthmap_pyr"]) + list(o["left_idepthmap_raw_pyr"]) + list(o["left_idepthmap_mask_pyr"])
names = ["idepth%d" % i for i in range(5)] + ["raw%d" % i for i in range(5)] + ["mask%d" % i for i in range(5)]
net.options.plan_max_chains = 0                                            # eager references
refs = [[t.clone() for t in flat(bench.run_forward(net, x, cfg["D"]))] for x in inps]
net.options.plan_max_chains = 16
for k, v in opts.items():
    setattr(net.options, k, v if not v.lstrip("-").isdigit() else int(v))
bad = 0
for i in range(reps):
    j = (i * 7 + i // 5) % K
    out = bench.run_forward(net, inps[j], cfg["D"])
    for nm, a, r in zip(names, flat(out), refs[j]):
        if not torch.equal(a, r):
            d = (a.float() - r.float()).abs()
            if bad < 12:
                print("iter", i, "set", j, nm, "differs: elems", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), flush=True)
            bad += 1
print(name, b, opts, "reps", reps, "mismatching tensors", bad, "status", net.engine().chain_status(), "replays", net.engine().replays)
