#!/usr/bin/env python3
"""Soak of the small-batch path (banded chain hand-offs, recorded plans / hipGraph replay, records hand-over).
For each (config, batch): K = 4 DIFFERENT input sets, each with an eager reference (plans off); then `reps` planned
forwards cycling through the sets in an irregular order -- every output must equal its set's reference bit for bit (a
replay that reads anything left over from the previous replay shows up, because the previous replay ran on other inputs),
the banded chain's status word must stay 0 and every depth map finite.  Prints one JSON line.
    python tools/soak.py [reps] [option=value ...] [cases=config3:1,config5:4] [graphed]     (default 400 per case)
`graphed`: every case runs through GraphedForward (the whole forward, ATen copies included, captured by
torch.cuda.graph) instead of the module's own recorded plan."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multi_view_stereonet_amd import MultiViewStereoNet, _native
from multi_view_stereonet_amd.graphed import GraphedForward
from multi_view_stereonet_amd.weights import load_weights
torch.set_grad_enabled(False)
args = sys.argv[1:]
reps = int(args[0]) if args and args[0].isdigit() else 400
opts = dict(kv.split("=") for kv in args if "=" in kv)
graphed = "graphed" in args
dev = torch.device("cuda")
K = 4
cases = [("headline", 1), ("headline", 2), ("headline", 8), ("config3", 1), ("config3", 3), ("config4", 1), ("config4", 16),
         ("config5", 1), ("config5", 4), ("headline", 128)]
LARGE = 8                                                    # a case of more than 16 chains runs reps / LARGE forwards
if "cases" in opts:                                          # cases=config3:1,config3:3
    cases = [(c.split(":")[0], int(c.split(":")[1])) for c in opts.pop("cases").split(",")]
FORMS = {_native.CHAIN_BANDED: "banded", _native.CHAIN_WINOGRAD: "winograd", _native.CHAIN_STEPWISE: "stepwise",
         _native.CHAIN_DIRECT: "direct"}


def flat(o):
    return list(o["left_idepthmap_pyr"]) + list(o["left_idepthmap_raw_pyr"]) + list(o["left_idepthmap_mask_pyr"])


res, t_all = [], time.perf_counter()
for name, b in cases:
    cfg = bench.CONFIGS[name]
    net = MultiViewStereoNet(); net.load_state_dict(load_weights(cfg["weights"]), strict=True); net = net.to(dev).eval()
    inps = [bench.config_inputs(cfg, b, r, dev)[1] for r in range(K)]
    # arithmetic options (conv_precision, winograd, towers ...) apply to the references too; the replay options
    # (plan_*, chain_form: same arithmetic, other mapping) only to the soaked forwards
    late = {k: v for k, v in opts.items() if k.startswith("plan_") or k == "chain_form"}
    for k, v in opts.items():
        if k not in late:
            setattr(net.options, k, int(v) if v.lstrip("-").isdigit() else v)
    keep = net.options.plan_max_chains
    net.options.plan_max_chains = 0
    refs = [[t.clone() for t in flat(bench.run_forward(net, x, cfg["D"]))] for x in inps]
    net.options.plan_max_chains = keep
    for k, v in late.items():
        setattr(net.options, k, int(v) if v.lstrip("-").isdigit() else v)
    bad, worst, repaired = 0, 0.0, 0
    run = lambda x: bench.run_forward(net, x, cfg["D"])
    if graphed:
        x0 = inps[0]
        gf = GraphedForward(net, x0["left_image_pyr"], x0["K_pyr"], x0["T_right_in_left"], x0["right_image_pyr"], cfg["D"])
        run = lambda x: gf(x["left_image_pyr"], x["K_pyr"], x["T_right_in_left"], x["right_image_pyr"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_rep = reps if b * cfg["S"] <= 16 else max(1, reps // LARGE)
    for i in range(n_rep):
        j = (i * 7 + i // 5) % K
        got = flat(run(inps[j]))
        same = all(torch.equal(a, r) for a, r in zip(got, refs[j]))
        if not same:
            bad += 1
            worst = max(worst, max(float((a.float() - r.float()).abs().max()) for a, r in zip(got, refs[j])))
        if i % 50 == 49 or not same:
            repaired += net.check_device_status()          # synchronises; counts forwards whose banded chain was repaired
    torch.cuda.synchronize()
    repaired += net.check_device_status()
    eng = net.engine()
    res.append({"repaired_forwards": repaired, "config": name, "batch": b, "chains": b * cfg["S"], "chain_form": FORMS.get(eng.last_chain_form),
                "forwards": n_rep, "graph_replays": eng.replays, "wrong_forwards": bad, "worst_abs_diff": worst,
                "ms_per_forward": round((time.perf_counter() - t0) / n_rep * 1e3, 3)})
    del net, inps, refs
    torch.cuda.empty_cache()
print(json.dumps({"soak": res, "options": opts, "graphed_forward": graphed, "input_sets": K, "seconds": round(time.perf_counter() - t_all, 1),
                  "all_bit_identical": all(r["wrong_forwards"] == 0 for r in res)}))
