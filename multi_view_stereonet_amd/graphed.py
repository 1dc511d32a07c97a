"""hipGraph replay of the whole forward (static shapes).

A forward is ~330 kernel launches on one stream; at batch 1 the launch gaps are a visible share of
the 8 ms latency, so the sequence can be captured once into a HIP graph and replayed
(torch.cuda.CUDAGraph is the hipGraph wrapper on ROCm; every libmvsn_hip.so call is a plain launch
on the capture stream, and all hipFuncSetAttribute opt-ins happen during the warm-up run).
Measured on MI355X (512x256, D=64, S=2): batch 1 8.42 -> 7.85 ms, batch 8 11.4 -> 11.0 ms, batch 128
unchanged (device-bound).  Since round 3 the module replays small batches from its own recorded plan as one hipGraph
launch (multi_view_stereonet.ForwardPlan); this wrapper remains for callers that want a graph of a large batch.

The captured graph holds ONLY libmvsn_hip.so kernels: the module's recorded plans are switched off while capturing
(their input / output copies are ATen kernels), and the caller's inputs are copied into the static buffers outside the
graph.  That is deliberate -- a captured hipGraph orders the caches between two kernel nodes by the buffers it can see
among their pointer arguments; ATen's kernels carry theirs inside by-value structs, and with such a writer in front of a
library kernel inside one graph about one replay in a thousand read a line of the previous replay's inputs
(tools/soak.py graphed; csrc/mvsn_common.h MVSN_VIS10 has the rule the library's own kernels follow).
"""
from typing import Dict, List

import torch


def _copy_tree(dst, src):
    if isinstance(dst, torch.Tensor):
        dst.copy_(src)
    else:
        for d, s in zip(dst, src):
            _copy_tree(d, s)


class GraphedForward:
    """Capture ``net(*inputs)`` once; ``__call__`` copies new inputs into the captured buffers and
    replays.  Inputs must keep the captured shapes; outputs are the captured tensors (overwritten by
    the next replay)."""

    def __init__(self, net, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, num_idepth_samples: int,
                 do_cost_volume_filter: bool = True, do_refiners=None):
        refs = [True] * 5 if do_refiners is None else list(do_refiners)
        clone = lambda t: t.clone()
        self.static = ([clone(x) for x in left_image_pyr], [clone(x) for x in K_pyr],
                       [clone(x) for x in T_right_in_lefts], [[clone(x) for x in p] for p in right_image_pyrs])
        self.args = (int(num_idepth_samples), bool(do_cost_volume_filter), refs)
        keep = net.options.plan_max_chains
        net.options.plan_max_chains = 0                    # library kernels only inside the graph (see above)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # warm-up: packs weights, opts in LDS sizes
                net(*self.static, *self.args)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.outputs: Dict[str, List[torch.Tensor]] = net(*self.static, *self.args)
        finally:
            net.options.plan_max_chains = keep

    def __call__(self, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs):
        _copy_tree(self.static, (left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs))
        self.graph.replay()
        return self.outputs
