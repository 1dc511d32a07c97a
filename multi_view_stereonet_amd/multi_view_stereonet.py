"""MultiViewStereoNet on MI355X: the reference's module boundary over libmvsn_hip.so.

Drop-in for multi_view_stereonet/multi_view_stereonet.py:494-695 of the reference:

* no-argument constructor, ``num_levels == 5``;
* ``forward(left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, num_idepth_samples,
  do_cost_volume_filter, do_refiners)`` with the same positional meaning (:538-545) and the same
  output dict of three 5-level pyramids (:686-693);
* the same 226-key ``state_dict`` (params.py), so pretrained tensors load with strict=True.

The modules hold parameters only.  All compute is enqueued on the current HIP stream through
the C ABI in include/mvsn_hip.h; PyTorch provides device memory and the stream, nothing else.
There is no CPU path: CPU tensors, or a missing library, raise.

Execution plan of one forward (B reference images, S sources each, N = S*B chains):
  1. mvsn_plane_sweep_setup         poses -> idepth samples, homography families       (a3, a4)
  2. mvsn_homography_warp           full-res source images at plane 0                   (a5)
  3. feature extractor, ONE batch of (1+S)*B images through mvsn_conv_forward           (a2)
  4. mvsn_incremental_cost_volume   the fused chain -> cost volume + mask               (a6-a8)
  5. cost-volume regulariser: five mvsn_conv_forward (3-D), GroupNorm folded into the
     next layer's tile load; mvsn_soft_argmin                                           (a9, a10)
  6. refiner 4 on all N chains, mvsn_fuse_sources                                       (a11, a13)
  7. levels 3..0: mvsn_upsample_bilinear / mvsn_upsample_mask + refiner                 (a12, a11)
"""
import ctypes
import threading
import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _native
from .params import REFINER_DILATIONS, build_parameter_tree

ListTensor = List[torch.Tensor]


class _Conv:
    """One convolution's packed weights + descriptor fields that do not depend on the input size."""

    def __init__(self, lib, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, dilation: int = 1):
        self.lib = lib
        self.cout, self.cin = int(weight.shape[0]), int(weight.shape[1])
        self.dims = weight.dim() - 2
        k = tuple(int(s) for s in weight.shape[2:])
        self.kd, self.kh, self.kw = (k if self.dims == 3 else (1,) + k)
        self.stride, self.dilation = stride, dilation
        self.bias = bias.detach().contiguous().float() if bias is not None else None
        self.weight = weight.detach().contiguous().float() if self.cout == 1 else None   # mvsn_conv_to1 takes it as is
        d = self.desc(1, 1 if self.dims == 2 else 2, 8, 32)
        n = lib.mvsn_conv_packed_floats(ctypes.byref(d))
        if n == 0:
            raise RuntimeError("mvsn_conv_packed_floats rejected the descriptor")
        self.packed = torch.empty(n, dtype=torch.float32, device=weight.device)
        w = weight.detach().contiguous().float()
        _native.check(lib.mvsn_conv_pack_weights(ctypes.byref(d), _native.ptr(w), _native.ptr(self.packed),
                                                 _native.stream()), "mvsn_conv_pack_weights")
        # second packing for the 3 x bf16 split tier, where the layer has one (32 -> 32, 3x3[x3], stride 1)
        self.packed_bx = None
        dbx = self.desc(1, 1 if self.dims == 2 else 2, 8, 32, _native.CONV_BF16X3)
        if lib.mvsn_conv_bf16x3_supported(ctypes.byref(dbx)):
            nbx = lib.mvsn_conv_packed_floats(ctypes.byref(dbx))
            self.packed_bx = torch.empty(nbx, dtype=torch.float32, device=weight.device)
            _native.check(lib.mvsn_conv_pack_weights(ctypes.byref(dbx), _native.ptr(w), _native.ptr(self.packed_bx),
                                                     _native.stream()), "mvsn_conv_pack_weights(bf16x3)")

        # third packing: Winograd F(2x2,3x3) coefficients for the 3x3 stride-1 layers that have that form (2-D with
        # dilation 1, 2, 4, 8 and the 3x3x3 volume layers; also what the plane-resident towers are packed from)
        self.packed_wino = None
        dwn = self.desc(1, 1, 8, 32, _native.CONV_FP32_WINO)
        if lib.mvsn_conv_winograd_supported(ctypes.byref(dwn)):
            nwn = lib.mvsn_conv_packed_floats(ctypes.byref(dwn))
            self.packed_wino = torch.empty(nwn, dtype=torch.float32, device=weight.device)
            _native.check(lib.mvsn_conv_pack_weights(ctypes.byref(dwn), _native.ptr(w), _native.ptr(self.packed_wino),
                                                     _native.stream()), "mvsn_conv_pack_weights(winograd)")

    def desc(self, n, depth, rows, cols, precision=_native.CONV_FP32):
        return _native.ConvDesc(n, self.cin, self.cout, depth, rows, cols, self.kd, self.kh, self.kw,
                                self.stride, self.dilation, precision)


class _Norm:
    def __init__(self, p):
        self.gamma = p.weight.detach().contiguous().float()
        self.beta = p.bias.detach().contiguous().float()


class EngineOptions:
    """Tuning switches of the launch sequence.  Owned by the MultiViewStereoNet, shared with whatever
    PlaneSweepEngine is current, copied with the module."""

    def __init__(self):
        # Residual blocks can be folded into the next convolution's tile load (no stand-alone
        # normalise/activate/add pass, 1/3 fewer launches).  Measured on MI355X the folded form is
        # 3 % slower end to end (the doubled staging loads are exposed), so it is off by default.
        self.fold_residual_blocks = False
        # Arithmetic of the 32 -> 32 channel 3x3 / 3x3x3 layers: "fp32" = exact fp32 MFMA;
        # "bf16x3" = 3 x bf16 split on the bf16 matrix cores (fp32-equivalent to ~2^-16 per product);
        # "bf16" = plain bf16 operands on the same kernels (BASELINE config 5's speed tier, outside the 1e-3 contract);
        # "bf16s" = the same plus bf16 STORAGE of the cost volume and of the regulariser's intermediate volumes (config 5's
        # "bf16 features": the chain writes the cost volume as bf16 -- mvsn_incremental_cost_volume_bf16 --, the first three
        # 3x3x3 layers read and write bf16, the fourth writes fp32 for the 32 -> 1 tail; GroupNorm statistics stay fp32 --
        # mvsn_conv_forward_bf16_storage).
        self.conv_precision = "fp32"
        # Winograd F(2x2,3x3) form of the 2-D 3x3 dilation-1 layers (fp32 throughout, 2.25x fewer multiplies).
        self.winograd = True
        # ... also where the previous layer's LeakyReLU(GN(.)) is applied on load (in LDS, by the fetching wave)
        self.winograd_with_input_transform = True
        # ... and the 3x3x3 layers of the cost-volume regulariser as 2-D Winograd products summed over the depth tap
        self.winograd_volume = True
        # ... and the extractor's 5x5 stride-2 32 -> 32 layers as F(2x2,3x3) on the input's four stride-2 phases
        # (392 multiplies per 2x2 outputs and channel pair instead of 800)
        self.winograd_stride2 = True
        self.volume_materialise = True     # LReLU(GN(.)) of the regulariser layers as one in-place pass (see cost_volume_filter)
        # Skip the stand-alone normalise/activate pass at both ends of a refiner tower (see
        # residual_tower_unfused); False keeps one pass per block (tests compare the two).
        self.trim_tower_ends = True
        # Small batches: hand a normalise/activate pass (or the folded 32 -> 1 tail) the producing convolution's RECORDS
        # instead of finalised statistics -- its workgroups run mvsn_groupnorm_finalize's code themselves (same bits) and
        # the dependent 7 us finalize launch in front of it disappears.  Limits: samples per launch, records per sample.
        self.lazy_stats_max_samples = 8
        self.lazy_stats_max_records = 2048
        # More than 2048 records per sample: the finalize launch reduces a sample's records in up to 16 slices
        # (mvsn_groupnorm_finalize_split) instead of one workgroup per sample reading up to 1.5 MB alone.
        self.split_finalize = True
        self.cat_free_heads = True         # refiner heads read [image, features, idepth] in place (no torch.cat)
        # The fused chain's three 3x3 convolutions: "auto" = one chain on SEVERAL workgroups ("banded") -- on 16x32 while
        # few chains are in flight (up to 64: one pass of thin bands), beyond that Winograd F(2x2,3x3) with the plane
        # resident in one CU ("winograd"); on 30x40 / 32x64 (no plane fits a CU) at ANY chain count: thin bands up to
        # 17 / 16 chains, the slab plan (3 / 4 LDS-resident fat bands per chain, 85 / 64 chains per pass) beyond.  Other
        # grids: the fused direct implicit GEMM.  "direct" / "winograd" / "stepwise" / "banded" force one form
        # ("stepwise" = one plane per round of full-chip launches, cols % 4 == 0: what the banded form falls back to
        # when several stream lanes share the device).
        self.chain_form = "auto"
        # Refiner towers on two batch slices, software-pipelined: slice B's convolution (matrix-pipe-bound) carries
        # slice A's normalise/activate/add pass (HBM-bound) inside its own launch (mvsn_conv_forward_carry).  Used
        # where a slice's activation tensor has at least `carry_min_bytes` bytes (small levels are launch-bound).
        self.carry_passes = True
        self.carry_min_bytes = 32 << 20
        # consecutive launches of a sliced tower walk their tiles in opposite directions: each starts on what the one
        # before it wrote / read last (38.55 -> 38.42 ms per step, five interleaved pairs; results identical)
        self.carry_alternate = True
        # ... the regulariser's three in-place passes the same way: measured +0.1-0.2 ms per 38.6 ms step only (a pass
        # costs the volume kernel about what it costs alone), so the regulariser stays one batch by default
        self.carry_volume_passes = False
        # Plane-resident towers on the 16x32 coarse grid (mvsn_tower_16x32): the level-4 refiner and the extractor's
        # residual stack as ONE launch each, one persistent workgroup per sample (15-21 launches otherwise)
        self.towers = True
        # Forwards of at most this many chains (S * B) are recorded once per input shape and replayed afterwards
        # (ForwardPlan: same calls, same arguments, ~2 us of host time per launch instead of ~23); 0 = always eager.
        self.plan_max_chains = 16
        self.plan_graph = True     # ... and from the second replay on the call list is launched as one hipGraph
        self.plan_max_bytes = 8 << 30   # intermediates all recorded plans together may keep alive
        # The banded chain's workgroups wait for each other: on a device SHARED with other work a hand-off can time out.
        # With this switch every banded call is followed, on the stream, by the single-launch form of the grid gated on
        # the banded status word (mvsn_incremental_cost_volume_guarded): a few microseconds when nothing happened, a full
        # recomputation when a hand-off timed out -- the forward's outputs are valid either way, without a host round
        # trip.  The module then stops choosing the banded form (see MultiViewStereoNet.check_device_status).
        self.banded_repair = True

    NAMES = ("split_finalize", "banded_repair", "towers", "plan_max_chains", "plan_graph", "plan_max_bytes", "carry_passes", "carry_min_bytes", "carry_volume_passes", "carry_alternate", "chain_form", "fold_residual_blocks", "conv_precision", "winograd", "winograd_with_input_transform",
             "winograd_volume", "winograd_stride2", "volume_materialise", "trim_tower_ends", "cat_free_heads", "lazy_stats_max_samples",
             "lazy_stats_max_records")


class _Records:
    """GroupNorm records of a convolution's output, (N, tiles, 4, 3), not finalised: a consumer that takes them
    (gn_lrelu, gn_lrelu_add2, conv_to1_block) forms the statistics inside its own launch."""
    __slots__ = ("partials", "tiles")

    def __init__(self, partials: torch.Tensor):
        self.partials, self.tiles = partials, int(partials.shape[1])


class _Job:
    """mvsn_apply_job + the tensors it points at: out = LReLU(GN(r)) [+ residual | + LReLU(GN_0(residual))], in place."""

    def __init__(self, r, stats, norm, residual=None, r_stats=None, r_norm=None):
        self.keep = (r, stats, norm, residual, r_stats, r_norm)
        n, spatial = r.shape[0], r[0, 0].numel()
        P = _native.ptr
        self.job = _native.ApplyJob(P(r), P(stats), P(norm.gamma), P(norm.beta), P(residual), P(r_stats),
                                    P(r_norm.gamma) if r_norm else None, P(r_norm.beta) if r_norm else None,
                                    P(r), n, 0, spatial)
        self.nbytes = 4.0 * r.numel() * (2 if residual is None else 3)

    def run_alone(self, eng):
        r, stats, norm, residual, r_stats, r_norm = self.keep
        if r_stats is not None:
            eng.gn_lrelu_add2(r, stats, norm, residual, r_stats, r_norm, out=r)
        else:
            eng.gn_lrelu(r, stats, norm, residual=residual, out=r)


class _SharedState:
    """What must outlive a PlaneSweepEngine (it is rebuilt on .to() / load_state_dict): the two status words the repair
    launches of the banded chain write (pinned host memory the device can reach: the host reads them WITHOUT
    synchronising), how many repairs have been noticed, and the latch that keeps AUTO off the banded form afterwards."""

    def __init__(self):
        self.words = None              # int32[4]: [0] |= status, [1] += 1 per repair (mvsn_chain.h: chain_gate_closed)
        self.words_np = None
        self.seen = 0
        self.banded_latched = False
        self.warned = False
        self.clean = 0                 # forwards since the latch was set (it expires: a transiently shared device)
        self.latch_forwards = self.LATCH_FORWARDS      # ... after this many forwards; doubled by every re-latch
        self.relatches = 0

    # A repaired forward keeps AUTO off the banded form for this many forwards, then the form is tried again (a device that
    # was shared for a moment should not cost the module its small-batch chain for good; a device that IS shared repairs
    # one forward in LATCH_FORWARDS and latches again).  net.reset_device_status() re-arms at once.
    # A device that stays shared would otherwise pay the spin-limit time-out (seconds) every LATCH_FORWARDS forwards for
    # ever: every re-latch doubles the wait up to LATCH_CAP, and the warning is issued once (reset_device_status re-arms both).
    LATCH_FORWARDS = 256
    LATCH_CAP = 1 << 16

    def tick(self):
        """Once per forward (after poll): lets the latch expire."""
        if self.banded_latched:
            self.clean += 1
            if self.clean >= self.latch_forwards:
                self.banded_latched, self.clean = False, 0

    def rearm(self):
        self.banded_latched, self.warned, self.clean, self.latch_forwards, self.relatches = False, False, 0, self.LATCH_FORWARDS, 0

    def status_ptr(self) -> int:
        if self.words is None:
            self.words = torch.zeros(4, dtype=torch.int32).pin_memory()
            self.words_np = self.words.numpy()
        return self.words.data_ptr()

    def poll(self) -> int:
        """Repairs since the last poll (no synchronisation: what the device has written so far)."""
        if self.words_np is None:
            return 0
        count = int(self.words_np[1])
        fresh, self.seen = count - self.seen, count
        if fresh:
            if self.relatches:         # repaired again after an expiry: the device IS shared -- wait longer next time
                self.latch_forwards = min(self.latch_forwards * 2, self.LATCH_CAP)
            self.relatches += 1
            self.banded_latched, self.clean = True, 0
            if not self.warned:
                self.warned = True
                warnings.warn(
                    "mvsn_incremental_cost_volume(banded): an inter-workgroup hand-off timed out (status %d) -- the chain's "
                    "workgroups were not co-resident: the device is shared with other work.  That forward's chain was "
                    "recomputed in-stream by the single-launch form (its outputs are valid); this module no longer "
                    "chooses the banded form (net.reset_device_status() re-enables it)." % int(self.words_np[0]),
                    RuntimeWarning, stacklevel=3)
        return fresh


class ForwardPlan:
    """One recorded forward: the library calls it consists of (function, arguments without the stream), every
    intermediate tensor those arguments point into, static copies of the inputs, and the output tensors.

    A forward at small batch is ~110 dependent launches of a few microseconds each; issued from Python one ctypes call
    at a time the host needs ~23 us per launch (argument checks, allocations, descriptor set-up: 2.6 ms per batch-1
    forward -- as long as the device takes).  All of that depends on the SHAPES only.  So the first forward of a shape
    is recorded while it runs, and later ones replay the list: inputs are copied into the plan's static tensors (one
    multi-tensor copy), each recorded call is re-issued with its recorded arguments on the current stream (~2 us of
    host time per call), and the outputs are copied into fresh tensors (so results stay valid after the next call,
    unlike a graph replay's).  Same kernels, same arguments, same order: bit-identical to the eager forward."""

    def __init__(self):
        self.calls, self.keep = [], []
        self.replayable = True
        self.static_inputs: List[torch.Tensor] = []
        self.outputs: Optional[dict] = None
        self.graph = None          # hipGraph of the call list (None: not captured yet, False: capture refused)
        self.uses = 0
        self.nbytes = 0            # bytes of intermediates the plan keeps alive
        self.chain_info = (None, None, None)   # (form, workspace, shape) of the recorded chain call

    def replay(self):
        stream = _native.stream()
        for fn, args, name in self.calls:
            rc = fn(*args, stream)
            if rc != 0:
                _native.check(rc, name)


class PlaneSweepEngine:
    """Packed weights + the launch sequence.  Built lazily from the module's parameters."""

    def __getattr__(self, name):            # only reached when normal lookup fails: the option switches
        if name in EngineOptions.NAMES:
            return getattr(self.__dict__["opt"], name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in EngineOptions.NAMES:
            setattr(self.__dict__["opt"], name, value)
        else:
            object.__setattr__(self, name, value)


    def __init__(self, net: "MultiViewStereoNet"):
        self.lib = lib = _native.load()
        self.carried_jobs = 0          # normalise/activate/add passes that travelled inside a convolution launch
        self.bf16_layers = 0           # convolution launches that ran on the bf16 / bf16x3 kernels
        # The banded chain form spins on its sibling workgroups: every workgroup of a launch must be resident, so two
        # such launches must not share the device (MultiViewStereoNet._forward_lanes clears this for its lanes).
        self.banded_ok = True
        self.net_state = net.__dict__.setdefault("_shared_state", _SharedState())   # survives rebuilds of this object
        self.last_chain_form, self.last_chain_workspace, self.last_chain_shape = None, None, None
        self.last_cost_dtype = torch.float32        # element type of the last chain call's cost volume (bf16: the "bf16s" tier)
        self.recording: Optional["ForwardPlan"] = None    # the plan a forward is being recorded into
        self.plans: Dict[tuple, Optional[ForwardPlan]] = {}   # shape key -> plan (None: that shape runs eagerly)
        self.replays = 0
        self._carried_before = {}
        # tuning switches live on the module (EngineOptions), so they survive every rebuild of this object
        # (.to(), load_state_dict, in-place parameter updates); `engine.<switch>` reads and writes through
        object.__setattr__(self, "opt", net.options)
        # When set to a list, every library call is bracketed by device events on the current
        # stream and appended as (kernel, start, end, algorithmic_flops, algorithmic_bytes).
        self.timeline: Optional[list] = None
        self.level_tag = ""            # timeline names of the refiner levels' launches carry " L<level>" (bench attribution)
        fe = net.left_feature_extractor
        self.fe_down = [_Conv(lib, getattr(fe, f"conv{i}").weight, None, stride=2) for i in range(4)]
        self.fe_res = [(_Conv(lib, getattr(fe, f"res{i}").conv1.weight, None), _Norm(getattr(fe, f"res{i}").bn1))
                       for i in range(6)]
        self.fe_final = _Conv(lib, fe.conv_final.weight, fe.conv_final.bias)

        r = net.right_feature_extractor.refiner
        dev = r.conv0.weight.device
        self.refiner_packed = torch.empty(lib.mvsn_feature_refiner_packed_floats(), dtype=torch.float32, device=dev)
        tensors = [r.conv0.weight, r.conv0.bias, r.bn0.weight, r.bn0.bias, r.res0.conv1.weight, r.res0.conv1.bias,
                   r.res0.bn1.weight, r.res0.bn1.bias, r.conv_final.weight, r.conv_final.bias]
        tensors = [t.detach().contiguous().float() for t in tensors]
        _native.check(lib.mvsn_pack_feature_refiner(*[_native.ptr(t) for t in tensors],
                                                    _native.ptr(self.refiner_packed), _native.stream()),
                      "mvsn_pack_feature_refiner")

        vf = net.volume_filter4
        self.vf_convs = [_Conv(lib, getattr(vf, f"conv{i}").weight, getattr(vf, f"conv{i}").bias) for i in range(5)]
        self.vf_norms = [_Norm(getattr(vf, f"bn{i}")) for i in range(4)]

        self.refiners = []
        for lvl in range(5):
            m = getattr(net, f"refiner{lvl}")
            self.refiners.append({
                "conv0": _Conv(lib, m.conv0.weight, m.conv0.bias),
                "bn0": _Norm(m.bn0),
                "res": [(_Conv(lib, getattr(m, f"res{i}").conv1.weight, getattr(m, f"res{i}").conv1.bias,
                               dilation=REFINER_DILATIONS[i]), _Norm(getattr(m, f"res{i}").bn1)) for i in range(6)],
                "final": _Conv(lib, m.conv_final.weight, m.conv_final.bias),
            })

        self._tower_packs = {}

    # ---- plane-resident towers (16x32) -----------------------------------------------------------
    @staticmethod
    def _tower_weights(convs):
        """Winograd-transformed weights of the layers in the tower kernel's order: per k-step
        [cout tile][xi row][lane][xi column] (mvsn_conv_pack_weights gives [xi][cout tile][lane])."""
        parts = [c.packed_wino.view(-1, 4, 4, 2, 64).permute(0, 3, 1, 4, 2).contiguous().view(-1) for c in convs]
        return torch.cat(parts).contiguous()

    def _tower_pack(self, which: str):
        if which in self._tower_packs:
            return self._tower_packs[which]
        pack = None
        if which == "refiner4":
            p = self.refiners[4]
            convs = [p["conv0"]] + [c for c, _ in p["res"]]
            norms = [p["bn0"]] + [n for _, n in p["res"]]
            fin = p["final"]
            if all(c.packed_wino is not None for c in convs) and fin.cout == 1 and fin.cin == 32 and p["conv0"].cin == 36:
                dev = fin.weight.device
                zeros = torch.zeros(32, dtype=torch.float32, device=dev)
                params = []
                for c, nm in zip(convs, norms):
                    params += [c.bias if c.bias is not None else zeros, nm.gamma, nm.beta]
                params += [fin.weight.reshape(-1), fin.bias if fin.bias is not None else zeros[:1]]
                pack = (self._tower_weights(convs), torch.cat([t.reshape(-1) for t in params]).contiguous(),
                        [c.dilation for c, _ in p["res"]])
        else:
            convs = [c for c, _ in self.fe_res]
            fin = self.fe_final
            if all(c.packed_wino is not None for c in convs + [fin]):
                dev = fin.packed_wino.device
                zeros = torch.zeros(32, dtype=torch.float32, device=dev)
                params = []
                for c, nm in self.fe_res:
                    params += [c.bias if c.bias is not None else zeros, nm.gamma, nm.beta]
                params += [fin.bias if fin.bias is not None else zeros]
                pack = (self._tower_weights(convs + [fin]), torch.cat([t.reshape(-1) for t in params]).contiguous(),
                        [c.dilation for c, _ in self.fe_res])
        self._tower_packs[which] = pack
        return pack

    def tower_extractor_tail(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """The six residual blocks + conv_final of FeatureNetwork.forward (:121-129) on (n, 32, 16, 32) as one launch."""
        pack = self._tower_pack("extractor")
        if pack is None or tuple(x.shape[1:]) != (32, 16, 32):
            return None
        U, params, dils = pack
        x = self.dense(x)
        out = self.empty(x.shape, x.dtype, x.device)
        d = _native.TowerDesc()
        d.inp[0], d.channels[0], d.sample_mod[0] = x.data_ptr(), 32, x.shape[0]
        d.head_chunks, d.n_blocks, d.tail_mode = 0, len(dils), 0
        for i, v in enumerate(dils):
            d.dilation[i] = v
        d.weights, d.params, d.out = U.data_ptr(), params.data_ptr(), out.data_ptr()
        self._call("mvsn_tower_16x32[extractor blocks]", self.lib.mvsn_tower_16x32, ctypes.byref(d), x.shape[0],
                   _native.stream(), flops=2.0 * 9 * 32 * 32 * 512 * 7 * x.shape[0], nbytes=8.0 * x.numel())
        return out

    def tower_refiner4(self, image4, feats4, prior, fx) -> Optional[torch.Tensor]:
        """IDepthmapRefiner.forward at level 4 (:468-484, gain :607-611) for N chains sharing B reference images."""
        pack = self._tower_pack("refiner4")
        if pack is None or tuple(prior.shape[1:]) != (1, 16, 32) or tuple(feats4.shape[1:]) != (32, 16, 32):
            return None
        U, params, dils = pack
        image4, feats4, prior, fx = self.f32c(image4), self.f32c(feats4), self.f32c(prior), self.f32c(fx)
        N, B = prior.shape[0], image4.shape[0]
        out = self.empty(prior.shape, prior.dtype, prior.device)
        d = _native.TowerDesc()
        for b, (t, c, m) in enumerate(((image4, 3, B), (feats4, 32, B), (prior, 1, N))):
            d.inp[b], d.channels[b], d.sample_mod[b] = t.data_ptr(), c, m
        d.block_scale, d.scale_mod, d.scale_block = fx.data_ptr(), B, 2
        d.head_chunks, d.n_blocks, d.tail_mode = 9, len(dils), 1
        for i, v in enumerate(dils):
            d.dilation[i] = v
        d.weights, d.params = U.data_ptr(), params.data_ptr()
        d.prior, d.fx, d.fx_mod, d.out = prior.data_ptr(), fx.data_ptr(), B, out.data_ptr()
        self._call("mvsn_tower_16x32[refiner 4]", self.lib.mvsn_tower_16x32, ctypes.byref(d), N, _native.stream(),
                   flops=2.0 * 9 * 32 * 512 * (36 + 6 * 32 + 1) * N, nbytes=4.0 * (36 + 1) * 512 * N)
        return out

    # ---- primitive wrappers ------------------------------------------------------------------
    def empty(self, shape, dtype=torch.float32, device=None, **_):
        """Device allocation of the launch sequence.  While a forward is being recorded into a plan (ForwardPlan) the
        tensor is kept alive by the plan: its address is what the recorded calls hold."""
        t = torch.empty(tuple(shape) if not isinstance(shape, int) else shape, dtype=dtype, device=device)
        if self.recording is not None:
            self.recording.keep.append(t)
        return t

    def _aten(self):
        """Marks the forward being recorded as not replayable: an ATen kernel took part in it (a fallback path)."""
        if self.recording is not None:
            self.recording.replayable = False

    def f32c(self, t: torch.Tensor) -> torch.Tensor:
        """t as dense fp32 -- free when it already is (the normal case), an ATen copy otherwise."""
        if t.dtype == torch.float32 and t.is_contiguous():
            return t
        self._aten()
        return t.float().contiguous()

    def dense(self, t: torch.Tensor) -> torch.Tensor:
        """t as a dense tensor -- itself when it already is; an ATen copy (which makes a forward being recorded
        non-replayable: the copy is not part of the plan) otherwise."""
        if t.is_contiguous():
            return t
        self._aten()
        return t.contiguous()

    def copy_into(self, dst: torch.Tensor, src: torch.Tensor):
        assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
        self._call("mvsn_copy", self.lib.mvsn_copy, _native.ptr(dst), _native.ptr(src),
                   dst.numel() * dst.element_size(), _native.stream(), nbytes=2.0 * dst.numel() * dst.element_size())

    def copy_many(self, dsts, srcs):
        """dst[i] <- src[i] for dense same-shape pairs, eight pairs per launch with every buffer visible to the runtime
        as a kernel argument (mvsn_copy_many); pairs of different dtype / layout fall back to Tensor.copy_."""
        pairs = []
        for d, s in zip(dsts, srcs):
            if d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous() and s.is_contiguous() and s.is_cuda:
                pairs.append((d, s))
            else:
                d.copy_(s)
        if not pairs:
            return
        n = len(pairs)
        dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
        sp = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
        nb = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in pairs])
        _native.check(self.lib.mvsn_copy_many(dp, sp, nb, n, _native.stream()), "mvsn_copy_many")

    def cat0(self, parts) -> torch.Tensor:
        """torch.cat(parts, 0) of dense fp32 tensors as device copies into one allocation."""
        parts = [self.f32c(t) for t in parts]
        if len(parts) == 1:
            return parts[0]
        out = self.empty((sum(t.shape[0] for t in parts),) + tuple(parts[0].shape[1:]), torch.float32, parts[0].device)
        at = 0
        for t in parts:
            self.copy_into(out[at:at + t.shape[0]], t)
            at += t.shape[0]
        return out

    def focal(self, K: torch.Tensor) -> torch.Tensor:
        """K[:, 0, 0] of (B, 4, 4) intrinsics as a dense (B,) tensor."""
        K = self.f32c(K)
        out = self.empty((K.shape[0],), torch.float32, K.device)
        self._call("mvsn_gather_strided", self.lib.mvsn_gather_strided, _native.ptr(K), K.shape[0], K[0].numel(),
                   _native.ptr(out), _native.stream())
        return out

    def _lvl(self) -> str:
        return f"[{self.level_tag.strip()}]" if self.level_tag else ""

    def _call(self, kernel: str, fn, *args, flops: float = 0.0, nbytes: float = 0.0):
        if self.timeline is None:
            _native.check(fn(*args), kernel)
            if self.recording is not None:
                self.recording.calls.append((fn, args[:-1], kernel))      # (the last argument is always the stream)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _native.check(fn(*args), kernel)
        b.record()
        self.timeline.append((kernel, a, b, float(flops), float(nbytes)))

    def conv(self, c: _Conv, x: torch.Tensor, in_stats=None, in_norm: Optional[_Norm] = None, want_stats=False,
             in_residual: Optional[torch.Tensor] = None, write_staged: bool = False, carry: Optional["_Job"] = None,
             out: Optional[torch.Tensor] = None, prefer_fp32_wino: bool = False, lazy_stats: bool = False):
        """x (N,C,[D,]H,W) -> (out, stats or None[, staged]).

        `lazy_stats`: the caller's consumer of the statistics takes records (`_Records`); returned instead of the
        finalised tensor when the batch is small enough for that to pay (lazy_stats_max_*).

        `in_stats`/`in_norm` fold LReLU(GN(x)) into the tile load; `in_residual` adds the residual
        branch on top (a whole SimpleBasicBlock folded into the NEXT layer's load); `write_staged`
        returns that folded input as a tensor (it is the block's output, needed as the next residual).
        `carry`: an independent normalise/activate/add job executed with this launch (inside it where the
        layer's kernel can, as its own launch otherwise); `out`: write into this tensor.
        """
        lib = self.lib
        if isinstance(x, (list, tuple)):
            # input as channel blocks: the Winograd head reads them in place, anything else gets the concatenation
            res = self.conv_blocks(c, x, want_stats, out=out) if (in_stats is None and in_residual is None and
                                                                  not write_staged) else None
            if res is not None:
                if carry is not None:
                    carry.run_alone(self)
                return res
            self._aten()
            x = torch.cat(list(x), 1)
        n = x.shape[0]
        depth = x.shape[2] if c.dims == 3 else 1
        rows, cols = x.shape[-2], x.shape[-1]
        d = c.desc(n, depth, rows, cols)
        packed = c.packed
        # the bf16 tiers: every 32 -> 32 3x3 / 3x3x3 layer that has the kernels -- except where the caller is a SLICED
        # tower (`prefer_fp32_wino`, set by residual_tower_sliced only): there the fp32 Winograd launches that carry
        # the other slice's pass beat bf16 kernels followed by a stand-alone pass (15.1 against 17.3 ms per step for
        # the refiner blocks).  Unsliced towers (small batches, small levels, odd shapes) and the extractor keep the
        # bf16 kernels; `bf16_layers` counts the launches that ran on them.
        if self.conv_precision in ("bf16x3", "bf16", "bf16s") and c.packed_bx is not None and in_residual is None and \
                not write_staged and carry is None and not prefer_fp32_wino:
            dbx = c.desc(n, depth, rows, cols,
                         _native.CONV_BF16X3 if self.conv_precision == "bf16x3" else _native.CONV_BF16)
            if lib.mvsn_conv_bf16x3_supported(ctypes.byref(dbx)):
                d, packed = dbx, c.packed_bx
                self.bf16_layers += 1
        if d.precision == _native.CONV_FP32 and self.winograd and c.packed_wino is not None and in_residual is None and not write_staged and \
                (in_stats is None or self.winograd_with_input_transform) and \
                (c.dims == 2 or self.winograd_volume) and \
                (c.stride == 1 or (self.winograd_stride2 and in_stats is None and not want_stats and carry is None and
                                   self.conv_precision not in ("bf16", "bf16s"))):
            # (the plain-bf16 tier keeps the direct extractor its error budget was measured with: its per-pixel p99.9 on
            # config 5 sits at the budget -- 1.9e-2 / 2.1e-2 of 2e-2 with the direct / phase-Winograd extractor, whose
            # features differ by fp32 rounding only)
            dwn = c.desc(n, depth, rows, cols, _native.CONV_FP32_WINO)
            if lib.mvsn_conv_winograd_supported(ctypes.byref(dwn)):
                d, packed = dwn, c.packed_wino
        ro, co = (rows - 1) // c.stride + 1, (cols - 1) // c.stride + 1
        shape = (n, c.cout, depth, ro, co) if c.dims == 3 else (n, c.cout, ro, co)
        if out is None:
            out = self.empty(shape, dtype=torch.float32, device=x.device)
        assert tuple(out.shape) == shape and out.is_contiguous()
        staged = self.empty(x.shape, x.dtype, x.device) if write_staged else None
        partials = None
        if want_stats:
            tiles = lib.mvsn_conv_num_tiles(ctypes.byref(d))
            partials = self.empty((n, tiles, 4, 3), dtype=torch.float32, device=x.device)
        taps = c.kd * c.kh * c.kw
        tag = (f"conv{c.dims}d k{c.kh}" + (f"s{c.stride}" if c.stride > 1 else "") +
               (f"d{c.dilation}" if c.dilation > 1 else "") + f" {c.cin}->{c.cout}" +
               (" bf16x3" if d.precision == _native.CONV_BF16X3 else "") +
               (" bf16" if d.precision == _native.CONV_BF16 else "") +
               (" wino" if d.precision == _native.CONV_FP32_WINO else "") + self.level_tag)
        nbytes = 4.0 * (x.numel() * (2 if in_residual is not None else 1) + out.numel() +
                        (staged.numel() if staged is not None else 0))
        if carry is not None:
            assert in_residual is None and not write_staged
            carried = ctypes.c_int(0)
            # (timeline label: whether the pass travels inside the launch is decided by the library from the same
            # shapes every step, so last step's answer names this one)
            went = self._carried_before.get(tag, True)
            self._call("mvsn_conv_forward[" + tag + (" +pass]" if went else "] + pass"), lib.mvsn_conv_forward_carry, ctypes.byref(d),
                       _native.ptr(x), _native.ptr(packed), _native.ptr(c.bias), _native.ptr(in_stats),
                       _native.ptr(in_norm.gamma) if in_norm else None, _native.ptr(in_norm.beta) if in_norm else None,
                       _native.ptr(out), _native.ptr(partials), ctypes.byref(carry.job), ctypes.byref(carried),
                       _native.stream(), flops=2.0 * c.cin * taps * c.cout * out[:, 0].numel(),
                       nbytes=nbytes + carry.nbytes)
            self.carried_jobs += carried.value
            self._carried_before[tag] = bool(carried.value)
        else:
            self._call("mvsn_conv_forward[" + tag + "]", lib.mvsn_conv_forward, ctypes.byref(d), _native.ptr(x),
                       _native.ptr(packed), _native.ptr(c.bias), _native.ptr(in_stats),
                       _native.ptr(in_norm.gamma) if in_norm else None, _native.ptr(in_norm.beta) if in_norm else None,
                       _native.ptr(in_residual), _native.ptr(staged), _native.ptr(out), _native.ptr(partials),
                       _native.stream(), flops=2.0 * c.cin * taps * c.cout * out[:, 0].numel(), nbytes=nbytes)
        stats = None
        if want_stats:
            # (records handed to the consumer are reduced in the ONE-workgroup order; the finalize launch switches to the
            # sliced order above a record count the library decides: a sample's statistics must not depend on which of
            # the two ran -- planned vs eager, batch 1 vs 8 --, so the lazy path ends where the sliced order begins
            # whatever lazy_stats_max_records says)
            if (lazy_stats and n <= self.lazy_stats_max_samples and partials.shape[1] <= self.lazy_stats_max_records
                    and not (self.split_finalize and
                             self.lib.mvsn_groupnorm_finalize_split_workspace_bytes(n, partials.shape[1]))):
                stats = _Records(partials)
            else:
                stats = self.finalize_stats(partials)
        if write_staged:
            return out, stats, staged
        return out, stats

    def finalize_stats(self, partials: torch.Tensor) -> torch.Tensor:
        """GroupNorm records (N, R, 4, 3) -> (N, 4, 2) {mean, rstd}; many records per sample are reduced in slices
        (mvsn_groupnorm_finalize_split: the slice count depends on R alone, so a sample's statistics do not depend on its
        batch)."""
        n, records = partials.shape[0], partials.shape[1]
        stats = self.empty((n, 4, 2), dtype=torch.float32, device=partials.device)
        ws_bytes = self.lib.mvsn_groupnorm_finalize_split_workspace_bytes(n, records) if self.split_finalize else 0
        if ws_bytes:
            ws = self.empty(ws_bytes, dtype=torch.uint8, device=partials.device)
            self._call("mvsn_groupnorm_finalize", self.lib.mvsn_groupnorm_finalize_split, _native.ptr(partials), n, records,
                       _native.ptr(stats), _native.ptr(ws), ws_bytes, _native.stream())
        else:
            self._call("mvsn_groupnorm_finalize", self.lib.mvsn_groupnorm_finalize, _native.ptr(partials), n, records,
                       _native.ptr(stats), _native.stream())
        return stats

    def conv_blocks(self, c: _Conv, blocks, want_stats=False, out: Optional[torch.Tensor] = None):
        """3x3 layer on the channel-wise concatenation of up to three tensors without assembling it
        (mvsn_conv_forward_blocks); None when the layer / shape has no such path."""
        lib = self.lib
        x0 = blocks[0]
        if not (self.winograd and self.cat_free_heads and c.packed_wino is not None and 1 <= len(blocks) <= 3
                and (self.conv_precision == "fp32" or c.packed_bx is None)     # (the heads have no bf16 form)
                and all(b.dtype == torch.float32 for b in blocks)):
            return None
        n, rows, cols = x0.shape[0], x0.shape[-2], x0.shape[-1]
        d = c.desc(n, 1, rows, cols, _native.CONV_FP32_WINO)
        if not lib.mvsn_conv_winograd_supported(ctypes.byref(d)) or sum(b.shape[1] for b in blocks) != c.cin:
            return None
        blocks = [self.dense(b) for b in blocks]
        if any(b.data_ptr() % 16 for b in blocks):
            return None
        if out is None:
            out = self.empty((n, c.cout, rows, cols), dtype=torch.float32, device=x0.device)
        assert tuple(out.shape) == (n, c.cout, rows, cols) and out.is_contiguous()
        partials = None
        if want_stats:
            partials = self.empty((n, lib.mvsn_conv_num_tiles(ctypes.byref(d)), 4, 3), dtype=torch.float32,
                                   device=x0.device)
        ptrs = (ctypes.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
        chans = (ctypes.c_int * len(blocks))(*[b.shape[1] for b in blocks])
        self._keep(blocks)
        self._call(f"mvsn_conv_forward_blocks[conv2d k3 {c.cin}->{c.cout} wino{self.level_tag}]", lib.mvsn_conv_forward_blocks,
                   ctypes.byref(d), ptrs, chans, len(blocks), _native.ptr(c.packed_wino), _native.ptr(c.bias),
                   _native.ptr(out), _native.ptr(partials), _native.stream(),
                   flops=2.0 * c.cin * 9 * c.cout * out[:, 0].numel(),
                   nbytes=4.0 * (sum(b.numel() for b in blocks) + out.numel()))
        stats = None
        if want_stats:
            stats = self.finalize_stats(partials)
        return out, stats

    def conv_to1(self, c: _Conv, x: torch.Tensor, prior: Optional[torch.Tensor] = None,
                 fx: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """32 -> 1 layer on the vector path (HBM-bound); None when the shape needs the MFMA kernel.
        With `prior`/`fx` the refiner epilogue relu(prior*fx + conv)/fx is applied in the same pass."""
        rows, cols = x.shape[-2], x.shape[-1]
        if c.cin != 32 or c.cout != 1 or c.dilation != 1 or c.stride != 1 or \
                not self.lib.mvsn_conv_to1_supported(rows, cols):
            return None
        n = x.shape[0]
        depth = x.shape[2] if c.dims == 3 else 1
        shape = (n, 1, depth, rows, cols) if c.dims == 3 else (n, 1, rows, cols)
        out = self.empty(shape, dtype=torch.float32, device=x.device)
        self._call(f"mvsn_conv_to1[{c.dims}d]", self.lib.mvsn_conv_to1, _native.ptr(x), _native.ptr(c.weight),
                   _native.ptr(c.bias), _native.ptr(prior), _native.ptr(fx), n, depth, rows, cols,
                   3 if c.dims == 3 else 1, _native.ptr(out), _native.stream(),
                   flops=2.0 * 32 * c.kd * 9 * out.numel(), nbytes=4.0 * (x.numel() + out.numel()))
        return out

    def residual_tower_unfused(self, x, first, blocks, final: _Conv, prior=None, fx=None):
        """first? -> [x + LReLU(GN(conv(x)))]* -> final conv, one stand-alone float4 normalise/activate/add
        pass per block -- except at the two ends of a refiner, where that pass would only feed one consumer:
        the head's activation is never written (block 1 reads conv0's raw output through the conv's input
        transform and as a transformed residual), and the last block is folded into the 32 -> 1 layer's load."""
        blocks = list(blocks)
        pend = None                      # (raw, stats, norm): LReLU(GN(raw)) not materialised
        if first is not None:
            r0, st0 = self.conv(first[0], x, want_stats=True)
            if blocks and self.trim_tower_ends:
                pend = (r0, st0, first[1])
            else:
                x = self.gn_lrelu(r0, st0, first[1], out=r0)
        shp = x[0].shape if isinstance(x, (list, tuple)) else x.shape
        to1_ok = (final.cout == 1 and final.cin == 32 and final.dilation == 1 and final.stride == 1 and
                  final.dims == 2 and self.lib.mvsn_conv_to1_supported(shp[-2], shp[-1]))
        for i, (conv, norm) in enumerate(blocks):
            if pend is not None:
                r0, st0, n0 = pend
                r, st = self.conv(conv, r0, in_stats=st0, in_norm=n0, want_stats=True, lazy_stats=True)
            else:
                r, st = self.conv(conv, x, want_stats=True, lazy_stats=True)
            if i == len(blocks) - 1 and to1_ok and self.trim_tower_ends and pend is None:
                return self.conv_to1_block(final, r, st, norm, x, prior, fx), prior is not None
            if pend is not None:
                x = self.gn_lrelu_add2(r, st, norm, *pend, out=r)
                pend = None
            else:
                x = self.gn_lrelu(r, st, norm, residual=x, out=r)
        if final.cout == 1:
            out = self.conv_to1(final, x, prior, fx)
            if out is not None:
                return out, prior is not None
        out, _ = self.conv(final, x)
        return out, False

    def residual_tower_sliced(self, blocks_in, first, blocks, final: _Conv, prior, fx):
        """residual_tower_unfused (trimmed ends, 32 -> 1 tail with the refiner epilogue) on TWO batch slices,
        software-pipelined so that every stand-alone normalise/activate/add pass of one slice travels inside the
        other slice's next convolution launch (`carry`): the launch order is
            head(A) head(B) c1(A) c1(B)+p1(A) c2(A)+p1(B) c2(B)+p2(A) ... c6(A)+p5(B) c6(B) tail(A) tail(B)
        with p_k = the pass that turns block k's raw output into its activation.  Same kernels, same arithmetic
        per sample as the unsliced tower: bit-identical results.  Reference: IDepthmapRefiner.forward
        (multi_view_stereonet.py:448-484), SimpleBasicBlock (:38-48); the samples of a batch are independent there
        (GroupNorm statistics are per sample), which is what makes the slicing legal."""
        n = blocks_in[0].shape[0]
        h = (n + 1) // 2
        bounds = ((0, h), (h, n))
        rows, cols = blocks_in[0].shape[-2], blocks_in[0].shape[-1]
        dev = blocks_in[0].device
        conv0, bn0 = first
        r0 = self.empty((n, 32, rows, cols), dtype=torch.float32, device=dev)
        st0 = [self.conv(conv0, [b[a:e] for b in blocks_in], want_stats=True, out=r0[a:e])[1] for a, e in bounds]
        x = [None, None]                 # the slice's current block input (materialised by a carried pass)
        job = None
        last = len(blocks) - 1
        tails = []
        flip = 0
        for i, (conv, norm) in enumerate(blocks):
            r = self.empty((n, 32, rows, cols), dtype=torch.float32, device=dev)
            for s_, (a, e) in enumerate(bounds):
                if job is not None and self.carry_alternate:
                    job.job.reverse = flip      # every other launch walks its tiles (and the job) from the end
                flip ^= 1
                if i == 0:
                    _, st = self.conv(conv, r0[a:e], in_stats=st0[s_], in_norm=bn0, want_stats=True, carry=job,
                                      out=r[a:e], prefer_fp32_wino=True)
                else:
                    _, st = self.conv(conv, x[s_], want_stats=True, carry=job, out=r[a:e], prefer_fp32_wino=True)
                if i == last:
                    job = None
                    tails.append((r[a:e], st, norm, x[s_]))
                elif i == 0:
                    job = _Job(r[a:e], st, norm, r0[a:e], st0[s_], bn0)      # x1 = LReLU(GN(r1)) + LReLU(GN(r0))
                    x[s_] = r[a:e]
                else:
                    job = _Job(r[a:e], st, norm, x[s_])                      # x_k = x_{k-1} + LReLU(GN(r_k))
                    x[s_] = r[a:e]
        out = self.empty((n, 1, rows, cols), dtype=torch.float32, device=dev)
        for (a, e), (r_, st, norm, x_) in zip(bounds, tails):
            self.conv_to1_block(final, r_, st, norm, x_, prior[a:e], fx[a:e], out=out[a:e])
        return out

    def gn_lrelu_add2(self, r, st, norm: _Norm, r0, st0, norm0: _Norm, out=None):
        n, spatial = r.shape[0], r[0, 0].numel()
        out = self.empty(r.shape, r.dtype, r.device) if out is None else out
        if isinstance(st, _Records):
            self._call("mvsn_groupnorm_lrelu_add2", self.lib.mvsn_groupnorm_lrelu_apply_records, _native.ptr(r),
                       _native.ptr(st.partials), st.tiles, _native.ptr(norm.gamma), _native.ptr(norm.beta),
                       _native.ptr(r0), _native.ptr(st0), _native.ptr(norm0.gamma), _native.ptr(norm0.beta), n, spatial,
                       _native.ptr(out), _native.stream(), nbytes=4.0 * r.numel() * 3)
            return out
        self._call("mvsn_groupnorm_lrelu_add2", self.lib.mvsn_groupnorm_lrelu_add2, _native.ptr(r), _native.ptr(st),
                   _native.ptr(norm.gamma), _native.ptr(norm.beta), _native.ptr(r0), _native.ptr(st0),
                   _native.ptr(norm0.gamma), _native.ptr(norm0.beta), n, spatial, _native.ptr(out), _native.stream(),
                   nbytes=4.0 * r.numel() * 3)
        return out

    def conv_to1_block(self, c: _Conv, r, st, norm: _Norm, x, prior=None, fx=None, out=None):
        """conv_to1 on x + LReLU(GN(r)) without materialising it."""
        n, rows, cols = r.shape[0], r.shape[-2], r.shape[-1]
        if out is None:
            out = self.empty((n, 1, rows, cols), dtype=torch.float32, device=r.device)
        assert tuple(out.shape) == (n, 1, rows, cols) and out.is_contiguous()
        if isinstance(st, _Records):
            self._call("mvsn_conv_to1_block" + self._lvl(), self.lib.mvsn_conv_to1_block_records, _native.ptr(r),
                       _native.ptr(st.partials), st.tiles, _native.ptr(norm.gamma), _native.ptr(norm.beta),
                       _native.ptr(x), _native.ptr(c.weight), _native.ptr(c.bias), _native.ptr(prior), _native.ptr(fx),
                       n, rows, cols, _native.ptr(out), _native.stream(), flops=2.0 * 32 * 9 * out.numel(),
                       nbytes=4.0 * (2 * r.numel() + out.numel()))
            return out
        self._call("mvsn_conv_to1_block" + self._lvl(), self.lib.mvsn_conv_to1_block, _native.ptr(r), _native.ptr(st),
                   _native.ptr(norm.gamma), _native.ptr(norm.beta), _native.ptr(x), _native.ptr(c.weight),
                   _native.ptr(c.bias), _native.ptr(prior), _native.ptr(fx), n, rows, cols, _native.ptr(out),
                   _native.stream(), flops=2.0 * 32 * 9 * out.numel(), nbytes=4.0 * (2 * r.numel() + out.numel()))
        return out

    def residual_tower(self, x, first, blocks, final: _Conv):
        """first? -> [x + LReLU(GN(conv(x)))]* -> final conv, with every normalise/activate/add folded
        into the following convolution's tile load (no stand-alone elementwise pass).

        `x` is a plain tensor when `first` is None (feature extractor), otherwise `first` = (conv0, bn0)
        and the tower starts with x0 = LReLU(GN(conv0(x))) (idepth refiner).
        """
        if first is not None:
            conv0, bn0 = first
            r, st = self.conv(conv0, x, want_stats=True)
            pend = (r, st, bn0, None)        # x0 = LReLU(GN(r)) not materialised yet
        else:
            pend = None
        cur = x
        for conv, norm in blocks:
            if pend is None:
                r, st = self.conv(conv, cur, want_stats=True)
            else:
                pr, pst, pnorm, pres = pend
                r, st, cur = self.conv(conv, pr, in_stats=pst, in_norm=pnorm, in_residual=pres, want_stats=True,
                                       write_staged=True)
            pend = (r, st, norm, cur)
        pr, pst, pnorm, pres = pend
        out, _ = self.conv(final, pr, in_stats=pst, in_norm=pnorm, in_residual=pres)
        return out

    def gn_lrelu(self, r: torch.Tensor, stats: torch.Tensor, norm: _Norm, residual: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None):
        n = r.shape[0]
        spatial = r[0, 0].numel()
        out = self.empty(r.shape, r.dtype, r.device) if out is None else out
        if isinstance(stats, _Records):
            self._call("mvsn_groupnorm_lrelu_apply" + self._lvl(), self.lib.mvsn_groupnorm_lrelu_apply_records, _native.ptr(r),
                       _native.ptr(stats.partials), stats.tiles, _native.ptr(norm.gamma), _native.ptr(norm.beta),
                       _native.ptr(residual), None, None, None, n, spatial, _native.ptr(out), _native.stream(),
                       nbytes=4.0 * r.numel() * (3 if residual is not None else 2))
            return out
        self._call("mvsn_groupnorm_lrelu_apply" + self._lvl(), self.lib.mvsn_groupnorm_lrelu_apply, _native.ptr(r),
                   _native.ptr(stats), _native.ptr(norm.gamma), _native.ptr(norm.beta), _native.ptr(residual), n,
                   spatial, _native.ptr(out), _native.stream(),
                   nbytes=4.0 * r.numel() * (3 if residual is not None else 2))
        return out

    def use_towers(self) -> bool:
        """The plane-resident towers are fp32 Winograd kernels: `winograd = False` (direct-form numerics end to end) and
        `fold_residual_blocks` switch them off with the layers they replace.  They do NOT follow `conv_precision` (the
        bf16 tiers keep these two small stages in fp32 -- more exact, and faster than 15-21 bf16 launches) nor
        `trim_tower_ends` (there is no stand-alone pass inside them to trim)."""
        return bool(self.towers and self.winograd and not self.fold_residual_blocks)

    def feature_network(self, image: torch.Tensor) -> List[torch.Tensor]:
        pyr = [image]
        x = image
        for i in range(3):
            x, _ = self.conv(self.fe_down[i], x)
            pyr.append(x)
        x, _ = self.conv(self.fe_down[3], x)
        if self.use_towers():
            feats = self.tower_extractor_tail(x)
            if feats is not None:
                pyr.append(feats)
                return pyr
        if self.fold_residual_blocks:
            pyr.append(self.residual_tower(x, None, self.fe_res, self.fe_final))
        else:
            pyr.append(self.residual_tower_unfused(x, None, self.fe_res, self.fe_final)[0])
        return pyr

    def cost_volume_filter(self, cost: torch.Tensor) -> torch.Tensor:
        n, _, depth, rows, cols = cost.shape
        to1 = bool(self.lib.mvsn_conv_to1_supported(rows, cols))
        if (self.carry_passes and self.carry_volume_passes and self.volume_materialise and self.winograd and self.winograd_volume and
                self.conv_precision == "fp32" and to1 and n >= 2 and (depth * rows * cols) % 256 == 0 and
                (n // 2) * 128 * depth * rows * cols >= self.carry_min_bytes):
            return self.cost_volume_filter_sliced(cost)
        if self.conv_precision == "bf16s":
            out = self.cost_volume_filter_bf16_storage(cost, to1)
            if out is not None:
                return out
        if cost.dtype != torch.float32:     # (a bf16 cost volume with no bf16-storage plan for the regulariser)
            self._aten()
            cost = cost.float()
        mat = self.volume_materialise and self.winograd and self.winograd_volume and self.conv_precision == "fp32"
        x, st = self.conv(self.vf_convs[0], cost, want_stats=True, lazy_stats=mat)
        for i in range(1, 4):
            if mat:
                # the volume Winograd kernel fetches every plane for three output planes: normalising it once in
                # place (one HBM pass) is cheaper than three in-LDS passes inside the convolution
                x = self.gn_lrelu(x, st, self.vf_norms[i - 1], out=x)
                x, st = self.conv(self.vf_convs[i], x, want_stats=True, lazy_stats=i < 3)
            else:
                x, st = self.conv(self.vf_convs[i], x, in_stats=st, in_norm=self.vf_norms[i - 1], want_stats=True)
        last = self.vf_convs[4]
        if to1:
            return self.conv_to1_volume_norm(last, x, st, self.vf_norms[3])
        out, _ = self.conv(last, x, in_stats=st, in_norm=self.vf_norms[3])
        return out[:, 0]

    def bf16_storage_supported(self, n: int, depth: int, rows: int, cols: int) -> bool:
        """Do the regulariser's four 3x3x3 layers have the bf16-storage kernels for this volume shape?"""
        convs = self.vf_convs[:4]
        d = convs[0].desc(n, depth, rows, cols, _native.CONV_BF16)
        return (all(c.packed_bx is not None for c in convs) and bool(self.lib.mvsn_conv_bf16x3_supported(ctypes.byref(d)))
                and cols % 2 == 0)

    def cost_volume_filter_bf16_storage(self, cost: torch.Tensor, to1: bool) -> Optional[torch.Tensor]:
        """The regulariser with its intermediate volumes stored as bf16 (`conv_precision = "bf16s"`, BASELINE config 5's
        "bf16 features", reference: multi_view_stereonet.py:341-353): layer 0 bf16 (the chain's cost volume; fp32 when a
        capture dict asks for the fp32 tensors) -> bf16, layers 1, 2 bf16 -> bf16 with the
        previous layer's LeakyReLU(GroupNorm(.)) applied on load, layer 3 bf16 -> fp32, then the usual 32 -> 1 tail.  21.5
        instead of 34.4 bytes per voxel and channel move through HBM; statistics are fp32 from the unrounded accumulators.
        None when a layer has no bf16 kernel for this shape (the caller falls back to the bf16-operand path)."""
        n, _, depth, rows, cols = cost.shape
        lib = self.lib
        convs = self.vf_convs[:4]
        d = convs[0].desc(n, depth, rows, cols, _native.CONV_BF16)
        if not (to1 and self.bf16_storage_supported(n, depth, rows, cols)):
            return None
        tiles = lib.mvsn_conv_num_tiles(ctypes.byref(d))
        x, st = cost, None
        in16_0 = cost.dtype == torch.bfloat16      # the chain stored the cost volume as bf16 (forward, step 4)
        for i, c in enumerate(convs):
            out16 = i < 3
            out = self.empty((n, 32, depth, rows, cols), dtype=torch.bfloat16 if out16 else torch.float32, device=cost.device)
            partials = self.empty((n, tiles, 4, 3), dtype=torch.float32, device=cost.device)
            nrm = self.vf_norms[i - 1] if i else None
            self._call("mvsn_conv_forward[conv3d k3 32->32 bf16 storage]", lib.mvsn_conv_forward_bf16_storage, ctypes.byref(d),
                       _native.ptr(x), 1 if (i or in16_0) else 0, _native.ptr(c.packed_bx), _native.ptr(c.bias), _native.ptr(st),
                       _native.ptr(nrm.gamma) if nrm else None, _native.ptr(nrm.beta) if nrm else None, _native.ptr(out),
                       1 if out16 else 0, _native.ptr(partials), _native.stream(),
                       flops=2.0 * 32 * 27 * 32 * out[:, 0].numel(),
                       nbytes=float(x.numel() * x.element_size() + out.numel() * out.element_size()))
            self.bf16_layers += 1
            st = self.finalize_stats(partials)
            x = out
        return self.conv_to1_volume_norm(self.vf_convs[4], x, st, self.vf_norms[3])

    def conv_to1_volume_norm(self, last: _Conv, x, st, nrm: _Norm, out=None):
        """The HBM-bound 32 -> 1 pass; applies LReLU(GN(.)) of the fourth layer while it loads the raw volume."""
        n, _, depth, rows, cols = x.shape
        if out is None:
            out = self.empty((n, depth, rows, cols), dtype=torch.float32, device=x.device)
        self._call("mvsn_conv_to1_volume_norm", self.lib.mvsn_conv_to1_volume_norm, _native.ptr(x), _native.ptr(st),
                   _native.ptr(nrm.gamma), _native.ptr(nrm.beta), _native.ptr(last.weight), _native.ptr(last.bias),
                   n, depth, rows, cols, _native.ptr(out), _native.stream(),
                   flops=2.0 * 32 * 27 * out.numel(), nbytes=4.0 * (x.numel() + out.numel()))
        return out

    def cost_volume_filter_sliced(self, cost: torch.Tensor) -> torch.Tensor:
        """The regulariser (CostVolumeFilter.forward, multi_view_stereonet.py:341-353) on two slices of the chains,
        pipelined like residual_tower_sliced: the in-place LReLU(GN(.)) pass of one slice's layer travels inside the
        other slice's next convolution launch
            c0(A) c0(B)+p0(A) c1(A)+p0(B) c1(B)+p1(A) c2(A)+p1(B) c2(B)+p2(A) c3(A)+p2(B) c3(B) tail(A) tail(B)."""
        n, _, depth, rows, cols = cost.shape
        h = (n + 1) // 2
        bounds = ((0, h), (h, n))
        x = [cost[a:e] for a, e in bounds]
        job = None
        stats = [None, None]
        for i in range(4):
            r = self.empty((n, 32, depth, rows, cols), dtype=torch.float32, device=cost.device)
            for s_, (a, e) in enumerate(bounds):
                _, stats[s_] = self.conv(self.vf_convs[i], x[s_], want_stats=True, carry=job, out=r[a:e])
                job = _Job(r[a:e], stats[s_], self.vf_norms[i]) if i < 3 else None
                x[s_] = r[a:e]
        out = self.empty((n, depth, rows, cols), dtype=torch.float32, device=cost.device)
        for s_, (a, e) in enumerate(bounds):
            self.conv_to1_volume_norm(self.vf_convs[4], x[s_], stats[s_], self.vf_norms[3], out=out[a:e])
        return out

    def idepth_refiner(self, level: int, guide, prior: torch.Tensor, fx: torch.Tensor,
                       scaled: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`guide` is a tensor or a list of channel blocks (image, features): the refiner input
        [guide..., prior * fx] is assembled with ONE concatenation.  `scaled` = prior * fx when the caller already
        has it (upsample_prior forms it in the upsampling pass)."""
        self.level_tag = f" L{level}"
        try:
            return self._idepth_refiner(level, guide, prior, fx, scaled)
        finally:
            self.level_tag = ""

    def _idepth_refiner(self, level: int, guide, prior: torch.Tensor, fx: torch.Tensor,
                        scaled: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self.refiners[level]
        prior, fx = self.dense(prior), self.dense(fx)
        n, pixels = prior.shape[0], prior[0].numel()
        if scaled is None:
            scaled = self.empty(prior.shape, prior.dtype, prior.device)
            self._call("mvsn_idepth_scale", self.lib.mvsn_idepth_scale, _native.ptr(prior), _native.ptr(fx), n, pixels,
                       _native.ptr(scaled), _native.stream(), nbytes=8.0 * prior.numel())
        x_in = (list(guide) if isinstance(guide, (list, tuple)) else [guide]) + [scaled]
        rows, cols = prior.shape[-2], prior.shape[-1]
        if (self.carry_passes and not self.fold_residual_blocks and self.trim_tower_ends and n >= 2 and
                len(x_in) <= 3 and self.winograd and self.cat_free_heads and len(p["res"]) >= 2 and
                (n // 2) * 128 * rows * cols >= self.carry_min_bytes and (rows * cols) % 512 == 0 and
                all(b.dtype == torch.float32 and b.is_contiguous() for b in x_in) and
                self.lib.mvsn_conv_to1_supported(rows, cols) and p["final"].dilation == 1):
            return self.residual_tower_sliced(x_in, (p["conv0"], p["bn0"]), p["res"], p["final"], prior, fx)
        if self.fold_residual_blocks or len(x_in) > 3:
            self._aten()                    # (an ATen kernel: a forward being recorded is not replayable)
            x_in = torch.cat(x_in, 1)       # those towers take one tensor
        if self.fold_residual_blocks:
            delta, done = self.residual_tower(x_in, (p["conv0"], p["bn0"]), p["res"], p["final"]), False
        else:
            delta, done = self.residual_tower_unfused(x_in, (p["conv0"], p["bn0"]), p["res"], p["final"],
                                                      prior=prior, fx=fx)
        if done:
            return delta            # epilogue relu(prior*fx + delta)/fx already applied in the kernel
        out = self.empty(prior.shape, prior.dtype, prior.device)
        self._call("mvsn_refiner_epilogue", self.lib.mvsn_refiner_epilogue, _native.ptr(prior), _native.ptr(fx),
                   _native.ptr(self.dense(delta)), n, pixels, _native.ptr(out), _native.stream(),
                   nbytes=12.0 * prior.numel())
        return out

    def homography_warp(self, image: torch.Tensor, H: torch.Tensor, out: Optional[torch.Tensor] = None):
        B, C, rows, cols = image.shape
        n = H.shape[1]
        vol = self.empty((B, C, n, rows, cols), dtype=torch.float32, device=image.device) if out is None else out
        assert vol.shape == (B, C, n, rows, cols) and vol.is_contiguous()
        mask = self.empty((B, n, rows, cols), dtype=torch.bool, device=image.device)
        self._call("mvsn_homography_warp", self.lib.mvsn_homography_warp, _native.ptr(image), _native.ptr(H), B, C, n,
                   rows, cols, _native.ptr(vol), _native.ptr(mask), _native.stream(),
                   nbytes=4.0 * (image.numel() + vol.numel()) + mask.numel())
        return vol, mask

    def plane_sweep_setup(self, T: torch.Tensor, K0: torch.Tensor, K4: torch.Tensor, rows4: int, cols4: int, D: int):
        N, dev = T.shape[0], T.device
        f = dict(dtype=torch.float32, device=dev)
        samples = self.empty((N, D), **f)
        H4 = self.empty((N, D, 3, 3), **f)
        Hinc = self.empty((N, D, 3, 3), **f)
        H0 = self.empty((N, 1, 3, 3), **f)
        base = self.empty((N,), **f)
        self._call("mvsn_plane_sweep_setup", self.lib.mvsn_plane_sweep_setup, _native.ptr(T), _native.ptr(K0),
                   _native.ptr(K4), N, rows4, cols4, D, _native.ptr(samples), _native.ptr(H4), _native.ptr(Hinc),
                   _native.ptr(H0), _native.ptr(base), _native.stream())
        return samples, H4, Hinc, H0, base

    def plane_sweep_setup_sources(self, Ts, K0: torch.Tensor, K4: torch.Tensor, rows4: int, cols4: int, D: int):
        """plane_sweep_setup for chains n = s * B + b straight from the per-source pose tensors and the batch's
        intrinsics (no cat / repeat)."""
        S = len(Ts)
        if S > 8:
            return self.plane_sweep_setup(self.cat0(list(Ts)), self.cat0([K0] * S), self.cat0([K4] * S), rows4, cols4, D)
        Ts = [self.f32c(t) for t in Ts]
        K0, K4 = self.f32c(K0), self.f32c(K4)
        B, dev = K0.shape[0], K0.device
        N = S * B
        f = dict(dtype=torch.float32, device=dev)
        samples, H4, Hinc = self.empty((N, D), **f), self.empty((N, D, 3, 3), **f), self.empty((N, D, 3, 3), **f)
        H0, base = self.empty((N, 1, 3, 3), **f), self.empty((N,), **f)
        ptrs = (ctypes.c_void_p * S)(*[_native.ptr(t) for t in Ts])
        self._keep(Ts)
        self._call("mvsn_plane_sweep_setup", self.lib.mvsn_plane_sweep_setup_sources, ptrs, S, _native.ptr(K0),
                   _native.ptr(K4), B, rows4, cols4, D, _native.ptr(samples), _native.ptr(H4), _native.ptr(Hinc),
                   _native.ptr(H0), _native.ptr(base), _native.stream())
        return samples, H4, Hinc, H0, base

    def focal_pyramid(self, K_pyr) -> torch.Tensor:
        """(levels, B) focal lengths K_pyr[l][:, 0, 0]."""
        Ks = [self.f32c(k) for k in K_pyr]
        L, B = len(Ks), Ks[0].shape[0]
        out = self.empty((L, B), torch.float32, Ks[0].device)
        if L > 8:
            for l in range(L):
                self.copy_into(out[l], self.focal(Ks[l]))
            return out
        ptrs = (ctypes.c_void_p * L)(*[_native.ptr(k) for k in Ks])
        self._keep(Ks)
        self._call("mvsn_gather_focal", self.lib.mvsn_gather_focal, ptrs, L, B, _native.ptr(out), _native.stream())
        return out

    def _keep(self, tensors):
        if self.recording is not None:
            self.recording.keep.extend(tensors)

    def incremental_cost_volume(self, src4, H4, Hinc, plane0, left_feats, want_features=False, cost_bf16=False):
        """cost_bf16: the cost volume stored as bf16 (the bf16 feature tier, `conv_precision = "bf16s"`:
        mvsn_incremental_cost_volume_bf16; the stepwise form has no such variant: fp32 call + conversion)."""
        N, _, rows, cols = src4.shape
        B = left_feats.shape[0]
        D = H4.shape[1]
        dev = src4.device
        form = {"auto": _native.CHAIN_AUTO, "direct": _native.CHAIN_DIRECT, "winograd": _native.CHAIN_WINOGRAD,
                "stepwise": _native.CHAIN_STEPWISE, "banded": _native.CHAIN_BANDED}[self.chain_form]
        if form == _native.CHAIN_AUTO:
            form = self.lib.mvsn_incremental_cost_volume_form_for(N, rows, cols)
            if form == _native.CHAIN_BANDED and not (self.banded_ok and not self.net_state.banded_latched
                                                     and _native.coresident_right(dev.index)):
                # lanes on several streams (see forward), a latched module, or ANOTHER PROCESS owning this device's
                # co-resident launches (_native.coresident_right): what AUTO picks once the banded form is out of reach
                form = self.lib.mvsn_incremental_cost_volume_form_for(1 << 20, rows, cols)
                if form == _native.CHAIN_BANDED:       # (30x40 / 32x64: AUTO is the banded form at any chain count)
                    form = _native.CHAIN_STEPWISE if cols % 4 == 0 else _native.CHAIN_DIRECT
        if form == _native.CHAIN_WINOGRAD and self.lib.mvsn_incremental_cost_volume_form(rows, cols) != form:
            form = _native.CHAIN_DIRECT        # no Winograd plan for this coarse grid
        if form == _native.CHAIN_STEPWISE and cols % 4 != 0:
            form = _native.CHAIN_DIRECT        # the Winograd convolutions of the stepwise form need cols % 4 == 0
        if cost_bf16 and form == _native.CHAIN_STEPWISE:
            cost, mask, fvol = self.incremental_cost_volume(src4, H4, Hinc, plane0, left_feats, want_features)
            self._aten()                    # (an ATen kernel: a forward being recorded is not replayable)
            self.last_cost_dtype = torch.bfloat16
            return cost.to(torch.bfloat16), mask, fvol
        cost = self.empty((N, 32, D, rows, cols), dtype=torch.bfloat16 if cost_bf16 else torch.float32, device=dev)
        mask = self.empty((N, D, rows, cols), dtype=torch.bool, device=dev)
        fvol = self.empty(cost.shape, torch.float32, dev) if want_features else None      # (the features stay fp32)
        ws_bytes = self.lib.mvsn_incremental_cost_volume_workspace_bytes_for(N, D, rows, cols, form)
        ws = self.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        self.last_chain_form, self.last_chain_workspace, self.last_chain_shape = form, ws, (N, rows, cols)
        self.last_cost_dtype = cost.dtype
        P = rows * cols
        common = (_native.ptr(src4), _native.ptr(H4), _native.ptr(Hinc), _native.ptr(plane0), _native.ptr(left_feats),
                  _native.ptr(self.refiner_packed), N, B, D, rows, cols, _native.ptr(cost), _native.ptr(mask),
                  _native.ptr(fvol), _native.ptr(ws), ws_bytes, form)
        acct = dict(flops=N * (D - 1) * 2.0 * 9 * 32 * (35 + 32 + 32) * P,
                    nbytes=N * (4.0 * 67 * P + (64.0 if cost_bf16 else 128.0) * D * P + D * P))  # Kernel A algorithmic bytes
        if cost_bf16:
            # (always the guarded call: a banded launch is followed by its gated repair launch)
            rws_bytes = (self.lib.mvsn_incremental_cost_volume_repair_workspace_bytes(N, rows, cols)
                         if form == _native.CHAIN_BANDED else 0)
            rws = self.empty(rws_bytes, dtype=torch.uint8, device=dev) if rws_bytes else None
            self._call("mvsn_incremental_cost_volume", self.lib.mvsn_incremental_cost_volume_bf16, *common,
                       _native.ptr(rws), rws_bytes, self.net_state.status_ptr(), _native.stream(), **acct)
        elif form == _native.CHAIN_BANDED and self.banded_repair:
            # followed by the gated single-launch form: valid outputs even if a hand-off times out (EngineOptions)
            rws_bytes = self.lib.mvsn_incremental_cost_volume_repair_workspace_bytes(N, rows, cols)
            rws = self.empty(rws_bytes, dtype=torch.uint8, device=dev) if rws_bytes else None
            self._call("mvsn_incremental_cost_volume", self.lib.mvsn_incremental_cost_volume_guarded, *common,
                       _native.ptr(rws), rws_bytes, self.net_state.status_ptr(), _native.stream(), **acct)
        else:
            self._call("mvsn_incremental_cost_volume", self.lib.mvsn_incremental_cost_volume, *common,
                       _native.stream(), **acct)
        return cost, mask, fvol

    def chain_status(self) -> int:
        """Status word of the last banded chain launch (synchronises): 0 = every inter-workgroup hand-off completed."""
        if self.last_chain_form != _native.CHAIN_BANDED or self.last_chain_workspace is None:
            return 0
        ws = self.last_chain_workspace
        n, rows, cols = self.last_chain_shape
        off = self.lib.mvsn_incremental_cost_volume_status_offset(n, rows, cols)
        return int(ws[off:off + 4].view(torch.int32).item())

    def soft_argmin(self, cost: torch.Tensor, samples: torch.Tensor) -> torch.Tensor:
        N, D, rows, cols = cost.shape
        out = self.empty((N, 1, rows, cols), dtype=torch.float32, device=cost.device)
        self._call("mvsn_soft_argmin", self.lib.mvsn_soft_argmin, _native.ptr(self.dense(cost)), _native.ptr(samples),
                   N, D, rows * cols, _native.ptr(out), _native.stream(), nbytes=4.0 * (cost.numel() + out.numel()))
        return out

    def upsample(self, x: torch.Tensor, size) -> torch.Tensor:
        n, c, h, w = x.shape
        out = self.empty((n, c, int(size[0]), int(size[1])), dtype=torch.float32, device=x.device)
        self._call("mvsn_upsample_bilinear", self.lib.mvsn_upsample_bilinear, _native.ptr(x), n, c, h, w, int(size[0]),
                   int(size[1]), _native.ptr(out), _native.stream(), nbytes=4.0 * (x.numel() + out.numel()))
        return out

    def upsample_prior(self, x: torch.Tensor, fx: torch.Tensor, size):
        """upsample(x) and upsample(x) * fx[n] (the next refiner's prior and its input channel) in one launch."""
        n, c, h, w = x.shape
        assert c == 1
        out = self.empty((n, 1, int(size[0]), int(size[1])), dtype=torch.float32, device=x.device)
        scaled = self.empty(out.shape, out.dtype, out.device)
        self._call("mvsn_upsample_prior", self.lib.mvsn_upsample_prior, _native.ptr(x), _native.ptr(fx), n, h, w,
                   int(size[0]), int(size[1]), _native.ptr(out), _native.ptr(scaled), _native.stream(),
                   nbytes=4.0 * (x.numel() + 2 * out.numel()))
        return out, scaled

    def upsample_mask(self, m: torch.Tensor, size) -> torch.Tensor:
        n, c, h, w = m.shape
        out = self.empty((n, c, int(size[0]), int(size[1])), dtype=torch.bool, device=m.device)
        self._call("mvsn_upsample_mask", self.lib.mvsn_upsample_mask, _native.ptr(m), n, c, h, w, int(size[0]),
                   int(size[1]), _native.ptr(out), _native.stream(), nbytes=float(m.numel() + out.numel()))
        return out

    def fuse_sources(self, raw, refined, baseline, mask, S, B, alias):
        N, D, rows, cols = mask.shape
        dev = raw.device
        raw_out = self.empty((B, 1, rows, cols), dtype=torch.float32, device=dev)
        ref_out = self.empty((B, 1, rows, cols), dtype=torch.float32, device=dev)
        mask_out = self.empty((B, D, rows, cols), dtype=torch.bool, device=dev)
        self._call("mvsn_fuse_sources", self.lib.mvsn_fuse_sources, _native.ptr(raw), _native.ptr(refined),
                   _native.ptr(baseline), _native.ptr(mask), S, B, D, rows * cols, 1 if alias else 0,
                   _native.ptr(raw_out), _native.ptr(ref_out), _native.ptr(mask_out), _native.stream(),
                   nbytes=float(mask.numel() + mask_out.numel()))
        return raw_out, ref_out, mask_out

    # ---- the forward -------------------------------------------------------------------------
    def forward(self, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, D, do_filter, do_refiners,
                capture: Optional[dict] = None):
        S = len(T_right_in_lefts)
        left0 = self.f32c(left_image_pyr[0])
        B = left0.shape[0]
        rows4, cols4 = left_image_pyr[-1].shape[-2:]

        # 1. set-up for all N = S*B chains (chain n = s*B + b)
        samples, H4, Hinc, H0, baseline = self.plane_sweep_setup_sources(T_right_in_lefts, K_pyr[0], K_pyr[-1], rows4,
                                                                         cols4, D)
        fx_all = self.focal_pyramid(K_pyr)          # (levels, B): every level's focal lengths in one launch

        # 2. full-resolution source images on plane 0, 3. one extractor batch
        #    (each source is warped straight into its slot of the extractor's frame batch)
        frames = self.empty(((S + 1) * B,) + tuple(left0.shape[1:]), dtype=torch.float32, device=left0.device)
        self.copy_into(frames[:B], left0)
        for s_, pyr in enumerate(right_image_pyrs):
            self.homography_warp(self.f32c(pyr[0]), H0[s_ * B:(s_ + 1) * B],
                                 out=frames[(s_ + 1) * B:(s_ + 2) * B].unsqueeze(2))
        warped0 = frames[B:].unsqueeze(2)
        feats = self.feature_network(frames)
        left_feats = [f[:B] for f in feats]        # (leading slices of dense tensors: dense)
        plane0 = feats[-1][B:]

        # 4. the fused chain
        src4 = self.cat0([p[-1] for p in right_image_pyrs])
        # (bf16 feature tier: the chain stores the cost volume as bf16 where the regulariser's bf16-storage kernels read it)
        cost16 = (self.conv_precision == "bf16s" and do_filter and capture is None and
                  bool(self.lib.mvsn_conv_to1_supported(rows4, cols4)) and self.bf16_storage_supported(S * B, D, rows4, cols4))
        cost, mask, fvol = self.incremental_cost_volume(src4, H4, Hinc, plane0, left_feats[-1],
                                                        want_features=capture is not None, cost_bf16=cost16)
        # 5. regularise + soft-argmin
        if do_filter:
            filtered = self.cost_volume_filter(cost)
        else:
            filtered = self.empty((cost.shape[0],) + tuple(cost.shape[2:]), dtype=torch.float32, device=cost.device)
            self._call("mvsn_channel_l2_norm", self.lib.mvsn_channel_l2_norm, _native.ptr(cost), cost.shape[0],
                       cost.shape[1], cost[0, 0].numel(), _native.ptr(filtered), _native.stream(),
                       nbytes=4.0 * (cost.numel() + filtered.numel()))
        raw = self.soft_argmin(filtered, samples)

        # 6. level-4 refinement per chain, then fuse the sources
        if do_refiners[4]:
            refined = None
            if self.use_towers():
                refined = self.tower_refiner4(self.f32c(left_image_pyr[-1]), left_feats[-1], raw, fx_all[-1])
            if refined is None:
                # per chain: the guide blocks of its reference image (device copies, no concatenated tensor)
                img4 = self.cat0([left_image_pyr[-1]] * S)
                feats4 = self.cat0([left_feats[-1]] * S)
                fx4 = self.cat0([fx_all[-1]] * S)
                refined = self.idepth_refiner(4, [img4, feats4], raw, fx4)
        else:
            refined = None
        raw4, idepth4, mask4 = self.fuse_sources(raw, refined, baseline, mask, S, B, alias=refined is None)
        if capture is not None:
            capture.update(idepth_samples=samples, H=H4, H_inc=Hinc, H_lvl0_plane0=H0, baseline=baseline,
                           warped_fullres=warped0, left_features=left_feats, plane0_features=plane0,
                           cost_volume=cost, mask_volume=mask, feature_volume=fvol, filtered_cost=filtered,
                           raw_per_chain=raw)

        # 7. coarse-to-fine
        idepth = [None] * 5
        prior = [None] * 5
        masks = [None] * 5
        prior[4], idepth[4], masks[4] = raw4, idepth4, mask4
        for lvl in (3, 2, 1, 0):
            size = left_image_pyr[lvl].shape[-2:]
            if do_refiners[lvl]:
                fx = fx_all[lvl]
                prior[lvl], scaled = self.upsample_prior(idepth[lvl + 1], fx, size)
            else:
                prior[lvl] = self.upsample(idepth[lvl + 1], size)
            masks[lvl] = self.upsample_mask(masks[lvl + 1], size)
            if do_refiners[lvl]:
                img = self.f32c(left_image_pyr[lvl])
                guide = [img] if lvl == 0 else [img, left_feats[lvl]]
                idepth[lvl] = self.idepth_refiner(lvl, guide, prior[lvl], fx, scaled=scaled)
            else:
                idepth[lvl] = prior[lvl]
        return {"left_idepthmap_pyr": idepth, "left_idepthmap_raw_pyr": prior, "left_idepthmap_mask_pyr": masks}


class MultiViewStereoNet(nn.Module):
    """Multi-view plane-sweep stereo network; see the module docstring for the contract."""

    num_levels = 5

    def __init__(self):
        super().__init__()
        self.min_idepth = 0.0
        build_parameter_tree(self)
        self.options = EngineOptions()
        self._engine: Optional[PlaneSweepEngine] = None
        self._engine_key = None
        self.stream_lanes = 1          # >1: cut the batch into that many slices on separate HIP streams
        self._lane_streams: List[torch.cuda.Stream] = []

    # The engine holds ctypes function pointers and device-side packed weights, the lanes HIP streams:
    # neither can be pickled or deep-copied.  Copies drop them and rebuild lazily on their first forward.
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"], state["_engine_key"], state["_lane_streams"] = None, None, []
        state.pop("_plist", None)
        state.pop("_shared_state", None)
        state.pop("_lock", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # parameters changed -> packed copies are stale
    def _invalidate(self):
        self._engine = None
        self._engine_key = None
        self.__dict__.pop("_plist", None)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._invalidate()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def register_parameter(self, name, param):
        super().register_parameter(name, param)
        self.__dict__.pop("_plist", None)

    def refresh(self):
        """Drop the packed weight copies now (after replacing Parameter objects by hand)."""
        self._invalidate()

    def check_device_status(self, synchronize: bool = True) -> int:
        """Where the wrappers synchronise anyway (multi_view_forward's timer, evaluate's final sync): has a banded chain
        (one chain on several workgroups, the small-batch default) run into a hand-off that timed out -- its workgroups
        were not co-resident, i.e. the device is shared with other work?

        With `options.banded_repair` (default) such a forward was recomputed in-stream by the single-launch form: its
        outputs are VALID; this call returns the number of repaired forwards since the last call, warns once and latches
        the module off the banded form (every later forward polls the same words without synchronising, so callers that
        never come here -- `net(...)` through torch.jit.load -- are latched one forward later).  Without the repair launch
        (`banded_repair = False`) a timed-out forward's outputs are NaN and this call raises."""
        eng = self._engine
        if eng is None:
            return 0
        if synchronize:                # (callers that have just synchronised pass False: two host reads remain)
            torch.cuda.synchronize()
        fresh = eng.net_state.poll()
        # (without the repair launch the status word lives in device memory: reading it synchronises -- a caller that
        # passes synchronize=False has done so itself, see multi_view_forward / metrics.evaluate)
        if not self.options.banded_repair:
            status = eng.chain_status()
            if status:
                raise RuntimeError(
                    "mvsn_incremental_cost_volume(banded): inter-workgroup hand-off %d timed out -- the chain's workgroups "
                    "were not co-resident (is the device shared with other work?).  The outputs of that forward are "
                    "NaN; set net.options.banded_repair = True (the default) or net.options.chain_form = 'winograd'." % status)
        return fresh

    def reset_device_status(self):
        """Re-enable the banded chain form after check_device_status / a forward latched it off."""
        st = self.__dict__.get("_shared_state")
        if st is not None:
            st.rearm()

    def engine(self) -> PlaneSweepEngine:
        # The packed copies go stale when a parameter is rebound (.to(), load_state_dict: both invalidate above) or
        # updated in place (bumps its version counter).  Walking the module tree for 202 (data_ptr, version) pairs on
        # every forward cost ~0.1 ms of a 4 ms batch-1 forward: the parameter list is cached; the key is the version
        # sum (versions only grow) plus the storage addresses (a rebound .data).
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        key = (sum(p._version for p in plist), tuple(p.data_ptr() for p in plist))
        # A Parameter OBJECT that was replaced on a submodule (module.weight = nn.Parameter(...), pruning /
        # parametrize utilities) changes neither a version nor -- necessarily -- an address the cached list knows:
        # the module tree is re-walked every 16th call (~6 us per call amortised) and the engine rebuilt when the
        # objects differ.  `net.refresh()` forces it at once.
        self.__dict__["_plist_age"] = self.__dict__.get("_plist_age", 0) + 1
        if self.__dict__["_plist_age"] >= 16:
            self.__dict__["_plist_age"] = 0
            fresh = list(self.parameters())
            if len(fresh) != len(plist) or any(a is not b for a, b in zip(fresh, plist)):
                plist = self.__dict__["_plist"] = fresh
                key = (sum(p._version for p in plist), tuple(p.data_ptr() for p in plist))
                self._engine = None
        if self._engine is None or key != self._engine_key:
            self._engine = PlaneSweepEngine(self)
            self._engine_key = key
        return self._engine

    @torch.no_grad()
    def forward(self, left_image_pyr: ListTensor, K_pyr: ListTensor, T_right_in_lefts: ListTensor,
                right_image_pyrs: List[ListTensor], num_idepth_samples: int, do_cost_volume_filter: bool,
                do_refiners: List[bool], capture: Optional[dict] = None) -> Dict[str, List[Optional[torch.Tensor]]]:
        assert len(K_pyr) == self.num_levels
        assert len(left_image_pyr) == self.num_levels
        assert len(T_right_in_lefts) == len(right_image_pyrs) and len(T_right_in_lefts) >= 1
        assert len(do_refiners) == self.num_levels
        if not left_image_pyr[0].is_cuda:
            raise RuntimeError("MultiViewStereoNet (MI355X build) runs on HIP devices only: move the module and "
                               "its inputs to 'cuda'; there is no CPU implementation of the plane-sweep path")
        if next(self.parameters()).device != left_image_pyr[0].device:
            raise RuntimeError("module parameters and inputs are on different devices")
        lock = self.__dict__.get("_lock")
        if lock is None:
            lock = self.__dict__.setdefault("_lock", threading.RLock())
        with torch.cuda.device(left_image_pyr[0].device), lock:
            # (the lock: a forward's launches -- and a recorded plan's copy-in / replay / copy-out -- are enqueued as a
            # unit; two threads sharing this module interleave whole forwards, never launches)
            eng = self.engine()
            eng.net_state.poll()       # a repaired banded chain since the last forward latches AUTO off that form
            eng.net_state.tick()       # ... for _SharedState.LATCH_FORWARDS forwards
            args = (int(num_idepth_samples), bool(do_cost_volume_filter), list(do_refiners))
            B = left_image_pyr[0].shape[0]
            lanes = min(self.stream_lanes, B) if capture is None else 1
            if lanes <= 1:
                S = len(T_right_in_lefts)
                if capture is None and eng.timeline is None and 0 < B * S <= self.options.plan_max_chains:
                    return self._forward_planned(eng, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, args)
                return eng.forward(left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, *args, capture)
            return self._forward_lanes(eng, lanes, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, args)

    def _forward_planned(self, eng, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, args):
        """Small batches: record the forward once per shape, replay it afterwards (ForwardPlan)."""
        S, L = len(T_right_in_lefts), len(left_image_pyr)
        flat = list(left_image_pyr) + list(K_pyr) + list(T_right_in_lefts) + [x for p in right_image_pyrs for x in p]
        opts = tuple(getattr(self.options, k) for k in EngineOptions.NAMES)
        # A plan owns its buffers (static inputs, intermediates, chain workspace) and its hipGraph: forwards of one shape
        # on DIFFERENT streams would race on them, so the stream is part of the key (each stream records its own plan);
        # forwards on one stream are ordered by the stream, and their enqueueing by the engine's lock.
        key = (tuple((tuple(t.shape), t.dtype) for t in flat), S, args[0], args[1], tuple(args[2]), opts, eng.banded_ok,
               eng.net_state.banded_latched, torch.cuda.current_stream().cuda_stream)
        plan = eng.plans.get(key, False)

        def unflatten(ts):
            rp = ts[2 * L + S:]
            return ts[:L], ts[L:2 * L], ts[2 * L:2 * L + S], [rp[i * L:(i + 1) * L] for i in range(S)]

        if plan is None:                         # this shape took an ATen fallback somewhere: always eager
            return eng.forward(left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, *args, None)
        if plan is False:
            plan = ForwardPlan()
            plan.static_inputs = [t.detach().clone(memory_format=torch.contiguous_format) for t in flat]
            eng.recording = plan
            try:
                plan.outputs = eng.forward(*unflatten(plan.static_inputs), *args, None)
            finally:
                eng.recording = None
            plan.nbytes = sum(t.numel() * t.element_size() for t in plan.keep)
            plan.chain_info = (eng.last_chain_form, eng.last_chain_workspace, eng.last_chain_shape)
            # a handful of shapes at most, and at most ~8 GB of kept intermediates: drop the oldest plans beyond that
            while eng.plans and (len(eng.plans) >= 8 or plan.nbytes +
                                 sum(p.nbytes for p in eng.plans.values() if p is not None) > self.options.plan_max_bytes):
                eng.plans.pop(next(iter(eng.plans)))
            eng.plans[key] = plan if plan.replayable else None
            if not plan.replayable:
                plan.keep.clear()
        else:
            eng.copy_many(plan.static_inputs, flat)
            plan.uses += 1
            capturing = torch.cuda.is_current_stream_capturing()     # (the caller is building a graph of its own)
            if plan.graph is None and self.options.plan_graph and plan.uses >= 2 and not capturing:
                # every address in the recorded calls is static: from its second replay on, the call list runs as
                # ONE hipGraph launch (captured by replaying it under stream capture)
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        plan.replay()
                    plan.graph = g
                except Exception:                       # capture refused (e.g. a foreign capture in progress): stay on the list
                    plan.graph = False
            if plan.graph and not capturing:
                plan.graph.replay()
            else:
                plan.replay()
            eng.replays += 1
            # (chain_status / check_device_status look at the chain call of THIS forward, not of the last recording)
            eng.last_chain_form, eng.last_chain_workspace, eng.last_chain_shape = plan.chain_info
        # fresh output tensors (a caller may keep them across forwards)
        out, news, olds = {}, [], []
        for k, lst in plan.outputs.items():
            out[k] = []
            for t in lst:
                if t is None:
                    out[k].append(None)
                    continue
                n = torch.empty_like(t)
                out[k].append(n)
                news.append(n)
                olds.append(t)
        eng.copy_many(news, olds)
        return out

    def _forward_lanes(self, eng, lanes, left_image_pyr, K_pyr, T_right_in_lefts, right_image_pyrs, args):
        """Images are independent, so the batch is cut into `lanes` slices that run on their own HIP
        streams: one slice's HBM-bound kernels (normalise/activate, upsamples) overlap another
        slice's MFMA-bound convolutions.  All slices join the caller's stream before returning."""
        B = left_image_pyr[0].shape[0]
        main = torch.cuda.current_stream()
        if len(self._lane_streams) < lanes:
            self._lane_streams = [torch.cuda.Stream() for _ in range(lanes)]
        bounds = [(B * i) // lanes for i in range(lanes + 1)]
        parts = []
        eng.banded_ok = False           # concurrent lanes: the banded chain needs the device to itself
        try:
            for i in range(lanes):
                lo, hi = bounds[i], bounds[i + 1]
                st = self._lane_streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    parts.append(eng.forward([x[lo:hi] for x in left_image_pyr], [k[lo:hi] for k in K_pyr],
                                             [t[lo:hi] for t in T_right_in_lefts],
                                             [[x[lo:hi] for x in p] for p in right_image_pyrs], *args, None))
        finally:
            eng.banded_ok = True        # (also when a lane's forward raised)
        out = {}
        for st in self._lane_streams[:lanes]:
            main.wait_stream(st)
        for key in parts[0]:
            out[key] = []
            for lvl in range(self.num_levels):
                pieces = [p[key][lvl] for p in parts]
                for t in pieces:
                    t.record_stream(main)
                out[key].append(torch.cat(pieces, 0))
        return out
