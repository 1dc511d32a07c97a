"""Parameter tree of MultiViewStereoNet: the reference's checkpoint layout, nothing else.

The reference's modules double as parameter containers and as the eager compute graph.  Here
they are containers only -- the compute lives in the HIP library -- but their attribute names
reproduce the reference's 226 ``state_dict`` keys exactly, including the second registration
of the left extractor under ``right_feature_extractor.feature_extractor``
(multi_view_stereonet/multi_view_stereonet.py:506-507, :243), so ``load_state_dict(strict=True)``
works in both directions.

Layout (reference lines): FeatureNetwork :91-105, FeatureRefiner :409-420, CostVolumeFilter
:325-337, IDepthmapRefiner :453-464, SimpleBasicBlock utils/resnet.py:85-86.
"""
from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn as nn

FEATURE_CHANNELS = 32
GN_GROUPS = 4
GN_EPS = 1e-5
LRELU_SLOPE = 0.2
REFINER_DILATIONS = (1, 2, 4, 8, 1, 1)


class ConvParams(nn.Module):
    """weight (cout, cin, k[, k[, k]]) and optional bias; init N(0, 0.01) / zeros (:40,:47)."""

    def __init__(self, cin: int, cout: int, k: int, bias: bool, dims: int = 2):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((cout, cin) + (k,) * dims).normal_(0.0, 0.01))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)


class NormParams(nn.Module):
    """GroupNorm(4, 32) affine parameters (:25-31)."""

    def __init__(self, channels: int = FEATURE_CHANNELS):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))


class ResBlockParams(nn.Module):
    def __init__(self, bias: bool):
        super().__init__()
        self.conv1 = ConvParams(32, 32, 3, bias)
        self.bn1 = NormParams()


class FeatureNetworkParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = ConvParams(3, 32, 5, False)
        self.conv1 = ConvParams(32, 32, 5, False)
        self.conv2 = ConvParams(32, 32, 5, False)
        self.conv3 = ConvParams(32, 32, 5, False)
        for i in range(6):
            setattr(self, f"res{i}", ResBlockParams(False))
        self.conv_final = ConvParams(32, 32, 3, True)


class FeatureRefinerParams(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0 = ConvParams(35, 32, 3, True)
        self.bn0 = NormParams()
        self.res0 = ResBlockParams(True)
        self.conv_final = ConvParams(32, 32, 3, True)


class SourceExtractorParams(nn.Module):
    def __init__(self, shared: FeatureNetworkParams):
        super().__init__()
        self.feature_extractor = shared
        self.refiner = FeatureRefinerParams()


class CostVolumeFilterParams(nn.Module):
    def __init__(self):
        super().__init__()
        for i in range(4):
            setattr(self, f"conv{i}", ConvParams(32, 32, 3, True, dims=3))
            setattr(self, f"bn{i}", NormParams())
        self.conv4 = ConvParams(32, 1, 3, True, dims=3)


class IDepthRefinerParams(nn.Module):
    def __init__(self, guide_channels: int):
        super().__init__()
        self.conv0 = ConvParams(guide_channels + 1, 32, 3, True)
        self.bn0 = NormParams()
        for i in range(6):
            setattr(self, f"res{i}", ResBlockParams(True))
        self.conv_final = ConvParams(32, 1, 3, True)


def build_parameter_tree(root: nn.Module) -> None:
    """Attach the reference-named submodules to ``root``."""
    root.left_feature_extractor = FeatureNetworkParams()
    root.right_feature_extractor = SourceExtractorParams(root.left_feature_extractor)
    root.volume_filter4 = CostVolumeFilterParams()
    root.refiner4 = IDepthRefinerParams(35)
    root.refiner3 = IDepthRefinerParams(35)
    root.refiner2 = IDepthRefinerParams(35)
    root.refiner1 = IDepthRefinerParams(35)
    root.refiner0 = IDepthRefinerParams(3)


def parameter_shapes() -> Dict[str, Tuple[int, ...]]:
    """All 226 state_dict keys -> shapes."""
    holder = nn.Module()
    build_parameter_tree(holder)
    return OrderedDict((k, tuple(v.shape)) for k, v in holder.state_dict().items())
