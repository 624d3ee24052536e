"""Layer factories (reference models/modules/common.py:20-188): `conv`, `conv_tr`,
`get_norm`, `ConvType`, `NormType`.  Every ConvType the hot path uses maps to a
HYPER_CUBE region (reference common.py:58-67)."""
from enum import Enum

from ... import MinkowskiEngine as ME


class NormType(Enum):
    BATCH_NORM = 0
    INSTANCE_NORM = 1
    INSTANCE_BATCH_NORM = 2


class ConvType(Enum):
    HYPERCUBE = 0
    SPATIAL_HYPERCUBE = 1
    SPATIO_TEMPORAL_HYPERCUBE = 2
    HYPERCROSS = 3
    SPATIAL_HYPERCROSS = 4
    SPATIO_TEMPORAL_HYPERCROSS = 5
    SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS = 6


_CUBE_TYPES = {ConvType.HYPERCUBE, ConvType.SPATIAL_HYPERCUBE, ConvType.SPATIO_TEMPORAL_HYPERCUBE,
               ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS}


def get_norm(norm_type, n_channels, D, bn_momentum=0.1):
    if norm_type != NormType.BATCH_NORM:
        raise ValueError(f"Norm type: {norm_type} not supported on the MI355X hot path")
    return ME.MinkowskiBatchNorm(n_channels, momentum=bn_momentum)


def _generator(conv_type, kernel_size, stride, dilation, D):
    if conv_type not in _CUBE_TYPES:
        raise ValueError(f"conv_type {conv_type} is not used by UnScene3D's shipped configs")
    return ME.KernelGenerator(kernel_size, stride, dilation, region_type=ME.RegionType.HYPER_CUBE, dimension=D)


def conv(in_planes, out_planes, kernel_size, stride=1, dilation=1, bias=False, conv_type=ConvType.HYPERCUBE, D=-1):
    assert D > 0, "Dimension must be a positive integer"
    return ME.MinkowskiConvolution(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size,
                                   stride=stride, dilation=dilation, bias=bias,
                                   kernel_generator=_generator(conv_type, kernel_size, stride, dilation, D),
                                   dimension=D)


def conv_tr(in_planes, out_planes, kernel_size, upsample_stride=1, dilation=1, bias=False,
            conv_type=ConvType.HYPERCUBE, D=-1):
    assert D > 0, "Dimension must be a positive integer"
    return ME.MinkowskiConvolutionTranspose(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size,
                                            stride=upsample_stride, dilation=dilation, bias=bias,
                                            kernel_generator=_generator(conv_type, kernel_size, upsample_stride,
                                                                        dilation, D),
                                            dimension=D)


def avg_pool(kernel_size, stride=1, dilation=1, conv_type=ConvType.HYPERCUBE, in_coords_key=None, D=-1):
    assert D > 0, "Dimension must be a positive integer"
    return ME.MinkowskiAvgPooling(kernel_size=kernel_size, stride=stride, dilation=dilation, dimension=D)
