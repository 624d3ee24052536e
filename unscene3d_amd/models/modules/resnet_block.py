"""Residual blocks (reference models/modules/resnet_block.py:7-137).

Same submodule names (conv1/norm1/conv2/norm2/downsample) so checkpoints load;
forward fuses BN + residual add + ReLU into one statistics pass and one
elementwise pass on the device."""
import torch.nn as nn

from ... import units
from ...MinkowskiEngine import MinkowskiBatchNorm, MinkowskiReLU, conv_bn_act
from .common import ConvType, NormType, conv, get_norm


def _residual(block, x):
    if block.downsample is None:
        return x
    ds_conv, ds_norm = block.downsample[0], block.downsample[1]
    return ds_norm(ds_conv(x))


class BasicBlockBase(nn.Module):
    expansion = 1
    NORM_TYPE = NormType.BATCH_NORM

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, conv_type=ConvType.HYPERCUBE,
                 bn_momentum=0.1, D=3):
        super().__init__()
        self.conv1 = conv(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, conv_type=conv_type, D=D)
        self.norm1 = get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=1, dilation=dilation, bias=False,
                          conv_type=conv_type, D=D)
        self.norm2 = get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def _native(self, x):
        """The whole block as one autograd node (units._BasicBlock) when every piece is the plain stride-1 form."""
        ds = self.downsample
        convs = (self.conv1, self.conv2) + ((ds[0],) if ds is not None else ())
        if not units.usable(x.F, *convs):
            return False
        if self.conv1.stride != 1 or self.conv2.stride != 1 or self.conv1.kernel_volume == 1 or \
                self.conv1.ksize != self.conv2.ksize or not isinstance(self.norm1, MinkowskiBatchNorm):
            return False
        return ds is None or (len(ds) == 2 and ds[0].stride == 1 and ds[0].kernel_volume == 1
                              and isinstance(ds[1], MinkowskiBatchNorm))

    def forward(self, x):
        if self._native(x):
            cm, ts = x.coordinate_manager, x._ts()
            return x._like(units.basic_block(x.F, self, cm.kmap_cube(ts, self.conv1.ksize), cm.kmap_identity(ts)))
        out = self.norm1(self.conv1(x), relu=True)
        out = self.conv2(out)
        # norm2 -> `out += residual` -> relu  (reference :56-62), fused
        return self.norm2(out, residual=_residual(self, x), relu=True)


class BasicBlock(BasicBlockBase):
    NORM_TYPE = NormType.BATCH_NORM


class BottleneckBase(nn.Module):
    expansion = 4
    NORM_TYPE = NormType.BATCH_NORM

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, conv_type=ConvType.HYPERCUBE,
                 bn_momentum=0.1, D=3):
        super().__init__()
        self.conv1 = conv(inplanes, planes, kernel_size=1, D=D)
        self.norm1 = get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=stride, dilation=dilation, conv_type=conv_type, D=D)
        self.norm2 = get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
        self.conv3 = conv(planes, planes * self.expansion, kernel_size=1, D=D)
        self.norm3 = get_norm(self.NORM_TYPE, planes * self.expansion, D, bn_momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = conv_bn_act(self.conv1, self.norm1, x, relu=True)
        out = conv_bn_act(self.conv2, self.norm2, out, relu=True)
        res = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        return conv_bn_act(self.conv3, self.norm3, out, residual=res, relu=True)


class Bottleneck(BottleneckBase):
    NORM_TYPE = NormType.BATCH_NORM
