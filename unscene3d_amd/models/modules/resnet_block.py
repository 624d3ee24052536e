"""Residual blocks (reference models/modules/resnet_block.py:7-137).

Same submodule names (conv1/norm1/conv2/norm2/downsample) so checkpoints load;
forward fuses BN + residual add + ReLU into one statistics pass and one
elementwise pass on the device."""
import torch.nn as nn

from ...MinkowskiEngine import MinkowskiReLU
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

    def forward(self, x):
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
        out = self.norm1(self.conv1(x), relu=True)
        out = self.norm2(self.conv2(out), relu=True)
        out = self.conv3(out)
        return self.norm3(out, residual=_residual(self, x), relu=True)


class Bottleneck(BottleneckBase):
    NORM_TYPE = NormType.BATCH_NORM
