"""ResNetBase (reference models/resnet.py:9-163): `_make_layer` (1x1 conv + norm
downsample when the width changes) and BN weight initialisation.  Only what
Res16UNet needs is kept; the ResNet14-101 classification variants are not on the
hot path."""
import torch.nn as nn

from .. import MinkowskiEngine as ME
from .model import Model
from .modules.common import ConvType, NormType, conv, get_norm


class ResNetBase(Model):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)
    OUT_PIXEL_DIST = 32
    HAS_LAST_BLOCK = False
    CONV_TYPE = ConvType.HYPERCUBE

    def __init__(self, in_channels, out_channels, config, D=3, **kwargs):
        assert self.BLOCK is not None
        assert self.OUT_PIXEL_DIST > 0
        super().__init__(in_channels, out_channels, config, D, **kwargs)
        self.network_initialization(in_channels, out_channels, config, D)
        self.weight_initialization()

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, norm_type=NormType.BATCH_NORM,
                    bn_momentum=0.1):
        width = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != width:
            downsample = nn.Sequential(
                conv(self.inplanes, width, kernel_size=1, stride=stride, bias=False, D=self.D),
                get_norm(norm_type, width, D=self.D, bn_momentum=bn_momentum),
            )
        # NB (reference resnet.py:124-146): the blocks themselves are built WITHOUT bn_momentum,
        # i.e. their norms keep the default 0.1 while stem/downsample norms use config.bn_momentum.
        layers = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=downsample,
                        conv_type=self.CONV_TYPE, D=self.D)]
        self.inplanes = width
        layers += [block(self.inplanes, planes, stride=1, dilation=dilation, conv_type=self.CONV_TYPE, D=self.D)
                   for _ in range(1, blocks)]
        return nn.Sequential(*layers)
