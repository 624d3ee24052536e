"""Res16UNet family (reference models/res16unet.py:9-508) on the MI355X façade.

U-Net with 4 strided (k2,s2) down-convs, 4 transposed up-convs and residual
BasicBlocks; `Res16UNet34C`: LAYERS=(2,3,4,6,2,2,2,2),
PLANES=(32,64,128,256,256,128,96,96) (reference :310-312, :373-374).  Module names
equal the reference's, so `state_dict` keys match published checkpoints
(conv0p1s1, bn0, conv{1..4}p{1,2,4,8}s2, bn{1..4}, block{1..8},
convtr{4..7}p{16,8,4,2}s2, bntr{4..7}, final).  `final` is constructed but — as in
the reference forward (:224-297) — never applied when `out_fpn` is used.
"""
from .. import MinkowskiEngine as ME
from ..MinkowskiEngine import MinkowskiOps as me
from .modules.common import ConvType, NormType, conv, conv_tr, get_norm
from .modules.resnet_block import BasicBlock, Bottleneck
from .resnet import ResNetBase


class Res16UNetBase(ResNetBase):
    BLOCK = None
    PLANES = (32, 64, 128, 256, 256, 256, 256, 256)
    DILATIONS = (1, 1, 1, 1, 1, 1, 1, 1)
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
    INIT_DIM = 32
    OUT_PIXEL_DIST = 1
    NORM_TYPE = NormType.BATCH_NORM
    NON_BLOCK_CONV_TYPE = ConvType.SPATIAL_HYPERCUBE
    CONV_TYPE = ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS

    # encoder stage i: (conv name, norm name, block name); pixel distance doubles each stage
    _DOWN = (("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"),
             ("conv3p4s2", "bn3", "block3"), ("conv4p8s2", "bn4", "block4"))
    # decoder stage j: (convtr name, norm name, block name, index into PLANES, skip width source)
    _UP = (("convtr4p16s2", "bntr4", "block5"), ("convtr5p8s2", "bntr5", "block6"),
           ("convtr6p4s2", "bntr6", "block7"), ("convtr7p2s2", "bntr7", "block8"))

    def __init__(self, in_channels, out_channels, config, D=3, out_fpn=False, **kwargs):
        super().__init__(in_channels, out_channels, config, D)
        self.out_fpn = out_fpn

    def network_initialization(self, in_channels, out_channels, config, D):
        assert D == 3, "the MI355X hot path is 3-D (reference configs use D=3)"
        mom = config.bn_momentum
        P, L, E = self.PLANES, self.LAYERS, self.BLOCK.expansion

        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = conv(in_channels, self.inplanes, kernel_size=config.conv1_kernel_size, stride=1,
                              dilation=1, conv_type=self.NON_BLOCK_CONV_TYPE, D=D)
        self.bn0 = get_norm(self.NORM_TYPE, self.inplanes, D, bn_momentum=mom)

        for i, (cname, nname, bname) in enumerate(self._DOWN):
            setattr(self, cname, conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dilation=1,
                                      conv_type=self.NON_BLOCK_CONV_TYPE, D=D))
            setattr(self, nname, get_norm(self.NORM_TYPE, self.inplanes, D, bn_momentum=mom))
            setattr(self, bname, self._make_layer(self.BLOCK, P[i], L[i], dilation=self.DILATIONS[i],
                                                  norm_type=self.NORM_TYPE, bn_momentum=mom))

        # skip widths seen by the decoder blocks: block3, block2, block1 outputs, then the stem
        skips = (P[2] * E, P[1] * E, P[0] * E, self.INIT_DIM)
        for j, (cname, nname, bname) in enumerate(self._UP):
            width = P[4 + j]
            setattr(self, cname, conv_tr(self.inplanes, width, kernel_size=2, upsample_stride=2, dilation=1,
                                         bias=False, conv_type=self.NON_BLOCK_CONV_TYPE, D=D))
            setattr(self, nname, get_norm(self.NORM_TYPE, width, D, bn_momentum=mom))
            self.inplanes = width + skips[j]
            setattr(self, bname, self._make_layer(self.BLOCK, width, L[4 + j], dilation=self.DILATIONS[4 + j],
                                                  norm_type=self.NORM_TYPE, bn_momentum=mom))

        self.final = conv(P[7], out_channels, kernel_size=1, stride=1, bias=True, D=D)
        self.relu = ME.MinkowskiReLU(inplace=True)

    # -- shared trunk ----------------------------------------------------------
    def _trunk(self, x):
        """-> (stride-1 output, [s16, s8, s4, s2, s1] block outputs)."""
        # all coordinate / kernel maps of the pyramid first (their host read-backs would otherwise stall the
        # convolution pipeline four times; see CoordinateManager.prepare)
        x.coordinate_manager.prepare(x.tensor_stride[0], n_down=len(self._DOWN), ksize=3)
        # the whole trunk as one step program each way (one autograd node, two C calls forward + one per stage
        # backward: unscene3d_amd/program.py) when every piece is the plain form; else module by module below
        from .. import program
        lv = program.trunk(self, x)
        if lv is not None:
            cm, ts, top = x.coordinate_manager, x._ts(), len(self._DOWN)
            levels = [ME.SparseTensor(features=f, coordinate_manager=cm,
                                      coordinate_map_key=ME.CoordinateMapKey(ts << (top - k))) for k, f in enumerate(lv)]
            return levels[-1], levels
        skip = [ME.conv_bn_act(self.conv0p1s1, self.bn0, x, relu=True)]          # out_p1
        out = skip[0]
        for cname, nname, bname in self._DOWN:
            out = ME.conv_bn_act(getattr(self, cname), getattr(self, nname), out, relu=True)
            out = getattr(self, bname)(out)
            skip.append(out)                                     # out_b1p2, out_b2p4, out_b3p8, (s16)
        levels = [out]                                           # pixel_dist 16
        for j, (cname, nname, bname) in enumerate(self._UP):
            out = ME.conv_bn_act(getattr(self, cname), getattr(self, nname), out, relu=True)
            out = me.cat(out, skip[3 - j])
            out = getattr(self, bname)(out)
            levels.append(out)                                   # pixel_dist 8, 4, 2, 1
        return out, levels

    def forward(self, x):
        out, feature_maps = self._trunk(x)
        if not self.out_fpn:
            return out
        return out, feature_maps


class Res16UNet14(Res16UNetBase):
    BLOCK = BasicBlock
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class Res16UNet18(Res16UNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class Res16UNet34(Res16UNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet50(Res16UNetBase):
    BLOCK = Bottleneck
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet14A(Res16UNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet18A(Res16UNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class Res16UNet34A(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 64)


class Res16UNet34B(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 32)


class Res16UNet34C(Res16UNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


class Res16UNet34CMultiRes(Res16UNet34C):
    """Pseudo-mask feature extractor (reference :428-505): returns `final(out)` and the
    per-resolution block outputs res_1 … res_16."""

    def forward(self, x):
        out, levels = self._trunk(x)
        res_16, res_8, res_4, res_2, res_1 = levels
        return self.final(out), {"res_1": res_1, "res_2": res_2, "res_4": res_4, "res_8": res_8, "res_16": res_16}
