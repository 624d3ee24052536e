"""MinkowskiEngine-compatible operator façade over libusc3d_hip.so.

Mirrors the subset of the MinkowskiEngine 0.5.4 Python API that UnScene3D's hot
path calls (reference call sites: models/modules/common.py:20-188,
models/res16unet.py:1-2,222,259-289, models/modules/resnet_block.py:2,45,
models/mask3d.py:4-5,131,206-215,424-436, trainer/trainer.py:115-117,
datasets/utils.py:408,430): same class names, constructor arguments, parameter
names/layouts (`kernel` f32[K,Cin,Cout], `bias` f32[1,Cout], `bn.*`) and error
behaviour, so `models/res16unet.py`-shaped code runs unchanged.

Coordinates, kernel maps and all feature arithmetic live on the MI355X; this
module only does object bookkeeping.  There is no CPU backend.
"""
from __future__ import annotations

import math
from enum import Enum

import torch
import torch.nn as nn

from .. import ops
from . import utils  # noqa: F401  (ME.utils.sparse_quantize / sparse_collate)

__version__ = "0.5.4+usc3d"


class RegionType(Enum):
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


class KernelGenerator:
    """Reference: models/modules/common.py:137-144 builds one per conv."""

    def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPER_CUBE,
                 region_offsets=None, expand_coordinates=False, axis_types=None, dimension=-1):
        assert dimension > 0
        self.dimension = dimension
        self.kernel_size = _to_list(kernel_size, dimension)
        self.kernel_stride = _to_list(stride, dimension)
        self.kernel_dilation = _to_list(dilation, dimension)
        self.region_type = region_type
        self.axis_types = axis_types
        if region_type != RegionType.HYPER_CUBE:
            raise NotImplementedError("only RegionType.HYPER_CUBE is on UnScene3D's hot path (common.py:58-67)")
        self.kernel_volume = int(torch.prod(torch.tensor(self.kernel_size)))


def _to_list(v, D):
    if isinstance(v, (list, tuple)):
        assert len(v) == D
        return [int(x) for x in v]
    return [int(v)] * D


class CoordinateMapKey:
    def __init__(self, tensor_stride):
        self._ts = tuple(_to_list(tensor_stride, 3))

    def get_tensor_stride(self):
        return list(self._ts)

    def get_key(self):
        return (list(self._ts), "")

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and self._ts == other._ts

    def __hash__(self):
        return hash(self._ts)

    def __repr__(self):
        return f"CoordinateMapKey(tensor_stride={list(self._ts)})"


class CoordinateManager:
    """Owns the coordinate maps and kernel maps of one batch (lifetime = the batch,
    like ME's CoordinateManager).  Maps are keyed by (isotropic) tensor stride."""

    def __init__(self, D=3):
        self.D = D
        self._maps = {}       # ts -> ops.CoordMap
        self._down = {}       # fine ts -> dict(parent, nbr2, kidx, rulebook)
        self._cube = {}       # (ts, ksize) -> dict(nbr, rulebook)
        self._batch_rows = {}  # ts -> list of (start, end) per batch, or index lists
        self._sorted_input = None
        self.single_scene = False   # set by whoever KNOWS the batch holds one scene (datasets.prefetch: len(samples) == 1)
        self._kmaps = {}      # ("cube", ts, ksize) / ("down", ts) / ("id", ts) -> units.KMapRef (native issue path)

    # -- maps
    def insert(self, coords: torch.Tensor, tensor_stride: int = 1):
        cmap, unique_idx, inverse = ops.coordmap_build(coords, quant=1, tensor_stride=tensor_stride)
        self._maps[tensor_stride] = cmap
        return cmap, unique_idx, inverse

    def coord_map(self, ts: int) -> ops.CoordMap:
        return self._maps[ts]

    def get_coordinates(self, key: CoordinateMapKey) -> torch.Tensor:
        return self._maps[key.get_tensor_stride()[0]].coords

    def stride_map(self, ts: int):
        """Coarser map at 2*ts (built once, cached) + child table of the k2/s2 kernel."""
        if ts not in self._down:
            fine = self._maps[ts]
            coarse, _, parent = ops.coordmap_build(fine.coords, quant=2 * ts, tensor_stride=2 * ts)
            if 2 * ts not in self._maps:
                self._maps[2 * ts] = coarse
            else:  # a map at that stride already exists (decoder side); it must be the same set
                coarse = self._maps[2 * ts]
            nbr2, kidx = ops.kernel_map_down2(fine, parent, coarse)
            self._down[ts] = {"parent": parent, "nbr2": nbr2, "kidx": kidx, "rulebook": None}
        return self._down[ts]

    def down_rulebook(self, ts: int):
        d = self.stride_map(ts)
        if d["rulebook"] is None:
            d["rulebook"] = ops.rulebook_compact(d["nbr2"])
        return d["rulebook"]

    def cube_map(self, ts: int, ksize: int = 3):
        key = (ts, ksize)
        if key not in self._cube:
            self._cube[key] = {"nbr": ops.kernel_map_cube(self._maps[ts], ksize), "rulebook": None}
        return self._cube[key]

    def cube_rulebook(self, ts: int, ksize: int = 3):
        d = self.cube_map(ts, ksize)
        if d["rulebook"] is None:
            d["rulebook"] = ops.rulebook_compact(d["nbr"])
        return d["rulebook"]

    # -- kernel-map descriptors of the native issue path (csrc/units.hip); built once per batch
    def kmap_cube(self, ts: int, ksize: int = 3):
        key = ("cube", ts, ksize)
        if key not in self._kmaps:
            from .. import units
            n = self._maps[ts].n
            self._kmaps[key] = units.kmap_from_table(self.cube_map(ts, ksize)["nbr"], self.cube_rulebook(ts, ksize), n)
        return self._kmaps[key]

    def kmap_down(self, ts: int):
        """k=2,s=2 map between ts (fine, n_in) and 2ts (coarse, n_out); the transposed conv uses it too."""
        key = ("down", ts)
        if key not in self._kmaps:
            from .. import units
            d = self.stride_map(ts)
            self._kmaps[key] = units.kmap_from_table(d["nbr2"], self.down_rulebook(ts), self._maps[ts].n)
        return self._kmaps[key]

    def kmap_identity(self, ts: int):
        key = ("id", ts)
        if key not in self._kmaps:
            from .. import units
            self._kmaps[key] = units.kmap_identity(self._maps[ts].n)
        return self._kmaps[key]

    def prepare(self, ts: int, n_down: int, ksize: int = 3):
        """Build the whole pyramid a U-Net needs in one go: maps at ts, 2ts, ... (n_down halvings), their k2/s2
        child tables, the ksize^3 neighbour tables with their mask-sorted row orders, and the per-scene row
        ranges.  Map building reads counts back to the host; done lazily, each read-back would wait for all the
        convolution work queued before it and leave the GPU idle while the host catches up (measured: ~18 ms of
        idle per 71 ms profiled step, concentrated in the forward pass).  Done here, the device queue holds only
        the short map kernels."""
        t = ts
        for lvl in range(n_down + 1):
            if t not in self._maps:
                break
            nbr = self.cube_map(t, ksize)["nbr"]
            ops.rowsort(nbr)
            if lvl < n_down:
                d = self.stride_map(t)
                ops.rowsort(d["nbr2"])
            self.batch_slices(t)
            if torch.is_grad_enabled():
                # the pair lists (weight gradients, transposed convs) and the native path's map descriptors as well:
                # on the prefetcher's side stream this is off the step's critical path
                self.kmap_cube(t, ksize)
                self.kmap_identity(t)
                if lvl < n_down:
                    self.kmap_down(t)
            t *= 2

    # -- batch decomposition
    def batch_slices(self, ts: int):
        """Row ranges of every scene in the map at tensor stride ts.  Rows are grouped by batch in every
        map this manager builds (first-occurrence order of batch-sorted input), so one bincount and one
        host read per level suffice; unsorted input falls back to index lists."""
        if ts not in self._batch_rows and self.single_scene:
            # one scene: every row of every map belongs to it — no bincount, no read-back (five of each per batch)
            self._batch_rows[ts] = [slice(0, int(self._maps[ts].n))] if int(self._maps[ts].n) > 0 else []
        if ts not in self._batch_rows:
            b = self._maps[ts].coords[:, 0]
            if b.numel() == 0:
                self._batch_rows[ts] = []
            else:
                if self._sorted_input is None:
                    b1 = self._maps[min(self._maps)].coords[:, 0]
                    self._sorted_input = bool((b1[1:] >= b1[:-1]).all().item()) if b1.numel() > 1 else True
                if self._sorted_input:
                    ends = torch.cumsum(torch.bincount(b.long()), 0).tolist()
                    starts = [0] + ends[:-1]
                    self._batch_rows[ts] = [slice(s, e) for s, e in zip(starts, ends)]
                else:
                    nb = int(b.max().item()) + 1
                    self._batch_rows[ts] = [torch.nonzero(b == i).reshape(-1) for i in range(nb)]
        return self._batch_rows[ts]


class SparseTensor:
    """ME.SparseTensor subset.  `SparseTensor(features, coordinates=…, device=…)` inserts the
    (already unique) coordinates as the stride-1 map in the given row order — the identity
    ordering that reference models/mask3d.py:206-209 relies on;
    `SparseTensor(features=…, coordinate_manager=…, coordinate_map_key=…)` re-uses a map."""

    def __init__(self, features=None, coordinates=None, tensor_stride=1, coordinate_map_key=None,
                 coordinate_manager=None, device=None, **kwargs):
        if features is None:
            raise ValueError("features required")
        if device is not None:
            features = features.to(device)
        if not features.is_cuda:
            raise RuntimeError("unscene3d_amd SparseTensor lives on the HIP device; pass device='cuda'")
        if coordinate_map_key is None:
            if coordinates is None:
                raise ValueError("either coordinates or coordinate_map_key must be given")
            coordinates = coordinates.to(features.device).to(torch.int32).contiguous()
            ts = tensor_stride if isinstance(tensor_stride, int) else tensor_stride[0]
            coordinate_manager = coordinate_manager or CoordinateManager(D=coordinates.shape[1] - 1)
            cmap, unique_idx, _ = coordinate_manager.insert(coordinates, ts)
            if cmap.n != coordinates.shape[0]:
                # ME's default quantization mode would sub-sample duplicates; the reference
                # always passes unique coordinates (sparse_quantize ran in the collate).
                features = features[unique_idx]
            coordinate_map_key = CoordinateMapKey(ts)
        self._F = features
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key

    # --- attributes used by the reference
    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.get_coordinates(self.coordinate_map_key)

    @property
    def coordinates(self):
        return self.C

    @property
    def features(self):
        return self._F

    @property
    def device(self):
        return self._F.device

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def shape(self):
        return self._F.shape

    def _ts(self):
        return self.coordinate_map_key.get_tensor_stride()[0]

    @property
    def decomposed_features(self):
        F, n = self._F, self._F.shape[0]
        # a slice that covers every row (one scene per batch) is the tensor itself as a view: `F[0:n]` would be a
        # SliceBackward node whose backward zero-fills an [n, C] tensor and copies the gradient into it (41 us at
        # 148 k voxels x 128 channels, once per training step)
        return [F.view(F.shape) if (isinstance(s, slice) and s.start in (0, None) and s.stop == n and s.step in (1, None))
                else F[s] for s in self.coordinate_manager.batch_slices(self._ts())]

    @property
    def decomposed_coordinates(self):
        C = self.C
        return [C[s][:, 1:] for s in self.coordinate_manager.batch_slices(self._ts())]

    def float(self):
        return self if self._F.dtype == torch.float32 else self._like(self._F.float())

    def _like(self, feats):
        return SparseTensor(features=feats, coordinate_manager=self.coordinate_manager,
                            coordinate_map_key=self.coordinate_map_key)

    def __iadd__(self, other):  # reference resnet_block.py:61 `out += residual`
        assert other.coordinate_map_key == self.coordinate_map_key
        self._F = self._F + other._F
        return self

    def __add__(self, other):
        assert other.coordinate_map_key == self.coordinate_map_key
        return self._like(self._F + other._F)

    def __len__(self):
        return self._F.shape[0]

    def __repr__(self):
        return f"SparseTensor(F={tuple(self._F.shape)}, tensor_stride={self.tensor_stride})"


class MinkowskiNetwork(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.D = D


class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, is_transpose=False, expand_coordinates=False, dimension=-1):
        super().__init__()
        assert dimension == 3, "UnScene3D's hot path is 3-D"
        if kernel_generator is None:
            kernel_generator = KernelGenerator(kernel_size=kernel_size, stride=stride, dilation=dilation,
                                               dimension=dimension)
        self.is_transpose = is_transpose
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_generator = kernel_generator
        self.dimension = dimension
        ks, st, dl = kernel_generator.kernel_size, kernel_generator.kernel_stride, kernel_generator.kernel_dilation
        if len(set(ks)) != 1 or len(set(st)) != 1 or set(dl) != {1}:
            raise NotImplementedError("anisotropic / dilated kernels are not on UnScene3D's hot path")
        self.ksize, self.stride = ks[0], st[0]
        self.kernel_volume = kernel_generator.kernel_volume
        self.use_mm = self.kernel_volume == 1 and self.stride == 1
        shape = (in_channels, out_channels) if self.use_mm else (self.kernel_volume, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))          # [ME] layout
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self, is_transpose=False):
        # [ME] MinkowskiConvolutionBase.reset_parameters: U(-1/sqrt(n), 1/sqrt(n)),
        # n = (out if transpose else in) * kernel_volume
        with torch.no_grad():
            n = (self.out_channels if self.is_transpose else self.in_channels) * self.kernel_volume
            stdv = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.ksize}, "
                f"stride={self.stride}, transpose={self.is_transpose}")


class MinkowskiConvolution(_ConvBase):
    """Reference: created by models/modules/common.py:146 (`conv`)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=-1):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=False, dimension=dimension)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x._ts()
        if self.stride == 1:
            if self.kernel_volume == 1:
                out = ops.conv_same(x.F, self.kernel, self.bias, None, None)
            else:
                nbr = cm.cube_map(ts, self.ksize)["nbr"]
                out = ops.conv_same(x.F, self.kernel, self.bias, nbr, lambda: cm.cube_rulebook(ts, self.ksize))
            return x._like(out)
        if self.stride == 2 and self.ksize == 2:
            d = cm.stride_map(ts)
            out = ops.conv_down2(x.F, self.kernel, d["nbr2"], lambda: cm.down_rulebook(ts))
            if self.bias is not None:
                out = out + self.bias
            return SparseTensor(features=out, coordinate_manager=cm, coordinate_map_key=CoordinateMapKey(2 * ts))
        raise NotImplementedError(f"conv kernel_size={self.ksize} stride={self.stride} is not on the hot path")


class MinkowskiConvolutionTranspose(_ConvBase):
    """Reference: created by models/modules/common.py:179 (`conv_tr`); k=2, upsample stride 2.
    The output map is the cached finer map (tensor stride ts/2), so `me.cat` with the encoder
    skip is row-aligned (reference models/res16unet.py:259-289)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=-1):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias, kernel_generator,
                         is_transpose=True, dimension=dimension)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x._ts()
        if not (self.stride == 2 and self.ksize == 2):
            raise NotImplementedError("only k=2, s=2 transposed convs are on the hot path")
        fine_ts = ts // 2
        if fine_ts < 1 or fine_ts not in cm._down:
            raise RuntimeError("MinkowskiConvolutionTranspose: no cached finer coordinate map to upsample onto")
        d = cm.stride_map(fine_ts)
        n_fine = cm.coord_map(fine_ts).n
        out = ops.conv_tr_up2(x.F, self.kernel, d["nbr2"], lambda: cm.down_rulebook(fine_ts), n_fine)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(features=out, coordinate_manager=cm, coordinate_map_key=CoordinateMapKey(fine_ts))


class MinkowskiBatchNorm(nn.Module):
    """= BatchNorm1d over feature rows; parameters under `.bn.*` like ME
    (reference models/modules/common.py:22, models/resnet.py:90-94)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x: SparseTensor, residual: SparseTensor = None, relu: bool = False) -> SparseTensor:
        bn = self.bn
        training = bn.training or not bn.track_running_stats
        # nn.BatchNorm's batch counter is bumped inside the statistics launch (one tiny add kernel per layer otherwise)
        nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None
        out = ops.batch_norm_act(x.F, bn.weight, bn.bias, None if residual is None else residual.F, relu, bn.eps,
                                 bn.running_mean, bn.running_var, bn.momentum, training, nbt)
        return x._like(out)


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x: SparseTensor) -> SparseTensor:
        return x._like(ops.relu(x.F))


class MinkowskiAvgPooling(nn.Module):
    """kernel_size=2, stride=2 average pooling over present children
    (reference models/mask3d.py:131; forward only — the reference detaches its output)."""

    def __init__(self, kernel_size=-1, stride=1, dilation=1, kernel_generator=None, dimension=-1):
        super().__init__()
        ks = _to_list(kernel_size, dimension)
        st = _to_list(stride, dimension)
        if set(ks) != {2} or set(st) != {2}:
            raise NotImplementedError("only MinkowskiAvgPooling(kernel_size=2, stride=2) is on the hot path")

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, ts = x.coordinate_manager, x._ts()
        d = cm.stride_map(ts)
        out = ops.avgpool_down2(x.F.detach().contiguous(), d["nbr2"])
        return SparseTensor(features=out, coordinate_manager=cm, coordinate_map_key=CoordinateMapKey(2 * ts))


def cat(*tensors):
    """MinkowskiOps.cat: channel concat of tensors on the same map (reference res16unet.py:259-289)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tensors[0]
    key = tensors[0].coordinate_map_key
    for t in tensors:
        if t.coordinate_map_key != key:
            raise RuntimeError("cat: tensors live on different coordinate maps")
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))


def conv_bn_act(conv, norm, x: SparseTensor, residual: SparseTensor = None, relu: bool = True) -> SparseTensor:
    """`norm(conv(x))` (+ residual) (+ ReLU) — MinkowskiConvolution[Transpose] followed by MinkowskiBatchNorm
    (reference models/res16unet.py:231-297, models/modules/resnet_block.py:48-64) through the native issue path
    (one C call each way, unscene3d_amd/units.py) when it applies, else through the two modules."""
    from .. import units
    if isinstance(norm, MinkowskiBatchNorm) and units.usable(x.F, conv):
        cm, ts = x.coordinate_manager, x._ts()
        res = None if residual is None else residual.F
        if isinstance(conv, MinkowskiConvolutionTranspose):
            if conv.stride == 2 and conv.ksize == 2 and ts // 2 >= 1 and (ts // 2) in cm._down:
                out = units.conv_bn_act(x.F, conv.kernel, norm.bn, cm.kmap_down(ts // 2), units.UP, res, relu)
                return SparseTensor(features=out, coordinate_manager=cm, coordinate_map_key=CoordinateMapKey(ts // 2))
        elif conv.stride == 1:
            kmap = cm.kmap_identity(ts) if conv.kernel_volume == 1 else cm.kmap_cube(ts, conv.ksize)
            return x._like(units.conv_bn_act(x.F, conv.kernel, norm.bn, kmap, units.SAME, res, relu))
        elif conv.stride == 2 and conv.ksize == 2:
            out = units.conv_bn_act(x.F, conv.kernel, norm.bn, cm.kmap_down(ts), units.DOWN, res, relu)
            return SparseTensor(features=out, coordinate_manager=cm, coordinate_map_key=CoordinateMapKey(2 * ts))
    return norm(conv(x), residual=residual, relu=relu)


# The two submodule paths the reference imports — `import MinkowskiEngine.MinkowskiOps as me` (models/res16unet.py:1,
# models/mask3d.py:4: `me.cat`, `me.SparseTensor`) and `from MinkowskiEngine.MinkowskiPooling import MinkowskiAvgPooling`
# (models/mask3d.py:5) — are views of this module, registered as importable submodules.
def _submodule(name, **members):
    import sys
    import types
    m = types.ModuleType(f"{__name__}.{name}")
    m.__dict__.update(members)
    sys.modules[m.__name__] = m
    return m


MinkowskiOps = _submodule("MinkowskiOps", SparseTensor=SparseTensor, cat=cat)
MinkowskiPooling = _submodule("MinkowskiPooling", MinkowskiAvgPooling=MinkowskiAvgPooling)
