"""Mask3D (reference models/mask3d.py:16-664) on the MI355X operator façade.

Same constructor signature, sub-module names (=> state_dict keys) and output dictionary as the
reference.  Backbone, 1x1 mask head, segment means, FPS, Fourier encodings, attention-mask
pooling and row gathers run on the hand-written HIP kernels; the 100-query transformer decoder
(nn.MultiheadAttention / Linear / LayerNorm: ~1 GFLOP per pass) stays on PyTorch-ROCm's stock
kernels, as planned in SURVEY.md §7.6.

Randomness: the reference sub-samples cross-attention keys with `torch.randperm` (mask3d.py:325);
`self.randperm` can be replaced to inject fixed indices (parity tests do that).
"""
import contextlib
import os

import torch
import torch.nn as nn
from torch.nn import functional as F

from .. import MinkowskiEngine as ME
from .. import ops
from ..MinkowskiEngine import MinkowskiOps as me
from ..MinkowskiEngine.MinkowskiPooling import MinkowskiAvgPooling
from ..pointnet2_utils import furthest_point_sample
from .modules.common import conv
from .modules.helpers_3detr import GenericMLP
from .position_embedding import PositionEmbeddingCoordsSine

SINGLE_POINT_ERROR = "only a single point gives nans in cross-attention"   # trainer.py:125 string-matches this
# the residual connection of a decoder block routed through the block's first projection node (one autograd add per
# block and pass less: 36 launches per step)
_RESIDUAL_IN_PROJECTION = os.environ.get("USC3D_RESIDUAL_IN_PROJECTION", "1") == "1"
_PADDED_MASK_EMBED = os.environ.get("USC3D_PADDED_MASK_EMBED", "1") == "1"
_LAZY_HOST_COPIES = os.environ.get("USC3D_LAZY_HOST_COPIES", "1") == "1"
_FUSED_KEY_SAMPLING = os.environ.get("USC3D_FUSED_KEY_SAMPLING", "1") == "1"
_GATHER_INTO_GRAPH_INPUTS = os.environ.get("USC3D_GATHER_INTO_GRAPH_INPUTS", "1") == "1"
_GRAD_SINKS = os.environ.get("USC3D_GRAD_SINKS", "1") == "1"
_LN_PASSTHROUGH = os.environ.get("USC3D_LN_PASSTHROUGH", "1") == "1"


class Mask3D(nn.Module):
    def __init__(self, config, hidden_dim, num_queries, num_heads, dim_feedforward, sample_sizes, shared_decoder,
                 num_classes, num_decoders, dropout, pre_norm, positional_encoding_type, non_parametric_queries,
                 train_on_segments, normalize_pos_enc, use_level_embed, scatter_type, hlevels, use_np_features,
                 voxel_size, max_sample_size, random_queries, gauss_scale, random_query_both, random_normal):
        super().__init__()
        self.random_normal, self.random_query_both, self.random_queries = random_normal, random_query_both, random_queries
        self.max_sample_size, self.gauss_scale, self.voxel_size = max_sample_size, gauss_scale, voxel_size
        self.scatter_type, self.hlevels, self.use_level_embed = scatter_type, list(hlevels), use_level_embed
        self.train_on_segments, self.normalize_pos_enc = train_on_segments, normalize_pos_enc
        self.num_decoders, self.num_classes, self.dropout, self.pre_norm = num_decoders, num_classes, dropout, pre_norm
        self.shared_decoder, self.sample_sizes = shared_decoder, list(sample_sizes)
        self.non_parametric_queries, self.use_np_features = non_parametric_queries, use_np_features
        self.mask_dim, self.num_heads, self.num_queries = hidden_dim, num_heads, num_queries
        self.pos_enc_type = positional_encoding_type

        self.backbone = config.backbone if hasattr(config, "backbone") else config["backbone"]
        self.num_levels = len(self.hlevels)
        sizes = self.backbone.PLANES[-5:]

        self.mask_features_head = conv(self.backbone.PLANES[7], self.mask_dim, kernel_size=1, stride=1, bias=True, D=3)
        if scatter_type not in ("mean", "max"):
            raise ValueError(f"Scatter function not known: {scatter_type!r}")          # reference :68-69 asserts
        assert (not use_np_features) or non_parametric_queries, "np features only with np queries"

        if non_parametric_queries:
            self.query_projection = GenericMLP(input_dim=self.mask_dim, hidden_dims=[self.mask_dim],
                                               output_dim=self.mask_dim, use_conv=True, output_use_activation=True,
                                               hidden_use_bias=True)
            if use_np_features:
                self.np_feature_projection = nn.Sequential(nn.Linear(sizes[-1], hidden_dim), nn.ReLU(),
                                                           nn.Linear(hidden_dim, hidden_dim))
        elif random_query_both:
            self.query_projection = GenericMLP(input_dim=2 * self.mask_dim, hidden_dims=[2 * self.mask_dim],
                                               output_dim=2 * self.mask_dim, use_conv=True,
                                               output_use_activation=True, hidden_use_bias=True)
        else:
            self.query_feat = nn.Embedding(num_queries, hidden_dim)
            self.query_pos = nn.Embedding(num_queries, hidden_dim)
        if use_level_embed:
            self.level_embed = nn.Embedding(self.num_levels, hidden_dim)

        self.mask_embed_head = nn.Sequential(Linear(hidden_dim, hidden_dim), nn.ReLU(),
                                             Linear(hidden_dim, hidden_dim))
        self.class_embed_head = nn.Linear(hidden_dim, self.num_classes)

        if positional_encoding_type not in ("fourier", "sine"):
            raise NotImplementedError("positional_encoding_type 'legacy' is not used by the shipped configs")
        self.pos_enc = PositionEmbeddingCoordsSine(pos_type=positional_encoding_type, d_pos=self.mask_dim,
                                                   gauss_scale=gauss_scale, normalize=normalize_pos_enc)
        self.pooling = MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=3)

        self.masked_transformer_decoder = nn.ModuleList()
        self.cross_attention, self.self_attention = nn.ModuleList(), nn.ModuleList()
        self.ffn_attention, self.lin_squeeze = nn.ModuleList(), nn.ModuleList()
        for _ in range(1 if shared_decoder else num_decoders):
            ca, sa, ffn, sq = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            for hlevel in self.hlevels:
                ca.append(CrossAttentionLayer(d_model=self.mask_dim, nhead=num_heads, dropout=dropout,
                                              normalize_before=pre_norm))
                sq.append(Linear(sizes[hlevel], self.mask_dim))
                sa.append(SelfAttentionLayer(d_model=self.mask_dim, nhead=num_heads, dropout=dropout,
                                             normalize_before=pre_norm))
                ffn.append(FFNLayer(d_model=self.mask_dim, dim_feedforward=dim_feedforward, dropout=dropout,
                                    normalize_before=pre_norm))
            self.cross_attention.append(ca)
            self.self_attention.append(sa)
            self.ffn_attention.append(ffn)
            self.lin_squeeze.append(sq)
        self.decoder_norm = LayerNorm(hidden_dim)
        self.randperm = lambda n, device: torch.randperm(n, device=device)

    # ------------------------------------------------------------------
    def _eager_pass(self, dec, i):
        return _DecoderPass(self.cross_attention[dec][i], self.self_attention[dec][i], self.ffn_attention[dec][i],
                            self.num_heads)

    def _key_prep(self, dec, i):
        """The query-independent half of pass (dec, i): lin_squeeze (+ level embedding) and the key / value projections."""
        # the embedding MODULE, not a slice of its weight: autograd differentiates with respect to the live parameter
        return _KeyPrep(self.lin_squeeze[dec][i], self.cross_attention[dec][i],
                        self.level_embed if self.use_level_embed else None, i)

    def _side_stream(self, device):
        """The HIP stream the query-independent key preparation of the decoder passes runs on (one per device)."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        pool = self.__dict__.setdefault("_usc_side_streams", {})
        if key not in pool:
            from .. import streams       # (measured to overlap with the compute and prefetch streams: streams.py)
            pool[key] = streams.pick(device, "keys")
            ops.SIDE_STREAMS.append(pool[key])
        return pool[key]

    def _decoder_pass(self, decoder_counter, dec, i):
        """The callable for pass (decoder_counter, level i): a captured HIP graph when enabled, else eager."""
        graphs = getattr(self, "_graphed_passes", None)
        if graphs is not None:
            return graphs[decoder_counter * self.num_levels + i]
        return self._eager_pass(dec, i)

    def enable_decoder_graphs(self, batch_size: int, device):
        """Capture the 12 decoder passes (forward and backward) as HIP graphs for the static shapes
        [batch_size, sample_sizes[hlevel]] — valid while every scene has at least that many voxels per level
        (checked per call; otherwise the eager path runs).  Weights stay shared: the graphs read the live
        parameter tensors and add their gradients straight into `p.grad` (unscene3d_amd/graphs.py), so gradient
        buffers must stay allocated: `optimizer.zero_grad(set_to_none=False)`."""
        from ..graphs import forget_grad_buffers
        forget_grad_buffers()          # a re-capture must not leave the old passes' buffers registered (stale pointers)
        passes, samples = [], []
        sizes = self.backbone.PLANES[-5:]
        B, Q, d = batch_size, self.num_queries, self.mask_dim
        for decoder_counter in range(self.num_decoders):
            dec = 0 if self.shared_decoder else decoder_counter
            for i, hlevel in enumerate(self.hlevels):
                K = self.sample_sizes[hlevel]
                passes.append(self._eager_pass(dec, i))
                samples.append((torch.zeros(B, Q, d, device=device, requires_grad=True),
                                torch.zeros(Q, B, d, device=device, requires_grad=True),
                                torch.zeros(K, B, d, device=device, requires_grad=True),       # keys   (ops.in_proj_kv,
                                torch.zeros(K, B, d, device=device, requires_grad=True),       # values  outside the graph)
                                torch.zeros(B, K, Q, device=device, dtype=torch.bool)))
        from ..graphs import capture_passes
        # query_pos (input 1) is one tensor for all passes of a step; the queries (input 0) are the previous pass's output
        graphed = capture_passes(passes, samples, shared_inputs=(1,), chain_input=0)
        self._graph_shapes = [tuple(a.shape for a in smp) for smp in samples]
        object.__setattr__(self, "_graphed_passes", list(graphed))

    def disable_decoder_graphs(self):
        object.__setattr__(self, "_graphed_passes", None)
        from ..graphs import forget_grad_buffers
        forget_grad_buffers()

    def get_pos_encs(self, coords):
        """Per level, per scene Fourier encodings [N_l, d] of the pooled raw coordinates
        (reference :183-198; same [level][0][scene] nesting)."""
        out = []
        for level in coords:
            per_scene = []
            for xyz in level.decomposed_features:
                mn, mx = _col_minmax(xyz)
                per_scene.append(self.pos_enc.fourier_rows(xyz, mn, mx)
                                 if self.pos_enc_type == "fourier" else
                                 self.pos_enc(xyz[None].float(), input_range=[mn[None], mx[None]])
                                 .squeeze(0).permute(1, 0))
            out.append([per_scene])
        return out

    @torch.no_grad()
    def precompute_geometry(self, x, raw_coordinates, point2segment=None, num_segments=None, n_levels=5,
                            is_eval=False):
        """Everything of the forward pass that depends only on the scene's GEOMETRY — coordinates, raw coordinates,
        segment ids — and on no learnt parameter (reference :205-241): the pooled raw coordinates of every level, their
        Fourier encodings, the segment CSRs, the farthest-point query seeds and their encodings.  ~0.7 ms of
        latency-bound launches (100 dependent FPS rounds, counting sorts, min/max reductions) that the scene prefetcher
        issues on its side stream under the previous step's backward (datasets/prefetch.py); `forward` picks the result
        up from the input tensor `x`, or calls this itself.  Needs the coordinate maps of the pyramid (prepare())."""
        cm = x.coordinate_manager
        n_scenes = len(x.decomposed_coordinates)
        coordinates = me.SparseTensor(features=raw_coordinates.float().contiguous(), coordinate_manager=cm,
                                      coordinate_map_key=x.coordinate_map_key)
        coords = [coordinates]
        for _ in range(n_levels - 1):
            coords.append(self.pooling(coords[-1]))
        coords.reverse()
        geo = {"n_levels": n_levels, "coordinates": coordinates, "coords": coords,
               "pos_encodings_pcd": self.get_pos_encs(coords), "seg_csr": None}
        if self.train_on_segments and point2segment is not None:
            if num_segments is None:
                num_segments = [None] * len(point2segment)
            geo["seg_csr"] = [ops.segment_csr(p2s.to(torch.int64).contiguous(),
                                              int(p2s.max().item()) + 1 if ns is None else int(ns))
                              for p2s, ns in zip(point2segment, num_segments)]
        if self.non_parametric_queries:
            dec_coords = x.decomposed_coordinates
            fps_idx = [furthest_point_sample(dec_coords[i][None].float().contiguous(), self.num_queries)
                       .squeeze(0).long() for i in range(n_scenes)]
            raw_per_scene = coordinates.decomposed_features
            sampled_coords = _stack([raw_per_scene[i][fps_idx[i]] for i in range(n_scenes)])
            mm = [_col_minmax(r) for r in raw_per_scene]
            mins = _stack([m[0] for m in mm])
            maxs = _stack([m[1] for m in mm])
            geo.update(fps_idx=fps_idx, sampled_coords=sampled_coords,
                       query_pos_enc=self.pos_enc(sampled_coords.float(), input_range=[mins, maxs]))     # B, d, Q
        # the cross-attention key samples of all passes (reference :325 draws them inside the decoder loop; they depend
        # on the level sizes only): ~9 launches each (arange, random keys, sort, de-duplication) taken off the
        # decoder's critical path.  `forward` consumes them ONCE (a second forward over the same maps draws again).
        # With random query initialisation (`random_queries` / `random_query_both`) the reference draws the queries
        # BEFORE the key samples (mask3d.py:285-325): leave the draw to forward(), after its torch.rand / randn, so that a
        # seeded run consumes the global generator in the reference's order.
        rng_queries = (not self.non_parametric_queries) and (self.random_queries or self.random_query_both)
        geo["key_samples"] = None if rng_queries else self._draw_key_samples(coords, is_eval)
        if geo["key_samples"] is not None:
            self._gather_pos_keys(geo["key_samples"], geo["pos_encodings_pcd"], geo)
        if self.train_on_segments and point2segment is not None and _FUSED_ATTN_MASK and len(point2segment) >= 1:
            # the mask module's child -> segment-row table (used by all 12 attention-mask chains of the step)
            if len(point2segment) == 1:
                rows = point2segment[0].to(torch.int64).contiguous()
            else:
                off, parts = 0, []
                for p2s, csr in zip(point2segment, geo["seg_csr"]):
                    parts.append(p2s.to(torch.int64) + off)
                    off += csr.S
                rows = torch.cat(parts).contiguous()
                cm._usc_p2s_batched = rows
            geo["p2s_rows"] = rows
            _child_segment_table(cm, x._ts(), rows)
        # kept on the INPUT tensor, not on the coordinate manager: the dict holds SparseTensors of this manager, and
        # manager -> geometry -> SparseTensor -> manager was a reference cycle — every batch's maps, rulebooks and
        # encodings (~190 MB at 150 k voxels) then lived until a generation-2 collection (tools/soak.py: the allocator
        # grew by 40 MB per step for 300 steps)
        x._usc_geometry = geo
        return geo

    def _gather_pos_keys(self, samples, pos_encodings_pcd, geo):
        """samples["pos_keys"][pass] = positional encodings of the pass's sampled voxels, f32[B, K, d] (reference :334-335):
        geometry only, so the scene prefetcher issues these twelve gathers with the key samples."""
        out = []
        for pidx, plan in enumerate(samples["passes"]):
            hlevel = self.hlevels[pidx % self.num_levels]
            per_scene = pos_encodings_pcd[hlevel][0]
            n_scenes, K = len(per_scene), plan["k"]
            if plan["gidx"] is not None and per_scene[0].is_cuda:
                if n_scenes == 1:
                    pos_l = per_scene[0]
                else:
                    cat = geo.setdefault("pos_cat", {})
                    if hlevel not in cat:
                        cat[hlevel] = torch.cat(per_scene)
                    pos_l = cat[hlevel]
                out.append(ops.gather_rows(pos_l.contiguous(), plan["gidx"]).view(n_scenes, K, -1))
            else:
                out.append(_stack([per_scene[k][plan["rand_idx"][k], :] for k in range(n_scenes)]))
        samples["pos_keys"] = out

    def _graph_key(self):
        g = getattr(self, "_graphed_passes", None)
        return None if g is None else id(g)

    def _draw_key_samples(self, coords, is_eval):
        """Per decoder pass (decoder-major, level-minor: the order the reference draws in): the number of keys, the
        per-scene row indices (random subset or everything + padding) and padding masks, the batch-wide row index and
        whether every scene was sub-sampled (reference :306-343)."""
        dev = coords[0].F.device
        level_sizes = [[f.shape[0] for f in lv.decomposed_features] for lv in coords]
        slices = [lv.coordinate_manager.batch_slices(lv._ts()) for lv in coords]
        graphed = getattr(self, "_graphed_passes", None) is not None
        passes = []
        for decoder_counter in range(self.num_decoders):
            for i, hlevel in enumerate(self.hlevels):
                sizes = level_sizes[hlevel]
                curr = max(sizes)
                if not (self.max_sample_size or is_eval):
                    curr = min(curr, self.sample_sizes[hlevel])
                if graphed:
                    want = self._graph_shapes[decoder_counter * self.num_levels + i][2][0]
                    if curr < want:
                        # a level smaller than the captured key count: pad up to it (row 0, masked like the padding of
                        # a ragged batch) so the captured pass still applies.  Masked keys carry softmax weight 0: the
                        # pass output is that of the unpadded keys, and the all-masked-row rule is unchanged because
                        # the padding repeats row 0's mask bits.
                        curr = want
                rand_idx, mask_idx = [], []
                for pcd_size in sizes:
                    if pcd_size <= curr:              # take everything, pad with row 0 and mask the padding
                        idx, midx = _padded_index(pcd_size, curr, dev)
                    else:                             # random subset, nothing masked
                        idx = self.randperm(pcd_size, dev)[:curr]
                        midx = _padded_index(curr, curr, dev)[1]
                    rand_idx.append(idx)
                    mask_idx.append(midx)
                gidx = None
                if all(isinstance(sl, slice) for sl in slices[hlevel]):      # scenes are contiguous row ranges
                    gidx = rand_idx[0] if len(sizes) == 1 else torch.cat(
                        [rand_idx[k] + slices[hlevel][k].start for k in range(len(sizes))])
                passes.append({"k": curr, "rand_idx": rand_idx, "mask_idx": mask_idx, "gidx": gidx,
                               "all_sampled": all(n > curr for n in sizes), "sizes": sizes})
        return {"passes": passes, "is_eval": bool(is_eval), "graph_key": self._graph_key()}

    def forward(self, x, point2segment=None, raw_coordinates=None, is_eval=False, num_segments=None):
        """`num_segments` (optional, one int per scene): the number of segments = point2segment.max() + 1 when the
        caller already knows it on the host (the collate does: it relabels the segments with torch.unique).  Without
        it the count is read back from the device here, which makes the host wait for the whole backbone forward
        before it can issue the decoder."""
        pcd_features, aux = self.backbone(x)
        n_scenes = len(x.decomposed_coordinates)

        geo = getattr(x, "_usc_geometry", None)
        if geo is None or geo.get("n_levels") != len(aux):
            geo = self.precompute_geometry(x, raw_coordinates, point2segment, num_segments, n_levels=len(aux),
                                           is_eval=is_eval)
        coordinates, coords, pos_encodings_pcd = geo["coordinates"], geo["coords"], geo["pos_encodings_pcd"]

        mask_features = self.mask_features_head(pcd_features)
        mask_segments, seg_csr = None, None
        if self.train_on_segments:
            seg_csr = geo["seg_csr"]
            if self.scatter_type == "mean":
                mask_segments = [ops.segment_mean(f, csr) for f, csr in zip(mask_features.decomposed_features, seg_csr)]
            else:
                # scatter_type 'max' (reference :66-67, :223: torch_scatter.scatter_max(...)[0]; not used by the shipped
                # configs, conf/model/mask3d.yaml:31): per-segment channel maximum, empty segments 0, the gradient goes to
                # the maximal row — plain device tensor ops
                mask_segments = [_segment_max(f, csr.seg, csr.S) for f, csr in zip(mask_features.decomposed_features, seg_csr)]

        sampled_coords = None
        if self.non_parametric_queries:
            fps_idx, sampled_coords = geo["fps_idx"], geo["sampled_coords"]
            query_pos = self.query_projection(geo["query_pos_enc"])
            if self.use_np_features:
                queries = _stack([pcd_features.decomposed_features[i][fps_idx[i]] for i in range(n_scenes)])
                queries = self.np_feature_projection(queries)
            else:
                queries = torch.zeros_like(query_pos).permute(0, 2, 1)
            # [Q, B, d] materialised ONCE: every pass adds it to its queries / keys (an eager pass would otherwise make
            # its own contiguous copy three times, 36 small copies per step)
            query_pos = query_pos.permute(2, 0, 1).contiguous()
        elif self.random_queries:
            query_pos = torch.rand(n_scenes, self.mask_dim, self.num_queries, device=x.device) - 0.5
            queries = torch.zeros_like(query_pos).permute(0, 2, 1)
            query_pos = query_pos.permute(2, 0, 1)
        elif self.random_query_both:
            shape = (n_scenes, 2 * self.mask_dim, self.num_queries)
            qpf = torch.randn(*shape, device=x.device) if self.random_normal else torch.rand(*shape, device=x.device) - 0.5
            queries = qpf[:, :self.mask_dim, :].permute(0, 2, 1)
            query_pos = qpf[:, self.mask_dim:, :].permute(2, 0, 1)
        else:
            queries = self.query_feat.weight.unsqueeze(0).repeat(n_scenes, 1, 1)
            query_pos = self.query_pos.weight.unsqueeze(1).repeat(1, n_scenes, 1)

        samples = geo.pop("key_samples", None)         # drawn with the geometry (prefetch stream), used once
        if samples is None or samples["is_eval"] != bool(is_eval) or samples["graph_key"] != self._graph_key():
            samples = self._draw_key_samples(coords, is_eval)
        if samples.get("pos_keys") is None:
            self._gather_pos_keys(samples, pos_encodings_pcd, geo)

        predictions_class, predictions_mask = [], []
        p2s_arg = point2segment if self.train_on_segments else None
        sinks = {}      # one gradient buffer per backbone level for the num_decoders key samples taken from it
        # The keys and values of a pass — sampled voxel rows -> lin_squeeze (+ level embedding) -> key / value projections
        # (reference :306-354, :547-605) — depend on the backbone's output and the sampled indices, not on the queries:
        # they are issued on a SIDE stream, pass by pass, and run beside the query chain (mask module -> attention mask ->
        # pass), forward and backward (autograd runs a node's backward on the stream of its forward and orders the
        # streams itself).  What stays on the chain per pass: the thresholded attention-mask rows.
        use_side = _KV_SIDE_STREAM and x.F.is_cuda
        main = torch.cuda.current_stream() if x.F.is_cuda else None
        side = self._side_stream(x.device) if use_side else None
        if use_side:
            ready = torch.cuda.Event()
            ready.record(main)                 # backbone features exist; the previous step's backward is behind this point too
            side.wait_event(ready)
        for decoder_counter in range(self.num_decoders):
            dec = 0 if self.shared_decoder else decoder_counter
            for i, hlevel in enumerate(self.hlevels):
                pidx = decoder_counter * self.num_levels + i
                plan = samples["passes"][pidx]
                decomposed_aux = aux[hlevel].decomposed_features
                sizes = [f.shape[0] for f in decomposed_aux]
                if min(sizes) == 1:
                    raise RuntimeError(SINGLE_POINT_ERROR)
                if plan["sizes"] != sizes:
                    raise RuntimeError(f"key samples were drawn for level sizes {plan['sizes']}, the backbone produced "
                                       f"{sizes}")
                curr_sample_size, rand_idx, mask_idx = plan["k"], plan["rand_idx"], plan["mask_idx"]
                n_valid = [min(n, curr_sample_size) for n in sizes]
                graph_shapes = (self._graph_shapes[pidx] if getattr(self, "_graphed_passes", None) is not None else None)
                step_fn = self._decoder_pass(decoder_counter, dec, i)
                bufs = None
                d_model = self.mask_dim
                if graph_shapes is not None:
                    have = (queries.shape, query_pos.shape, (curr_sample_size, n_scenes, d_model),
                            (curr_sample_size, n_scenes, d_model), (n_scenes, curr_sample_size, self.num_queries))
                    if tuple(tuple(h) for h in have) != tuple(tuple(w) for w in graph_shapes):
                        step_fn = self._eager_pass(dec, i)
                    elif _GATHER_INTO_GRAPH_INPUTS and plan["gidx"] is not None:
                        bufs = step_fn.input_buffers    # keys / values / mask written straight into the captured pass's inputs
                feats_l = aux[hlevel].F
                fused = (_FUSED_KEY_SAMPLING and plan["gidx"] is not None and feats_l.is_cuda
                         and feats_l.dtype == torch.float32 and n_scenes <= 16 and feats_l.shape[1] % 4 == 0
                         and self.num_queries <= 128)

                # ---- query-independent half (side stream): rows of the level's features -> keys, values
                # (the non-fused gather indexes per-scene slices that a compute-stream kernel may only just have produced —
                #  after `ready` — and that no stream record covers: it stays on the compute stream; round-5 advice)
                on_side = use_side and fused
                if on_side:
                    # geometry tensors come from the prefetch stream's pool and are released as soon as the host has
                    # ISSUED their last reader (autograd drops a node's saved tensors after running it): the allocator
                    # must know that this stream reads them too, or the next scene's prefetch overwrites the indices
                    # under a backward kernel still in flight here (seen as a memory access fault)
                    for t in (plan["gidx"], samples["pos_keys"][pidx], feats_l):
                        if t is not None and t.is_cuda:
                            t.record_stream(side)
                with (torch.cuda.stream(side) if on_side else contextlib.nullcontext()):
                    if fused:
                        batched_aux = ops.sample_keys(
                            feats_l.contiguous(), None, None, plan["gidx"], n_scenes, curr_sample_size, n_valid,
                            unique=plan["all_sampled"], valid_unique=True,       # (the plan's keys: distinct rows, then masked padding)
                            sink=sinks.setdefault((hlevel, bool(plan["all_sampled"])), ops.GradSink()) if _GRAD_SINKS else None)
                    else:
                        batched_aux = _stack([ops.gather_rows(decomposed_aux[k].contiguous(), rand_idx[k],
                                                              unique=sizes[k] > curr_sample_size) for k in range(n_scenes)])
                    k_keys, v_keys = self._key_prep(dec, i)(batched_aux, samples["pos_keys"][pidx],
                                                            outs=None if bufs is None else (bufs[2], bufs[3]))
                    if on_side:
                        kv_done = torch.cuda.Event()
                        kv_done.record(side)
                        if bufs is None:               # fresh tensors of the side stream's pool, read on the compute stream
                            k_keys.record_stream(main)
                            v_keys.record_stream(main)

                # ---- the query chain (compute stream)
                normed, queries = self._norm_queries(queries)
                output_class, outputs_mask, attn_mask = self.mask_module(
                    queries, mask_features, mask_segments, len(aux) - hlevel - 1, ret_attn_mask=True,
                    point2segment=p2s_arg, coords=coords, defer_class=True, normed=normed)
                decomposed_attn = attn_mask.decomposed_features
                if fused and attn_mask.F.dtype == torch.bool and attn_mask.F.shape[1] <= 128:
                    # the mask rows, the all-masked-query rule (reference :346) and the padding mask (:343): two launches
                    batched_attn = ops.sample_keys(None, attn_mask.F.contiguous(), None, plan["gidx"], n_scenes,
                                                   curr_sample_size, n_valid,
                                                   outs=None if bufs is None else (None, bufs[4], None))
                else:
                    batched_attn = _stack([decomposed_attn[k][rand_idx[k], :] for k in range(n_scenes)])
                    # a query whose sampled keys are all masked attends to everything (reference :346)
                    batched_attn.permute(0, 2, 1)[batched_attn.sum(1) == curr_sample_size] = False
                    if not plan["all_sampled"]:        # (every scene sampled: no padding rows to mask)
                        batched_attn = torch.logical_or(batched_attn, _stack(mask_idx)[..., None])
                    if bufs is not None:
                        bufs[4].copy_(batched_attn)
                        batched_attn = bufs[4]

                rec = getattr(self, "attn_mask_record", None)
                if rec is not None:          # parity tests: the thresholded masks are discrete decisions
                    rec.append(batched_attn.detach().clone())
                if on_side:
                    main.wait_event(kv_done)
                queries = step_fn(queries, query_pos, k_keys, v_keys, batched_attn.contiguous())

                predictions_class.append(output_class)
                predictions_mask.append(outputs_mask)

        output_class, outputs_mask = self.mask_module(queries, mask_features, mask_segments, 0, ret_attn_mask=False,
                                                      point2segment=p2s_arg, coords=coords, defer_class=True)
        predictions_class.append(output_class)
        predictions_mask.append(outputs_mask)
        # the class head over every call's normalised queries at once (reference :428 per call)
        predictions_class = list(self.class_embed_head(torch.stack(predictions_class)).unbind(0))

        return {
            "pred_logits": predictions_class[-1],
            "pred_masks": predictions_mask[-1],
            "aux_outputs": self._set_aux_loss(predictions_class, predictions_mask),
            "sampled_coords": _host_array(sampled_coords) if sampled_coords is not None else None,
            "backbone_features": pcd_features,
        }

    def _norm_queries(self, queries):
        """decoder_norm(queries) for mask_module AND the queries for the next decoder layer as outputs of ONE autograd
        node: the gradient that comes back from the layer is then summed inside the norm's backward launch
        (ops.layer_norm(passthrough=True)) instead of by an autograd add per pass -> (normed, queries)."""
        norm = self.decoder_norm
        if (_LN_PASSTHROUGH and queries.is_cuda and queries.dtype == torch.float32 and queries.requires_grad
                and isinstance(norm, LayerNorm) and norm.elementwise_affine and norm.bias is not None
                and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] in ops._LN_DIMS):
            return ops.layer_norm(queries, norm.weight, norm.bias, norm.eps, passthrough=True)
        return norm(queries), queries

    def mask_module(self, query_feat, mask_features, mask_segments, num_pooling_steps, ret_attn_mask=True,
                    point2segment=None, coords=None, defer_class=False, normed=None):
        query_feat = self.decoder_norm(query_feat) if normed is None else normed
        head = self.mask_embed_head
        Q = query_feat.shape[-2]
        q_pad = (-Q) % 32
        if query_feat.is_cuda and query_feat.dtype == torch.float32:      # Linear + ReLU in one launch
            if defer_class and _RESIDUAL_IN_PROJECTION and query_feat.requires_grad:
                # the normalised queries feed the mask head AND the class head: routed through the first projection's
                # node, the class head's gradient is summed inside that projection's input-gradient launch
                hidden, query_feat = ops.linear(query_feat, head[0].weight, head[0].bias, relu=True, passthrough=True)
            else:
                hidden = ops.linear(query_feat, head[0].weight, head[0].bias, relu=True)
            if (q_pad and _PADDED_MASK_EMBED and query_feat.dim() == 3 and query_feat.shape[0] == 1
                    and isinstance(head[2], Linear) and head[2].out_features % 32 == 0):
                # one scene: the embeddings come out zero-extended to a multiple of 32 rows (what the logits product
                # below wants) from the launch that computes them
                mask_embed = ops.linear(hidden, head[2].weight, head[2].bias, pad_rows_to=Q + q_pad)
                mask_embed = _PaddedRows(mask_embed, Q)
            else:
                mask_embed = head[2](hidden)
        else:
            mask_embed = head(query_feat)
        # defer_class: hand back the normalised queries; forward() runs the class head ONCE over all 13 calls' queries
        # (row-wise linear: same numbers; 12 fewer head launches and 24 fewer gradient accumulations per step)
        outputs_class = query_feat if defer_class else self.class_embed_head(query_feat)

        output_masks, output_segments = [], []
        if (point2segment is not None and ret_attn_mask and num_pooling_steps >= 1 and query_feat.is_cuda
                and _FUSED_ATTN_MASK):
            # the per-voxel logits are rows of the [segments, Q] logits; the first pooling step reads them through
            # point2segment (no [voxels, Q] table: 59 MB written and read back per call at 150 k voxels) and the last
            # one applies sigmoid < 0.5 (reference :418-436).  Several scenes: the segment tables are stacked and the
            # row index carries each scene's segment offset (the voxel rows of a batch are one table already).
            for i, seg_feat in enumerate(mask_segments):
                if _CHAIN_SEGMENT_GRADS:
                    logits, mask_segments[i] = _mask_logits(seg_feat, mask_embed[i], chain=True)   # (the caller's list)
                else:
                    logits = _mask_logits(seg_feat, mask_embed[i])
                output_segments.append(logits)
            cm, ts = mask_features.coordinate_manager, mask_features._ts()
            if len(output_segments) == 1:
                pooled = output_segments[0].detach()
                rows = point2segment[0].to(torch.int64).contiguous()
            else:
                Q = output_segments[0].shape[1]
                pooled = torch.cat([getattr(o, "_usc_padded", o).detach() for o in output_segments])[:, :Q]
                rows = getattr(cm, "_usc_p2s_batched", None)
                if rows is None or rows.shape[0] != mask_features.F.shape[0]:
                    off, parts = 0, []
                    for p2s, seg in zip(point2segment, mask_segments):
                        parts.append(p2s.to(torch.int64) + off)
                        off += seg.shape[0]
                    rows = torch.cat(parts).contiguous()
                    cm._usc_p2s_batched = rows          # geometry only: built once per batch
            for step in range(num_pooling_steps):
                if step == 0:
                    # child table -> segment rows, folded once per batch (geometry only): the kernel then reads
                    # child -> logits row in two dependent loads instead of three, in all 12 calls of the step
                    pooled = ops.avgpool_down2(pooled, _child_segment_table(cm, ts, rows),
                                               threshold=num_pooling_steps == 1)
                else:
                    pooled = ops.avgpool_down2(pooled, cm.stride_map(ts)["nbr2"],
                                               threshold=step == num_pooling_steps - 1)
                ts *= 2
            attn_mask = me.SparseTensor(features=pooled, coordinate_manager=cm,
                                        coordinate_map_key=ME.CoordinateMapKey(ts))
            return outputs_class, output_segments, attn_mask
        if point2segment is not None:
            for i, seg_feat in enumerate(mask_segments):
                output_segments.append(_mask_logits(seg_feat, mask_embed[i]))
                if ret_attn_mask:   # per-voxel logits only feed the (detached) attention masks
                    with torch.no_grad():
                        output_masks.append(ops.gather_rows(output_segments[-1].detach().contiguous(),
                                                            point2segment[i].to(torch.int64).contiguous()))
        else:
            per_scene = mask_features.decomposed_features
            for i in range(len(per_scene)):
                output_masks.append(_mask_logits(per_scene[i], mask_embed[i]))

        if ret_attn_mask:
            attn_mask = me.SparseTensor(features=torch.cat(output_masks).detach(),
                                        coordinate_manager=mask_features.coordinate_manager,
                                        coordinate_map_key=mask_features.coordinate_map_key)
            for _ in range(num_pooling_steps):
                attn_mask = self.pooling(attn_mask.float())
            attn_mask = me.SparseTensor(features=(attn_mask.F.detach().sigmoid() < 0.5),
                                        coordinate_manager=attn_mask.coordinate_manager,
                                        coordinate_map_key=attn_mask.coordinate_map_key)
            if point2segment is not None:
                return outputs_class, output_segments, attn_mask
            return outputs_class, output_masks, attn_mask
        if point2segment is not None:
            return outputs_class, output_segments
        return outputs_class, output_masks

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_seg_masks):
        return [{"pred_logits": a, "pred_masks": b} for a, b in zip(outputs_class[:-1], outputs_seg_masks[:-1])]


class HostArrayLater:
    """The query seed coordinates as the reference returns them — a numpy array (models/mask3d.py:467) — without making
    the host wait for the whole forward pass: the device-to-host copy goes into pinned memory behind the forward's
    kernels and is waited for when somebody LOOKS at the array (np.asarray, indexing, .numpy()).  A blocking copy
    here was the one host/device synchronisation left on the training step's compute stream: the host could not issue
    the criterion and the backward pass until the device had drained, and the device then idled while they were being
    issued."""

    def __init__(self, t):
        t = t.detach()
        self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self._host.copy_(t, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self.shape, self.dtype = tuple(t.shape), self._host.numpy().dtype

    def numpy(self):
        self._event.synchronize()
        return self._host.numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, i):
        return self.numpy()[i]

    def __len__(self):
        return self.shape[0]


def _host_array(t):
    if t.is_cuda and _LAZY_HOST_COPIES:
        return HostArrayLater(t)
    return t.detach().cpu().numpy()


class Linear(nn.Linear):
    """nn.Linear whose few-row inputs (the 100 queries) run on the wave-per-tile kernels of csrc/decoder.hip."""

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            return ops.linear(x, self.weight, self.bias)
        return super().forward(x)


class _PaddedRows:
    """mask_embed[i] -> (the zero-extended [Qp, d] table of scene i, Q)."""

    def __init__(self, table, q):
        self.table, self.q = table, q

    def __getitem__(self, i):
        t = self.table
        if t.shape[0] == 1 and i == 0:
            # a view, not a select: SelectBackward would zero-fill a [1, Qp, d] tensor and copy the gradient into it
            # (two stock launches per mask_module call)
            return (t.view(t.shape[1], t.shape[2]), self.q)
        return (t[i], self.q)


def _mask_logits(feats, mask_embed, chain=False):
    """feats [S, d] @ mask_embed[Q, d]^T -> [S, Q] (reference mask3d.py:425,430).  On the device the Q query
    embeddings are padded to a multiple of 32 so that the product runs on this library's row GEMM kernels (forward,
    d feats, d mask_embed); the padded columns are cut off again as a view.
    chain: -> (logits, feats'), feats' = `feats` as a second output of the product's autograd node: the NEXT consumer
    of the segment features takes feats', and its gradient is then summed inside this product's input-gradient launch
    (the segment table has 13 consumers per step: 12 autograd adds of [S, d] otherwise)."""
    if isinstance(mask_embed, tuple):          # already zero-extended by the producing launch (_PaddedRows)
        W, Q = mask_embed
        pad = W.shape[0] - Q
        if not (feats.is_cuda and feats.dtype == torch.float32 and feats.shape[1] % 32 == 0):
            out = feats @ W[:Q].T
            return (out, feats) if chain else out
    else:
        Q = mask_embed.shape[0]
        if not (feats.is_cuda and feats.dtype == torch.float32 and feats.shape[1] % 32 == 0):
            out = feats @ mask_embed.T
            return (out, feats) if chain else out
        pad = (-Q) % 32
        W = F.pad(mask_embed, (0, 0, 0, pad)) if pad else mask_embed
    nxt = feats
    if chain and feats.requires_grad and torch.is_grad_enabled():
        out, nxt = ops.linear(feats, W.contiguous(), passthrough=True)
    else:
        out = ops.linear(feats, W.contiguous())
    if pad:
        view = out[:, :Q]
        view._usc_padded = out        # the device criterion reads (and differentiates) the padded table directly
        out = view
    return (out, nxt) if chain else out


def multihead_attention(mha: nn.MultiheadAttention, query, key, value, attn_mask=None, mask_bsl=None, pos_q=None,
                        pos_k=None, residual=False):
    """nn.MultiheadAttention.forward(query + pos_q, key + pos_k, value, attn_mask=…, need_weights=False)[0] for the
    sequence-first layout, dropout 0 and a boolean mask (True = masked), with the input / output projections
    through ops.in_proj / ops.linear (same parameters, same state_dict); the positional adds of the reference
    (`with_pos_embed`, :485 / :517) happen inside the projection launches."""
    L, B, E = query.shape
    S = key.shape[0]
    H = mha.num_heads
    hd = E // H
    # residual: -> (output, query'), query' = `query` routed through the projection node so that the gradient of the
    # block's residual connection is summed inside the projection's input-gradient launch (no separate add)
    res = None
    if residual:
        q, k, v, res = ops.in_proj(query, key, value, mha.in_proj_weight, mha.in_proj_bias, pos_q=pos_q, pos_k=pos_k,
                                   residual=True)
    else:
        q, k, v = ops.in_proj(query, key, value, mha.in_proj_weight, mha.in_proj_bias, pos_q=pos_q, pos_k=pos_k)
    done = (lambda o: (o, res)) if residual else (lambda o: o)
    if mask_bsl is not None and hd == 16 and L <= 128:
        # `mask_bsl` = the decoder's bool[B, S, L] mask (same for every head): fused HIP kernels, no score tensor
        out = ops.masked_cross_attention(q, k, v, mask_bsl, H)
        return done(ops.linear(out, mha.out_proj.weight, mha.out_proj.bias))
    if mask_bsl is None and attn_mask is None and hd == 16 and L == S and L <= 128:
        # the decoder's self attention (100 queries): one HIP launch each way instead of the library's fused kernels
        out = ops.self_attention(q, k, v, H)
        return done(ops.linear(out, mha.out_proj.weight, mha.out_proj.bias))
    if mask_bsl is not None:
        attn_mask = mask_bsl.repeat_interleave(H, dim=0).permute(0, 2, 1)
    q = q.reshape(L, B * H, hd).transpose(0, 1).reshape(B, H, L, hd)
    k = k.reshape(S, B * H, hd).transpose(0, 1).reshape(B, H, S, hd)
    v = v.reshape(S, B * H, hd).transpose(0, 1).reshape(B, H, S, hd)
    if attn_mask is None:
        out = F.scaled_dot_product_attention(q, k, v)                      # 100 x 100: the fused kernel is fine
    else:
        # masked cross attention, 100 queries x up to 12 800 keys: softmax(q k^T / sqrt(hd) + mask) v as two batched
        # GEMMs and one softmax.  (The fused memory-efficient kernel that scaled_dot_product_attention selects for
        # contiguous operands with an arbitrary mask measured 122 us forward / 135 us backward per call here —
        # 7.5 ms per training step; this path is the one nn.MultiheadAttention's strided operands fell back to.)
        mask = torch.zeros(attn_mask.shape, dtype=q.dtype, device=q.device).masked_fill_(attn_mask, float("-inf"))
        scores = torch.baddbmm(mask, q.reshape(B * H, L, hd), k.reshape(B * H, S, hd).transpose(1, 2),
                               alpha=1.0 / (hd ** 0.5))
        out = torch.bmm(torch.softmax(scores, dim=-1), v.reshape(B * H, S, hd)).reshape(B, H, L, hd)
    out = out.permute(2, 0, 1, 3).reshape(L, B, E)
    return done(ops.linear(out, mha.out_proj.weight, mha.out_proj.bias))


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict keys) computed by the HIP kernels of csrc/decoder.hip when the
    width allows it; other widths use PyTorch's stock operator."""

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and len(self.normalized_shape) == 1
                and self.normalized_shape[0] in ops._LN_DIMS and self.bias is not None):
            return ops.layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


class _KeyPrep:
    """lin_squeeze (+ level embedding) and the key / value thirds of the cross attention's input projection over the
    sampled voxels of one (decoder, level) pass (reference mask3d.py:351-354 and the k / v part of :547-605): everything
    of a pass that does not depend on the queries.  [B, K, C], [B, K, d] -> keys, values [K, B, d]."""

    def __init__(self, squeeze, cross, level_embed, level):
        self.squeeze, self.cross, self.level_embed, self.level = squeeze, cross, level_embed, level

    def __call__(self, batched_aux, batched_pos, outs=None):
        src = self.squeeze(batched_aux.permute(1, 0, 2))
        if self.level_embed is not None:
            src = src + self.level_embed.weight[self.level]          # reference mask3d.py:353-354
        mha = self.cross.multihead_attn
        pos = batched_pos.permute(1, 0, 2)
        if src.is_cuda and src.dtype == torch.float32:
            return ops.in_proj_kv(src, mha.in_proj_weight, mha.in_proj_bias, pos=pos.contiguous(), outs=outs)
        E = src.shape[-1]
        W, b = mha.in_proj_weight, mha.in_proj_bias
        k, v = F.linear(src + pos, W[E:2 * E], b[E:2 * E]), F.linear(src, W[2 * E:], b[2 * E:])
        if outs is not None:
            outs[0].copy_(k), outs[1].copy_(v)
            k, v = outs
        return k, v


class _DecoderPass(nn.Module):
    """masked cross attention (on prepared keys / values) -> self attention -> FFN of one (decoder, level) pass
    (reference mask3d.py:355-373).  Pure tensor-in / tensor-out with static shapes whenever every scene
    has at least `sample_sizes[hlevel]` voxels at that level, which makes it capturable as a HIP graph
    (Mask3D.enable_decoder_graphs): ~16 forward and ~25 backward launches per pass become one graph launch
    each, removing most of the host launch overhead of the 12 passes.  The keys and values come from _KeyPrep."""

    def __init__(self, cross, self_attn, ffn, num_heads):
        super().__init__()
        self.cross, self.self_attn, self.ffn, self.num_heads = cross, self_attn, ffn, num_heads

    def forward(self, queries, query_pos, k, v, batched_attn):
        out = self.cross.forward_kv(queries.permute(1, 0, 2), k, v, batched_attn, query_pos)
        out = self.self_attn(out, tgt_mask=None, tgt_key_padding_mask=None, query_pos=query_pos)
        return self.ffn(out).permute(1, 0, 2)


_KV_SIDE_STREAM = os.environ.get("USC3D_KV_SIDE_STREAM", "1") != "0"


def set_kv_side_stream(on: bool):
    """Switch the key-preparation stream at run time (streams.recheck_under_collective, tests).  Captured decoder passes
    do not depend on it: the key / value side is issued eagerly, pass by pass, on whichever stream."""
    global _KV_SIDE_STREAM
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _KV_SIDE_STREAM = bool(on)
_FUSED_ATTN_MASK = os.environ.get("USC3D_FUSED_ATTN_MASK", "1") != "0"
_CHAIN_SEGMENT_GRADS = os.environ.get("USC3D_CHAIN_SEGMENT_GRADS", "1") != "0"


_PAD_CACHE = {}


def _padded_index(n, size, device):
    """(idx i64[size] = 0..n-1 then zeros, mask bool[size] = False for the first n, True for the padding): constants
    of (n, size), kept per device instead of being rebuilt from five small fills / copies per scene and pass (the
    callers only read them)."""
    key = (int(n), int(size), str(device))
    hit = _PAD_CACHE.get(key)
    if hit is None:
        if len(_PAD_CACHE) > 256:
            _PAD_CACHE.clear()
        idx = torch.zeros(size, dtype=torch.long, device=device)
        midx = torch.ones(size, dtype=torch.bool, device=device)
        idx[:n] = torch.arange(n, device=device)
        midx[:n] = False
        hit = _PAD_CACHE[key] = (idx, midx)
    return hit


def _child_segment_table(cm, ts, rows):
    """i32[8, n_coarse]: for every coarse voxel the SEGMENT-TABLE rows of its (present) children, -1 where a child is
    absent — `point2segment` folded into the k2/s2 child table of the map at tensor stride ts.  Cached on the coordinate
    manager (one per batch; `rows` is the batch-wide row index into the stacked segment tables)."""
    cache = cm.__dict__.setdefault("_usc_child_segments", {})
    hit = cache.get(ts)
    if hit is None or hit[0] != rows.data_ptr():
        nbr2 = cm.stride_map(ts)["nbr2"]
        tab = torch.where(nbr2 >= 0, rows.to(torch.int32)[nbr2.clamp(min=0).long()], nbr2).contiguous()
        hit = cache[ts] = (rows.data_ptr(), tab, rows)
    return hit[1]


def _segment_max(feats, seg, S):
    out = torch.zeros((int(S), feats.shape[1]), dtype=feats.dtype, device=feats.device)
    return out.scatter_reduce(0, seg.to(torch.int64)[:, None].expand(-1, feats.shape[1]), feats, "amax", include_self=False)


def _stack(tensors):
    """torch.stack, except that ONE tensor (one scene per GPU, the data-parallel configuration) becomes a view instead
    of a copy: ~60 small copy launches per training step."""
    tensors = list(tensors)
    return tensors[0].unsqueeze(0) if len(tensors) == 1 else torch.stack(tensors)


def _col_minmax(x):
    """(x.min(0)[0], x.max(0)[0]) of an [N, 3] tensor: one reduction over contiguous rows of the transpose
    instead of two strided ones (39 us each on 148 k points)."""
    mn, mx = torch.aminmax(x.t().contiguous(), dim=1)
    return mn, mx


def _residual_norm(layer, tgt, upd):
    """tgt + dropout(upd), then the layer's post-norm (reference :493-494, :523-524, :543-544); with dropout 0 on the
    device the add and the LayerNorm are one launch."""
    norm = layer.norm
    if (not layer.normalize_before and (layer.dropout.p == 0.0 or not layer.training) and tgt.is_cuda and tgt.dtype == torch.float32
            and isinstance(norm, LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] in ops._LN_DIMS
            and tgt.shape == upd.shape):
        return ops.add_layer_norm(tgt, upd, norm.weight, norm.bias, norm.eps)
    out = tgt + layer.dropout(upd)
    return out if layer.normalize_before else norm(out)


def _with_pos(t, pos):
    return t if pos is None else t + pos


def _xavier(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class SelfAttentionLayer(nn.Module):
    """Post-norm (default) / pre-norm self attention block (reference :491-545)."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        _xavier(self)

    def forward(self, tgt, tgt_mask=None, tgt_key_padding_mask=None, query_pos=None):
        src = self.norm(tgt) if self.normalize_before else tgt
        # need_weights=False: same output; skips materialising/averaging the [B,Q,K] attention weights the
        # reference computes and discards (`[0]`), and lets PyTorch take its fused SDPA path
        if tgt_key_padding_mask is None and self.self_attn.dropout == 0.0 and src.is_cuda:
            if not self.normalize_before and _RESIDUAL_IN_PROJECTION:
                upd, tgt = multihead_attention(self.self_attn, src, src, src, attn_mask=tgt_mask, pos_q=query_pos,
                                               pos_k=query_pos, residual=True)
            else:
                upd = multihead_attention(self.self_attn, src, src, src, attn_mask=tgt_mask, pos_q=query_pos,
                                          pos_k=query_pos)
        else:
            q = k = _with_pos(src, query_pos)
            upd = self.self_attn(q, k, value=src, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask,
                                 need_weights=False)[0]
        return _residual_norm(self, tgt, upd)


class CrossAttentionLayer(nn.Module):
    """Masked cross attention block (reference :547-605)."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        _xavier(self)

    def forward(self, tgt, memory, memory_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None,
                memory_mask_bsl=None):
        """`memory_mask_bsl`: the same mask as `memory_mask` in the decoder's own bool[B, S, L] layout (not repeated
        per head); when given on a HIP device the fused attention kernels are used."""
        src = self.norm(tgt) if self.normalize_before else tgt
        if memory_mask is None and memory_mask_bsl is not None and not src.is_cuda:
            memory_mask = memory_mask_bsl.repeat_interleave(self.multihead_attn.num_heads, dim=0).permute(0, 2, 1)
        if memory_key_padding_mask is None and self.multihead_attn.dropout == 0.0 and src.is_cuda:
            if not self.normalize_before and _RESIDUAL_IN_PROJECTION:
                upd, tgt = multihead_attention(self.multihead_attn, src, memory, memory, attn_mask=memory_mask,
                                               mask_bsl=memory_mask_bsl, pos_q=query_pos, pos_k=pos, residual=True)
            else:
                upd = multihead_attention(self.multihead_attn, src, memory, memory, attn_mask=memory_mask,
                                          mask_bsl=memory_mask_bsl, pos_q=query_pos, pos_k=pos)
        else:
            upd = self.multihead_attn(query=_with_pos(src, query_pos), key=_with_pos(memory, pos), value=memory,
                                      attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask,
                                      need_weights=False)[0]
        return _residual_norm(self, tgt, upd)


def _attention_on_projected(q, k, v, mask_bsl, H, dropout_p, training):
    """softmax(q k^T / sqrt(hd) + mask) v on already projected q [L,B,E], k / v [S,B,E] (sequence-first), mask bool[B,S,L]
    (True = masked): the fused HIP kernels for head dim 16 and <= 128 queries, else plain tensor ops."""
    L, B, E = q.shape
    S = k.shape[0]
    hd = E // H
    if q.is_cuda and q.dtype == torch.float32 and hd == 16 and L <= 128 and mask_bsl is not None and dropout_p == 0.0:
        return ops.masked_cross_attention(q, k.contiguous(), v.contiguous(), mask_bsl, H)
    qh = q.reshape(L, B * H, hd).transpose(0, 1).reshape(B, H, L, hd)
    kh = k.reshape(S, B * H, hd).transpose(0, 1).reshape(B, H, S, hd)
    vh = v.reshape(S, B * H, hd).transpose(0, 1).reshape(B, H, S, hd)
    scores = (qh @ kh.transpose(-1, -2)) / (hd ** 0.5)
    if mask_bsl is not None:
        scores = scores.masked_fill(mask_bsl.permute(0, 2, 1)[:, None], float("-inf"))
    att = F.dropout(torch.softmax(scores, dim=-1), dropout_p, training)
    return (att @ vh).permute(2, 0, 1, 3).reshape(L, B, E)


def _cross_forward_kv(self, tgt, k, v, mask_bsl, query_pos):
    """CrossAttentionLayer on PREPARED keys / values (ops.in_proj_kv over the sampled voxels, computed apart from the
    query chain): query projection, masked attention, output projection, residual + post-norm (reference :547-605)."""
    mha = self.multihead_attn
    src = self.norm(tgt) if self.normalize_before else tgt
    W, b = mha.in_proj_weight, mha.in_proj_bias
    E = src.shape[-1]
    if src.is_cuda and src.dtype == torch.float32:
        if not self.normalize_before and _RESIDUAL_IN_PROJECTION and (mha.dropout == 0.0 or not self.training):
            q, tgt = ops.in_proj_q(src, W, b, pos=query_pos, residual=True)
        else:
            q = ops.in_proj_q(src, W, b, pos=query_pos)
        out = _attention_on_projected(q, k, v, mask_bsl, mha.num_heads, mha.dropout if self.training else 0.0, self.training)
        upd = ops.linear(out, mha.out_proj.weight, mha.out_proj.bias)
    else:
        q = F.linear(_with_pos(src, query_pos), W[:E], b[:E])
        out = _attention_on_projected(q, k, v, mask_bsl, mha.num_heads, mha.dropout if self.training else 0.0, self.training)
        upd = F.linear(out, mha.out_proj.weight, mha.out_proj.bias)
    return _residual_norm(self, tgt, upd)


class FFNLayer(nn.Module):
    """Feed-forward block (reference :607-651)."""

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm = LayerNorm(d_model)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        _xavier(self)

    def forward(self, tgt):
        src = self.norm(tgt) if self.normalize_before else tgt
        if src.is_cuda and src.dtype == torch.float32 and self.activation is F.relu and (self.dropout.p == 0.0
                                                                                          or not self.training):
            if not self.normalize_before and _RESIDUAL_IN_PROJECTION:
                # ReLU in the same launch; `tgt` comes back through the layer's node: the residual's gradient is summed
                # inside linear1's input-gradient launch
                hidden, tgt = ops.linear(src, self.linear1.weight, self.linear1.bias, relu=True, passthrough=True)
            else:
                hidden = ops.linear(src, self.linear1.weight, self.linear1.bias, relu=True)   # ReLU in the same launch
        else:
            hidden = self.dropout(self.activation(self.linear1(src)))
        return _residual_norm(self, tgt, self.linear2(hidden))


CrossAttentionLayer.forward_kv = _cross_forward_kv


def _get_activation_fn(activation):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
