"""IDOL's per-frame deformable transformer on the MI355X op (SURVEY.md section 8 row a5 callers).

Module tree and parameter names are those of
projects/IDOL/idol/models/deformable_transformer.py (encoder.layers.N.self_attn,
decoder.layers.N.{cross_attn,self_attn,...}, level_embed, reference_points) so reference
checkpoints load unchanged; `two_stage` is never enabled by IDOL's configs and is not built.
Frames are independent here (no clip axis): the op batch is the number of frames in the call
-- key + reference frames in training, up to BATCH_INFER_LEN frames in inference.

Not reproduced: the decoder's top-30 sample keeper (:352-358).  Its output (`inter_samples`)
is returned by the reference transformer and dropped by every caller
(segmentation_condInst.py:142-143, :281); computing it costs a 128-wide top-k per query per
layer.  `return_samples=True` brings it back for callers that want it.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from ..ops.fused_ffn import autocast_once, ffn_block
from ..ops import shadow_weights
from ..ops.fused_norm import add_dropout_norm
from ..ops.self_attention import query_self_attention_block
from ..ops.modules import MSDeformAttnIDOL
from .seqformer_transformer import DeformableTransformerEncoder as _ClipEncoder
from .seqformer_transformer import _get_activation_fn, _get_clones
from .transformer_common import ReferenceScaler, flatten_levels, refined_boxes


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttnIDOL(d_model, n_levels, n_heads, n_points)
        self.self_attn.defer_output_bias = True      # added in norm1's pass (forward below)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        shadow_weights.install(self)     # under autocast: this layer's GEMM weights cast once, together

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        q = src if pos is None else src + pos
        src2 = self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask)[0]
        src = add_dropout_norm(src, src2, self.dropout1, self.norm1, r_bias=self.self_attn.output_proj.bias)
        # norm2(src + dropout3(linear2(dropout2(activation(linear1(src)))))) -- vnext_amd/ops/fused_ffn.py
        return ffn_block(src, self.linear1, self.activation, self.dropout2, self.linear2, self.dropout3, self.norm2)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    get_reference_points = staticmethod(_ClipEncoder.get_reference_points)   # same construction (:249-261)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                spatial_shapes_list=None):
        shapes = spatial_shapes_list if spatial_shapes_list is not None else spatial_shapes.tolist()
        reference_points = self.get_reference_points(shapes, valid_ratios, device=src.device)
        for layer in self.layers:
            src = layer(src, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return src


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttnIDOL(d_model, n_levels, n_heads, n_points)
        self.cross_attn.defer_output_bias = True     # added in norm1's pass (forward below)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        shadow_weights.install(self)     # under autocast: this layer's GEMM weights cast once, together

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None):
        # norm2(tgt + dropout2(self_attn(tgt + pos, tgt + pos, tgt))): vnext_amd/ops/self_attention.py
        tgt = query_self_attention_block(tgt, query_pos, self.self_attn, self.dropout2, self.norm2)
        tgt2, loc, w = self.cross_attn(tgt if query_pos is None else tgt + query_pos, reference_points, src,
                                       src_spatial_shapes, level_start_index, src_padding_mask)
        tgt = add_dropout_norm(tgt, tgt2, self.dropout1, self.norm1, r_bias=self.cross_attn.output_proj.bias)
        return ffn_block(tgt, self.linear1, self.activation, self.dropout3, self.linear2, self.dropout4, self.norm3), loc, w


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False, return_samples=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.return_samples = return_samples
        self.bbox_embed = None      # installed by the detector for iterative box refinement
        self.class_embed = None

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None):
        """tgt [N, Q, C], reference_points [N, Q, 2] -> stacked per-layer (queries [Ld, N, Q, C], references [Ld, N, Q, 4],
        kept sampling points [Ld, N, Q, 30, 2] | None, the layers' box predictions [Ld, N, Q, 4] | None), or the last layer's
        (queries, references).  The box predictions are the refined references before they are detached: what the detector's
        box head would compute again (projects/IDOL/idol/models/deformable_detr.py:214-232; see seqformer_transformer.py)."""
        scaled = ReferenceScaler(src_valid_ratios, extra_axes=1)       # [N, 1, L, 2|4] against [N, Q, 1, 2|4]
        src = autocast_once(src)      # every layer's cross attention projects it: one cast under autocast, not one per call
        unscale = src_valid_ratios[:, None, None, None, :, :]
        queries = tgt
        kept, kept_refs, kept_samples, kept_boxes = [], [], [], []
        for lid, layer in enumerate(self.layers):
            queries, loc, w = layer(queries, query_pos, scaled(reference_points), src, src_spatial_shapes,
                                    src_level_start_index, src_padding_mask)
            if self.return_samples:   # the 30 heaviest sampling points of every query, in unpadded image coordinates
                flat = (loc / unscale).flatten(2, 4)
                heaviest = w.flatten(2).topk(30, dim=2)[1]
                kept_samples.append(torch.gather(flat, 2, heaviest.unsqueeze(-1).expand(-1, -1, -1, 2)))
            if self.bbox_embed is not None:
                boxes = refined_boxes(self.bbox_embed[lid](queries), reference_points)
                reference_points = boxes.detach()
                kept_boxes.append(boxes)
            if self.return_intermediate:
                kept.append(queries)
                kept_refs.append(reference_points)
        if not self.return_intermediate:
            return queries, reference_points
        return (torch.stack(kept), torch.stack(kept_refs), (torch.stack(kept_samples) if kept_samples else None),
                (torch.stack(kept_boxes) if kept_boxes else None))


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_frames=1,
                 num_feature_levels=4, dec_n_points=4, enc_n_points=4, two_stage=False, return_samples=False):
        super().__init__()
        if two_stage:
            raise NotImplementedError("two_stage is never enabled by IDOL's configs")
        self.d_model, self.nhead, self.two_stage = d_model, nhead, False
        self.num_frames = 1
        self.num_feature_levels = num_feature_levels
        enc = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels,
                                                nhead, enc_n_points)
        self.encoder = DeformableTransformerEncoder(enc, num_encoder_layers)
        dec = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation, num_feature_levels,
                                                nhead, dec_n_points)
        self.decoder = DeformableTransformerDecoder(dec, num_decoder_layers, return_intermediate_dec, return_samples)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()
        for m in self.encoder.modules():   # the encoder never reads locations / weights: fused prologue
            if isinstance(m, MSDeformAttnIDOL):
                m.return_samples = False
        if not return_samples:             # nor does the decoder, unless the sample keeper is on
            for m in self.decoder.modules():
                if isinstance(m, MSDeformAttnIDOL):
                    m.return_samples = False

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttnIDOL):
                m._reset_parameters()
        xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        constant_(self.reference_points.bias.data, 0.)
        normal_(self.level_embed)

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def forward(self, srcs, masks, pos_embeds, query_embed=None):
        """srcs: per level [N, C, H_l, W_l]; -> (hs [Ld, N, Q, C], memory [N, S, C], init_reference
        [N, Q, 2], inter_references [Ld, N, Q, 4], inter_samples | None, the layers' box predictions (with their graph) | None,
        None)  (:135-198)"""
        assert query_embed is not None
        memory_in, padding, pos, shapes_t, start_t, sizes = flatten_levels(srcs, masks, pos_embeds, self.level_embed)
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        memory = self.encoder(memory_in, shapes_t, start_t, valid_ratios, pos, padding, spatial_shapes_list=sizes)
        images, channels = memory.shape[0], memory.shape[-1]
        query_pos = query_embed[:, :channels].unsqueeze(0).expand(images, -1, -1)
        tgt = query_embed[:, channels:].unsqueeze(0).expand(images, -1, -1)
        init_reference = self.reference_points(query_pos).sigmoid()
        hs, inter_references, inter_samples, inter_boxes = self.decoder(tgt, init_reference, memory, shapes_t, start_t,
                                                                        valid_ratios, query_pos, padding)
        return hs, memory, init_reference, inter_references, inter_samples, inter_boxes, None
