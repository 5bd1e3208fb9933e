"""SeqFormer's deformable transformer on the MI355X op (SURVEY.md section 8 row a4 callers, (f) rank 2).

Module tree and parameter names are those of
projects/SeqFormer/seqformer/models/deformable_transformer.py (encoder.layers.N.self_attn,
decoder.layers.N.{cross_attn,self_attn,self_attn_box,...}, level_embed, reference_points), so
reference checkpoints load unchanged.  Behaviour is the reference's; two things are mapped
differently onto the machine:
  * every MSDeformAttn call folds the frame axis into the op batch (one launch per layer,
    vnext_amd/ops/modules/ms_deform_attn.py);
  * the decoder's per-frame box-query self-attention, a Python loop over frames in the reference
    (:291-297), runs as one batched call over N*T sequences -- and every self-attention sub-layer as
    `query_self_attention_block` (vnext_amd/ops/self_attention.py: one attention launch for all heads, the output
    projection's bias folded into the LayerNorm pass; the module stays an nn.MultiheadAttention).
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from ..ops.fused_ffn import autocast_once, ffn_block
from ..ops import shadow_weights
from ..ops.fused_norm import add_dropout_norm
from ..ops.decoder_glue import time_weighted_sum
from ..ops.modules import MSDeformAttnSeqFormer
from ..ops.self_attention import query_self_attention_block
from .transformer_common import ReferenceScaler, flatten_levels, inverse_sigmoid, refined_boxes  # noqa: F401  (inverse_sigmoid: re-exported)



def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttnSeqFormer(d_model, n_levels, n_heads, n_points, 'encode')
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

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        # src2 = linear2(dropout2(activation(linear1(src)))); norm2(src + dropout3(src2)) -- vnext_amd/ops/fused_ffn.py
        return ffn_block(src, self.linear1, self.activation, self.dropout2, self.linear2, self.dropout3, self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(self.with_pos_embed(src, pos), None, reference_points, src, spatial_shapes,
                              level_start_index, padding_mask)
        src = add_dropout_norm(src, src2, self.dropout1, self.norm1, r_bias=self.self_attn.output_proj.bias)
        return self.forward_ffn(src)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        # pixel centres / (valid ratio * size), then * valid ratio  (reference :183-196)
        refs = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            H_, W_ = int(H_), int(W_)
            ref_y, ref_x = torch.meshgrid(
                torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            refs.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(refs, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                spatial_shapes_list=None):
        shapes_iter = spatial_shapes_list if spatial_shapes_list is not None else spatial_shapes.tolist()
        reference_points = self.get_reference_points(shapes_iter, valid_ratios, device=src.device)
        output = src
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return output


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4):
        super().__init__()
        self.cross_attn = MSDeformAttnSeqFormer(d_model, n_levels, n_heads, n_points, 'decode')
        self.cross_attn.defer_output_bias = True     # both output projections: added in norm1's / norm1_box's pass (forward below)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.dropout1_box = nn.Dropout(dropout)
        self.norm1_box = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.self_attn_box = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2_box = nn.Dropout(dropout)
        self.norm2_box = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.linear1_box = nn.Linear(d_model, d_ffn)
        self.activation_box = _get_activation_fn(activation)
        self.dropout3_box = nn.Dropout(dropout)
        self.linear2_box = nn.Linear(d_ffn, d_model)
        self.dropout4_box = nn.Dropout(dropout)
        self.norm3_box = nn.LayerNorm(d_model)
        self.time_attention_weights = nn.Linear(d_model, 1)
        shadow_weights.install(self)     # under autocast: this layer's GEMM weights cast once, together

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        return ffn_block(tgt, self.linear1, self.activation, self.dropout3, self.linear2, self.dropout4, self.norm3)

    def forward_ffn_box(self, tgt):
        return ffn_block(tgt, self.linear1_box, self.activation_box, self.dropout3_box, self.linear2_box,
                         self.dropout4_box, self.norm3_box)

    def forward(self, tgt, tgt_box, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None):
        # self attention of the mask & class queries: norm2(tgt + dropout2(self_attn(tgt + pos, tgt + pos, tgt)))
        tgt = query_self_attention_block(tgt, query_pos, self.self_attn, self.dropout2, self.norm2)

        if tgt_box.dim() == 3:  # first layer: box queries still shared by the frames [N, Q, C]
            tgt_box = query_self_attention_block(tgt_box, query_pos, self.self_attn_box, self.dropout2_box, self.norm2_box)
            box_query = self.with_pos_embed(tgt_box, query_pos)
        else:  # [N, T, Q, C]: every frame attends over its own box queries -- one batched call, pos shared by a clip's frames
            N, nf, num_q, C = tgt_box.shape
            tgt_box = query_self_attention_block(tgt_box.reshape(N * nf, num_q, C), query_pos, self.self_attn_box,
                                                 self.dropout2_box, self.norm2_box).view(N, nf, num_q, C)
            box_query = tgt_box if query_pos is None else tgt_box + query_pos.unsqueeze(1)

        # (the decoder's cross attention samples around the BOX queries only; its `query` argument is read for nothing --
        #  ms_deform_attn.py decode_forward, reference :120-217 -- so the reference's `with_pos_embed(tgt, query_pos)` is not evaluated)
        tgt2, tgt2_box, sampling_locations, attention_weights = self.cross_attn(
            tgt, box_query, reference_points, src, src_spatial_shapes, level_start_index, src_padding_mask)

        # (the cross attention returned both projections without their biases: defer_output_bias)
        if tgt_box.dim() == 3:      # first layer: the shared box queries broadcast over the frames
            tgt_box = self.norm1_box(tgt_box.unsqueeze(1) + self.dropout1_box(tgt2_box + self.cross_attn.output_proj_box.bias))
        else:
            tgt_box = add_dropout_norm(tgt_box, tgt2_box, self.dropout1_box, self.norm1_box,
                                       r_bias=self.cross_attn.output_proj_box.bias)
        tgt_box = self.forward_ffn_box(tgt_box)

        # tgt2 = (tgt2 * softmax(time_attention_weights(tgt_box), over the frames)).sum(frames): one launch (ops/decoder_glue.py).
        # The weights of a query sum to one: output_proj's bias passes through unchanged.
        tgt2 = time_weighted_sum(tgt2, self.time_attention_weights(tgt_box))
        tgt = add_dropout_norm(tgt, tgt2, self.dropout1, self.norm1, r_bias=self.cross_attn.output_proj.bias)
        return self.forward_ffn(tgt), tgt_box, sampling_locations, attention_weights


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.bbox_embed = None
        self.class_embed = None

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos=None, src_padding_mask=None):
        """tgt [N, Q, C] (both query streams start from it), reference_points [N, T, Q, 2] ->
        stacked per-layer (mask/class queries [Ld, N, Q, C], box queries [Ld, N, T, Q, C], references [Ld, N, T, Q, 4], the
        layers' box predictions [Ld, N, T, Q, 4] | None), or the last layer's (queries, references) without
        `return_intermediate`.  The box predictions are the refined references BEFORE they are detached: the detector's box
        head evaluates exactly this expression again for its loss (reference: deformable_detr.py:195-213) -- here it takes
        them from this loop instead (six box MLPs, logits and sigmoids less per step)."""
        scaled = ReferenceScaler(src_valid_ratios, extra_axes=2)       # [N, 1, 1, L, 2|4] against [N, T, Q, 1, 2|4]
        src = autocast_once(src)      # every layer's cross attention projects it: one cast under autocast, not one per call
        queries, box_queries = tgt, tgt
        kept = ([], [], [], [])
        for lid, layer in enumerate(self.layers):
            queries, box_queries, _, _ = layer(queries, box_queries, query_pos, scaled(reference_points), src,
                                               src_spatial_shapes, src_level_start_index, src_padding_mask)
            boxes = None
            if self.bbox_embed is not None:
                boxes = refined_boxes(self.bbox_embed[lid](box_queries), reference_points)
                reference_points = boxes.detach()
            if self.return_intermediate:
                for store, item in zip(kept, (queries, box_queries, reference_points, boxes)):
                    store.append(item)
        if not self.return_intermediate:
            return queries, reference_points
        return tuple(torch.stack(store) for store in kept[:3]) + (torch.stack(kept[3]) if self.bbox_embed is not None else None,)


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_frames=1,
                 num_feature_levels=4, dec_n_points=4, enc_n_points=4):
        super().__init__()
        self.d_model = d_model
        self.nhead = nhead
        self.num_feature_levels = num_feature_levels
        encoder_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                          num_feature_levels, nhead, enc_n_points)
        self.encoder = DeformableTransformerEncoder(encoder_layer, num_encoder_layers)
        decoder_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                          num_feature_levels, nhead, dec_n_points)
        self.decoder = DeformableTransformerDecoder(decoder_layer, num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()
        for m in self.modules():     # nothing here reads the modules' locations / weights: fused prologue
            if isinstance(m, MSDeformAttnSeqFormer):
                m.return_samples = False

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttnSeqFormer):
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
        """srcs / pos_embeds per level [N, T, C, H_l, W_l], masks [N, T, H_l, W_l], query_embed [Q, 2C] ->
        (hs, hs_box, memory [N, T, S, C], initial references [N, T, Q, 2], per-layer references, per-layer box predictions
        (with their graph; None without a box head on the decoder), None, valid_ratios)."""
        assert query_embed is not None
        memory_in, padding, pos, shapes_t, start_t, sizes = flatten_levels(srcs, masks, pos_embeds, self.level_embed)
        valid_ratios = torch.stack([self.get_valid_ratio(m[:, 0]) for m in masks], 1)      # of the clip's first frame
        memory = self.encoder(memory_in, shapes_t, start_t, valid_ratios, pos, padding, spatial_shapes_list=sizes)

        clips, frames, channels = memory.shape[0], memory.shape[1], memory.shape[-1]
        # (materialised once: every layer's kernels want dense rows, and an expanded view would be copied per use)
        query_pos = query_embed[:, :channels].unsqueeze(0).expand(clips, -1, -1).contiguous()
        tgt = query_embed[:, channels:].unsqueeze(0).expand(clips, -1, -1).contiguous()
        # one learned reference point per query, the same in every frame of the clip to begin with
        init_reference = self.reference_points(query_pos).sigmoid().unsqueeze(1).repeat(1, frames, 1, 1)
        hs, hs_box, inter_references, inter_boxes = self.decoder(tgt, init_reference, memory, shapes_t, start_t,
                                                                 valid_ratios, query_pos, padding)
        return hs, hs_box, memory, init_reference, inter_references, inter_boxes, None, valid_ratios
