"""SeqFormer's deformable transformer on the MI355X op (SURVEY.md section 8 row a4 callers, (f) rank 2).

Module tree and parameter names are those of
projects/SeqFormer/seqformer/models/deformable_transformer.py (encoder.layers.N.self_attn,
decoder.layers.N.{cross_attn,self_attn,self_attn_box,...}, level_embed, reference_points), so
reference checkpoints load unchanged.  Behaviour is the reference's; two things are mapped
differently onto the machine:
  * every MSDeformAttn call folds the frame axis into the op batch (one launch per layer,
    vnext_amd/ops/modules/ms_deform_attn.py);
  * the decoder's per-frame box-query self-attention, a Python loop over frames in the reference
    (:291-297), runs as one batched nn.MultiheadAttention call over N*T sequences.
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from ..ops.functions import level_tensors
from ..ops.fused_ffn import ffn_block
from ..ops.fused_norm import add_dropout_norm
from ..ops.modules import MSDeformAttnSeqFormer


def inverse_sigmoid(x, eps=1e-5):
    # projects/SeqFormer/seqformer/util/misc.py:493-497
    x = x.clamp(min=0, max=1)
    x1 = x.clamp(min=eps)
    x2 = (1 - x).clamp(min=eps)
    return torch.log(x1 / x2)


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
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        # src2 = linear2(dropout2(activation(linear1(src)))); norm2(src + dropout3(src2)) -- vnext_amd/ops/fused_ffn.py
        return ffn_block(src, self.linear1, self.activation, self.dropout2, self.linear2, self.dropout3, self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        src2 = self.self_attn(self.with_pos_embed(src, pos), None, reference_points, src, spatial_shapes,
                              level_start_index, padding_mask)
        src = add_dropout_norm(src, src2, self.dropout1, self.norm1)
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
        # self attention of the mask & class queries
        q1 = k1 = self.with_pos_embed(tgt, query_pos)
        tgt2 = self.self_attn(q1.transpose(0, 1), k1.transpose(0, 1), tgt.transpose(0, 1))[0].transpose(0, 1)
        tgt = add_dropout_norm(tgt, tgt2, self.dropout2, self.norm2)

        if tgt_box.dim() == 3:  # first layer: box queries still shared by the frames [N, Q, C]
            q_box = k_box = self.with_pos_embed(tgt_box, query_pos)
            tgt2_box = self.self_attn_box(q_box.transpose(0, 1), k_box.transpose(0, 1),
                                          tgt_box.transpose(0, 1))[0].transpose(0, 1)
            tgt_box = add_dropout_norm(tgt_box, tgt2_box, self.dropout2_box, self.norm2_box)
            box_query = self.with_pos_embed(tgt_box, query_pos)
        else:  # [N, T, Q, C]: every frame attends over its own box queries -- one batched call
            N, nf, num_q, C = tgt_box.shape
            flat = tgt_box.reshape(N * nf, num_q, C)
            pos = None if query_pos is None else query_pos.unsqueeze(1).expand(N, nf, num_q, C).reshape(N * nf, num_q, C)
            q_box = k_box = self.with_pos_embed(flat, pos)
            t2 = self.self_attn_box(q_box.transpose(0, 1), k_box.transpose(0, 1), flat.transpose(0, 1))[0].transpose(0, 1)
            tgt_box = add_dropout_norm(flat, t2, self.dropout2_box, self.norm2_box).view(N, nf, num_q, C)
            box_query = tgt_box if query_pos is None else tgt_box + query_pos.unsqueeze(1)

        tgt2, tgt2_box, sampling_locations, attention_weights = self.cross_attn(
            self.with_pos_embed(tgt, query_pos), box_query, reference_points, src, src_spatial_shapes,
            level_start_index, src_padding_mask)

        if tgt_box.dim() == 3:      # first layer: the shared box queries broadcast over the frames
            tgt_box = self.norm1_box(tgt_box.unsqueeze(1) + self.dropout1_box(tgt2_box))
        else:
            tgt_box = add_dropout_norm(tgt_box, tgt2_box, self.dropout1_box, self.norm1_box)
        tgt_box = self.forward_ffn_box(tgt_box)

        time_weight = F.softmax(self.time_attention_weights(tgt_box), 1)   # softmax over the frames
        tgt2 = (tgt2 * time_weight).sum(1)
        tgt = add_dropout_norm(tgt, tgt2, self.dropout1, self.norm1)
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
        output = tgt
        output_box = tgt
        intermediate, intermediate_box, intermediate_reference_points = [], [], []
        for lid, layer in enumerate(self.layers):
            if reference_points.shape[-1] == 4:
                reference_points_input = reference_points[:, :, :, None] \
                    * torch.cat([src_valid_ratios, src_valid_ratios], -1)[:, None, None]
            else:
                assert reference_points.shape[-1] == 2
                reference_points_input = reference_points[:, :, :, None] * src_valid_ratios[:, None, None]
            output, output_box, _, _ = layer(output, output_box, query_pos, reference_points_input, src,
                                             src_spatial_shapes, src_level_start_index, src_padding_mask)
            if self.bbox_embed is not None:
                tmp = self.bbox_embed[lid](output_box)
                if reference_points.shape[-1] == 4:
                    new_reference_points = (tmp + inverse_sigmoid(reference_points)).sigmoid()
                else:
                    new_reference_points = tmp
                    new_reference_points[..., :2] = tmp[..., :2] + inverse_sigmoid(reference_points)
                    new_reference_points = new_reference_points.sigmoid()
                reference_points = new_reference_points.detach()
            if self.return_intermediate:
                intermediate.append(output)
                intermediate_box.append(output_box)
                intermediate_reference_points.append(reference_points)
        if self.return_intermediate:
            return (torch.stack(intermediate), torch.stack(intermediate_box),
                    torch.stack(intermediate_reference_points), None)
        return output, reference_points


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
        assert query_embed is not None
        src_flatten, mask_flatten, lvl_pos_embed_flatten, spatial_shapes = [], [], [], []
        for lvl, (src, mask, pos_embed) in enumerate(zip(srcs, masks, pos_embeds)):
            bs, nf, c, h, w = src.shape
            spatial_shapes.append((h, w))
            src_flatten.append(src.flatten(3).transpose(2, 3))
            mask_flatten.append(mask.flatten(2))
            lvl_pos_embed_flatten.append(pos_embed.flatten(3).transpose(2, 3) + self.level_embed[lvl].view(1, 1, 1, -1))
        src_flatten = torch.cat(src_flatten, 2)
        mask_flatten = torch.cat(mask_flatten, 2)
        lvl_pos_embed_flatten = torch.cat(lvl_pos_embed_flatten, 2)
        shapes_list = spatial_shapes
        # cached device tensors, tagged as packed (the op's backward then skips the general-path
        # launches) and with their host-side sizes (no device read for the length checks)
        spatial_shapes, level_start_index = level_tensors(shapes_list, src_flatten.device)
        valid_ratios = torch.stack([self.get_valid_ratio(m[:, 0]) for m in masks], 1)

        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, lvl_pos_embed_flatten,
                              mask_flatten, spatial_shapes_list=shapes_list)

        bs, nf, _, c = memory.shape
        query_embed, tgt = torch.split(query_embed, c, dim=1)
        query_embed = query_embed.unsqueeze(0).expand(bs, -1, -1)
        tgt = tgt.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_embed).sigmoid()
        reference_points = reference_points.unsqueeze(1).repeat(1, nf, 1, 1)
        init_reference_out = reference_points
        hs, hs_box, inter_references, inter_samples = self.decoder(
            tgt, reference_points, memory, spatial_shapes, level_start_index, valid_ratios, query_embed, mask_flatten)
        return hs, hs_box, memory, init_reference_out, inter_references, inter_samples, None, valid_ratios
