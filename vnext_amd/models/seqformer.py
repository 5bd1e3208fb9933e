"""`SeqFormer` meta-architecture on the MI355X hot path (b).

Registered under the reference's name on the META_ARCH_REGISTRY surface
(projects/SeqFormer/seqformer/seqformer.py:74-75) with the same construction contract
`SeqFormer(cfg)` and the same `forward(batched_inputs)` I/O: a list of dicts with an "image"
list of T frames each; eval returns {"image_size", "pred_scores", "pred_labels", "pred_masks"}
(:403-408), training returns a dict of already-weighted losses (:221-225).

What is inside follows the reference's model tree (`detr` = CondInst_segm {`detr` =
DeformableDETR {transformer, class_embed, bbox_embed, query_embed, input_proj, backbone},
controller, mask_head}) where the hot path lives; what SURVEY section 2 marks out
of scope is deliberately thin:
  * backbone: a plain-PyTorch ResNet-50 trunk (convolutions run on MIOpen), frozen BN;
  * training loss: the reference's objective (vnext_amd/models/criterion.py: clip-level
    Hungarian matching, focal / L1 / GIoU / mask focal + dice with deep supervision), with the
    dynamic mask head of ALL decoder layers' matched instances on ALL frames in one fused
    forward launch and one fused backward launch (the reference: 6 layers x T frames x
    [MaskHeadSmallConv + repeat/cat + 3 grouped convs + pads/interpolate], and their autograd
    twins).  Clips without annotations ("instances" absent) train on an empty target set.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..heads import dynamic_mask_head, dynamic_mask_with_coords
from .criterion import HungarianMatcher, SetCriterion, box_xyxy_to_cxcywh
from ..registry import META_ARCH_REGISTRY
from .seqformer_transformer import DeformableTransformer, inverse_sigmoid


_SCALES = {}


def scale_tensor(values, device):
    """A small constant tensor (image width / height factors) on `device`, built once per value
    and reused: `torch.tensor([...], device=cuda)` is a synchronous pageable host->device copy
    (~0.4 ms on this box) and the training step made ~20 of them."""
    key = (tuple(float(v) for v in values), str(device))
    t = _SCALES.get(key)
    if t is None:
        if len(_SCALES) > 256:
            _SCALES.clear()
        t = _SCALES[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
    return t


# TRAINING steps run the ResNet trunk in channels-last memory format on the GPU: MIOpen's fp32 convolutions of these shapes
# are NHWC implicit-GEMM kernels in the backward either way, and given NCHW tensors it transposes in and out around each of
# them (156 + 88 transpose / sub-tensor launches, ~2.2 ms of a 66 ms SeqFormer step on MI355X).  PyTorch only hands MIOpen an
# NHWC problem when PYTORCH_MIOPEN_SUGGEST_NHWC is set.  Measured on one box (bench.py model legs, ms per step, NCHW -> this):
# SeqFormer fp32 two clips 66.2 -> 64.3, one clip 52.4 -> 50.6, bf16 autocast 76.9 -> 74.3, IDOL bf16 pair 67.1 -> 65.8.
# Inference keeps NCHW (forward-only NHWC is slower: SeqFormer 12.0 -> 12.9 ms per clip, IDOL 720p 185 -> 172 frames/s),
# which is why the filters stay NCHW (channels-last filters select NHWC whatever the input is).  A training-only process can
# set VNX_CHANNELS_LAST_WEIGHTS=1 to store the filters channels-last as well (no per-call filter conversion: IDOL pair
# 65.8 -> 60.7 ms, bf16 autocast 74.3 -> 69.7; fp32 SeqFormer unchanged).
# OFF unless the process asks for it: vnext_amd.train.enable_channels_last() sets the MIOpen variable and this switch
# together (importing this module changes nothing process-wide; until round 4 it did both at import time).
CHANNELS_LAST = False


class FrozenBatchNorm2d(nn.Module):
    """models/backbone.py:27-64: fixed statistics and affine parameters.  As a standalone module it
    is `x * scale + shift`; inside the trunk below the two constants are folded into the preceding
    convolution instead (`folded()`), which removes two full-resolution elementwise passes per
    convolution from the forward and one from the backward."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._fold = None

    def folded(self):
        """(scale [C], shift [C]), recomputed only when a buffer changed (load_state_dict, .to())."""
        key = tuple((t.data_ptr(), t._version) for t in (self.weight, self.bias, self.running_mean, self.running_var))
        if self._fold is None or self._fold[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + 1e-5).rsqrt()
                self._fold = (key, scale, self.bias - self.running_mean * scale)
        return self._fold[1], self._fold[2]

    def expanded_scale(self, weight):
        """the scale broadcast to the shape of the preceding convolution's filters (a cached constant: the
        multi-tensor multiply of `fold_all` wants equal shapes)"""
        scale, _ = self.folded()
        cached = getattr(self, "_expanded", None)
        if cached is None or cached[0] is not scale or cached[1].shape != weight.shape or cached[1].device != weight.device:
            fmt = torch.channels_last if weight.dim() == 4 and weight.is_contiguous(memory_format=torch.channels_last) \
                and not weight.is_contiguous() else torch.contiguous_format
            cached = self._expanded = (scale, scale.reshape(-1, 1, 1, 1).expand_as(weight).contiguous(memory_format=fmt))
        return cached[1]

    def forward(self, x):
        scale, shift = self.folded()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def conv_bn(x, conv, bn, folded_weight=None):
    """bn(conv(x)) for a frozen bn: one convolution with scaled filters and the shift as its bias
    (the filters stay trainable: the scaling is a differentiable [Cout,1,1,1] multiply on the weights).
    `folded_weight`: the scaled filters -- or a (filters, shift) pair -- when the caller has folded all convolutions at
    once (`fold_all`)."""
    if isinstance(folded_weight, tuple):
        folded_weight, shift = folded_weight
    else:
        scale, shift = bn.folded()
        if folded_weight is None:
            folded_weight = conv.weight * scale.reshape(-1, 1, 1, 1)
    return F.conv2d(x, folded_weight, shift, conv.stride, conv.padding)


class _FoldScales(torch.autograd.Function):
    """weights[i] * scales[i] for a list of filters in one multi-tensor launch (and one in the backward)
    instead of one small launch per convolution: 53 + 43 launches -> a handful per training step."""

    @staticmethod
    def forward(ctx, scales, out_dtype, *weights):
        ctx.scales = scales
        ctx.set_materialize_grads(False)
        return tuple(_cast_all(torch._foreach_mul(list(weights), scales), out_dtype))

    @staticmethod
    def backward(ctx, *grads):
        live = [i for i, g in enumerate(grads) if g is not None]
        out = [None] * len(grads)
        if live:
            gs = _cast_all([grads[i] for i in live], torch.float32)      # (16-bit under autocast: one multi-tensor cast back)
            for i, g in zip(live, torch._foreach_mul(gs, [ctx.scales[i] for i in live])):
                out[i] = g
        return (None, None, *out)


def _cast_all(tensors, dtype):
    """the list in `dtype` with ONE multi-tensor launch (torch._foreach_copy_ converts), itself when it already is"""
    if dtype is None or not tensors or all(t.dtype == dtype for t in tensors):
        return list(tensors)
    out = [torch.empty_like(t, dtype=dtype) for t in tensors]
    torch._foreach_copy_(out, list(tensors))
    return out


def fold_all(pairs):
    """[(conv, frozen bn)] -> {conv: scaled filters}.  Trainable filters go through `_FoldScales`
    (differentiable), frozen ones through a plain multi-tensor multiply.  Under torch.autocast the filters come out in the
    autocast dtype (round 6): autocast would otherwise cast every convolution's filters at its call -- 53 launches forward and
    53 casts of their gradients backward per step where two multi-tensor launches do."""
    out = {}
    adt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() and pairs and pairs[0][0].weight.is_cuda else None
    for trainable in (True, False):
        group = [(c, b) for c, b in pairs if c.weight.requires_grad == trainable]
        if not group:
            continue
        scales = [b.expanded_scale(c.weight) for c, b in group]
        weights = [c.weight for c, _ in group]
        if trainable and torch.is_grad_enabled():
            scaled = _FoldScales.apply(scales, adt, *weights)
        else:
            with torch.no_grad():
                scaled = _cast_all(torch._foreach_mul(weights, scales), adt)
        out.update({c: w for (c, _), w in zip(group, scaled)})
    if adt is not None:      # the shifts (the convolutions' biases) too: one multi-tensor cast instead of one per convolution
        convs = [c for c, _ in pairs]
        with torch.no_grad():
            shifts = _cast_all([b.folded()[1] for _, b in pairs], adt)
        out = {c: (out[c], sh) for c, sh in zip(convs, shifts)}
    return out


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False); self.bn1 = FrozenBatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride, 1, bias=False); self.bn2 = FrozenBatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = FrozenBatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))

    def pairs(self):
        out = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        return out + ([(self.down[0], self.down[1])] if self.down is not None else [])

    def forward(self, x, folded=None):
        w = (lambda c: folded[c]) if folded is not None else (lambda c: None)
        y = F.relu(conv_bn(x, self.conv1, self.bn1, w(self.conv1)))
        y = F.relu(conv_bn(y, self.conv2, self.bn2, w(self.conv2)))
        y = conv_bn(y, self.conv3, self.bn3, w(self.conv3))
        return F.relu(y + (x if self.down is None else conv_bn(x, self.down[0], self.down[1], w(self.down[0]))))


class ResNet50Trunk(nn.Module):
    """res3/res4/res5 features (strides 8/16/32, 512/1024/2048 channels)."""
    strides = (8, 16, 32)
    num_channels = (512, 1024, 2048)

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), FrozenBatchNorm2d(64), nn.ReLU(),
                                  nn.MaxPool2d(3, 2, 1))
        def stage(cin, mid, cout, n, stride):
            return nn.Sequential(*[_Bottleneck(cin if i == 0 else cout, mid, cout, stride if i == 0 else 1)
                                   for i in range(n)])
        self.res2 = stage(64, 64, 256, 3, 1)
        self.res3 = stage(256, 128, 512, 4, 2)
        self.res4 = stage(512, 256, 1024, 6, 2)
        self.res5 = stage(1024, 512, 2048, 3, 2)
        if CHANNELS_LAST and os.environ.get("VNX_CHANNELS_LAST_WEIGHTS", "0") == "1":
            self.to(memory_format=torch.channels_last)

    def freeze(self, freeze_at=2):
        """detectron2's MODEL.BACKBONE.FREEZE_AT (default 2, kept by the reference's configs): the
        stem and res2 are not trained, so the backward stops at res3's input."""
        stages = [self.stem, self.res2, self.res3, self.res4, self.res5]
        for stage in stages[:freeze_at]:
            for p in stage.parameters():
                p.requires_grad_(False)
        return self

    def forward(self, x):
        if CHANNELS_LAST and self.training and torch.is_grad_enabled():
            x = x.contiguous(memory_format=torch.channels_last)
        blocks = [b for stage in (self.res2, self.res3, self.res4, self.res5) for b in stage]
        folded = fold_all([(self.stem[0], self.stem[1])] + [p for b in blocks for p in b.pairs()])
        x = F.max_pool2d(F.relu(conv_bn(x, self.stem[0], self.stem[1], folded[self.stem[0]])), 3, 2, 1)
        outs = []
        for stage in (self.res2, self.res3, self.res4, self.res5):
            for b in stage:
                x = b(x, folded)
            outs.append(x)
        return outs[1:]


def sine_position(mask, num_pos_feats=128, temperature=10000):
    """models/position_encoding.py:35-55 with normalize=True."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < len(self.layers) - 1 else layer(x)
        return x


class MaskHeadSmallConv(nn.Module):
    """segmentation_condInst.py:504-575 (fpns=None path): 3x3 convs on the stride-8/16/32 encoder
    memories, fused top-down at stride 8, down to 8 mask-feature channels."""

    def __init__(self, dim):
        super().__init__()
        self.lay1 = nn.Conv2d(dim, dim // 4, 3, padding=1)
        self.lay2 = nn.Conv2d(dim // 4, dim // 32, 3, padding=1)
        self.lay3 = nn.Conv2d(dim, dim // 4, 3, padding=1)
        self.lay4 = nn.Conv2d(dim, dim // 4, 3, padding=1)
        self.dcn = nn.Conv2d(dim // 4, dim // 4, 3, padding=1)

    def forward(self, feats):
        fused = F.relu(self.lay3(feats[-1]))
        fused = F.relu(self.lay4(feats[-2])) + F.interpolate(fused, size=feats[-2].shape[-2:], mode="nearest")
        fused = F.relu(self.lay1(feats[-3])) + F.interpolate(fused, size=feats[-3].shape[-2:], mode="nearest")
        return self.lay2(F.relu(self.dcn(fused)))


class DeformableDETR(nn.Module):
    def __init__(self, backbone, transformer, num_classes, num_frames, num_queries, num_feature_levels, hidden):
        super().__init__()
        self.backbone = backbone
        self.transformer = transformer
        self.num_frames, self.num_queries, self.num_feature_levels = num_frames, num_queries, num_feature_levels
        nd = transformer.decoder.num_layers
        self.class_embed = nn.ModuleList([nn.Linear(hidden, num_classes) for _ in range(nd)])
        self.bbox_embed = nn.ModuleList([MLP(hidden, hidden, 4, 3) for _ in range(nd)])
        self.query_embed = nn.Embedding(num_queries, hidden * 2)
        proj = [nn.Sequential(nn.Conv2d(c, hidden, 1), nn.GroupNorm(32, hidden)) for c in backbone.num_channels]
        cin = backbone.num_channels[-1]
        for _ in range(num_feature_levels - len(proj)):  # deformable_detr.py:66-76
            proj.append(nn.Sequential(nn.Conv2d(cin, hidden, 3, 2, 1), nn.GroupNorm(32, hidden)))
            cin = hidden
        self.input_proj = nn.ModuleList(proj)
        bias_value = -math.log((1 - 0.01) / 0.01)
        for ce in self.class_embed:
            ce.bias.data.fill_(bias_value)
        for be in self.bbox_embed:
            nn.init.constant_(be.layers[-1].weight.data, 0)
            nn.init.constant_(be.layers[-1].bias.data, 0)
        nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
        self.transformer.decoder.bbox_embed = self.bbox_embed   # iterative box refinement (:95-103)


class CondInstSegm(nn.Module):
    def __init__(self, detr, hidden):
        super().__init__()
        self.detr = detr
        self.controller = MLP(hidden, hidden, 169, 3)
        self.mask_head = MaskHeadSmallConv(hidden)


class _TrainTrunk(nn.Module):
    """The shape-static, sync-free part of a training step -- normalise + pad, backbone, input
    projections, 6+6 transformer layers, class / box heads, mask-feature convs -- as one module, so
    that `torch.cuda.make_graphed_callables` can capture its forward AND its backward into two
    hipGraphs.  Shares the owner's parameters; not registered on the owner."""

    def __init__(self, owner):
        super().__init__()
        self.detr = owner.detr
        object.__setattr__(self, "_owner", owner)

    def forward(self, stack):
        from ..ops.fused_norm import step_scope
        with step_scope(stack.device):     # the fused dropout sites read a device-side step seed bumped HERE per replay
            return self._forward(stack)

    def _forward(self, stack):
        o = self._owner
        h, w = stack.shape[-2:]
        H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        x = stack.new_zeros(stack.shape[0], 3, H, W)
        x[:, :, :h, :w] = (stack - o.pixel_mean) / o.pixel_std
        mask = torch.ones(stack.shape[0], H, W, dtype=torch.bool, device=stack.device)
        mask[:, :h, :w] = False
        srcs, masks, poss = o._features(x, mask)
        d = self.detr.detr
        hs, hs_box, memory, init_ref, inter_refs, inter_boxes, _, _ = d.transformer(srcs, masks, poss, d.query_embed.weight)
        logits, boxes = o._heads(hs, hs_box, init_ref, inter_refs, inter_boxes)
        return (hs, logits, boxes, inverse_sigmoid(init_ref), inverse_sigmoid(inter_refs),
                o._mask_features(srcs, memory))


def graphed_callable(cache, make_module, stack, keep=2):
    """`make_module()(stack)` replayed from forward + backward hipGraphs captured per (input shape, autocast dtype); `cache`: the
    owner's dict (at most `keep` entries).  Under torch.autocast the capture -- warm-up iterations included -- runs with
    autocast's weight cache off: a weight cast cached BEFORE the capture is a tensor outside the graph's memory pool, and one
    cached DURING it would be served, stale, to the eager code after it.  Replays run no Python of the module, so they need
    nothing from the ambient autocast state."""
    amp = torch.is_autocast_enabled() and stack.is_cuda
    adt = torch.get_autocast_dtype("cuda") if amp else None
    key = (tuple(stack.shape), adt)
    fn = cache.get(key)
    if fn is None:
        if len(cache) >= keep:
            cache.pop(next(iter(cache)))
        with torch.autocast("cuda", dtype=adt, enabled=amp, cache_enabled=False):
            fn = cache[key] = torch.cuda.make_graphed_callables(make_module(), (stack.clone(),), allow_unused_input=True)
    return fn(stack)


@META_ARCH_REGISTRY.register()
class SeqFormer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        m = cfg.MODEL.SeqFormer
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.num_frames = cfg.INPUT.SAMPLING_FRAME_NUM
        self.num_classes = m.NUM_CLASSES
        self.mask_stride = m.MASK_STRIDE
        hidden = m.HIDDEN_DIM
        transformer = DeformableTransformer(
            d_model=hidden, nhead=m.NHEADS, num_encoder_layers=m.ENC_LAYERS, num_decoder_layers=m.DEC_LAYERS,
            dim_feedforward=m.DIM_FEEDFORWARD, dropout=m.DROPOUT, activation="relu", return_intermediate_dec=True,
            num_frames=self.num_frames, num_feature_levels=m.NUM_FEATURE_LEVELS, dec_n_points=m.DEC_N_POINTS,
            enc_n_points=m.ENC_N_POINTS)
        detr = DeformableDETR(ResNet50Trunk().freeze(2), transformer, m.NUM_CLASSES, self.num_frames, m.NUM_OBJECT_QUERIES,
                              m.NUM_FEATURE_LEVELS, hidden)
        self.detr = CondInstSegm(detr, hidden)
        weights = {"loss_ce": m.CLASS_WEIGHT, "loss_bbox": m.L1_WEIGHT, "loss_giou": m.GIOU_WEIGHT,
                   "loss_mask": m.MASK_WEIGHT, "loss_dice": m.DICE_WEIGHT}
        if m.DEEP_SUPERVISION:   # seqformer.py:183-187
            weights.update({f"{k}_{i}": v for i in range(m.DEC_LAYERS - 1) for k, v in list(weights.items())})
        matcher = HungarianMatcher(multi_frame=True, cost_class=m.SET_COST_CLASS, cost_bbox=m.SET_COST_BOX,
                                   cost_giou=m.SET_COST_GIOU)
        self.criterion = SetCriterion(m.NUM_CLASSES, matcher, weights, ["labels", "boxes", "masks"],
                                      mask_out_stride=m.MASK_STRIDE, focal_alpha=m.FOCAL_ALPHA,
                                      num_frames=self.num_frames)
        self.deep_supervision = m.DEEP_SUPERVISION
        self.multi_cls, self.cls_thres = m.MULTI_CLS_ON, m.APPLY_CLS_THRES
        self.clip_matching, self.clip_length, self.clip_stride = m.CLIP_MATCHING, m.CLIP_LENGTH, m.CLIP_STRIDE
        self.graph_inference = True     # replay the inference trunk from a hipGraph (per clip shape)
        self.graph_training = False     # capture the training trunk's forward and backward (opt-in)
        self._graphs = {}
        self._train_trunks = {}
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1), persistent=False)
        self.to(self.device)

    # ---- shared trunk -------------------------------------------------------------------------
    def _preprocess(self, batched_inputs):
        """normalise + pad to a multiple of 32 (seqformer.py:413-429, util/misc.py:298-302)."""
        frames = [f.to(self.device, torch.float32) for clip in batched_inputs for f in clip["image"]]
        frames = [(f - self.pixel_mean) / self.pixel_std for f in frames]
        H = max(f.shape[-2] for f in frames)
        W = max(f.shape[-1] for f in frames)
        H, W = (H + 31) // 32 * 32, (W + 31) // 32 * 32
        x = frames[0].new_zeros(len(frames), 3, H, W)
        mask = torch.ones(len(frames), H, W, dtype=torch.bool, device=self.device)
        for i, f in enumerate(frames):
            x[i, :, :f.shape[-2], :f.shape[-1]] = f
            mask[i, :f.shape[-2], :f.shape[-1]] = False
        return x, mask

    def _features(self, x, mask):
        d = self.detr.detr
        T = self.num_frames
        N = x.shape[0] // T
        feats = d.backbone(x)
        srcs, masks, poss = [], [], []
        for l, f in enumerate(feats):
            srcs.append(d.input_proj[l](f))
        for l in range(len(feats), d.num_feature_levels):
            srcs.append(d.input_proj[l](feats[-1] if l == len(feats) else srcs[-1]))
        for s in srcs:
            mk = F.interpolate(mask[None].float(), size=s.shape[-2:]).to(torch.bool)[0]
            masks.append(mk)
            poss.append(sine_position(mk, s.shape[1] // 2).to(s.dtype))
        fold = lambda t: t.reshape(N, T, *t.shape[1:])  # noqa: E731
        return [fold(s) for s in srcs], [fold(mk) for mk in masks], [fold(p) for p in poss]

    def _heads(self, hs, hs_box, init_reference, inter_references, inter_boxes=None):
        """Class logits [Ld, N, Q, K] and boxes [Ld, N, T, Q, 4] of every decoder layer (deformable_detr.py:195-213).
        inter_boxes: the box predictions the decoder's refinement loop already made with these same box heads and
        references (seqformer_transformer.py) -- then only the class heads run here."""
        d = self.detr.detr
        if inter_boxes is not None and d.transformer.decoder.bbox_embed is d.bbox_embed:
            return torch.stack([d.class_embed[lvl](h) for lvl, h in enumerate(hs.unbind(0))]), inter_boxes
        classes, coords = [], []
        for lvl, (h, h_box) in enumerate(zip(hs.unbind(0), hs_box.unbind(0))):   # unbind: ONE stack in the backward
            reference = init_reference if lvl == 0 else inter_references[lvl - 1]
            reference = inverse_sigmoid(reference)
            classes.append(d.class_embed[lvl](h))
            tmp = d.bbox_embed[lvl](h_box)
            if reference.shape[-1] == 4:
                tmp = tmp + reference
            else:
                tmp[..., :2] = tmp[..., :2] + reference
            coords.append(tmp.sigmoid())
        return torch.stack(classes), torch.stack(coords)

    def _run(self, batched_inputs, want_refs=False):
        x, mask = self._preprocess(batched_inputs)
        srcs, masks, poss = self._features(x, mask)
        hs, hs_box, memory, init_ref, inter_refs, inter_boxes, _, _ = self.detr.detr.transformer(
            srcs, masks, poss, self.detr.detr.query_embed.weight)
        logits, boxes = self._heads(hs, hs_box, init_ref, inter_refs, inter_boxes)
        if want_refs:  # the (pre-sigmoid) reference each decoder layer refined, [N, T, Q, 2 or 4]
            refs = [inverse_sigmoid(init_ref if l == 0 else inter_refs[l - 1]) for l in range(hs.shape[0])]
            return x, srcs, hs, memory, logits, boxes, refs
        return x, srcs, hs, memory, logits, boxes

    # ---- training -----------------------------------------------------------------------------
    def prepare_targets(self, batched_inputs):
        """Clip-level targets (seqformer.py:268-299).  A clip's "instances" is a list over frames
        of objects with gt_classes [n], gt_boxes (xyxy pixels; `.tensor` or a tensor), gt_masks
        (`.tensor` or a tensor) [n, H, W], gt_ids [n] (-1 = not visible) and image_size (h, w) --
        detectron2 `Instances` or anything with those attributes / keys."""
        def field(o, k):
            v = o[k] if isinstance(o, dict) else getattr(o, k)
            return getattr(v, "tensor", v)
        targets = []
        for clip in batched_inputs:
            frames = clip.get("instances")
            if not frames:
                h, w = clip["image"][0].shape[-2:]
                T = len(clip["image"])
                targets.append({"labels": torch.zeros(0, dtype=torch.int64, device=self.device),
                                "boxes": torch.zeros(0, T, 4, device=self.device),
                                "masks": torch.zeros(0, T, h, w, dtype=torch.bool, device=self.device),
                                "size": torch.as_tensor([h, w], dtype=torch.long)})     # host: only the host reads it
                continue
            boxes, masks, classes = [], [], []
            for fr in frames:
                h, w = field(fr, "image_size")
                scale = scale_tensor([w, h, w, h], self.device)
                boxes.append(box_xyxy_to_cxcywh(field(fr, "gt_boxes").to(self.device, torch.float32) / scale))
                masks.append(field(fr, "gt_masks").to(self.device))
                classes.append(field(fr, "gt_classes").to(self.device) * (field(fr, "gt_ids").to(self.device) != -1))
            targets.append({"labels": torch.stack(classes, 0).max(0)[0], "boxes": torch.stack(boxes, 1),
                            "masks": torch.stack(masks, 1),
                            "size": torch.as_tensor([h, w], dtype=torch.long)})
        return targets

    def _mask_features(self, srcs, memory):
        """[N, T, S, C] encoder memory -> stride-8 mask features of every frame [N*T, 8, H/8, W/8]
        (forward_mask_head_train's per-frame MaskHeadSmallConv calls, batched over N*T and run
        ONCE instead of once per decoder layer)."""
        N, T = memory.shape[:2]
        mem, start = [], 0
        for s in srcs[:3]:
            h, w = s.shape[-2:]
            mem.append(memory[:, :, start:start + h * w].reshape(N * T, h, w, -1).permute(0, 3, 1, 2))
            start += h * w
        return self.detr.mask_head(mem).float().contiguous()

    def _graphed_train_trunk(self, stack):
        """Forward + backward hipGraphs of the training trunk for this clip batch shape, captured
        on first use (3 eager warm-up iterations on a side stream, then capture -- what
        make_graphed_callables does)."""
        return graphed_callable(self._train_trunks, lambda: _TrainTrunk(self).train(), stack)

    def losses(self, batched_inputs):
        """CondInst_segm.forward (segmentation_condInst.py:69-207) + SetCriterion."""
        targets = self.prepare_targets(batched_inputs)
        frames = [f for clip in batched_inputs for f in clip["image"]]
        if self.graph_training and frames[0].is_cuda and all(f.shape == frames[0].shape for f in frames):
            hs, logits, boxes, ref0, ref_rest, feats = self._graphed_train_trunk(
                torch.stack([f.to(self.device, torch.float32) for f in frames]))
            refs = [ref0] + list(ref_rest[:hs.shape[0] - 1])
        else:
            x, srcs, hs, memory, logits, boxes, refs = self._run(batched_inputs, want_refs=True)
            feats = self._mask_features(srcs, memory)
        Ld, N, T = boxes.shape[:3]
        indices_list = self.criterion.matcher.match_all_layers(logits, boxes, targets)
        # the matched instances of every decoder layer, on every frame of their clip: one gather, one
        # controller call, one mask-head launch (the reference: a Python loop over layers x clips x frames)
        lay = torch.cat([torch.full_like(q, l) for l, ind in enumerate(indices_list) for q, _ in ind]).to(self.device)
        clip = torch.cat([torch.full_like(q, i) for ind in indices_list for i, (q, _) in enumerate(ind)]).to(self.device)
        qry = torch.cat([q for ind in indices_list for q, _ in ind]).to(self.device)
        params = self.detr.controller(hs[lay, clip, qry])                         # [Ld*n, 169]
        ref_xy = torch.stack([r[..., :2] for r in refs])                          # [Ld, N, T, Q, 2] pre-sigmoid
        sizes = torch.stack([scale_tensor(t["size"].flip(0).tolist(), self.device) for t in targets])   # [N, (w, h)]
        points = ref_xy[lay, clip, :, qry].sigmoid() * sizes[clip][:, None, :]    # [Ld*n, T, 2] image pixels
        image = (clip * T)[:, None] + torch.arange(T, device=self.device)[None, :]
        params = params[:, None].expand(-1, T, -1).flatten(0, 1)                   # [(l, i, inst, t), 169]
        points = points.flatten(0, 1)
        image = image.flatten().to(torch.int32)
        masks = dynamic_mask_head(feats, points.float(), params.float(), image, 8)  # [sum, H/4, W/4]
        masks = masks.view(-1, T, *masks.shape[-2:])
        if masks.shape[0] == 0:  # nothing matched anywhere: keep the mask branch in the autograd graph
            masks = masks + 0 * (feats.sum() + sum(p.sum() for p in self.detr.controller.parameters()))
        if self.deep_supervision:   # every decoder layer's losses in one pass over stacked tensors
            return self.criterion.forward_all_layers(logits, boxes, masks, targets, indices_list, weighted=True)
        else:
            n_last = sum(len(q) for q, _ in indices_list[-1])
            outputs = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "pred_masks": masks[masks.shape[0] - n_last:]}
            loss = self.criterion(outputs, targets, indices_list)
        w = self.criterion.weight_dict
        return {k: v * w[k] if k in w else v for k, v in loss.items()}

    # ---- the two branches -----------------------------------------------------------------------
    def forward(self, batched_inputs):
        if self.training:
            return self.losses(batched_inputs)
        return self.inference(batched_inputs)

    # ---- inference --------------------------------------------------------------------------
    def _clip_trunk(self, stack):
        """[T, 3, h, w] raw frames of ONE clip -> what `inference` needs from the network: class
        logits and (pre-sigmoid) reference of the last decoder layer, its query states and the
        stride-8 mask features.  Shape-static and free of host synchronisation, so it can be
        captured into a hipGraph; boxes of the other layers are not computed (whole-video
        inference never reads them, seqformer.py:351-410)."""
        h, w = stack.shape[-2:]
        H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        x = stack.new_zeros(stack.shape[0], 3, H, W)
        x[:, :, :h, :w] = (stack - self.pixel_mean) / self.pixel_std
        mask = torch.ones(stack.shape[0], H, W, dtype=torch.bool, device=stack.device)
        mask[:, :h, :w] = False
        srcs, masks, poss = self._features(x, mask)
        d = self.detr.detr
        hs, _, memory, init_ref, inter_refs, _, _, _ = d.transformer(srcs, masks, poss, d.query_embed.weight)
        last = hs.shape[0] - 1
        ref = inverse_sigmoid(init_ref if last == 0 else inter_refs[last - 1])
        return d.class_embed[last](hs[last]), ref, hs[last], self._mask_features(srcs, memory)

    def _clip_trunk_graphed(self, stack):
        """`_clip_trunk` replayed from a hipGraph captured per (T, h, w): the ~1 500 launches of
        backbone + 6+6 transformer layers + mask-feature convs become one graph launch
        (SURVEY section 8(f) rank 2).  Outputs live in the graph's static buffers until the next call."""
        key = tuple(stack.shape)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = stack.clone()
            side = torch.cuda.Stream(device=stack.device)
            side.wait_stream(torch.cuda.current_stream(stack.device))
            with torch.cuda.stream(side):           # warm-up outside capture: lazy inits, cached level tensors
                for _ in range(2):
                    self._clip_trunk(static_in)
            torch.cuda.current_stream(stack.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._clip_trunk(static_in)
            if len(self._graphs) >= 4:              # a few resolutions at most; drop the oldest
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(stack)
        graph.replay()
        return static_out

    def _top_instances(self, frames):
        """Frames of one clip (any length; the reference sets `num_frames` to it, seqformer.py:231,249)
        -> class probabilities [10, K] of the 10 best queries and their mask logits
        [10, T, H/4, W/4] at a quarter of the padded input size (`inference_clip`, :302-326)."""
        T = len(frames)
        keep, self.num_frames = self.num_frames, T
        try:
            if all(f.shape == frames[0].shape for f in frames):
                stack = torch.stack(frames)
                trunk = self._clip_trunk_graphed if (self.graph_inference and stack.is_cuda) else self._clip_trunk
                logits, ref_last, hs_last, feats = trunk(stack)
            else:   # frames of different sizes: the general (eager, padded-batch) path
                x, srcs, hs, memory, logits_all, _, refs = self._run([{"image": frames}], want_refs=True)
                logits, ref_last, hs_last, feats = logits_all[-1], refs[-1], hs[-1], self._mask_features(srcs, memory)
        finally:
            self.num_frames = keep
        ih, iw = frames[0].shape[-2:]                                         # size fed to the network
        prob = logits[0].sigmoid()                                            # [Q, classes]
        query = prob.max(1)[0].topk(min(10, prob.shape[0]))[1]
        params = self.detr.controller(hs_last[0, query])                     # [10, 169]
        scale = scale_tensor([iw, ih], self.device)
        ref = ref_last[0][:, query, :2].sigmoid() * scale                     # [T, 10, 2] image pixels
        n = len(query)
        logits_m = dynamic_mask_with_coords(feats, ref.reshape(1, T * n, 2).float(),
                                            params.float().repeat(T, 1)[None], [n] * T, 8)
        return prob[query], logits_m.view(T, n, *logits_m.shape[-2:]).transpose(0, 1)

    def _report(self, prob, mask_logits, image_size, out_size):
        """class probabilities [n, K] + mask logits [n, T, H/4, W/4] -> the output dict
        (`whole_video_inference` :351-410 == `clip_matching_postprocess` :328-349)."""
        if prob.shape[0] == 0:
            return {"image_size": out_size, "pred_scores": [], "pred_labels": [], "pred_masks": []}
        h, w = mask_logits.shape[-2:]
        masks = F.interpolate(mask_logits, size=(h * self.mask_stride, w * self.mask_stride), mode="bilinear",
                              align_corners=False).sigmoid()
        if self.multi_cls:
            who, label = torch.where(prob > self.cls_thres)
            score = prob[who, label]
            masks = masks[who]
        else:
            score, label = prob.max(-1)
        masks = F.interpolate(masks[:, :, :image_size[0], :image_size[1]], size=out_size, mode="nearest") > 0.5
        return {"image_size": out_size, "pred_scores": score.tolist(), "pred_labels": label.tolist(),
                "pred_masks": [m for m in masks.cpu()]}

    @torch.no_grad()
    def inference(self, batched_inputs):
        """One video (seqformer.py:227-264).  Default: the whole video as one clip -- the 10 queries
        with the best class score, their masks on every frame, every (query, class) pair above
        APPLY_CLS_THRES reported (the reference runs the mask head for all 300 queries of all 6 decoder
        layers and keeps 10 of the last).  With CLIP_MATCHING: overlapping clips of CLIP_LENGTH frames
        every CLIP_STRIDE, linked by mask sIoU (vnext_amd/models/clip_matching.py), tracks averaged."""
        assert len(batched_inputs) == 1
        video = batched_inputs[0]
        frames = [f.to(self.device, torch.float32) for f in video["image"]]
        ih, iw = frames[0].shape[-2:]
        out_size = (video.get("height", ih), video.get("width", iw))
        if not self.clip_matching:
            prob, mask_logits = self._top_instances(frames)
            return self._report(prob, mask_logits, (ih, iw), out_size)
        from types import SimpleNamespace
        from .clip_matching import Clips, Videos
        n_frames, merged = len(frames), None
        for start in range(0, n_frames, self.clip_stride):
            end, last = start + self.clip_length, False
            if end >= n_frames:
                start, end, last = max(0, n_frames - self.clip_length), n_frames, True
            idx = list(range(start, end))
            prob, mask_logits = self._top_instances([frames[i] for i in idx])
            if merged is None:
                merged = Videos(self.clip_length, n_frames, self.num_classes, mask_logits.shape[-2:], self.device)
            score, label = prob.max(-1)
            merged.update(Clips(idx, SimpleNamespace(pred_classes=label, scores=score, cls_probs=prob,
                                                     pred_masks=mask_logits)))
            if last:
                break
        cls, mask_logits = merged.get_result()
        return self._report(cls, mask_logits, (ih, iw), out_size)
