"""`IDOL` meta-architecture on the MI355X hot path (b).

Registered under the reference's name (projects/IDOL/idol/idol.py:71-72), built as `IDOL(cfg)`,
same `forward(batched_inputs)` I/O: training takes key/reference frame pairs (each item's
"image" / "instances" lists hold 2 frames) and returns the weighted loss dict (:228-232);
inference takes one video and returns {"image_size", "pred_scores", "pred_labels",
"pred_masks"} (:466-471).  Model tree (`detr` = CondInst_segm {`detr` = DeformableDETR {...},
controller, mask_head, reid_embed_head}) and parameter names follow the reference so its
checkpoints load.

Mapping onto the machine (the arithmetic and the decisions are the reference's):
  * key and reference frames go through backbone + transformer in ONE batch of 2*bz frames
    (the reference runs the transformer twice, segmentation_condInst.py:142-143);
  * matching for all decoder layers in one host round-trip (idol_criterion.OTAMatcher);
  * the dynamic mask head of all layers' matched queries in one fused launch (forward and
    backward), mask features computed once instead of once per layer (:157-200);
  * reid losses: one similarity launch per image on the matrix cores (heads.loss_reid);
  * inference: per-frame candidate selection for the whole chunk from one host copy, mask
    head only for the selected queries (the reference evaluates all 300 per frame, :296-318),
    tracker with one mask-IoU matrix and one association matrix per frame (models/tracker.py).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ..heads import dynamic_mask_head, loss_reid
from ..registry import META_ARCH_REGISTRY
from .criterion import box_cxcywh_to_xyxy, box_xyxy_to_cxcywh
from .idol_criterion import IDOLCriterion, OTAMatcher, reid_terms, select_pos_neg_masks
from .idol_transformer import DeformableTransformer
from .seqformer import MLP, DeformableDETR, MaskHeadSmallConv, ResNet50Trunk, scale_tensor, sine_position
from .seqformer_transformer import inverse_sigmoid
from .tracker import DeviceTracker, IDOL_Tracker


class CondInstSegmIDOL(nn.Module):
    def __init__(self, detr, hidden):
        super().__init__()
        self.detr = detr
        self.controller = MLP(hidden, hidden, 169, 3)
        for layer in self.controller.layers:     # segmentation_condInst.py:68-70
            nn.init.xavier_uniform_(layer.weight)
            nn.init.zeros_(layer.bias)
        self.mask_head = MaskHeadSmallConv(hidden)
        self.reid_embed_head = MLP(hidden, hidden, hidden, 3)


class _TrainTrunk(nn.Module):
    """The shape-static, sync-free part of an IDOL training step -- normalise + pad, backbone, input projections, 6 + 6
    transformer layers, the class / box heads of every decoder layer on the key frames, the stride-8 mask features, the
    reference frames' class scores and the reid embeddings -- as one module, so that `torch.cuda.make_graphed_callables` can
    capture its forward AND its backward into two hipGraphs (SeqFormer's twin:
    seqformer.py:_TrainTrunk).  Shares the owner's parameters; not registered on the owner.

    Why IDOL wants it more than SeqFormer: a key / reference pair is TWO frames -- the ~3 300 launches of a step carry 38 ms of
    kernels (720p, bf16) and the step took 52-56 ms: the host could not issue them fast enough (round 6)."""

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
        return o._train_trunk(x, mask)


def class_aware_nms(boxes_xyxy, scores, classes, thr):
    """torchvision.ops.batched_nms restated on host arrays (published algorithm: boxes of
    different classes never suppress each other; greedy by descending score; returns the kept
    indices in descending-score order)."""
    order = np.argsort(-scores, kind="stable")
    b = boxes_xyxy[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep = np.ones(len(order), dtype=bool)
    for i in range(len(order)):
        if not keep[i]:
            continue
        lt = np.maximum(b[i, :2], b[i + 1:, :2])
        rb = np.minimum(b[i, 2:], b[i + 1:, 2:])
        wh = np.clip(rb - lt, 0, None)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area[i + 1:] - inter)
        keep[i + 1:] &= ~((iou > thr) & (classes[order[i + 1:]] == classes[order[i]]))
    return order[keep]


@META_ARCH_REGISTRY.register()
class IDOL(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        m = cfg.MODEL.IDOL
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.num_frames = cfg.INPUT.SAMPLING_FRAME_NUM
        self.num_classes, self.mask_stride = m.NUM_CLASSES, m.MASK_STRIDE
        self.is_multi_cls, self.apply_cls_thres = m.MULTI_CLS_ON, m.APPLY_CLS_THRES
        self.temporal_score_type = m.TEMPORAL_SCORE_TYPE
        self.inference_select_thres = m.INFERENCE_SELECT_THRES
        self.inference_fw, self.inference_tw = m.INFERENCE_FW, m.INFERENCE_TW
        self.memory_len, self.nms_pre, self.add_new_score = m.MEMORY_LEN, m.NMS_PRE, m.ADD_NEW_SCORE
        self.batch_infer_len = m.BATCH_INFER_LEN
        hidden = m.HIDDEN_DIM
        transformer = DeformableTransformer(
            d_model=hidden, nhead=m.NHEADS, num_encoder_layers=m.ENC_LAYERS, num_decoder_layers=m.DEC_LAYERS,
            dim_feedforward=m.DIM_FEEDFORWARD, dropout=m.DROPOUT, activation="relu", return_intermediate_dec=True,
            num_frames=self.num_frames, num_feature_levels=m.NUM_FEATURE_LEVELS, dec_n_points=m.DEC_N_POINTS,
            enc_n_points=m.ENC_N_POINTS)
        detr = DeformableDETR(ResNet50Trunk().freeze(2), transformer, m.NUM_CLASSES, self.num_frames, m.NUM_OBJECT_QUERIES,
                              m.NUM_FEATURE_LEVELS, hidden)
        self.detr = CondInstSegmIDOL(detr, hidden)
        weights = {"loss_ce": m.CLASS_WEIGHT, "loss_bbox": m.L1_WEIGHT, "loss_giou": m.GIOU_WEIGHT,
                   "loss_reid": m.REID_WEIGHT, "loss_reid_aux": m.REID_WEIGHT * 1.5,
                   "loss_mask": m.MASK_WEIGHT, "loss_dice": m.DICE_WEIGHT}
        if m.DEEP_SUPERVISION:   # idol.py:183-187
            weights.update({f"{k}_{i}": v for i in range(m.DEC_LAYERS - 1) for k, v in list(weights.items())})
        matcher = OTAMatcher(multi_frame=True, cost_class=m.SET_COST_CLASS, cost_bbox=m.SET_COST_BOX,
                             cost_giou=m.SET_COST_GIOU)
        self.criterion = IDOLCriterion(m.NUM_CLASSES, matcher, weights, ["labels", "boxes", "masks", "reid"],
                                       mask_out_stride=m.MASK_STRIDE, focal_alpha=m.FOCAL_ALPHA,
                                       num_frames=self.num_frames)
        self.deep_supervision = m.DEEP_SUPERVISION
        self.graph_inference = True      # replay the per-chunk inference trunk from a hipGraph
        self.graph_training = False      # capture the training trunk's forward and backward (opt-in: the trunk's gradients then
        #                                  reach DDP's buckets together, at the end of the replayed backward)
        self._graphs = {}
        self._train_trunks = {}
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1), persistent=False)
        self.to(self.device)

    # ---- shared trunk -------------------------------------------------------------------------
    def _preprocess(self, frames):
        """normalise + pad to a multiple of 32 (idol.py:473-482, util/misc.py nested tensor)."""
        frames = [(f.to(self.device, torch.float32) - self.pixel_mean) / self.pixel_std for f in frames]
        H = (max(f.shape[-2] for f in frames) + 31) // 32 * 32
        W = (max(f.shape[-1] for f in frames) + 31) // 32 * 32
        x = frames[0].new_zeros(len(frames), 3, H, W)
        mask = torch.ones(len(frames), H, W, dtype=torch.bool, device=self.device)
        for i, f in enumerate(frames):
            x[i, :, :f.shape[-2], :f.shape[-1]] = f
            mask[i, :f.shape[-2], :f.shape[-1]] = False
        return x, mask

    def _encode_decode(self, x, mask):
        """-> srcs, hs [Ld, N, Q, C], memory [N, S, C], per-layer pre-sigmoid references, the refined references, and the box
        predictions of the decoder's refinement loop (what _box_heads would compute again; None if it used other heads)"""
        d = self.detr.detr
        feats = d.backbone(x)
        srcs = [d.input_proj[l](f) for l, f in enumerate(feats)]
        for l in range(len(feats), d.num_feature_levels):
            srcs.append(d.input_proj[l](feats[-1] if l == len(feats) else srcs[-1]))
        masks = [F.interpolate(mask[None].float(), size=s.shape[-2:]).to(torch.bool)[0] for s in srcs]
        poss = [sine_position(mk, s.shape[1] // 2).to(s.dtype) for s, mk in zip(srcs, masks)]
        hs, memory, init_ref, inter_refs, _, inter_boxes, _ = d.transformer(srcs, masks, poss, d.query_embed.weight)
        refs = [inverse_sigmoid(init_ref if l == 0 else inter_refs[l - 1]) for l in range(hs.shape[0])]
        if d.transformer.decoder.bbox_embed is not d.bbox_embed:
            inter_boxes = None        # the decoder's loop did not use the detector's box heads: they run in _box_heads
        return srcs, hs, memory, refs, inter_refs, inter_boxes        # inter_boxes [Ld, N, Q, 4] with their graph | None

    def _box_heads(self, hs, refs, layers, loop_boxes=None):
        """Class logits and boxes of the decoder layers `layers`.  loop_boxes [Ld, ...]: the box predictions the decoder's
        refinement loop made with these box heads and references (idol_transformer.py) -- then only the class heads run."""
        d = self.detr.detr
        if loop_boxes is not None:
            layers = list(layers)      # (plain indexing: a list index would copy an index tensor to the device -- not under capture)
            picked = loop_boxes if layers == list(range(loop_boxes.shape[0])) else torch.stack([loop_boxes[l] for l in layers])
            return torch.stack([d.class_embed[l](hs[l]) for l in layers]), picked
        logits, boxes = [], []
        for l in layers:
            tmp = d.bbox_embed[l](hs[l])
            if refs[l].shape[-1] == 4:
                tmp = tmp + refs[l]
            else:
                tmp[..., :2] = tmp[..., :2] + refs[l]
            logits.append(d.class_embed[l](hs[l]))
            boxes.append(tmp.sigmoid())
        return torch.stack(logits), torch.stack(boxes)

    def _mask_features(self, srcs, memory):
        """[N, S, C] memory -> stride-8 mask features [N, 8, H/8, W/8] (forward_mask_head_train :330-345)"""
        mem, start = [], 0
        for s in srcs[:3]:
            h, w = s.shape[-2:]
            mem.append(memory[:, start:start + h * w].reshape(memory.shape[0], h, w, -1).permute(0, 3, 1, 2))
            start += h * w
        return self.detr.mask_head(mem).float().contiguous()

    def _train_trunk(self, x, mask):
        """Padded frames of the key / reference pairs (key frames at even positions) -> everything `losses` needs from the
        network, as a flat tuple of tensors: query states of every decoder layer [Ld, N, Q, C]; class logits and boxes of every
        layer on the KEY frames; the layers' pre-sigmoid reference xy on the key frames [Ld, bz, Q, 2]; the last layer's
        refined references on the REFERENCE frames; the key frames' stride-8 mask features; the reference frames' class logits
        (last layer); the reid embeddings of all frames."""
        srcs, hs, memory, refs, inter_refs, inter_boxes = self._encode_decode(x, mask)
        loop_boxes = None if inter_boxes is None else inter_boxes[:, 0::2]
        logits, boxes = self._box_heads(hs[:, 0::2], [r[0::2] for r in refs], range(hs.shape[0]), loop_boxes)
        feats = self._mask_features([s[0::2] for s in srcs], memory[0::2])
        ref_xy = torch.stack([r[0::2, :, :2] for r in refs])
        ref_logits = self.detr.detr.class_embed[-1](hs[-1, 1::2])
        embeds = self.detr.reid_embed_head(hs[-1])
        return hs, logits, boxes, ref_xy, inter_refs[-1, 1::2], feats, ref_logits, embeds

    def _graphed_train_trunk(self, stack):
        """Forward + backward hipGraphs of `_train_trunk` for this batch shape (and autocast dtype), captured on first use
        (`torch.cuda.make_graphed_callables`: eager warm-up iterations on a side stream, then capture).  Under torch.autocast
        the capture runs with autocast's weight cache OFF: a cast cached before the capture would be a tensor outside the
        graph's memory pool."""
        from .seqformer import graphed_callable
        return graphed_callable(self._train_trunks, lambda: _TrainTrunk(self).train(), stack)

    # ---- training -----------------------------------------------------------------------------
    def prepare_targets(self, batched_inputs):
        """-> (det_targets, ref_targets): key / reference frame of every pair (idol.py:283-311)."""
        def field(o, k):
            v = o[k] if isinstance(o, dict) else getattr(o, k)
            return getattr(v, "tensor", v)
        per_frame = []
        for video in batched_inputs:
            for fr in video["instances"]:
                h, w = field(fr, "image_size")
                scale = scale_tensor([w, h, w, h], self.device)
                ids = field(fr, "gt_ids").to(self.device)
                per_frame.append({"labels": field(fr, "gt_classes").to(self.device),
                                  "boxes": box_xyxy_to_cxcywh(field(fr, "gt_boxes").to(self.device, torch.float32) / scale),
                                  "masks": field(fr, "gt_masks").to(self.device), "inst_id": ids, "valid": ids != -1})
        det, ref = per_frame[0::2], per_frame[1::2]
        for d_t, r_t in zip(det, ref):     # objects absent from the key frame are dropped from both
            if not bool(d_t["valid"].all()):
                keep = d_t["valid"].clone()
                for t in (d_t, r_t):
                    for k in list(t):
                        t[k] = t[k][keep]
        return det, ref

    def losses(self, batched_inputs):
        """CondInst_segm.forward (segmentation_condInst.py:78-231) + SetCriterion."""
        det_t, ref_t = self.prepare_targets(batched_inputs)
        frames = [f for video in batched_inputs for f in video["image"]]
        sizes = [tuple(f.shape[-2:]) for f in frames][0::2]
        if self.graph_training and frames[0].is_cuda and all(f.shape == frames[0].shape for f in frames):
            hs, logits, boxes, ref_xy, ref_last, feats, ref_logits, embeds = self._graphed_train_trunk(
                torch.stack([f.to(self.device, torch.float32) for f in frames]))
        else:
            hs, logits, boxes, ref_xy, ref_last, feats, ref_logits, embeds = self._train_trunk(*self._preprocess(frames))
        Ld, bz = hs.shape[0], len(det_t)
        indices_list, matched = self.criterion.matcher.match_all_layers(logits, boxes, det_t)
        # the selected queries of every decoder layer on every key frame: one gather, one controller
        # call, one mask-head launch
        q_host = [[torch.nonzero(sel).flatten() for sel, _ in ind] for ind in indices_list]
        lay = torch.cat([torch.full_like(q, l) for l, layer in enumerate(q_host) for q in layer]).to(self.device, non_blocking=True)
        img = torch.cat([torch.full_like(q, i) for layer in q_host for i, q in enumerate(layer)]).to(self.device, non_blocking=True)
        qry = torch.cat([q for layer in q_host for q in layer]).to(self.device, non_blocking=True)
        key_hs = hs[:, 0::2]                                                            # [Ld, bz, Q, C]
        scale = torch.stack([scale_tensor([sizes[i][1], sizes[i][0]], self.device) for i in range(bz)])   # [bz, (w, h)]
        params = self.detr.controller(key_hs[lay, img, qry])
        points = ref_xy[lay, img, qry].sigmoid() * scale[img]
        masks = dynamic_mask_head(feats, points.float(), params.float(), img.to(torch.int32), 8)
        if masks.shape[0] == 0:  # nothing matched: keep the mask branch in the autograd graph
            masks = masks + 0 * (feats.sum() + sum(p.sum() for p in self.detr.controller.parameters()))
        masks = masks[:, None]                                                          # [n, 1, H/4, W/4]
        # contrastive sets on the reference frames (last decoder layer), embeddings of both frames
        selections = select_pos_neg_masks(ref_last, ref_logits.sigmoid(), ref_t)
        qd = reid_terms(embeds[0::2], embeds[1::2], matched, selections, loss_reid)
        if self.deep_supervision:   # every decoder layer's losses in one pass over stacked tensors
            loss = self.criterion.forward_all_layers(logits, boxes, masks, det_t, indices_list, qd)
        else:
            n_last = sum(len(q) for q in q_host[-1])
            outputs = {"pred_logits": logits[-1], "pred_boxes": boxes[-1], "pred_masks": masks[masks.shape[0] - n_last:],
                       "pred_qd": qd}
            loss = self.criterion(outputs, det_t, ref_t, indices_list)
        if qd["count"] == 0:     # keep the reid head in the graph (static DDP graph)
            loss["loss_reid"] = loss["loss_reid"] + embeds.sum() * 0
        w = self.criterion.weight_dict
        return {k: v * w[k] if k in w else v for k, v in loss.items()}

    def forward(self, batched_inputs):
        if self.training:
            return self.losses(batched_inputs)
        return self.inference_video(batched_inputs)

    # ---- inference ----------------------------------------------------------------------------
    @torch.no_grad()
    def select_candidates(self, logits, boxes):
        """logits [F, Q, K], boxes [F, Q, 4] -> per frame the query indices that enter the tracker
        (idol.py:331-343): best class score above INFERENCE_SELECT_THRES (the top-1 query if none),
        then class-aware box NMS at 0.9; one host copy for the whole chunk."""
        best, label = logits.sigmoid().max(-1)
        host = torch.cat([best[..., None], label[..., None].to(best.dtype), box_cxcywh_to_xyxy(boxes)], -1).cpu().numpy()
        picks = []
        for f in range(host.shape[0]):
            score, cls, bx = host[f, :, 0], host[f, :, 1].astype(np.int64), host[f, :, 2:]
            cand = np.nonzero(score > self.inference_select_thres)[0]
            if len(cand) == 0:
                cand = np.array([int(np.argmax(score))])
            else:
                cand = cand[class_aware_nms(bx[cand], score[cand], cls[cand], 0.9)]
            picks.append(cand)
        return picks

    def _chunk_trunk(self, x, mask):
        """Padded frames -> (logits [F,Q,K], boxes [F,Q,4], query states, pre-sigmoid reference of the
        last decoder layer, stride-8 mask features): the shape-static, sync-free part of inference."""
        srcs, hs, memory, refs, _, inter_boxes = self._encode_decode(x, mask)
        last = hs.shape[0] - 1
        logits, boxes = self._box_heads(hs, refs, [last], inter_boxes)
        return logits[0], boxes[0], hs[last], refs[last], self._mask_features(srcs, memory)

    def _chunk_trunk_graphed(self, stack):
        """`_chunk_trunk` of same-sized frames replayed from a hipGraph captured per chunk shape
        (a video is chunks of BATCH_INFER_LEN frames + one shorter tail: two graphs)."""
        key = tuple(stack.shape)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = stack.clone()

            def run():
                h, w = static_in.shape[-2:]
                H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32
                x = static_in.new_zeros(static_in.shape[0], 3, H, W)
                x[:, :, :h, :w] = (static_in - self.pixel_mean) / self.pixel_std
                mask = torch.ones(static_in.shape[0], H, W, dtype=torch.bool, device=static_in.device)
                mask[:, :h, :w] = False
                return self._chunk_trunk(x, mask)
            side = torch.cuda.Stream(device=stack.device)
            side.wait_stream(torch.cuda.current_stream(stack.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    run()
            torch.cuda.current_stream(stack.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = run()
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(stack)
        graph.replay()
        return static_out

    @torch.no_grad()
    def inference_forward(self, frames):
        """One chunk of frames -> per-frame candidates (inference_forward :234-321 + the selection of
        idol.py:331-349): list over frames of dicts {indices, logits [n,K], boxes [n,4], embeds
        [n,C], masks [n,1,H/4,W/4]} for the queries that survive the score threshold and the
        class-aware box NMS (0.9)."""
        frames = [f.to(self.device, torch.float32) for f in frames]
        if self.graph_inference and frames[0].is_cuda and all(f.shape == frames[0].shape for f in frames):
            logits, boxes, hs_last, ref_last, feats = self._chunk_trunk_graphed(torch.stack(frames))
        else:
            logits, boxes, hs_last, ref_last, feats = self._chunk_trunk(*self._preprocess(frames))
        picks = self.select_candidates(logits, boxes)
        frame_of = torch.from_numpy(np.concatenate([np.full(len(c), f) for f, c in enumerate(picks)])).to(self.device)
        query = torch.from_numpy(np.concatenate(picks)).to(self.device)
        sel_hs = hs_last[frame_of, query]
        ih, iw = frames[0].shape[-2:]
        scale = scale_tensor([iw, ih], self.device)
        points = ref_last[frame_of, query, :2].sigmoid() * scale              # = inter_references[-2][..., :2]
        masks = dynamic_mask_head(feats, points.float(), self.detr.controller(sel_hs).float(),
                                  frame_of.to(torch.int32), 8)
        embeds = self.detr.reid_embed_head(sel_hs)
        out, start = [], 0
        for f, c in enumerate(picks):
            sl = slice(start, start + len(c))
            out.append({"indices": c.tolist(), "logits": logits[f, query[sl]], "boxes": boxes[f, query[sl]],
                        "embeds": embeds[sl], "masks": masks[sl][:, None]})
            start += len(c)
        return out

    @torch.no_grad()
    def inference_video(self, batched_inputs):
        """idol.py:236-281: chunks of BATCH_INFER_LEN frames through the network, then the tracker."""
        video = batched_inputs[0]["image"]
        per_frame = []
        for s in range(0, len(video), self.batch_infer_len):
            per_frame.extend(self.inference_forward(video[s:s + self.batch_infer_len]))
        args = dict(init_score_thr=0.2, obj_score_thr=0.1, nms_thr_pre=self.nms_pre, nms_thr_post=0.05,
                    addnew_score_thr=self.add_new_score, memo_tracklet_frames=10, memo_momentum=0.8,
                    long_match=self.inference_tw, frame_weight=(self.inference_tw | self.inference_fw),
                    temporal_weight=self.inference_tw, memory_len=self.memory_len)
        most = max((len(fr["indices"]) for fr in per_frame), default=0)
        on_device = self.device.type == "cuda" and DeviceTracker.supports(memory_len=self.memory_len, max_dets=most)
        tracker = (DeviceTracker if on_device else IDOL_Tracker)(**args)
        ih, iw = video[0].shape[-2:]
        oh, ow = batched_inputs[0].get("height", ih), batched_inputs[0].get("width", iw)
        return self.associate(per_frame, tracker, (oh, ow), (ih, iw), host_factory=lambda: IDOL_Tracker(**args))

    @torch.no_grad()
    def associate(self, per_frame, tracker, ori_size, image_size, host_factory=None):
        """IDOL.inference (idol.py:313-471) on the pre-selected candidates of every frame.

        With a `DeviceTracker` the association of the whole video is enqueued without a host round trip:
        every frame leaves its ids in device memory, they are read ONCE after the last frame, and the
        per-track bookkeeping (which only needs the ids) runs on that copy.  With the host-side
        `IDOL_Tracker` (CPU, differential tests) the ids arrive frame by frame, as in the reference.

        The device tracker has fixed limits (slots alive at a time, detections per frame, memory_len): a video that
        outgrows its slots is associated AGAIN by `host_factory()` (the host-side tracker, no limits) instead of
        being lost -- a tracklet that found no slot is never remembered, so the device ids up to that point may
        already differ from the reference's and are discarded as a whole."""
        n_frames = len(per_frame)
        on_device = isinstance(tracker, DeviceTracker)
        probs, frame_ids = [], []
        for t, fr in enumerate(per_frame):
            prob = fr["logits"].sigmoid()
            score, label = prob.max(1)
            det = torch.cat([fr["boxes"], score[:, None]], 1)
            probs.append(prob)
            if on_device:
                frame_ids.append(tracker.match_device(det, label, fr["masks"], fr["embeds"], t))
            else:
                _, _, ids, kept = tracker.match(bboxes=det, labels=label, masks=fr["masks"], track_feats=fr["embeds"],
                                                frame_id=t, indices=list(range(len(fr["indices"]))))
                full = torch.full((det.shape[0],), -3, dtype=torch.long)
                full[torch.tensor(kept, dtype=torch.long)] = ids
                frame_ids.append(full)
        if on_device and frame_ids:
            sizes = [len(x) for x in frame_ids]
            frame_ids = list(torch.cat(frame_ids).cpu().split(sizes))      # the one copy of the video
            if tracker.counters()[1]:
                if host_factory is None:
                    raise RuntimeError("DeviceTracker: more simultaneous tracklets than slots; raise `capacity`")
                return self.associate(per_frame, host_factory(), ori_size, image_size)
        video = {}
        for t, (fr, ids) in enumerate(zip(per_frame, frame_ids)):
            for row, k in enumerate(ids.tolist()):
                if k < 0:
                    continue
                v = video.setdefault(k, {"masks": [None] * t, "scores": [None] * t, "valid": 0})
                v["masks"].append(fr["masks"][row])
                v["scores"].append(probs[t][row])
                v["valid"] += 1
            for v in video.values():
                if len(v["masks"]) < t + 1:
                    v["masks"].append(None)
                    v["scores"].append(None)
            if t > 8:      # drop short noisy sequences (:386-393)
                for k in [k for k, v in video.items() if v["masks"][-1] is None and v["masks"][-2] is None and v["valid"] < 3]:
                    video.pop(k)
        cls, masks_out = [], []
        for v in video.values():
            s = torch.stack([x for x in v["scores"] if x is not None])
            cls.append(s.mean(0) if self.temporal_score_type == "mean" else s.max(0)[0])
            seen = [i for i, m in enumerate(v["masks"]) if m is not None]
            m = torch.stack([v["masks"][i] for i in seen])                        # [k, 1, H/4, W/4]: one pass per track
            h, w = m.shape[-2:]
            m = F.interpolate(m, size=(h * 4, w * 4), mode="bilinear", align_corners=False).sigmoid()
            m = (F.interpolate(m[:, :, :image_size[0], :image_size[1]], size=ori_size, mode="nearest") > 0.5)[:, 0].cpu()
            per = [None] * n_frames
            for j, i in enumerate(seen):
                per[i] = m[j]
            masks_out.append(per)
        if not cls:
            return {"image_size": ori_size, "pred_scores": [], "pred_labels": [], "pred_masks": []}
        cls = torch.stack(cls)
        if self.is_multi_cls:
            who, label = torch.where(cls > self.apply_cls_thres)
            score, masks_out = cls[who, label], [masks_out[i] for i in who.tolist()]
        else:
            score, label = cls.max(-1)
        return {"image_size": ori_size, "pred_scores": score.tolist(), "pred_labels": label.tolist(),
                "pred_masks": masks_out}
