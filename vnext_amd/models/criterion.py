"""Clip-level set criterion of SeqFormer: Hungarian matching + focal / L1 / GIoU / mask losses.

Host-side mirror of the reference's training objective (same loss names, same numbers):
  matcher     projects/SeqFormer/seqformer/models/matcher.py:25-96
  criterion   projects/SeqFormer/seqformer/models/deformable_detr.py:231-439
  focal/dice  projects/SeqFormer/seqformer/models/segmentation_condInst.py:680-723
  GIoU loss   fvcore.nn.giou_loss (fvcore 0.1.5): 1 - GIoU, eps 1e-7 on union and hull
This is the caller of the hot path on the training side (it decides which instances the dynamic
mask head runs for), not a kernel.  Differences in *how*, not in *what*:
  * the cost matrices of all decoder layers are built in one batched pass and cross to the
    host in ONE copy (`match_all_layers`); the reference matches inside the layer loop -> 6
    device syncs per step (deformable_detr.py / segmentation_condInst.py:146-147);
  * the per-frame GIoU loop (matcher.py:69-72) is one broadcast over the frame axis;
  * `num_boxes` stays a tensor -- no `.item()` sync (deformable_detr.py:417-419).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


def box_cxcywh_to_xyxy(b):
    c, wh = b[..., :2], b[..., 2:]
    return torch.cat([c - 0.5 * wh, c + 0.5 * wh], -1)


def box_xyxy_to_cxcywh(b):
    lo, hi = b[..., :2], b[..., 2:]
    return torch.cat([(lo + hi) / 2, hi - lo], -1)


def _area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def pairwise_giou(a, b):
    """a [..., N, 4], b [..., M, 4] (xyxy, broadcastable leading axes) -> GIoU [..., N, M]
    (util/box_ops.py:65-86: eps only on the enclosing area)."""
    a, b = a[..., :, None, :], b[..., None, :, :]
    wh = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = _area(a) + _area(b) - inter
    hull = (torch.maximum(a[..., 2:], b[..., 2:]) - torch.minimum(a[..., :2], b[..., :2])).clamp(min=0)
    hull = hull[..., 0] * hull[..., 1]
    return inter / union - (hull - union) / (hull + 1e-7)


def giou_loss(a, b, eps=1e-7):
    """element-wise 1 - GIoU of matched xyxy boxes (fvcore.nn.giou_loss, reduction 'none')."""
    lo, hi = torch.maximum(a[..., :2], b[..., :2]), torch.minimum(a[..., 2:], b[..., 2:])
    overlap = (hi > lo).all(-1)
    inter = torch.where(overlap, (hi - lo).prod(-1), torch.zeros_like(lo[..., 0]))
    union = _area(a) + _area(b) - inter
    hull = (torch.maximum(a[..., 2:], b[..., 2:]) - torch.minimum(a[..., :2], b[..., :2])).prod(-1)
    return 1 - (inter / (union + eps) - (hull - union) / (hull + eps))


def sigmoid_focal_loss(logits, targets, num_boxes, alpha=0.25, gamma=2.0):
    """segmentation_condInst.py:698-723: mean over the last axis, summed, / num_boxes"""
    p = logits.sigmoid()
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def dice_loss(logits, targets, num_boxes):
    """segmentation_condInst.py:680-695"""
    p = logits.sigmoid().flatten(1)
    num = 2 * (p * targets).sum(1)
    den = p.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_boxes


class HungarianMatcher(nn.Module):
    """One-to-one assignment of queries to ground-truth *clip* instances; a box cost is the
    distance over all frames of the clip (matcher.py:53-96)."""

    def __init__(self, multi_frame=True, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.multi_frame = multi_frame
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou

    @torch.no_grad()
    def cost(self, logits, boxes, targets):
        """logits [..., bs, Q, K], boxes [..., bs, nf, Q, 4] (any leading layer axes) ->
        cost [..., bs, Q, sum n] on the device."""
        tgt_ids = torch.cat([t["labels"] for t in targets])
        nf = boxes.shape[-3]
        tgt = torch.cat([t["boxes"] for t in targets]).reshape(len(tgt_ids), nf, 4).to(boxes.dtype)
        if logits.dtype in (torch.bfloat16, torch.float16) or boxes.dtype in (torch.bfloat16, torch.float16):
            logits, boxes = logits.float(), boxes.float()                    # autocast outputs: fp32 costs
        prob = logits.sigmoid()
        out = boxes.transpose(-3, -2)                                        # [..., bs, Q, nf, 4]
        # Euclidean distance over the clip's nf*4 coordinates (torch.cdist default p=2, matcher.py:66)
        c_box = torch.cdist(out.flatten(-2), tgt.flatten(1))
        tgt = tgt.clamp(min=1e-7, max=1)                                     # matcher.py:68
        g = pairwise_giou(box_cxcywh_to_xyxy(out.transpose(-3, -2)),         # frames leading: [..., bs, nf, Q, n]
                          box_cxcywh_to_xyxy(tgt.transpose(0, 1)))
        c_giou = -g.mean(-3)
        alpha, gamma = 0.25, 2.0
        neg = (1 - alpha) * prob ** gamma * -(1 - prob + 1e-8).log()
        pos = alpha * (1 - prob) ** gamma * -(prob + 1e-8).log()
        c_cls = (pos - neg)[..., tgt_ids]
        return self.cost_bbox * c_box + self.cost_class * c_cls + self.cost_giou * c_giou

    @staticmethod
    def _solve(cost_cpu, sizes):
        """cost_cpu [bs, Q, sum n] on the host -> [(query idx, target idx)] per clip"""
        out, start = [], 0
        for i, n in enumerate(sizes):
            q, t = linear_sum_assignment(cost_cpu[i, :, start:start + n])
            out.append((torch.as_tensor(q, dtype=torch.int64), torch.as_tensor(t, dtype=torch.int64)))
            start += n
        return out

    def forward(self, outputs, targets, nf=None, valid_ratios=None):
        """The reference call: one layer's {'pred_logits', 'pred_boxes'} -> indices."""
        sizes = [len(t["labels"]) for t in targets]
        c = self.cost(outputs["pred_logits"], outputs["pred_boxes"], targets).cpu().numpy()
        return self._solve(c, sizes)

    def match_all_layers(self, logits, boxes, targets):
        """logits [Ld, bs, Q, K], boxes [Ld, bs, nf, Q, 4] -> indices_list (one entry per decoder
        layer); one device->host copy for all layers."""
        sizes = [len(t["labels"]) for t in targets]
        c = self.cost(logits, boxes, targets).cpu().numpy()
        return [self._solve(c[l], sizes) for l in range(c.shape[0])]


class SetCriterion(nn.Module):
    """labels (focal), boxes (L1 + GIoU over the clip's frames), masks (focal + dice)."""

    def __init__(self, num_classes, matcher, weight_dict, losses, focal_alpha=0.25, mask_out_stride=4, num_frames=1):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.losses = num_classes, matcher, weight_dict, losses
        self.focal_alpha, self.mask_out_stride, self.num_frames = focal_alpha, mask_out_stride, num_frames

    @staticmethod
    def _src_idx(indices):
        return (torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)]),
                torch.cat([s for s, _ in indices]))

    @staticmethod
    def _tgt_idx(indices):
        return (torch.cat([torch.full_like(t, i) for i, (_, t) in enumerate(indices)]),
                torch.cat([t for _, t in indices]))

    def loss_labels(self, outputs, targets, indices, num_boxes, log=True):
        logits = outputs["pred_logits"]                                     # [bs, Q, K]
        b, q = self._src_idx(indices)
        cls = torch.cat([t["labels"][j.to(t["labels"].device)] for t, (_, j) in zip(targets, indices)])
        onehot = torch.zeros_like(logits)
        onehot[b.to(logits.device), q.to(logits.device), cls.to(logits.device)] = 1
        out = {"loss_ce": sigmoid_focal_loss(logits, onehot, num_boxes, self.focal_alpha, 2.0) * logits.shape[1]}
        if log:
            with torch.no_grad():
                sel = logits[b.to(logits.device), q.to(logits.device)]
                if cls.numel() == 0:
                    out["class_error"] = 100 - torch.zeros([], device=logits.device)
                else:
                    hit = (sel.argmax(-1) == cls.to(logits.device)).float().mean() * 100
                    out["class_error"] = 100 - hit
        return out

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        b, q = self._src_idx(indices)
        pred = outputs["pred_boxes"].transpose(1, 2)[b.to(outputs["pred_boxes"].device), q.to(outputs["pred_boxes"].device)]
        nf = pred.shape[1] if pred.dim() == 3 else outputs["pred_boxes"].shape[1]
        tgt = torch.cat([t["boxes"].reshape(-1, nf, 4)[j.to(t["boxes"].device)] for t, (_, j) in zip(targets, indices)])
        tgt = tgt.to(pred)
        l1 = F.l1_loss(pred.flatten(1), tgt.flatten(1), reduction="none") / nf
        g = giou_loss(box_cxcywh_to_xyxy(pred.flatten(0, 1)), box_cxcywh_to_xyxy(tgt.flatten(0, 1))) / nf
        return {"loss_bbox": l1.sum() / num_boxes, "loss_giou": g.sum() / num_boxes}

    def loss_masks(self, outputs, targets, indices, num_boxes):
        """pred_masks: list over clips of [1, n_i, nf, H/4, W/4] (instances in matched order) or one
        tensor [sum n, nf, H/4, W/4]."""
        src = outputs["pred_masks"]
        if isinstance(src, (list, tuple)):
            src = torch.cat(list(src), 1)[0]
        nf, h, w = src.shape[1:]
        s = self.mask_out_stride
        picked = []
        for t, (_, j) in zip(targets, indices):
            m = t["masks"][j.to(t["masks"].device)]                           # [n_i, nf, H_i, W_i]
            # ground truth sampled at the centre of each stride-4 cell of the /32-padded canvas
            # (deformable_detr.py:353-362): zero-pad to the canvas, then [s//2::s]
            m = m[..., s // 2::s, s // 2::s]
            assert m.shape[-2] <= h and m.shape[-1] <= w
            picked.append(F.pad(m.to(src.dtype), (0, w - m.shape[-1], 0, h - m.shape[-2])))
        tgt = torch.cat(picked) if picked else src.new_zeros((0, nf, h, w))
        if tgt.shape[0] == 0:
            zero = (src * 0).sum()
            return {"loss_mask": zero, "loss_dice": zero}
        src, tgt = src.flatten(1), tgt.flatten(1)
        return {"loss_mask": sigmoid_focal_loss(src, tgt, num_boxes), "loss_dice": dice_loss(src, tgt, num_boxes)}

    # ---- all decoder layers in one pass ------------------------------------------------------
    def forward_all_layers(self, logits, boxes, masks, targets, indices_list, weighted=False):
        """`weighted`: return the terms already multiplied by `weight_dict` (what the model's forward returns,
        seqformer.py:140-144 of the reference): one multiply per loss type on the per-layer vectors and one
        `unbind` each, instead of a multiply and a `select` per (type, layer) -- 60 forward and ~110 backward
        launches fewer per step.

        `forward` for every decoder layer at once: logits [Ld, N, Q, K], boxes [Ld, N, T, Q, 4],
        masks [Ld * n, T, h, w] (the matched instances' mask logits, layer-major, clips in order,
        instances in matched order -- what the fused mask head returns), indices_list[layer][clip] =
        (query idx, target idx).  Same names and numbers as `forward` with deep supervision; one set of
        kernels instead of one per layer (the Hungarian matching assigns every target in every layer,
        so each layer contributes the same number n of instances)."""
        Ld, N, Q, K = logits.shape
        T = boxes.shape[2]
        dev = logits.device
        num_boxes = torch.as_tensor([float(sum(len(t["labels"]) for t in targets))], device=dev)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(num_boxes)
            world = torch.distributed.get_world_size()
        num_boxes = torch.clamp(num_boxes / world, min=1)[0]
        # stacked index tensors, built on the host, one transfer each
        lay = torch.cat([torch.full_like(q, l) for l, ind in enumerate(indices_list) for q, _ in ind]).to(dev)
        clip = torch.cat([torch.full_like(q, i) for ind in indices_list for i, (q, _) in enumerate(ind)]).to(dev)
        qry = torch.cat([q for ind in indices_list for q, _ in ind]).to(dev)
        start = [0]
        for t in targets:
            start.append(start[-1] + len(t["labels"]))
        tgt = torch.cat([j + start[i] for ind in indices_list for i, (_, j) in enumerate(ind)]).to(dev)   # into the concatenated targets
        n = len(qry) // Ld
        names = [f"_{l}" for l in range(Ld - 1)] + [""]
        out = {}
        # labels (focal): mean over Q * Q = sum over Q
        all_labels = torch.cat([t["labels"] for t in targets]).to(dev)
        onehot = torch.zeros_like(logits)
        onehot[lay, clip, qry, all_labels[tgt]] = 1
        p = logits.sigmoid()
        ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
        p_t = p * onehot + (1 - p) * (1 - onehot)
        focal = ce * (1 - p_t) ** 2.0
        if self.focal_alpha >= 0:
            focal = (self.focal_alpha * onehot + (1 - self.focal_alpha) * (1 - onehot)) * focal
        loss_ce = focal.mean(2).sum((1, 2)) / num_boxes * Q
        with torch.no_grad():
            if n:
                sel = logits[-1][clip[-n:], qry[-n:]]
                out["class_error"] = 100 - (sel.argmax(-1) == all_labels[tgt[-n:]]).float().mean() * 100
            else:
                out["class_error"] = 100 - torch.zeros([], device=dev)
        # boxes (L1 + GIoU over the clip's frames)
        pred = boxes.transpose(2, 3)[lay, clip, qry]                                   # [Ld*n, T, 4]
        all_boxes = torch.cat([t["boxes"].reshape(-1, T, 4) for t in targets]).to(pred)
        want = all_boxes[tgt]
        l1 = (pred - want).abs().flatten(1).sum(1).view(Ld, n).sum(1) / T / num_boxes
        g = giou_loss(box_cxcywh_to_xyxy(pred.flatten(0, 1)), box_cxcywh_to_xyxy(want.flatten(0, 1)))
        g = g.view(Ld, n * T).sum(1) / T / num_boxes
        # masks (focal + dice)
        if n:
            h, w = masks.shape[-2:]
            s_ = self.mask_out_stride
            gt = []
            for t in targets:
                m = t["masks"][..., s_ // 2::s_, s_ // 2::s_]
                gt.append(F.pad(m.to(masks.dtype), (0, w - m.shape[-1], 0, h - m.shape[-2])))
            gt = torch.cat(gt).to(dev)[tgt].flatten(1)                                 # [Ld*n, T*h*w]
            src = masks.flatten(1)
            pm = src.sigmoid()
            ce_m = F.binary_cross_entropy_with_logits(src, gt, reduction="none")
            pt_m = pm * gt + (1 - pm) * (1 - gt)
            fm = (0.25 * gt + 0.75 * (1 - gt)) * ce_m * (1 - pt_m) ** 2.0
            loss_mask = fm.mean(1).view(Ld, n).sum(1) / num_boxes
            dice = 1 - (2 * (pm * gt).sum(1) + 1) / (pm.sum(1) + gt.sum(1) + 1)
            loss_dice = dice.view(Ld, n).sum(1) / num_boxes
        else:
            loss_mask = loss_dice = (masks * 0).sum() + torch.zeros(Ld, device=dev)
        for kind, per_layer in (("loss_ce", loss_ce), ("loss_bbox", l1), ("loss_giou", g), ("loss_mask", loss_mask),
                                ("loss_dice", loss_dice)):
            if weighted:
                per_layer = per_layer * self._layer_weights(kind, names, dev)
            for suffix, term in zip(names, per_layer.unbind(0)):
                out[kind + suffix] = term
        return out

    def _layer_weights(self, kind, names, dev):
        """[Ld] tensor of weight_dict[kind + suffix] (1 where the dict has no entry), cached per device."""
        cache = self.__dict__.setdefault("_lw_cache", {})
        key = (kind, tuple(names), str(dev), tuple(self.weight_dict.get(kind + s, 1.0) for s in names))
        if key not in cache:
            cache[key] = torch.tensor(key[3], dtype=torch.float32, device=dev)
        return cache[key]

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kw):
        table = {"labels": self.loss_labels, "boxes": self.loss_boxes, "masks": self.loss_masks}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, indices, num_boxes, **kw)

    def forward(self, outputs, targets, indices_list, valid_ratios=None):
        device = outputs["pred_logits"].device
        num_boxes = torch.as_tensor([float(sum(len(t["labels"]) for t in targets))], device=device)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(num_boxes)
            world = torch.distributed.get_world_size()
        num_boxes = torch.clamp(num_boxes / world, min=1)[0]
        losses = {}
        for name in self.losses:
            losses.update(self.get_loss(name, outputs, targets, indices_list[-1], num_boxes))
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            for name in self.losses:
                kw = {"log": False} if name == "labels" else {}
                part = self.get_loss(name, aux, targets, indices_list[i], num_boxes, **kw)
                losses.update({f"{k}_{i}": v for k, v in part.items()})
        return losses
