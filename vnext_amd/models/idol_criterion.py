"""IDOL's training-side association: simOTA matcher, contrastive positive/negative selection
and the criterion (SURVEY.md section 8 row a7 and the callers of rows a5/a6).

Host-side mirror of
  matcher            projects/IDOL/idol/models/matcher.py:45-170          (simOTA for DETR queries)
  select_pos_neg     projects/IDOL/idol/models/pos_neg_select.py:13-199
  SetCriterion       projects/IDOL/idol/models/deformable_detr.py:236-494
with the same loss names and numbers (tests/test_idol_criterion.py pins them to outputs of the
reference classes).  What changes is where the work runs:
  * the assignment logic is tiny integer/boolean work on [300, n_gt] matrices that the
    reference runs on the GPU with a host sync per ground-truth box (`.item()` in topk,
    `.any()` in the while loop): here the logits and boxes of ALL decoder layers cross to the
    host in one copy and the matching runs there (`match_all_layers`);
  * the reid losses use one similarity launch per image on the matrix cores
    (vnext_amd.heads.loss_reid) instead of two einsums + two normalisations per instance; the
    selection below only produces boolean masks for it.
Two quirks of the reference are kept on purpose, because they change results: the
"matched to several boxes" mask in the repair loop is the one computed BEFORE the loop
(matcher.py:156-158 reuses `anchor_matching_gt`), and the k=10 / k=100 selections of
pos_neg_select share one cost matrix, so repairs made by the first are seen by the second.
"""
from __future__ import annotations

import random as _random

import torch
import torch.nn as nn
import torch.nn.functional as F

from .criterion import box_cxcywh_to_xyxy, dice_loss, giou_loss, pairwise_giou, sigmoid_focal_loss


class _one_thread:
    """Context: ATen intra-op threads = 1 while the tiny host-side matching matrices are processed."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        torch.set_num_threads(1)

    def __exit__(self, *exc):
        torch.set_num_threads(self.n)
        return False


def _host32(t):
    """detach -> host; 16-bit (autocast) tensors as fp32, fp32 / fp64 as they are"""
    t = t.detach()
    return (t.float() if t.dtype in (torch.bfloat16, torch.float16) else t).cpu()


def _pairwise_iou(a, b):
    a, b = a[:, None, :], b[None, :, :]
    wh = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = lambda t: (t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1])  # noqa: E731
    return inter / (area(a) + area(b) - inter)


def in_boxes_info(boxes, gts, expanded_strides=32, center_radius=2.5):
    """boxes [Q, 4], gts [n, 4] (cxcywh, normalised) -> (query is a candidate [Q],
    centre inside box AND inside the fixed-radius centre region [Q, n])  (matcher.py:93-124)"""
    cx, cy = boxes[:, 0, None], boxes[:, 1, None]
    xy = box_cxcywh_to_xyxy(gts)
    in_box = (cx > xy[None, :, 0]) & (cx < xy[None, :, 2]) & (cy > xy[None, :, 1]) & (cy < xy[None, :, 3])
    r = center_radius / expanded_strides
    in_ctr = (cx > gts[None, :, 0] - r) & (cx < gts[None, :, 0] + r) & (cy > gts[None, :, 1] - r) & (cy < gts[None, :, 1] + r)
    return in_box.any(1) | in_ctr.any(1), in_box & in_ctr


def ota_cost(boxes, prob, gt_boxes, gt_labels):
    """-> (cost [Q, n], iou [Q, n])   (matcher.py:72-88 == pos_neg_select.py:86-103)"""
    fg, both = in_boxes_info(boxes, gt_boxes)
    a, b = box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(gt_boxes)
    iou = _pairwise_iou(a, b)
    alpha, gamma = 0.25, 2.0
    neg = (1 - alpha) * prob ** gamma * -(1 - prob + 1e-8).log()
    pos = alpha * (1 - prob) ** gamma * -(prob + 1e-8).log()
    cost = (pos - neg)[:, gt_labels] + 3.0 * -pairwise_giou(a, b) + 100.0 * (~both)
    cost[~fg] += 10000.0
    return cost, iou


def dynamic_k_matching(cost, iou, n_candidate_k):
    """-> matching [Q, n] (0/1); `cost` is modified in place exactly as the reference does
    (matcher.py:126-160, pos_neg_select.py:154-185)."""
    ks = torch.topk(iou, n_candidate_k, dim=0)[0].sum(0).int().clamp(min=1)
    # per box, the reference's own call (matcher.py:138-141): torch.topk's choice among tied costs is
    # what decides the assignment there, so it is what runs here (host tensors, a handful of columns)
    M = torch.zeros_like(cost)
    for g in range(cost.shape[1]):
        M[torch.topk(cost[:, g], k=int(ks[g]), largest=False)[1], g] = 1.0
    multi = M.sum(1) > 1                                   # queries claimed by several boxes
    if multi.any():
        keep = cost[multi].argmin(1)
        M[multi] = 0
        M[multi, keep] = 1
    while (M.sum(0) == 0).any():                           # boxes that lost all their queries
        cost[M.sum(1) > 0] += 100000.0
        for g in torch.nonzero(M.sum(0) == 0).flatten().tolist():
            M[cost[:, g].argmin(), g] = 1
        if (M.sum(1) > 1).any():                           # (the mask of BEFORE the loop, see above)
            keep = cost[multi].argmin(1)
            M[multi] = 0
            M[multi, keep] = 1
    return M


class OTAMatcher(nn.Module):
    """simOTA assignment of queries to the key frame's boxes: several queries per box.
    Per image -> ((selected [Q] bool, box index of each selected query), best query per box)."""

    def __init__(self, multi_frame=True, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
        super().__init__()
        self.multi_frame, self.cost_class, self.cost_bbox, self.cost_giou = multi_frame, cost_class, cost_bbox, cost_giou

    @staticmethod
    def _one(boxes, prob, target, nf):
        """Host tensors in, host tensors out.  The matrices are [300, n_gt]: run the ATen CPU ops on
        one thread (their default fan-out over all host cores costs ~1 ms per op on a 256-core box)."""
        n = len(target["labels"])
        if n == 0:
            return (torch.zeros(prob.shape[0], dtype=torch.bool), torch.zeros(0, dtype=torch.int64)), \
                torch.zeros(0, dtype=torch.int64)
        gt = target["boxes"].reshape(n, nf, 4)[:, 0].to(boxes)
        cost, iou = ota_cost(boxes, prob, gt, target["labels"])
        M = dynamic_k_matching(cost, iou, 10)
        selected = M.sum(1) > 0
        gt_idx = M[selected].max(1)[1]
        cost[M == 0] += float("inf")
        return (selected, gt_idx), cost.argmin(0)

    @torch.no_grad()
    def forward(self, outputs, targets, nf=1):
        prob = _host32(outputs["pred_logits"].detach().sigmoid())
        boxes = _host32(outputs["pred_boxes"])
        tg = [{k: t[k].cpu() for k in ("labels", "boxes")} for t in targets]
        with _one_thread():
            res = [self._one(boxes[i], prob[i], tg[i], nf) for i in range(len(tg))]
        return [r[0] for r in res], [r[1] for r in res]

    @torch.no_grad()
    def match_all_layers(self, logits, boxes, targets, nf=1):
        """logits [Ld, bz, Q, K], boxes [Ld, bz, Q, 4] -> (indices_list over layers, matched ids
        of the last layer); one device->host copy."""
        prob = _host32(logits.detach().float().sigmoid() if logits.dtype in (torch.bfloat16, torch.float16)
                       else logits.detach().sigmoid())
        boxes = _host32(boxes)
        tg = [{k: t[k].cpu() for k in ("labels", "boxes")} for t in targets]
        with _one_thread():
            out = [[self._one(boxes[l, i], prob[l, i], tg[i], nf) for i in range(len(tg))] for l in range(prob.shape[0])]
        return [[r[0] for r in layer] for layer in out], [r[1] for r in out[-1]]


@torch.no_grad()
def select_pos_neg_masks(ref_boxes, ref_prob, ref_targets, rng=_random):
    """The reference frame's contrastive sets (pos_neg_select.py:13-67, 72-124).

    ref_boxes [bz, Q, 4], ref_prob [bz, Q, K] (last decoder layer on the reference frame),
    ref_targets: per image {"boxes" [n,4], "labels" [n], "valid" [n]} -- same instance order as
    the key frame's targets.  -> per image (inst [I] indices of the valid instances,
    pos [Q, I], neg [Q, I], aux [Q, I] bool masks); aux = positives + the negatives drawn with
    `rng.sample` (host RNG, same call sequence as the reference)."""
    out = []
    ref_boxes, ref_prob = _host32(ref_boxes), _host32(ref_prob)
    with _one_thread():
        for i, t in enumerate(ref_targets):
            valid = t["valid"].cpu().bool()
            Q = ref_boxes.shape[1]
            inst = torch.nonzero(valid).flatten()
            I = len(inst)
            pos = torch.zeros(Q, I, dtype=torch.bool)
            neg = torch.zeros(Q, I, dtype=torch.bool)
            aux = torch.zeros(Q, I, dtype=torch.bool)
            if I > 0:
                gt = t["boxes"].cpu().reshape(-1, 4)[valid].to(ref_boxes)
                cost, iou = ota_cost(ref_boxes[i], ref_prob[i], gt, t["labels"].cpu()[valid])
                pos = dynamic_k_matching(cost, iou, 10) > 0
                neg = ~(dynamic_k_matching(cost, iou, 100) > 0)       # same (already repaired) cost matrix
                for c in range(I):
                    P, N = int(pos[:, c].sum()), int(neg[:, c].sum())
                    k = 10 if P == 0 else (N if P * 10 >= N else P * 10)
                    picked = rng.sample(list(range(N)), k)
                    neg_rows = torch.nonzero(neg[:, c]).flatten()
                    aux[:, c] = pos[:, c]
                    aux[neg_rows[picked], c] = True
            out.append((inst, pos, neg, aux))
    return out


class IDOLCriterion(nn.Module):
    """labels / boxes / masks on the key frame with simOTA indices + the reid losses between
    key and reference frame (deformable_detr.py:236-494)."""

    def __init__(self, num_classes, matcher, weight_dict, losses, focal_alpha=0.25, mask_out_stride=4, num_frames=1):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.losses = num_classes, matcher, weight_dict, losses
        self.focal_alpha, self.mask_out_stride, self.num_frames = focal_alpha, mask_out_stride, num_frames

    @staticmethod
    def _on_device(indices, device):
        """[(selected [Q] bool, gt idx)] on the host -> [(query idx, gt idx)] on the device.  Indexing a
        device tensor with a boolean mask makes the host wait for the device (the output size is data);
        with index tensors built on the host it does not."""
        return [(torch.nonzero(sel).flatten().to(device, non_blocking=True), gt.to(device, non_blocking=True))
                for sel, gt in indices]

    def loss_labels(self, outputs, targets, ref_targets, indices, num_boxes, log=True):
        logits = outputs["pred_logits"]
        onehot = torch.zeros_like(logits)
        count = 0
        for i, (q, gt) in enumerate(self._on_device(indices, logits.device)):
            if len(gt):
                onehot[i, q, targets[i]["labels"].to(logits.device)[gt]] = 1
                count += len(gt)
        return {"loss_ce": sigmoid_focal_loss(logits, onehot, max(count, 1), self.focal_alpha, 2.0) * logits.shape[1]}

    def loss_boxes(self, outputs, targets, ref_targets, indices, num_boxes):
        boxes = outputs["pred_boxes"]
        dev = self._on_device(indices, boxes.device)
        pred = [boxes[i][q] for i, (q, gt) in enumerate(dev) if len(gt)]
        if not pred:
            zero = boxes.sum() * 0
            return {"loss_bbox": zero, "loss_giou": zero}
        pred = torch.cat(pred)
        tgt = torch.cat([targets[i]["boxes"].to(boxes)[gt] for i, (_, gt) in enumerate(dev) if len(gt)])
        n = pred.shape[0]
        return {"loss_bbox": F.l1_loss(pred, tgt, reduction="none").sum() / n,
                "loss_giou": giou_loss(box_cxcywh_to_xyxy(pred), box_cxcywh_to_xyxy(tgt)).sum() / n}

    def loss_masks(self, outputs, targets, ref_targets, indices, num_boxes):
        src = outputs["pred_masks"]
        if isinstance(src, (list, tuple)):
            src = torch.cat(list(src), 1)[0]
        h, w = src.shape[-2:]
        s = self.mask_out_stride
        picked = []
        for i, (_, gt) in enumerate(indices):
            if len(gt):
                m = targets[i]["masks"][gt.to(targets[i]["masks"].device)][..., s // 2::s, s // 2::s]
                picked.append(F.pad(m.to(src.dtype), (0, w - m.shape[-1], 0, h - m.shape[-2]))[:, None])
        if not picked:
            zero = (src * 0).sum()
            return {"loss_mask": zero, "loss_dice": zero}
        tgt = torch.cat(picked)
        n = src.shape[0]
        assert src.shape == tgt.shape
        return {"loss_mask": sigmoid_focal_loss(src.flatten(1), tgt.flatten(1), n),
                "loss_dice": dice_loss(src.flatten(1), tgt.flatten(1), n)}

    def loss_reid(self, outputs, targets, ref_targets, indices, num_boxes):
        """outputs['pred_qd'] = {"contrast": sum over instances, "aux": sum over instances,
        "count": number of instances} as produced by `reid_terms` below."""
        qd = outputs["pred_qd"]
        if qd["count"] == 0:
            zero = outputs["pred_logits"].sum() * 0
            return {"loss_reid": zero, "loss_reid_aux": zero}
        return {"loss_reid": qd["contrast"] / qd["count"], "loss_reid_aux": qd["aux"] / qd["count"]}

    # ---- all decoder layers in one pass ------------------------------------------------------
    def forward_all_layers(self, logits, boxes, masks, targets, indices_list, pred_qd):
        """`forward` for every decoder layer at once: logits [Ld, bz, Q, K], boxes [Ld, bz, Q, 4], masks
        [sum over layers of n_l, 1, h, w] (layer-major, images in order, queries ascending -- what the
        fused mask head returns), indices_list[layer][image] = (selected [Q] bool, gt idx).  simOTA
        selects a different number of queries per layer, so per-layer sums are segment sums over one
        flat list.  Same names and numbers as `forward` with deep supervision."""
        Ld, bz, Q, K = logits.shape
        dev = logits.device
        q_host = [[torch.nonzero(sel).flatten() for sel, _ in ind] for ind in indices_list]
        counts = [sum(len(q) for q in layer) for layer in q_host]
        lay = torch.cat([torch.full_like(q, l) for l, layer in enumerate(q_host) for q in layer]).to(dev, non_blocking=True)
        img = torch.cat([torch.full_like(q, i) for layer in q_host for i, q in enumerate(layer)]).to(dev, non_blocking=True)
        qry = torch.cat([q for layer in q_host for q in layer]).to(dev, non_blocking=True)
        start = [0]
        for t in targets:
            start.append(start[-1] + len(t["labels"]))
        tgt = torch.cat([gt.long() + start[i] for ind in indices_list for i, (_, gt) in enumerate(ind)]).to(dev, non_blocking=True)
        denom = torch.tensor([max(c, 1) for c in counts], dtype=logits.dtype, device=dev)
        present = torch.tensor([1.0 if c else 0.0 for c in counts], dtype=logits.dtype, device=dev)

        def per_layer(values):        # [sum n] -> [Ld] segment sums
            return torch.zeros(Ld, dtype=values.dtype, device=dev).index_add_(0, lay, values)
        # labels
        all_labels = torch.cat([t["labels"] for t in targets]).to(dev)
        onehot = torch.zeros_like(logits)
        onehot[lay, img, qry, all_labels[tgt]] = 1
        p = logits.sigmoid()
        ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
        focal = ce * (1 - (p * onehot + (1 - p) * (1 - onehot))) ** 2.0
        if self.focal_alpha >= 0:
            focal = (self.focal_alpha * onehot + (1 - self.focal_alpha) * (1 - onehot)) * focal
        loss_ce = focal.mean(2).sum((1, 2)) / denom * Q
        # boxes
        pred = boxes[lay, img, qry]
        want = torch.cat([t["boxes"].reshape(-1, 4) for t in targets]).to(pred)[tgt]
        l1 = per_layer((pred - want).abs().sum(1)) / denom * present
        giou = per_layer(giou_loss(box_cxcywh_to_xyxy(pred), box_cxcywh_to_xyxy(want))) / denom * present
        # masks
        h, w = masks.shape[-2:]
        s_ = self.mask_out_stride
        gt_masks = []
        for t in targets:
            m = t["masks"][..., s_ // 2::s_, s_ // 2::s_]
            gt_masks.append(F.pad(m.to(masks.dtype), (0, w - m.shape[-1], 0, h - m.shape[-2])))
        if sum(counts):
            gt_m = torch.cat(gt_masks).to(dev)[tgt].flatten(1)
            src = masks.flatten(1)
            pm = src.sigmoid()
            ce_m = F.binary_cross_entropy_with_logits(src, gt_m, reduction="none")
            fm = (0.25 * gt_m + 0.75 * (1 - gt_m)) * ce_m * (1 - (pm * gt_m + (1 - pm) * (1 - gt_m))) ** 2.0
            loss_mask = per_layer(fm.mean(1)) / denom * present
            dice = 1 - (2 * (pm * gt_m).sum(1) + 1) / (pm.sum(1) + gt_m.sum(1) + 1)
            loss_dice = per_layer(dice) / denom * present
        else:
            loss_mask = loss_dice = (masks * 0).sum() + torch.zeros(Ld, dtype=logits.dtype, device=dev)
        out = self.loss_reid({"pred_qd": pred_qd, "pred_logits": logits[-1]}, None, None, None, None)
        for l, suffix in enumerate([f"_{l}" for l in range(Ld - 1)] + [""]):
            out["loss_ce" + suffix], out["loss_bbox" + suffix], out["loss_giou" + suffix] = loss_ce[l], l1[l], giou[l]
            out["loss_mask" + suffix], out["loss_dice" + suffix] = loss_mask[l], loss_dice[l]
        return out

    def get_loss(self, loss, outputs, targets, ref_targets, indices, num_boxes, **kw):
        table = {"labels": self.loss_labels, "boxes": self.loss_boxes, "masks": self.loss_masks, "reid": self.loss_reid}
        assert loss in table, f"do you really want to compute {loss} loss?"
        return table[loss](outputs, targets, ref_targets, indices, num_boxes, **kw)

    def forward(self, outputs, targets, ref_targets, indices_list):
        losses = {}
        for name in self.losses:
            losses.update(self.get_loss(name, outputs, targets, ref_targets, indices_list[-1], None))
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            for name in self.losses:
                if name == "reid":
                    continue
                part = self.get_loss(name, aux, targets, ref_targets, indices_list[i], None)
                losses.update({f"{k}_{i}": v for k, v in part.items()})
        return losses


def reid_terms(key_embeds, ref_embeds, matched_ids, selections, loss_fn):
    """Sum the per-image reid terms.  key_embeds / ref_embeds [bz, Q, C]; matched_ids: best query
    per key-frame box (matcher); selections: select_pos_neg_masks(...) output;
    loss_fn(ref [Q,C], key [I,C], pos, neg, aux) -> (contrast sum, aux sum)."""
    contrast = aux = 0
    count = 0
    for i, (inst, pos, neg, aux_mask) in enumerate(selections):
        if len(inst) == 0:
            continue
        dev = key_embeds.device
        q = matched_ids[i][inst].to(dev)
        c, a = loss_fn(ref_embeds[i], key_embeds[i, q], pos.to(dev), neg.to(dev), aux_mask.to(dev))
        contrast, aux, count = contrast + c, aux + a, count + len(inst)
    return {"contrast": contrast, "aux": aux, "count": count}
