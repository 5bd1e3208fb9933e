"""oracle/make_golden_idol_inference.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs the reference's video-level post-processing `IDOL.inference`
(projects/IDOL/idol/idol.py:313-471: per-frame score threshold + class-aware box NMS, the
reference IDOL_Tracker, sequence filtering, temporal score, mask resizing) on synthetic network
outputs and stores inputs + the returned video_output in tests/golden/inference_idol.npz.
The method is cut out of idol.py with ast (the file imports detectron2) and bound to a minimal
`self`; torchvision.ops.batched_nms is restated from its published algorithm (per-class greedy
NMS on IoU, indices returned in descending-score order).

    python -m oracle.make_golden_idol_inference
"""
from __future__ import annotations

import os
import types

import numpy as np
import torch
import torch.nn.functional as F

from oracle.make_golden_tracker import load_reference as load_tracker
from oracle.ref_extract import extract

REF = "/root/reference/projects/IDOL/idol"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def batched_nms_published(boxes, scores, idxs, thr):
    order = torch.argsort(scores, descending=True, stable=True)
    keep = []
    alive = torch.ones(len(order), dtype=torch.bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for a in range(len(order)):
        if not alive[a]:
            continue
        i = order[a]
        keep.append(int(i))
        for b in range(a + 1, len(order)):
            j = order[b]
            if not alive[b] or idxs[i] != idxs[j]:
                continue
            lt, rb = torch.max(boxes[i, :2], boxes[j, :2]), torch.min(boxes[i, 2:], boxes[j, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[0] * wh[1]
            if inter / (area[i] + area[j] - inter) > thr:
                alive[b] = False
    return torch.tensor(keep, dtype=torch.long)


def synthetic_outputs(seed, frames=12, Q=60, K=6, C=16, h=12, w=20, objects=4):
    g = torch.Generator().manual_seed(seed)
    ident = 3.0 * torch.randn(objects, C, generator=g)
    cls = torch.randint(0, K, (objects,), generator=g)
    pos = torch.rand(objects, 2, generator=g) * torch.tensor([w - 7.0, h - 5.0])
    ys, xs = torch.arange(h)[:, None].float(), torch.arange(w)[None, :].float()
    logits = -4.0 + 0.5 * torch.randn(frames, Q, K, generator=g)
    boxes = torch.cat([0.2 + 0.6 * torch.rand(frames, Q, 2, generator=g), 0.05 + 0.2 * torch.rand(frames, Q, 2, generator=g)], -1)
    masks = -3.0 + 0.5 * torch.randn(frames, Q, h, w, generator=g)
    embeds = torch.randn(frames, Q, C, generator=g)
    for t in range(frames):
        slots = torch.randperm(Q, generator=g)
        s = 0
        for k in range(objects):
            if torch.rand(1, generator=g).item() < 0.2:
                continue
            for dup in range(2 if torch.rand(1, generator=g).item() < 0.4 else 1):   # duplicates: near-identical boxes
                q = int(slots[s]); s += 1
                logits[t, q, cls[k]] = 1.5 - 0.8 * dup + 0.5 * torch.randn(1, generator=g).item()
                p = pos[k] + 0.3 * t
                boxes[t, q] = torch.tensor([(p[0] + 3.5) / w, (p[1] + 2.5) / h, 7.0 / w, 5.0 / h]) + 0.004 * dup
                inside = (xs >= p[0]) & (xs < p[0] + 7) & (ys >= p[1]) & (ys < p[1] + 5)
                masks[t, q] = torch.where(inside, 3.0, -3.0) + 0.5 * torch.randn(h, w, generator=g)
                embeds[t, q] = ident[k] + 0.4 * torch.randn(C, generator=g)
    return {"pred_logits": logits, "pred_masks": masks[:, :, None], "pred_boxes": boxes, "pred_inst_embed": embeds}


def main():
    Tracker = load_tracker()
    ops = types.SimpleNamespace(batched_nms=batched_nms_published)

    def box_cxcywh_to_xyxy(x):
        c, wh = x[..., :2], x[..., 2:]
        return torch.cat([c - 0.5 * wh, c + 0.5 * wh], -1)
    fn = extract(f"{REF}/idol.py", ["inference"], {"ops": ops, "box_cxcywh_to_xyxy": box_cxcywh_to_xyxy})["inference"]
    d = {}
    for v, seed in enumerate((7, 8)):
        out = synthetic_outputs(seed)
        me = types.SimpleNamespace(inference_select_thres=0.1, temporal_score_type="mean", is_multi_cls=True,
                                   apply_cls_thres=0.05)
        tracker = Tracker(init_score_thr=0.2, obj_score_thr=0.1, nms_thr_pre=0.5, nms_thr_post=0.05,
                          addnew_score_thr=0.2, memo_tracklet_frames=10, memo_momentum=0.8, long_match=True,
                          frame_weight=True, temporal_weight=True, memory_len=3)
        h4, w4 = out["pred_masks"].shape[-2:]
        image_size, ori_size = (h4 * 4 - 5, w4 * 4 - 3), (h4 * 4 + 9, w4 * 4 + 13)   # network input (unpadded), original video
        inputs = {k: t.clone() for k, t in out.items()}
        res = fn(me, out, tracker, ori_size, image_size)
        for k, t in inputs.items():
            d[f"v{v}.{k}"] = t.numpy()
        d[f"v{v}.sizes"] = np.array([*ori_size, *image_size])
        d[f"v{v}.scores"] = np.array(res["pred_scores"])
        d[f"v{v}.labels"] = np.array(res["pred_labels"])
        F_ = inputs["pred_logits"].shape[0]
        present = np.zeros((len(res["pred_masks"]), F_), dtype=bool)
        stack = np.zeros((len(res["pred_masks"]), F_, *ori_size), dtype=bool)
        for i, track in enumerate(res["pred_masks"]):
            for t, m in enumerate(track):
                if m is not None:
                    present[i, t] = True
                    stack[i, t] = m.numpy()
        d[f"v{v}.present"], d[f"v{v}.masks"] = present, np.packbits(stack, axis=-1)
        print(f"video {v}: {len(res['pred_scores'])} (track, class) results, labels {sorted(set(res['pred_labels']))}, "
              f"{int(present.sum())} masks")
    path = os.path.join(OUT_DIR, "inference_idol.npz")
    np.savez_compressed(path, **d)
    print("IDOL inference fixture", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
