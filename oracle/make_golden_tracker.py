"""oracle/make_golden_tracker.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Drives the reference's IDOL_Tracker (projects/IDOL/idol/models/tracker.py:50-298; its only
missing import, torchvision.ops, is unused by the class and stubbed) through seeded synthetic
videos -- persistent objects with noisy identity embeddings, drop-outs, duplicate detections
and clutter -- and stores every frame's inputs and the ids it assigned in
tests/golden/tracker_idol.npz.

    python -m oracle.make_golden_tracker
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

from oracle.make_golden_criterion import _stub

REF = "/root/reference/projects/IDOL/idol/models/tracker.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

TRACKER_ARGS = dict(init_score_thr=0.2, obj_score_thr=0.1, nms_thr_pre=0.5, nms_thr_post=0.05, addnew_score_thr=0.2,
                    memo_tracklet_frames=10, memo_momentum=0.8, long_match=True, frame_weight=True,
                    temporal_weight=True, memory_len=3)          # idol.py:262-274 with the config defaults


def load_reference():
    tv = _stub("torchvision", __version__="0.15.0")
    tv.ops = _stub("torchvision.ops")
    spec = importlib.util.spec_from_file_location("_ref_tracker", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod.IDOL_Tracker


def synthetic_video(seed, frames=14, objects=5, C=16, h=16, w=24):
    """-> list over frames of (bboxes [n,5], labels [n], masks [n,1,h,w], embeds [n,C], indices)"""
    g = torch.Generator().manual_seed(seed)
    ident = 3.0 * torch.randn(objects, C, generator=g)
    pos = torch.rand(objects, 2, generator=g) * torch.tensor([w - 8.0, h - 6.0])
    vel = torch.randn(objects, 2, generator=g) * 0.6
    cls = torch.randint(0, 5, (objects,), generator=g)
    ys, xs = torch.arange(h)[:, None].float(), torch.arange(w)[None, :].float()
    out = []
    for t in range(frames):
        rows = []
        for k in range(objects):
            if torch.rand(1, generator=g).item() < 0.2 or (k == objects - 1 and t < 4):
                continue                                      # drop-out / late entry
            p = pos[k] + vel[k] * t
            def rect(dx=0.0):
                inside = (xs >= p[0] + dx) & (xs < p[0] + dx + 7) & (ys >= p[1]) & (ys < p[1] + 5)
                return torch.where(inside, 3.0, -3.0) + 0.5 * torch.randn(h, w, generator=g)
            score = 0.35 + 0.6 * torch.rand(1, generator=g).item()
            rows.append((score, int(cls[k]), rect(), ident[k] + 0.4 * torch.randn(C, generator=g)))
            if torch.rand(1, generator=g).item() < 0.35:      # a duplicate query on the same object
                rows.append((score * 0.8, int(cls[k]), rect(1.0), ident[k] + 0.6 * torch.randn(C, generator=g)))
        for _ in range(int(torch.randint(0, 3, (1,), generator=g))):   # clutter
            m = -3.0 + 0.5 * torch.randn(h, w, generator=g)
            y0, x0 = int(torch.randint(0, h - 3, (1,), generator=g)), int(torch.randint(0, w - 3, (1,), generator=g))
            m[y0:y0 + 3, x0:x0 + 3] += 6.0
            rows.append((0.1 + 0.25 * torch.rand(1, generator=g).item(), 0, m, 2.0 * torch.randn(C, generator=g)))
        rows.sort(key=lambda r: -r[0])
        n = len(rows)
        bboxes = torch.cat([torch.rand(n, 4, generator=g), torch.tensor([r[0] for r in rows])[:, None]], 1) if n else torch.zeros(0, 5)
        labels = torch.tensor([r[1] for r in rows], dtype=torch.long)
        masks = torch.stack([r[2] for r in rows])[:, None] if n else torch.zeros(0, 1, h, w)
        embeds = torch.stack([r[3] for r in rows]) if n else torch.zeros(0, C)
        indices = torch.randperm(300, generator=g)[:n].tolist()
        out.append((bboxes, labels, masks, embeds, indices))
    return out


def main():
    Tracker = load_reference()
    d = {}
    for v, seed in enumerate((3, 4, 5)):
        video = synthetic_video(seed)
        tr = Tracker(**TRACKER_ARGS)
        d[f"v{v}.frames"] = np.array(len(video))
        for t, (bboxes, labels, masks, embeds, indices) in enumerate(video):
            _, _, ids, kept = tr.match(bboxes=bboxes, labels=labels, masks=masks, track_feats=embeds, frame_id=t,
                                       indices=indices)
            p = f"v{v}.f{t}."
            d[p + "bboxes"], d[p + "labels"], d[p + "masks"] = bboxes.numpy(), labels.numpy(), masks.numpy()
            d[p + "embeds"], d[p + "indices"] = embeds.numpy(), np.array(indices, dtype=np.int64)
            d[p + "ids"], d[p + "kept"] = ids.numpy(), np.array(kept, dtype=np.int64)
        print(f"video {v}: {tr.num_tracklets} tracklets created,", len(tr.tracklets), "alive at the end")
    path = os.path.join(OUT_DIR, "tracker_idol.npz")
    np.savez_compressed(path, **d)
    print("tracker fixture", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
