"""oracle/make_golden_clip_matching.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Drives the reference's `Videos` / `Clips` (projects/SeqFormer/seqformer/models/clip_output.py; its
detectron2 imports are unused by the two classes and stubbed) through overlapping synthetic clips and
stores the clips and the merged result in tests/golden/clip_matching.npz.

    python -m oracle.make_golden_clip_matching
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

from oracle.make_golden_criterion import _stub

REF = "/root/reference/projects/SeqFormer/seqformer/models/clip_output.py"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    _stub("detectron2")
    _stub("detectron2.structures", Instances=object)
    _stub("detectron2.utils")
    _stub("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    pkg = "_ref_clip"
    for name in (pkg, pkg + ".models", pkg + ".util"):
        _stub(name)
    _stub(pkg + ".util.misc", interpolate=torch.nn.functional.interpolate)
    spec = importlib.util.spec_from_file_location(pkg + ".models.clip_output", REF)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = pkg + ".models"
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod.Videos, mod.Clips


def synthetic_clips(seed, video_length=9, clip_length=4, stride=2, K=5, h=10, w=14, objects=3, per_clip=4):
    """-> list of (frame_idx, cls_probs [n,K], mask_logits [n,T,h,w]) for overlapping clips of one video."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(objects, 2, generator=g) * torch.tensor([w - 6.0, h - 4.0])
    vel = torch.randn(objects, 2, generator=g) * 0.4
    cls = torch.softmax(3 * torch.randn(objects, K, generator=g), -1)
    ys, xs = torch.arange(h)[:, None].float(), torch.arange(w)[None, :].float()
    clips = []
    for start in range(0, video_length, stride):
        end = start + clip_length
        last = end >= video_length
        frames = list(range(max(0, video_length - clip_length), video_length)) if last else list(range(start, end))
        order = torch.randperm(objects, generator=g).tolist()
        rows = []
        for k in order:
            m = []
            for t in frames:
                p = pos[k] + vel[k] * t
                inside = (xs >= p[0]) & (xs < p[0] + 6) & (ys >= p[1]) & (ys < p[1] + 4)
                m.append(torch.where(inside, 3.0, -3.0) + 0.6 * torch.randn(h, w, generator=g))
            rows.append((cls[k] * (0.8 + 0.2 * torch.rand(1, generator=g)), torch.stack(m)))
        for _ in range(per_clip - objects):            # a spurious instance
            rows.append((torch.softmax(torch.randn(K, generator=g), -1) * 0.3, -3.0 + 0.6 * torch.randn(len(frames), h, w, generator=g)))
        clips.append((frames, torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])))
        if last:
            break
    return clips, video_length, clip_length, K, (h, w)


def main():
    Videos, Clips = load_reference()
    d = {}
    for v, seed in enumerate((2, 3)):
        clips, L, clen, K, size = synthetic_clips(seed)
        video = Videos(clen, L, K, size, "cpu")
        for c, (frames, cls_probs, logits) in enumerate(clips):
            res = types.SimpleNamespace(pred_classes=cls_probs.argmax(1), scores=cls_probs.max(1)[0], cls_probs=cls_probs,
                                        pred_masks=logits)
            video.update(Clips(frames, res))
            d[f"v{v}.c{c}.frames"], d[f"v{v}.c{c}.cls"], d[f"v{v}.c{c}.logits"] = np.array(frames), cls_probs.numpy(), logits.numpy()
        out_cls, out_logits = video.get_result()
        d[f"v{v}.cfg"] = np.array([len(clips), L, clen, K, *size])
        d[f"v{v}.out_cls"], d[f"v{v}.out_logits"] = out_cls.numpy(), out_logits.numpy()
        print(f"video {v}: {len(clips)} clips -> {out_cls.shape[0]} tracks")
    path = os.path.join(OUT_DIR, "clip_matching.npz")
    np.savez_compressed(path, **d)
    print("clip matching fixture", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
