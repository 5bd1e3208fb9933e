"""oracle/make_golden_idol_criterion.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs IDOL's training-side association logic from the reference on seeded inputs and stores
inputs + results in tests/golden/criterion_idol.npz:
  * simOTA matcher          projects/IDOL/idol/models/matcher.py:45-170
  * positive/negative pick  projects/IDOL/idol/models/pos_neg_select.py:13-199 (incl. the
    host-side `random.sample` of negatives -- the draw order is reproduced with random.seed)
  * SetCriterion            projects/IDOL/idol/models/deformable_detr.py:236-494
Stubs as in make_golden_criterion.py, plus torchvision.ops.box_iou (pairwise IoU, the published
definition).

    python -m oracle.make_golden_idol_criterion
"""
from __future__ import annotations

import ast
import importlib.util
import os
import random
import sys
import textwrap

import numpy as np
import torch

from oracle.make_golden_criterion import _stub, giou_loss_published

REF = "/root/reference/projects/IDOL/idol"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def box_iou_published(a, b):
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])  # noqa: E731
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area(a)[:, None] + area(b) - inter)


def load_reference():
    tv = _stub("torchvision", __version__="0.15.0")
    tv.ops = _stub("torchvision.ops", box_iou=box_iou_published,
                   boxes=_stub("torchvision.ops.boxes", box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])))
    tv.ops.misc = _stub("torchvision.ops.misc")
    _stub("fvcore")
    _stub("fvcore.nn", giou_loss=giou_loss_published, smooth_l1_loss=None)
    pkg = "_ref_idol"
    for name in (pkg, pkg + ".models", pkg + ".util"):
        _stub(name)

    def load(rel, modname):
        spec = importlib.util.spec_from_file_location(f"{pkg}.{modname}", f"{REF}/{rel}")
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = f"{pkg}.{modname}".rsplit(".", 1)[0]
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        return mod

    box_ops = load("util/box_ops.py", "util.box_ops")
    sys.modules[pkg + ".util"].box_ops = box_ops
    misc = load("util/misc.py", "util.misc")
    matcher = load("models/matcher.py", "models.matcher")
    pns = load("models/pos_neg_select.py", "models.pos_neg_select")

    def cut(path, names, kind):
        src = open(path).read()
        return {n.name: textwrap.dedent(ast.get_source_segment(src, n)) for n in ast.parse(src).body
                if isinstance(n, kind) and n.name in names}

    ns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "box_ops": box_ops,
          "giou_loss": giou_loss_published, "accuracy": misc.accuracy,
          "nested_tensor_from_tensor_list": misc.nested_tensor_from_tensor_list,
          "is_dist_avail_and_initialized": misc.is_dist_avail_and_initialized, "get_world_size": misc.get_world_size}
    for name, code in cut(f"{REF}/models/segmentation_condInst.py", ["sigmoid_focal_loss", "dice_loss"],
                          ast.FunctionDef).items():
        exec(compile(code, name, "exec"), ns)
    exec(compile(cut(f"{REF}/models/deformable_detr.py", ["SetCriterion"], ast.ClassDef)["SetCriterion"],
                 "SetCriterion", "exec"), ns)
    return matcher.HungarianMatcher, pns.select_pos_neg, ns["SetCriterion"]


def make_case(gen, bz, Q, K, H, W, counts, layers, C):
    def boxes(n):
        c = 0.2 + 0.6 * torch.rand(n, 2, generator=gen)
        wh = 0.1 + 0.3 * torch.rand(n, 2, generator=gen)
        return torch.cat([c, wh], -1)

    det, ref = [], []
    for n in counts:
        b = boxes(n)
        labels = torch.randint(0, K, (n,), generator=gen)
        ids = torch.arange(n)
        det.append({"labels": labels, "boxes": b, "masks": torch.rand(n, H, W, generator=gen) > 0.6,
                    "inst_id": ids, "valid": torch.ones(n, dtype=torch.bool)})
        valid = torch.ones(n, dtype=torch.bool)
        if n > 2:
            valid[1] = False                      # one object left the reference frame
        ref.append({"labels": labels.clone(), "boxes": (b + 0.03 * torch.randn(n, 4, generator=gen)).clamp(0.02, 0.98),
                    "masks": torch.rand(n, H, W, generator=gen) > 0.6, "inst_id": ids, "valid": valid})

    def preds(tgts):
        out = []
        for _ in range(layers):
            pb = torch.stack([boxes(Q) for _ in range(bz)])
            for i, t in enumerate(tgts):         # a few queries sit near every ground-truth box
                n = len(t["labels"])
                for r in range(4):
                    if n:
                        pb[i, r * n:(r + 1) * n] = (t["boxes"] + 0.02 * (r + 1) * torch.randn(n, 4, generator=gen)).clamp(0.02, 0.98)
            out.append({"pred_logits": torch.randn(bz, Q, K, generator=gen), "pred_boxes": pb})
        return out
    return det, ref, preds(det), preds(ref)[-1]


def main():
    torch.set_default_dtype(torch.float64)
    Matcher, select_pos_neg, Criterion = load_reference()
    gen = torch.Generator().manual_seed(23)
    bz, Q, K, H, W, layers, C = 3, 120, 6, 32, 64, 2, 16
    counts = [3, 0, 2]
    det, ref, outs, ref_out = make_case(gen, bz, Q, K, H, W, counts, layers, C)
    matcher = Matcher(multi_frame=True, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    weight = {}
    crit = Criterion(K, matcher, weight, ["labels", "boxes", "masks", "reid"], mask_out_stride=4, num_frames=1)
    indices_list, matched = [], None
    for o in outs:
        ind, matched = matcher(o, det)
        indices_list.append(ind)
    for o, ind in zip(outs, indices_list):
        o["pred_masks"] = [torch.randn(1, int(sel.sum()), 1, H // 4, W // 4, generator=gen) for sel, _ in ind]
    # reid branch: embeddings of key / reference frame queries through a fixed linear "head"
    hs_key, hs_ref = torch.randn(bz, Q, C, generator=gen), torch.randn(bz, Q, C, generator=gen)
    head_w = torch.randn(C, C, generator=gen) / C ** 0.5
    head = lambda x: x @ head_w.t()  # noqa: E731
    ref_cls = ref_out["pred_logits"].sigmoid()
    random.seed(5)
    items = select_pos_neg(ref_out["pred_boxes"], matched, ref, det, head, hs_key, hs_ref, ref_cls)
    outputs = dict(outs[-1])
    outputs["pred_qd"] = items
    outputs["aux_outputs"] = outs[:-1]
    losses = crit(outputs, det, ref, indices_list)

    d = {"cfg": np.array([bz, Q, K, H, W, layers, C]), "counts": np.array(counts), "head_w": head_w.numpy(),
         "hs_key": hs_key.numpy(), "hs_ref": hs_ref.numpy(), "ref_logits": ref_out["pred_logits"].numpy(),
         "ref_boxes": ref_out["pred_boxes"].numpy(), "n_items": np.array(len(items))}
    for name, tg in (("det", det), ("ref", ref)):
        for i, t in enumerate(tg):
            for k, v in t.items():
                d[f"{name}{i}.{k}"] = v.numpy()
    for l, (o, ind) in enumerate(zip(outs, indices_list)):
        d[f"l{l}.logits"], d[f"l{l}.boxes"] = o["pred_logits"].numpy(), o["pred_boxes"].numpy()
        for i, (sel, gt) in enumerate(ind):
            d[f"l{l}.sel{i}"], d[f"l{l}.gt{i}"] = sel.numpy(), gt.numpy()
            d[f"l{l}.masks{i}"] = o["pred_masks"][i].numpy()
    for i, m in enumerate(matched):
        d[f"matched{i}"] = m.numpy()
    for j, it in enumerate(items):
        for k, v in it.items():
            d[f"item{j}.{k}"] = v.numpy()
    for k, v in losses.items():
        d[f"loss.{k}"] = np.asarray(float(v))
    path = os.path.join(OUT_DIR, "criterion_idol.npz")
    np.savez_compressed(path, **d)
    print("IDOL criterion fixture", os.path.getsize(path) // 1024, "KiB;", len(items), "contrast items;",
          [int(s.sum()) for s, _ in indices_list[-1]], "matched queries")
    for k, v in losses.items():
        print(f"  {k:16s} {float(v):.6f}")


if __name__ == "__main__":
    main()
