"""oracle/make_golden_criterion.py -- TEST INFRASTRUCTURE ONLY (fixture generator).

Runs the reference's clip-level Hungarian matcher and SetCriterion
(projects/SeqFormer/seqformer/models/matcher.py:25-96, models/deformable_detr.py:231-439, with
sigmoid_focal_loss / dice_loss from models/segmentation_condInst.py:680-723) on seeded inputs
and stores inputs, the matching and every loss term in tests/golden/criterion_seqformer.npz.

The files import torchvision and fvcore, neither installed here.  They are loaded under a
synthetic package with two stubs:
  * torchvision.ops.boxes.box_area  -- (x1 - x0) * (y1 - y0), the published definition;
  * fvcore.nn.giou_loss (fvcore >= 0.1.5, < 0.1.6 per the reference's setup.py:179) -- restated
    from its published algorithm: loss = 1 - (IoU - (|C| - |A u B|) / (|C| + eps)), eps = 1e-7
    added to the union and to the enclosing-box area, reduction "none".
SetCriterion itself lives in a file whose other imports (backbone -> detectron2) cannot be
satisfied, so the class is cut out with ast and exec'd with the reference's own helpers.

    python -m oracle.make_golden_criterion
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch

REF = "/root/reference/projects/SeqFormer/seqformer"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def giou_loss_published(boxes1, boxes2, reduction="none", eps=1e-7):
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    assert (x2 >= x1).all() and (y2 >= y1).all()
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    loss = 1 - (iou - (area_c - union) / (area_c + eps))
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    """-> (HungarianMatcher, SetCriterion) classes of the reference."""
    box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])  # noqa: E731
    tv = _stub("torchvision", __version__="0.15.0")
    tv.ops = _stub("torchvision.ops", boxes=_stub("torchvision.ops.boxes", box_area=box_area))
    tv.ops.misc = _stub("torchvision.ops.misc")
    _stub("fvcore")
    _stub("fvcore.nn", giou_loss=giou_loss_published, smooth_l1_loss=None)
    pkg = "_ref_crit"
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

    def cut(path, names, kind):
        src = open(path).read()
        out = {}
        for node in ast.parse(src).body:
            if isinstance(node, kind) and node.name in names:
                out[node.name] = textwrap.dedent(ast.get_source_segment(src, node))
        return out

    ns = {"torch": torch, "nn": torch.nn, "F": torch.nn.functional, "box_ops": box_ops,
          "giou_loss": giou_loss_published, "accuracy": misc.accuracy,
          "nested_tensor_from_tensor_list": misc.nested_tensor_from_tensor_list,
          "is_dist_avail_and_initialized": misc.is_dist_avail_and_initialized,
          "get_world_size": misc.get_world_size}
    for name, code in cut(f"{REF}/models/segmentation_condInst.py", ["sigmoid_focal_loss", "dice_loss"],
                          ast.FunctionDef).items():
        exec(compile(code, name, "exec"), ns)
    exec(compile(cut(f"{REF}/models/deformable_detr.py", ["SetCriterion"], ast.ClassDef)["SetCriterion"],
                 "SetCriterion", "exec"), ns)
    return matcher.HungarianMatcher, ns["SetCriterion"]


def make_case(gen, bs, nf, Q, K, H, W, counts, layers):
    """Synthetic decoder outputs + clip targets.  Boxes cxcywh in (0, 1); masks binary [n, nf, H, W]."""
    def boxes(*shape):
        c = 0.2 + 0.6 * torch.rand(*shape, 2, generator=gen)
        wh = 0.05 + 0.3 * torch.rand(*shape, 2, generator=gen)
        return torch.cat([c, wh], -1)
    targets = []
    for n in counts:
        targets.append({"labels": torch.randint(0, K, (n,), generator=gen),
                        "boxes": boxes(n, nf),
                        "masks": (torch.rand(n, nf, H, W, generator=gen) > 0.6),
                        "size": torch.tensor([H, W])})
    outs = []
    for _ in range(layers):
        outs.append({"pred_logits": torch.randn(bs, Q, K, generator=gen),
                     "pred_boxes": boxes(bs, nf, Q)})
    return targets, outs


def main():
    torch.set_default_dtype(torch.float64)
    Matcher, Criterion = load_reference()
    gen = torch.Generator().manual_seed(11)
    bs, nf, Q, K, H, W, layers = 3, 2, 7, 5, 32, 64, 3
    counts = [2, 0, 3]
    targets, outs = make_case(gen, bs, nf, Q, K, H, W, counts, layers)
    matcher = Matcher(multi_frame=True, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    weight = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0, "loss_mask": 2.0, "loss_dice": 5.0}
    crit = Criterion(K, matcher, weight, ["labels", "boxes", "masks"], mask_out_stride=4, num_frames=nf)
    indices_list = [matcher(o, targets, nf, None) for o in outs]
    # predicted masks of the matched instances, per layer: list over clips of [1, n_i, nf, H/4, W/4]
    for o, ind in zip(outs, indices_list):
        o["pred_masks"] = [torch.randn(1, len(src), nf, H // 4, W // 4, generator=gen) for src, _ in ind]
    outputs = dict(outs[-1])
    outputs["aux_outputs"] = outs[:-1]
    losses = crit(outputs, targets, indices_list, None)

    d = {"cfg": np.array([bs, nf, Q, K, H, W, layers]), "counts": np.array(counts)}
    for i, t in enumerate(targets):
        d[f"t{i}.labels"], d[f"t{i}.boxes"], d[f"t{i}.masks"] = t["labels"].numpy(), t["boxes"].numpy(), t["masks"].numpy()
    for l, (o, ind) in enumerate(zip(outs, indices_list)):
        d[f"l{l}.logits"], d[f"l{l}.boxes"] = o["pred_logits"].numpy(), o["pred_boxes"].numpy()
        for i, (src, tgt) in enumerate(ind):
            d[f"l{l}.src{i}"], d[f"l{l}.tgt{i}"] = src.numpy(), tgt.numpy()
            d[f"l{l}.masks{i}"] = o["pred_masks"][i].numpy()
    for k, v in losses.items():
        d[f"loss.{k}"] = np.asarray(float(v))
    path = os.path.join(OUT_DIR, "criterion_seqformer.npz")
    np.savez_compressed(path, **d)
    print("criterion fixture", os.path.getsize(path) // 1024, "KiB")
    for k, v in losses.items():
        print(f"  {k:16s} {float(v):.6f}")


if __name__ == "__main__":
    main()
