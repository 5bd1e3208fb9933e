"""SeqFormer's clip-level matcher and criterion (callers of the hot path on the training side)
against outputs of the reference classes (oracle/make_golden_criterion.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from vnext_amd.models.criterion import HungarianMatcher, SetCriterion, giou_loss, pairwise_giou


@pytest.fixture(scope="module")
def case():
    g = dict(np.load(os.path.join(GOLDEN_DIR, "criterion_seqformer.npz")))
    bs, nf, Q, K, H, W, layers = (int(v) for v in g["cfg"])
    targets = [{"labels": torch.from_numpy(g[f"t{i}.labels"]), "boxes": torch.from_numpy(g[f"t{i}.boxes"]),
                "masks": torch.from_numpy(g[f"t{i}.masks"]), "size": torch.tensor([H, W])} for i in range(bs)]
    outs = [{"pred_logits": torch.from_numpy(g[f"l{l}.logits"]), "pred_boxes": torch.from_numpy(g[f"l{l}.boxes"])}
            for l in range(layers)]
    return g, targets, outs, (bs, nf, Q, K, H, W, layers)


def _matcher():
    return HungarianMatcher(multi_frame=True, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)


def test_matching_equals_reference_per_layer_and_batched(case):
    g, targets, outs, (bs, nf, Q, K, H, W, layers) = case
    m = _matcher()
    batched = m.match_all_layers(torch.stack([o["pred_logits"] for o in outs]),
                                 torch.stack([o["pred_boxes"] for o in outs]), targets)
    for l, o in enumerate(outs):
        single = m(o, targets, nf, None)
        for i in range(bs):
            for got in (single[i], batched[l][i]):
                np.testing.assert_array_equal(got[0].numpy(), g[f"l{l}.src{i}"])
                np.testing.assert_array_equal(got[1].numpy(), g[f"l{l}.tgt{i}"])


def test_losses_equal_reference(case):
    g, targets, outs, (bs, nf, Q, K, H, W, layers) = case
    m = _matcher()
    weight = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0, "loss_mask": 2.0, "loss_dice": 5.0}
    crit = SetCriterion(K, m, weight, ["labels", "boxes", "masks"], mask_out_stride=4, num_frames=nf)
    indices_list = []
    for l, o in enumerate(outs):
        indices_list.append([(torch.from_numpy(g[f"l{l}.src{i}"]), torch.from_numpy(g[f"l{l}.tgt{i}"])) for i in range(bs)])
        o["pred_masks"] = [torch.from_numpy(g[f"l{l}.masks{i}"]) for i in range(bs)]
    outputs = dict(outs[-1])
    outputs["aux_outputs"] = outs[:-1]
    losses = crit(outputs, targets, indices_list, None)
    want = {k[5:]: float(v) for k, v in g.items() if k.startswith("loss.")}
    assert set(losses) == set(want)
    for k, v in want.items():
        np.testing.assert_allclose(float(losses[k]), v, rtol=1e-10, atol=1e-12, err_msg=k)
    # masks handed over as one tensor (how the fused mask head returns them) give the same numbers
    outputs["pred_masks"] = torch.cat(outs[-1]["pred_masks"], 1)[0]
    again = crit.loss_masks(outputs, targets, indices_list[-1], torch.tensor(5.0, dtype=torch.float64))
    np.testing.assert_allclose(float(again["loss_dice"]), want["loss_dice"], rtol=1e-10)


def test_all_layers_in_one_pass_equals_the_per_layer_form(case):
    g, targets, outs, (bs, nf, Q, K, H, W, layers) = case
    crit = SetCriterion(K, _matcher(), {}, ["labels", "boxes", "masks"], mask_out_stride=4, num_frames=nf)
    indices_list = [[(torch.from_numpy(g[f"l{l}.src{i}"]), torch.from_numpy(g[f"l{l}.tgt{i}"])) for i in range(bs)]
                    for l in range(layers)]
    masks = torch.cat([torch.cat([torch.from_numpy(g[f"l{l}.masks{i}"]) for i in range(bs)], 1)[0] for l in range(layers)])
    got = crit.forward_all_layers(torch.stack([o["pred_logits"] for o in outs]), torch.stack([o["pred_boxes"] for o in outs]),
                                  masks, targets, indices_list)
    want = {k[5:]: float(v) for k, v in g.items() if k.startswith("loss.")}
    assert set(got) == set(want)
    for k, v in want.items():
        np.testing.assert_allclose(float(got[k]), v, rtol=1e-10, atol=1e-12, err_msg=k)


def test_no_targets_anywhere():
    K, nf, Q = 4, 2, 5
    crit = SetCriterion(K, _matcher(), {}, ["labels", "boxes", "masks"], num_frames=nf)
    targets = [{"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, nf, 4),
                "masks": torch.zeros(0, nf, 32, 32, dtype=torch.bool)}]
    out = {"pred_logits": torch.randn(1, Q, K), "pred_boxes": torch.rand(1, nf, Q, 4)}
    ind = crit.matcher.match_all_layers(out["pred_logits"][None], out["pred_boxes"][None], targets)
    assert ind[0][0][0].numel() == 0
    out["pred_masks"] = [torch.zeros(1, 0, nf, 8, 8)]
    losses = crit(out, targets, ind)
    assert float(losses["loss_bbox"]) == 0 and float(losses["loss_mask"]) == 0 and float(losses["loss_ce"]) > 0


def test_giou_forms_agree_on_overlapping_boxes():
    a = torch.tensor([[0.1, 0.1, 0.5, 0.6], [0.2, 0.3, 0.9, 0.8]], dtype=torch.float64)
    b = torch.tensor([[0.3, 0.2, 0.7, 0.7], [0.0, 0.0, 0.1, 0.1]], dtype=torch.float64)
    pw = pairwise_giou(a, b)
    np.testing.assert_allclose((1 - giou_loss(a, b)).numpy(), torch.diagonal(pw).numpy(), atol=1e-6)
    assert float(giou_loss(a[:1], a[:1])) < 1e-6
