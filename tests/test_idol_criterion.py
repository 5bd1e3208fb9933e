"""IDOL's simOTA matcher, positive/negative selection and criterion against outputs of the
reference classes (oracle/make_golden_idol_criterion.py)."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from vnext_amd.models.idol_criterion import IDOLCriterion, OTAMatcher, reid_terms, select_pos_neg_masks


def _loss_reid_torch(ref, key, pos, neg, aux):
    """tests-only restatement of vnext_amd.heads.loss_reid with torch matmuls (CPU)."""
    dot = ref @ key.t()
    cos = torch.nn.functional.normalize(ref, dim=1) @ torch.nn.functional.normalize(key, dim=1).t()
    lse_neg = torch.logsumexp(dot.masked_fill(~neg, float("-inf")), dim=0)
    lse_pos = torch.logsumexp((-dot).masked_fill(~pos, float("-inf")), dim=0)
    contrast = torch.nn.functional.softplus((lse_neg + lse_pos).clamp_min(torch.finfo(dot.dtype).min))
    a = (((cos - pos.to(cos.dtype)) ** 2) * aux).sum(0) / aux.sum(0).clamp_min(1)
    return contrast.sum(), a.sum()


@pytest.fixture(scope="module")
def case():
    g = dict(np.load(os.path.join(GOLDEN_DIR, "criterion_idol.npz")))
    bz, Q, K, H, W, layers, C = (int(v) for v in g["cfg"])
    def tg(name):
        return [{k: torch.from_numpy(g[f"{name}{i}.{k}"]) for k in ("labels", "boxes", "masks", "inst_id", "valid")}
                for i in range(bz)]
    outs = [{"pred_logits": torch.from_numpy(g[f"l{l}.logits"]), "pred_boxes": torch.from_numpy(g[f"l{l}.boxes"])}
            for l in range(layers)]
    return g, tg("det"), tg("ref"), outs, (bz, Q, K, H, W, layers, C)


def test_simota_matching_equals_reference(case):
    g, det, ref, outs, (bz, Q, K, H, W, layers, C) = case
    m = OTAMatcher()
    batched, matched_b = m.match_all_layers(torch.stack([o["pred_logits"] for o in outs]),
                                            torch.stack([o["pred_boxes"] for o in outs]), det)
    for l, o in enumerate(outs):
        single, matched = m(o, det)
        for i in range(bz):
            for sel, gt in (single[i], batched[l][i]):
                np.testing.assert_array_equal(sel.numpy(), g[f"l{l}.sel{i}"])
                np.testing.assert_array_equal(gt.numpy(), g[f"l{l}.gt{i}"])
    for i in range(bz):
        np.testing.assert_array_equal(matched[i].numpy(), g[f"matched{i}"])
        np.testing.assert_array_equal(matched_b[i].numpy(), g[f"matched{i}"])


def _selection(g, ref):
    random.seed(5)
    return select_pos_neg_masks(torch.from_numpy(g["ref_boxes"]), torch.from_numpy(g["ref_logits"]).sigmoid(), ref)


def test_contrastive_sets_equal_reference(case):
    g, det, ref, outs, (bz, Q, K, H, W, layers, C) = case
    sel = _selection(g, ref)
    head = torch.from_numpy(g["head_w"])
    key = torch.from_numpy(g["hs_key"]) @ head.t()
    refe = torch.from_numpy(g["hs_ref"]) @ head.t()
    j = 0
    for i, (inst, pos, neg, aux) in enumerate(sel):
        assert inst.tolist() == torch.nonzero(ref[i]["valid"]).flatten().tolist()
        for c, k in enumerate(inst.tolist()):
            label = g[f"item{j}.label"]
            assert int(pos[:, c].sum()) == int(label.sum()) and int(neg[:, c].sum()) == int((label == 0).sum())
            q = int(g[f"matched{i}"][k])
            rows = torch.cat([torch.nonzero(pos[:, c]).flatten(), torch.nonzero(neg[:, c]).flatten()])
            np.testing.assert_allclose((refe[i][rows] @ key[i, q]).numpy(), g[f"item{j}.contrast"][:, 0], rtol=1e-10, atol=1e-12)
            assert int(aux[:, c].sum()) == len(g[f"item{j}.aux_label"])
            j += 1
    assert j == int(g["n_items"])


def test_losses_equal_reference(case):
    g, det, ref, outs, (bz, Q, K, H, W, layers, C) = case
    crit = IDOLCriterion(K, OTAMatcher(), {}, ["labels", "boxes", "masks", "reid"], mask_out_stride=4)
    indices_list = []
    for l, o in enumerate(outs):
        indices_list.append([(torch.from_numpy(g[f"l{l}.sel{i}"]), torch.from_numpy(g[f"l{l}.gt{i}"])) for i in range(bz)])
        o["pred_masks"] = [torch.from_numpy(g[f"l{l}.masks{i}"]) for i in range(bz)]
    head = torch.from_numpy(g["head_w"])
    key = torch.from_numpy(g["hs_key"]) @ head.t()
    refe = torch.from_numpy(g["hs_ref"]) @ head.t()
    matched = [torch.from_numpy(g[f"matched{i}"]) for i in range(bz)]
    outputs = dict(outs[-1])
    outputs["pred_qd"] = reid_terms(key, refe, matched, _selection(g, ref), _loss_reid_torch)
    outputs["aux_outputs"] = outs[:-1]
    losses = crit(outputs, det, ref, indices_list)
    want = {k[5:]: float(v) for k, v in g.items() if k.startswith("loss.")}
    assert set(losses) == set(want)
    for k, v in want.items():
        # the reference computes the cosine term in fp32 (`.float()`, pos_neg_select.py:59-60)
        np.testing.assert_allclose(float(losses[k]), v, rtol=1e-6 if k == "loss_reid_aux" else 1e-9, atol=1e-12, err_msg=k)


def test_all_layers_in_one_pass_equals_the_per_layer_form(case):
    g, det, ref, outs, (bz, Q, K, H, W, layers, C) = case
    crit = IDOLCriterion(K, OTAMatcher(), {}, ["labels", "boxes", "masks", "reid"], mask_out_stride=4)
    indices_list = [[(torch.from_numpy(g[f"l{l}.sel{i}"]), torch.from_numpy(g[f"l{l}.gt{i}"])) for i in range(bz)]
                    for l in range(layers)]
    masks = torch.cat([torch.cat([torch.from_numpy(g[f"l{l}.masks{i}"]) for i in range(bz)], 1)[0] for l in range(layers)])
    head = torch.from_numpy(g["head_w"])
    key = torch.from_numpy(g["hs_key"]) @ head.t()
    refe = torch.from_numpy(g["hs_ref"]) @ head.t()
    matched = [torch.from_numpy(g[f"matched{i}"]) for i in range(bz)]
    qd = reid_terms(key, refe, matched, _selection(g, ref), _loss_reid_torch)
    got = crit.forward_all_layers(torch.stack([o["pred_logits"] for o in outs]), torch.stack([o["pred_boxes"] for o in outs]),
                                  masks, det, indices_list, qd)
    want = {k[5:]: float(v) for k, v in g.items() if k.startswith("loss.")}
    assert set(got) == set(want)
    for k, v in want.items():
        np.testing.assert_allclose(float(got[k]), v, rtol=1e-6 if k == "loss_reid_aux" else 1e-9, atol=1e-12, err_msg=k)


def test_no_objects_in_any_key_frame():
    K, Q = 4, 110
    crit = IDOLCriterion(K, OTAMatcher(), {}, ["labels", "boxes", "masks", "reid"])
    empty = {"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4),
             "masks": torch.zeros(0, 32, 32, dtype=torch.bool), "valid": torch.zeros(0, dtype=torch.bool)}
    out = {"pred_logits": torch.randn(1, Q, K), "pred_boxes": torch.rand(1, Q, 4)}
    ind, matched = crit.matcher.match_all_layers(out["pred_logits"][None], out["pred_boxes"][None], [empty])
    assert not ind[0][0][0].any() and matched[0].numel() == 0
    sel = select_pos_neg_masks(out["pred_boxes"], out["pred_logits"].sigmoid(), [empty])
    out["pred_masks"] = [torch.zeros(1, 0, 1, 8, 8)]
    out["pred_qd"] = reid_terms(torch.zeros(1, Q, 8), torch.zeros(1, Q, 8), matched, sel, _loss_reid_torch)
    losses = crit(out, [empty], [empty], ind)
    assert float(losses["loss_reid"]) == 0 and float(losses["loss_bbox"]) == 0 and float(losses["loss_ce"]) > 0


@pytest.mark.gpu
def test_reid_terms_on_the_hip_kernels_match_the_torch_restatement(case):
    from vnext_amd.heads import loss_reid
    g, det, ref, outs, (bz, Q, K, H, W, layers, C) = case
    head = torch.from_numpy(g["head_w"]).float()
    key = (torch.from_numpy(g["hs_key"]).float() @ head.t()).cuda()
    refe = (torch.from_numpy(g["hs_ref"]).float() @ head.t()).cuda()
    matched = [torch.from_numpy(g[f"matched{i}"]) for i in range(bz)]
    sel = _selection(g, ref)
    got = reid_terms(key, refe, matched, sel, loss_reid)
    want = reid_terms(key.cpu().double(), refe.cpu().double(), matched, sel, _loss_reid_torch)
    assert got["count"] == want["count"] == int(g["n_items"])
    np.testing.assert_allclose(float(got["contrast"]), float(want["contrast"]), rtol=2e-5)
    np.testing.assert_allclose(float(got["aux"]), float(want["aux"]), rtol=2e-5)
    np.testing.assert_allclose(float(got["contrast"]) / got["count"], float(g["loss.loss_reid"]), rtol=1e-4)
